#!/bin/bash
# usage: tools/pmc_gen_kernels.sh <tag> "<COUNTER COUNTER ...>" [bench_gen_kernels args]   (one PMC pass, per-kernel means)
tag=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/tools/bench_gen_kernels.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
python3 - <<PY
import csv,glob,collections,json
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag/*/*counter_collection.csv")
rows=list(csv.DictReader(open(f[0])))
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for r in rows:
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    a=agg[k][r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
out={}
for k,cs in agg.items():
    if "nnconv" in k:
        out[k]={c:v/n for c,(n,v) in cs.items()}
        d=out[k]
        if "GRBM_GUI_ACTIVE" in d:
            simd_cycles=d["GRBM_GUI_ACTIVE"]/8*1024
            if "SQ_VALU_MFMA_BUSY_CYCLES" in d: d["mfma_busy"]=round(d["SQ_VALU_MFMA_BUSY_CYCLES"]/simd_cycles,3)
            if "SQ_ACTIVE_INST_VALU" in d: d["valu_busy"]=round(4*d["SQ_ACTIVE_INST_VALU"]/simd_cycles,3)
print(json.dumps(out, indent=1))
json.dump(out, open("$GRAFT_REPO_ROOT/gpurun_out/$tag.json","w"), indent=1)
PY
