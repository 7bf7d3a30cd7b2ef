#!/bin/bash
# usage: tools/pmc_multi.sh <tag> "<COUNTER COUNTER ...>" [bench args]   (one PMC pass, several counters, per-kernel means)
tag=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
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
    if "nnconv" in k or "tconv" in k:
        out[k]={c:v/n for c,(n,v) in cs.items()}
print(json.dumps(out, indent=1))
json.dump(out, open("$GRAFT_REPO_ROOT/gpurun_out/$tag.json","w"), indent=1)
PY
