#!/bin/bash
# usage: tools/pmc_script.sh <tag> "<COUNTER COUNTER ...>" <python script relative to the repo root> [args]
# one rocprofv3 --pmc pass (+ kernel trace) of any tool; per-kernel means of the counters and of the duration, and the
# clock GRBM_GUI_ACTIVE / duration when that counter is in the list -> gpurun_out/<tag>.json
tag=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/$tag -- python3 $R/$1 "${@:2}" > $R/gpurun_out/$tag.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections,json
f=glob.glob("$R/gpurun_out/$tag/*/*counter_collection.csv")
rows=list(csv.DictReader(open(f[0])))
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
dur=collections.defaultdict(lambda:[0,0.0])
seen=set()
for r in rows:
    k=r["Kernel_Name"].split("(")[0].replace("void ","")[:60]
    a=agg[k][r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    did=r["Dispatch_Id"]
    if did not in seen and "Start_Timestamp" in r:
        seen.add(did); d=dur[k]; d[0]+=1; d[1]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
out={}
for k,cs in agg.items():
    o={c:v/n for c,(n,v) in cs.items()}
    if dur[k][0]:
        o["duration_us"]=dur[k][1]/dur[k][0]/1e3
        if "GRBM_GUI_ACTIVE" in o: o["clock_GHz"]=round(o["GRBM_GUI_ACTIVE"]/(o["duration_us"]*1e3),3)
    o["calls"]=max(n for n,_ in cs.values())
    out[k]=o
out=dict(sorted(out.items(), key=lambda kv:-kv[1].get("duration_us",0)*kv[1]["calls"])[:12])
print(json.dumps(out, indent=1))
json.dump(out, open("$R/gpurun_out/$tag.json","w"), indent=1)
PY
