#!/bin/bash
# usage: tools/pmc.sh <tag> <COUNTER>   (one PMC pass: per-dispatch counter rows for the eager bench)
tag=$1; ctr=$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag/*/*counter_collection.csv")
print(f)
rows=list(csv.DictReader(open(f[0])))
print(rows[0].keys())
agg=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    k=r.get("Kernel_Name","?")[:70]
    agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
for k,(n,v) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:14]:
    print(f"{k:70s} n={n:4d} $ctr/launch={v/n:14.1f}")
PY
