#!/bin/bash
# usage: tools/pmc_cfg.sh <tag> <COUNTER> <config name for tools/bench_configs.py>   (one PMC pass, per-kernel mean)
tag=$1; ctr=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag/*/*counter_collection.csv")
rows=list(csv.DictReader(open(f[0])))
agg=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    k=r.get("Kernel_Name","?")[:80]
    agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
for k,(n,v) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:12]:
    print(f"{k:80s} n={n:4d} $ctr/launch={v/n:14.1f}")
PY
