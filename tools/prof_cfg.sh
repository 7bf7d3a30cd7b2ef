#!/bin/bash
# usage: tools/prof_cfg.sh <tag> <config name for tools/bench_configs.py>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print("total kernel time us per step (13 steps):", tot/1e3/13, "launches per step:", calls/13)
for r in rows[:25]:
    print(f"{r['Name'][:90]:90s} n={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):5.1f}%")
PY
