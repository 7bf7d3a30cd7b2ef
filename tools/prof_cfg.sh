#!/bin/bash
# usage: tools/prof_cfg.sh <tag> <cfg...>   rocprofv3 kernel stats of tools/bench_configs.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    print(f"{r['Name'][:96]:96s} n={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):5.1f}%")
PY
