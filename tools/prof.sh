#!/bin/bash
# usage: tools_prof.sh <tag> [bench args...]   (runs on the GPU box; writes gpurun_out/<tag>/ + top kernels)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:28]:
    print(f"{r['Name'][:100]:100s} n={r['Calls']:>5s} tot_us={float(r['TotalDurationNs'])/1e3:9.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):5.1f}%")
PY
