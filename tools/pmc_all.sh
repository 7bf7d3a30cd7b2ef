#!/bin/bash
# usage: tools/pmc_all.sh <tag> <python script + args, relative to the repo root>
# three rocprofv3 runs of the same command (FETCH_SIZE pass, WRITE_SIZE pass, kernel trace + stats) -> gpurun_out/<tag>_*
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_$ctr -- python3 $R/$1 "${@:2}" > $R/gpurun_out/${tag}_$ctr.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python3 $R/$1 "${@:2}" > $R/gpurun_out/${tag}_stats.log 2>&1 || exit 1
F=$(ls $R/gpurun_out/${tag}_FETCH_SIZE/*/*counter_collection.csv | head -1)
W=$(ls $R/gpurun_out/${tag}_WRITE_SIZE/*/*counter_collection.csv | head -1)
S=$(ls $R/gpurun_out/${tag}_stats/*/*kernel_stats.csv | head -1)
cp $S $R/gpurun_out/${tag}_kernel_stats.csv
python3 $R/tools/traffic_table.py $F $W $S $R/gpurun_out/${tag}_traffic.json "command: $*"
