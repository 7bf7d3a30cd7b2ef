#!/bin/bash
# usage (in the container, after gpurun merged gpurun_out/): bash tools/collect_profiles.sh <tag>
t=$1
cp gpurun_out/${t}_bench.json profiles/${t}_bench.json
cp $(ls gpurun_out/${t}/*/*kernel_stats.csv | head -1) profiles/${t}_graph_step_kernel_stats.csv
cp gpurun_out/${t}_pmc.json profiles/${t}_pmc_mfma_valu_counters.json
python tools/busy_from_pmc.py gpurun_out/${t}_pmc.json profiles/${t}_pmc_mfma_valu_busy.json > /dev/null
cp gpurun_out/${t}_cfg3_traffic.json profiles/r04_cfg3_traffic.json      # the file bench.py reads
cp gpurun_out/${t}_cfg3_kernel_stats.csv profiles/r04_cfg3_kernel_stats.csv
cp gpurun_out/${t}_other_configs.jsonl profiles/${t}_other_configs.jsonl
cp gpurun_out/${t}_gemm.jsonl profiles/${t}_gemm.jsonl
ls -la profiles/${t}_*
