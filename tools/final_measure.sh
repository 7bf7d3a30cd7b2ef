#!/bin/bash
# usage (on the GPU box): bash tools/final_measure.sh <tag>      e.g. r04_g
# the measurement set a round's records come from: full bench line, replayed-step kernel stats, MFMA / VALU busy pass,
# cfg3 per-kernel traffic (three rocprofv3 passes), the other configs, the GEMM table.  Everything lands in gpurun_out/;
# tools/collect_profiles.sh copies the judged files to profiles/.
t=$1
python bench.py > gpurun_out/${t}_bench.json 2> gpurun_out/${t}_bench.err || exit 1
bash tools/prof.sh ${t} --no-lightpath --no-reference-scale --no-cpu-baseline > gpurun_out/${t}_prof.txt || exit 1
bash tools/pmc_multi.sh ${t}_pmc "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" > /dev/null || exit 1
bash tools/pmc_all.sh ${t}_cfg3 tools/bench_configs.py cfg3 > gpurun_out/${t}_cfg3_pmc.txt || exit 1
python tools/bench_configs.py > gpurun_out/${t}_other_configs.jsonl 2>/dev/null || exit 1
python tools/bench_gemm.py > gpurun_out/${t}_gemm.jsonl || exit 1
echo ALLOK
