# A/B of translation units built with other machine-scheduler strategies (-mllvm -amdgpu-sched-strategy=...):
#   bash tools/experiments/r03_sched_bench.sh base tconv_ilp head_ilp     (libraries tools/diag/libqot_<name>.so)
for v in "$@"; do
  if [ $v = base ]; then L=""; else L="QOT_LIB_PATH=tools/diag/libqot_$v.so"; fi
  env $L python bench.py --no-cpu-baseline --no-lightpath > gpurun_out/sched_$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/sched_$v.json')); print('$v', round(d['ms_per_step'],4), [(k['kernel'][:14], round(k['ms']*1e3,1)) for k in d['kernels'][:8]])"
done
