# A/B of the H = 64 NNConv kernels built with other machine-scheduler strategies (-mllvm -amdgpu-sched-strategy=...)
for v in base max-ilp max-memory-clause; do
  if [ $v = base ]; then L=""; else L="QOT_LIB_PATH=tools/diag/libqot_$v.so"; fi
  env $L python bench.py --no-cpu-baseline --no-lightpath > gpurun_out/sched_$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/sched_$v.json')); print('$v', round(d['ms_per_step'],4), [(k['kernel'][7:18], round(k['ms']*1e3,1)) for k in d['kernels'][:3]])"
done
