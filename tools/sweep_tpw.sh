#!/bin/bash
# Sweep of the minimum tiles per workgroup of the weight-gradient producers (slab volume vs parallelism). Run on the GPU box.
mkdir -p gpurun_out/tpw
for a in 2 8 16 32; do for b in 2 8 32; do
  SMT_SHIFT_MIN_TPW=$a SMT_FUSED_MIN_TPW=$b python bench.py --no_ragged --no_micro --no_fp32 --no_graph --no_cpu_baseline --steps 8 --warmup 2 > gpurun_out/tpw/b_${a}_${b}.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/tpw/b_${a}_${b}.json').read().strip().splitlines()[-1])
k={x['name']:x for x in d['kernels']}
n=d.get('kernel_event_steps',2)
g=lambda nm: round(k[nm]['total_ms']/n,2) if nm in k else None
print('shift_tpw=${a} fused_tpw=${b}', round(d['ms_per_step'],2), 'reduce', g('conv_wgrad_reduce'), 'shift', g('conv_wgrad_shift'), '1x1bwd', g('conv1x1_bwd'), 'k1bwd', g('conv_k1_bwd'), 'gatebwd', g('conv_gate_bwd'))
PY
done; done
