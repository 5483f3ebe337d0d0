#!/bin/bash
# the grouped dW1+dW2+dW3 launch in the other plane configurations: bf16 / f32x3 at B = 32 (and bf16 at B = 64 again), apart vs grouped
out=${1:-gpurun_out/r04/trio_other_configs.txt}
for cfg in "bf16 64" "bf16 32" "f32x3 32" "f32x3 64"; do
  set -- $cfg
  for trio in 0 1; do
    for rep in 1 2; do
      r=$(DPD_DW_TRIO=$trio python bench.py --dtype $1 --batch $2 --steps 200 --warmup 30 --no-cpu-baseline --no-other-dtypes 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d['roofline']['frac'])")
      echo "$1 B=$2 DPD_DW_TRIO=$trio : ms_per_step gemm_ms frac = $r" | tee -a $out
    done
  done
done
