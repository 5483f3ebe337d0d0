#!/bin/bash
# bf16 B=64 step under one-off plan overrides (op 16+n = plane call site n: 16 fwd L1, 17 fwd L2/3, 18 dH; 33 = grouped dW trio)
out=${1:-gpurun_out/r04/bf16_plan_sweep.txt}
for plan in ${PLANS:-"" "18:9" "18:10" "18:1" "18:13" "18:3" "17:9" "17:23" "16:22"}; do
  for rep in 1 2; do
    r=$(python bench.py --dtype bf16 --batch 64 --steps 200 --warmup 30 --no-cpu-baseline --no-other-dtypes ${plan:+--plan $plan} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d['roofline']['frac'])")
    echo "plan '${plan}' : ms_per_step gemm_ms frac = $r" | tee -a $out
  done
done
