#!/bin/bash
# bf16 B=64 training step (BASELINE config 3) under different plans of the backward weight-gradient GEMMs (round 4: in-launch split-K).
#   plan op 20 = dW1 of the plane compute types, 32 = grouped dW2 + dW3; tile:split (split > 1 in-launch, < -1 slabs + reduce launch)
#   DPD_RED_SC1=1 (default): slabs published by write-through stores; 0: plain stores + agent-scope release / acquire fences
out=${1:-gpurun_out/r04/bf16_bwd_sweep.txt}
PLANS=${PLANS:-"20:0:1,32:3:1 20:2:2,32:3:1 20:2:3,32:3:1 20:0:1,32:2:2 20:0:1,32:2:3 20:2:-2,32:3:1 20:2:-3,32:3:1 20:0:1,32:2:-2"}
for sc1 in ${SC1S:-1 0}; do
for plan in $PLANS; do
  for rep in 1 2; do
    r=$(DPD_RED_SC1=$sc1 python bench.py --dtype bf16 --batch 64 --steps 200 --warmup 30 --no-cpu-baseline --no-other-dtypes --plan "$plan" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d['roofline']['frac'])")
    echo "sc1=$sc1 plan $plan : ms_per_step gemm_ms frac = $r" | tee -a $out
  done
done
done
