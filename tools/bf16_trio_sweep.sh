#!/bin/bash
# bf16 B=64 step: the three weight gradients apart (round 3) vs in one grouped launch (round 4), by tile and in-launch split-K
out=${1:-gpurun_out/r04/bf16_trio_sweep.txt}
run() {
  for rep in 1 2; do
    r=$(env "$@" python bench.py --dtype bf16 --batch 64 --steps 200 --warmup 30 --no-cpu-baseline --no-other-dtypes --plan "$PLAN" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d['roofline']['frac'])")
    echo "$* plan $PLAN : ms_per_step gemm_ms frac = $r" | tee -a $out
  done
}
PLAN="20:0:1,32:3:1" run DPD_DW_TRIO=0
for PLAN in ${PLANS:-"33:2:1" "33:13:1" "33:14:1" "33:3:1"}; do run DPD_DW_TRIO=1; done
