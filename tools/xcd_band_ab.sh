#!/bin/bash
# fp32 step at B=32: the XCD-blocked tile map of the register-streamed GEMMs, off / dW only / every GEMM
out=${1:-gpurun_out/r04/xcd_band_ab.txt}
for plan in "40:0" "40:1" "40:2" "40:0" "40:1"; do
  r=$(python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-other-dtypes --plan "$plan" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d['roofline']['frac'])")
  echo "plan $plan : ms_per_step gemm_ms frac = $r" | tee -a $out
done
