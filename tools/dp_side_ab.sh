for e in "DPD_DP_ADAM_SIDE=0" "DPD_DP_ADAM_SIDE=1" "DPD_DP_ADAM_SIDE=0 DPD_DP_BACKEND=torch" "DPD_DP_ADAM_SIDE=1 DPD_DP_BACKEND=torch"; do
  for dtb in "f32 32" "bf16 64"; do set -- $dtb
    r=$(env $e DPD_FORCE_DIST=1 MASTER_PORT=29541 python bench.py --dtype $1 --batch $2 --steps 200 --warmup 30 --no-cpu-baseline --no-other-dtypes --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['dp']['exposed_comm_us_per_step'])")
    echo "$e $1 B=$2 : ms_per_step exposed_us = $r"
  done
done
