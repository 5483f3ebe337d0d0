mkdir -p gpurun_out/r02
for plan in "" "32:12" "32:5" "32:9" "32:2" "18:9" "20:9" "21:9" "17:8" "16:9,17:9"; do
  echo "plan=$plan $(python bench.py --no-cpu-baseline --no-other-dtypes --no-roofline --dtype bf16 --batch 64 --plan "$plan" | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done
