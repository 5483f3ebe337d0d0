#!/bin/bash
# Regenerate the per-round evidence under gpurun_out/<tag>/ (run on the GPU box through gpurun); tools/summarize_profiles.py condenses it
# into the tracked profiles/<tag>_* files (+ profiles/raw/<tag>_*: the raw rocprofv3 stats and a 200-row excerpt of every kernel trace).
#   bench.json / bench_driver_flags.json   python bench.py                 (value, roofline, cpu_baseline, fwd_only / as_loss legs)
#   stats*/                                rocprofv3 --kernel-trace --stats (per-kernel average durations), per compute type
#   pmc_*                                  separate --pmc passes            (FETCH_SIZE, WRITE_SIZE, MFMA busy; own runs, --kernel-trace only)
#   asloss_* / registration_*              the as-loss evaluation and the registration step (config 5): benches + kernel stats
#   dp_single_rank.txt, watchdog_*, bench_forced_dist_cfg4.json   the N > 1 skeleton on one GPU
# usage: tools/profile_round.sh r06 [quick]
set -u
TAG=${1:-r06}
QUICK=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
B="python $R/bench.py"
NOCPU="--no-cpu-baseline --no-live-pmc"
# the profiler passes must see the training step only: no spin-up GEMMs (bench.py: spinup_ms) in the traces
SHORT="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-other-dtypes --spinup-ms 0 --no-live-pmc"
LONG="--steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --spinup-ms 0 --no-live-pmc"
export TMPDIR=/tmp
cd /tmp
stats() { rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1 -o $TAG -- "${@:2}" > $OUT/$1.out 2>&1; }
pmc3() {      # three counter passes of one command: pmc3 <prefix> <cmd...>
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$1$c -o p -- "${@:2}" > /dev/null 2>&1
  done
  timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_$1MFMA -o p -- "${@:2}" > /dev/null 2>&1
}
stats stats $B $LONG
pmc3 "" $B $SHORT
# the PMC summary must exist in profiles/ BEFORE the bench line is emitted (bench.py reads roofline.traffic from it)
python $R/tools/summarize_profiles.py $TAG > /dev/null
$B > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
$B --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2>> $OUT/bench.err     # the command the driver runs
[ -n "$QUICK" ] && { python $R/tools/summarize_profiles.py $TAG; exit 0; }
# compute types on the bf16 matrix cores (same step, --dtype) and the B=64 configuration of BASELINE configs 3-4
for dt in f32x3 bf16; do
  $B --dtype $dt $NOCPU --no-other-dtypes > $OUT/bench_$dt.json 2>> $OUT/bench.err
  stats stats_$dt $B $LONG --dtype $dt
done
$B --dtype bf16 --batch 64 $NOCPU --no-other-dtypes > $OUT/bench_bf16_b64.json 2>> $OUT/bench.err
$B --batch 64 $NOCPU --no-other-dtypes > $OUT/bench_f32_b64.json 2>> $OUT/bench.err
stats stats_bf16_b64 $B $LONG --dtype bf16 --batch 64
pmc3 bf16_b64_ $B $SHORT --dtype bf16 --batch 64
{ for t in "stats f32_B32" "stats_f32x3 f32x3_B32" "stats_bf16_b64 bf16_B64"; do set -- $t; echo "== $2"; python $R/tools/step_timeline.py $OUT/$1/${TAG}_kernel_trace.csv; done; } > $OUT/step_timeline.txt 2>&1
# the step's remaining forms (each line: form, ms/step, GEMM ms/step)
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s ms_per_step %.4f  gemm_ms_per_step %s  value %.0f' % ('$1', d['ms_per_step'], (d.get('roofline') or {}).get('gemm_ms_per_step'), d['value']))"; }
{
  $B --steps 200 --warmup 30 $NOCPU --no-other-dtypes 2>/dev/null | line default
  $B --steps 200 --warmup 30 $NOCPU --no-other-dtypes 2>/dev/null | line default_again
  $B --steps 200 --warmup 30 $NOCPU --no-other-dtypes --prefetch 2>/dev/null | line prefetch
  $B --steps 200 --warmup 30 $NOCPU --no-other-dtypes --plan 4:33:0 2>/dev/null | line dw1_tail_split
  $B --steps 200 --warmup 30 $NOCPU --no-other-dtypes --dtype bf16 --batch 64 2>/dev/null | line bf16_b64
} > $OUT/variants.txt 2>&1
python $R/tools/gemm_bench.py --tiles 8,9,30,31,32,33 --splits 1,3 > $OUT/gemm_bench.txt 2>&1
python $R/tools/x3_bench.py 2 3 5 > $OUT/x3_bench.txt 2>&1
python $R/tools/determinism_check.py f32 > $OUT/determinism.txt 2>&1
# SURVEY 8(d)'s as-loss figures: the engine (default) at the PCRNet batch and at 32, the entry-by-entry node beside it, the backward's
# non-GEMM tail launch by launch, and the kernel stats of the evaluation
{ for b in 16 32; do for dt in f32 f32x3 bf16; do python $R/tools/asloss_bench.py --batch $b --dtype $dt | grep mode; done; done; } > $OUT/asloss_bench.txt 2>/dev/null
{ for b in 16 32; do for dt in f32 f32x3 bf16; do for e in 0 1; do python $R/tools/asloss_bench.py --batch $b --dtype $dt --engine $e | grep mode; done; done; done; } > $OUT/asloss_engine_ab.txt 2>/dev/null
{ python $R/tools/asloss_tail_bench.py --batch 16; python $R/tools/asloss_tail_bench.py --batch 32; } > $OUT/asloss_tail_bench.txt 2>/dev/null
for dt in f32 bf16; do stats stats_asloss_$dt python $R/tools/asloss_bench.py --batch 16 --dtype $dt --steps 100 --warmup 20; done
# config 5: the registration step in its forms (eager torch ... the captured default), the eager default's kernel stats, the demo
python $R/tools/registration_step_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/registration_step_bench.txt
# the same captured step with DPDist on the bf16 matrix cores (a numerics choice of the consumer: 5e-2 output tolerance; not the default)
{ for dt in f32x3 bf16; do echo "dpdist_dtype=$dt $(python $R/tools/registration_step_bench.py --forms graph --dtype $dt 2>&1 | grep '^graph')"; done; } > $OUT/registration_step_bench_dtypes.txt
stats stats_registration python $R/tools/registration_step_bench.py --forms eager_native --steps 50 --train-only
( cd $R; DPD_FORCE_DIST=1 MASTER_PORT=29535 timeout 900 python tools/registration_demo.py --loss ours 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > $OUT/registration_demo.txt )
python $R/tools/ramp_probe.py > $OUT/ramp_probe.txt 2>/dev/null
{ python $R/tools/host_rate.py f32 32; python $R/tools/host_rate.py bf16 64; } > $OUT/host_rate.txt 2>/dev/null
# the N > 1 skeleton on one GPU: every communication form on a single-rank group, the watchdog -> fallback path, the config-4 legs
{ for e in "DPD_FORCE_DIST=0" "DPD_FORCE_DIST=1" "DPD_FORCE_DIST=1 DPD_DP_BACKEND=torch" "DPD_FORCE_DIST=1 DPD_DP_SCHEDULE=late" "DPD_FORCE_DIST=1 DPD_DP_MODE=rs_ag" "DPD_FORCE_DIST=1 DPD_DP_WIRE=bf16" "DPD_FORCE_DIST=1 DPD_DP_MODE=zero1" "DPD_FORCE_DIST=1 DPD_DP_MODE=zero1 DPD_DP_BACKEND=torch"; do
    env $e MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 $B --steps 200 --warmup 30 $NOCPU --no-other-dtypes --no-roofline 2>/dev/null | tail -1 | line "$e" ; done; } > $OUT/dp_single_rank.txt 2>&1
( cd $R; DPD_FORCE_DIST=1 DPD_WD_INJECT_HANG=timed DPD_WD_LIMITS=timed=8 MASTER_PORT=29531 $B --steps 20 --warmup 5 $NOCPU --no-other-dtypes 2> $OUT/watchdog_fallback.err | tail -1 > $OUT/watchdog_fallback.json
  DPD_FORCE_DIST=1 MASTER_PORT=29533 $B --cfg4 --steps 50 --warmup 10 $NOCPU 2>/dev/null | tail -1 > $OUT/bench_forced_dist_cfg4.json )
python $R/tools/summarize_profiles.py $TAG
