#!/bin/bash
# Regenerate the per-round evidence under gpurun_out/<tag>/ (run on the GPU box through gpurun):
#   bench.json                     python bench.py                       (value, roofline, cpu_baseline)
#   stats/*_kernel_stats.csv       rocprofv3 --kernel-trace --stats      (per-kernel average durations)
#   pmc_FETCH_SIZE / pmc_WRITE_SIZE  separate --pmc passes               (HBM/fabric bytes per launch)
#   pmc_MFMA                       MFMA-busy cycles per kernel           (own pass, --kernel-trace only)
#   variants.txt                   the opt-in step variants, measured     (prefetch / fused gather / hipGraph / tail split ...)
# usage: tools/profile_round.sh r03 [quick]
set -u
TAG=${1:-r03}
QUICK=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
B="python $R/bench.py"
NOCPU="--no-cpu-baseline"
SHORT="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-other-dtypes"
export TMPDIR=/tmp
# the profiler passes must see the training step only: no spin-up GEMMs (bench.py: spinup_ms) in the traces
PROF="rocprofv3"
NOSPIN="--spinup-ms 0"
cd /tmp
$PROF --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- $B --steps 20 --warmup 5 $NOCPU --no-other-dtypes > $OUT/bench_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 240 $PROF --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- $B $SHORT > /dev/null 2>&1
done
timeout 240 $PROF --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_MFMA -o p -- $B $SHORT > /dev/null 2>&1
# the PMC summary must exist in profiles/ BEFORE the bench line is emitted (bench.py reads roofline.traffic from it)
python $R/tools/summarize_profiles.py $TAG > /dev/null
$B > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
$B --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2>> $OUT/bench.err     # the command the driver runs
[ -n "$QUICK" ] && exit 0
# compute types on the bf16 matrix cores (same step, --dtype) and the B=64 configuration of BASELINE configs 3-4
for dt in f32x3 bf16; do
  $B --dtype $dt $NOCPU --no-other-dtypes > $OUT/bench_$dt.json 2>> $OUT/bench.err
done
$B --dtype bf16 --batch 64 $NOCPU --no-other-dtypes > $OUT/bench_bf16_b64.json 2>> $OUT/bench.err
$B --batch 64 $NOCPU --no-other-dtypes > $OUT/bench_f32_b64.json 2>> $OUT/bench.err
for dt in f32x3 bf16; do
  $PROF --kernel-trace --stats --output-format csv -d $OUT/stats_$dt -o $TAG -- $B --steps 20 --warmup 5 $NOCPU --no-other-dtypes --dtype $dt > /dev/null 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 240 $PROF --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${dt}_$c -o p -- $B $SHORT --dtype $dt > /dev/null 2>&1
  done
  timeout 240 $PROF --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_${dt}_MFMA -o p -- $B $SHORT --dtype $dt > /dev/null 2>&1
done
$PROF --kernel-trace --stats --output-format csv -d $OUT/stats_bf16_b64 -o $TAG -- $B --steps 20 --warmup 5 $NOCPU --no-other-dtypes --dtype bf16 --batch 64 > /dev/null 2>&1
# BASELINE config 3 (bf16, 64 pairs): MFMA-busy and fabric bytes of its own launches, and the per-launch timeline of the three steps
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 240 $PROF --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_bf16_b64_$c -o p -- $B $SHORT --dtype bf16 --batch 64 > /dev/null 2>&1
done
timeout 240 $PROF --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_bf16_b64_MFMA -o p -- $B $SHORT --dtype bf16 --batch 64 > /dev/null 2>&1
{ for t in "stats f32_B32" "stats_f32x3 f32x3_B32" "stats_bf16_b64 bf16_B64"; do set -- $t; echo "== $2"; python $R/tools/step_timeline.py $OUT/$1/${TAG}_kernel_trace.csv; done; } > $OUT/step_timeline.txt 2>&1
# opt-in variants of the same step (each line: variant, ms/step, GEMM ms/step)
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s ms_per_step %.4f  gemm_ms_per_step %s  value %.0f' % ('$1', d['ms_per_step'], (d.get('roofline') or {}).get('gemm_ms_per_step'), d['value']))"; }
{
  $B --steps 200 --warmup 30 $NOCPU --no-other-dtypes 2>/dev/null | line default
  $B --steps 200 --warmup 30 $NOCPU --no-other-dtypes 2>/dev/null | line default_again
  $B --steps 200 --warmup 30 $NOCPU --no-other-dtypes --prefetch 2>/dev/null | line prefetch
  DPD_FUSED_GATHER=1 $B --steps 200 --warmup 30 $NOCPU --no-other-dtypes 2>/dev/null | line fused_gather
  DPD_GRAPH=1 $B --steps 200 --warmup 30 $NOCPU --no-other-dtypes 2>/dev/null | line hipgraph
  DPD_FUSE_LOSS=0 $B --steps 200 --warmup 30 $NOCPU --no-other-dtypes 2>/dev/null | line separate_l1_loss
  DPD_DET_DB=0 $B --steps 200 --warmup 30 $NOCPU --no-other-dtypes 2>/dev/null | line atomic_bias_grads
  $B --steps 200 --warmup 30 $NOCPU --no-other-dtypes --plan 8:30:0 2>/dev/null | line dw1_tail_split
} > $OUT/variants.txt 2>&1
python $R/tools/gemm_bench.py --tiles 8,9,30,31,32,33 --splits 1,3 > $OUT/gemm_bench.txt 2>&1
python $R/tools/x3_bench.py 2 3 5 > $OUT/x3_bench.txt 2>&1
{ B64=1 python $R/tools/x3_bench.py 2 9 20 21 23; NP=3 python $R/tools/p8_probe.py 2 3 24 25; } 2>&1 | grep -v 'rc=-3' > $OUT/x3_bench_p8.txt
{ python $R/tools/gather_bench.py 64; python $R/tools/gather_bench.py 32; DPD_GATHER_PLANES_V1=1 python $R/tools/gather_bench.py 64; } > $OUT/gather_bench.txt 2>&1
python $R/tools/event_cost.py > $OUT/event_cost.txt 2>/dev/null
hipcc --offload-arch=gfx950 -O3 $R/tools/launch_floor.hip -o /tmp/launch_floor 2>/dev/null && /tmp/launch_floor > $OUT/launch_floor.txt 2>&1
python $R/tools/determinism_check.py f32 > $OUT/determinism.txt 2>&1
{ for b in 16 32; do for dt in f32 f32x3 bf16; do python $R/tools/asloss_bench.py --batch $b --dtype $dt | tail -1; done; done; } > $OUT/asloss_bench.txt 2>/dev/null
# round 5: the chained persistent decoder launches against the separate launches (A/B + stamps), the as-loss engine on / off, the
# registration loop's step rate with and without it
{ python $R/tools/chain_bench.py 64 100; python $R/tools/chain_bench.py 32 100; } 2>&1 | grep -v amdgpu.ids > $OUT/chain_bench.txt
{ for b in 16 32; do for dt in f32 f32x3 bf16; do for e in 0 1; do DPD_ASLOSS_ENGINE=$e python $R/tools/asloss_bench.py --batch $b --dtype $dt | grep mode; done; done; done; } > $OUT/asloss_engine_ab.txt 2>/dev/null
{ for e in 0 1; do echo "== DPD_ASLOSS_ENGINE=$e"; ( cd $R; DPD_ASLOSS_ENGINE=$e timeout 600 python tools/registration_demo.py --loss ours --dp_steps 1500 --reg_steps 1500 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['pcrnet_ours']; print(json.dumps({k: d[k] for k in ('train_s','host_gen_s','steps','pairs_per_s','eval_loss','rot_err_median_deg') if k in d}), 'gpu+host ms per registration step (7 pose-only refinements + 1 DPDist forward/backward):', round((d['train_s']-d['host_gen_s'])/d['steps']*1e3,3))" ); done; } > $OUT/registration_engine_ab.txt 2>&1
[ "${SWEEPS:-1}" = "0" ] || {
# round 4: the bf16 backward -- weight gradients apart / grouped / split-K in the launch, dH || dW on two streams, tile plans
( cd $R; tools/bf16_trio_sweep.sh $OUT/bf16_trio_sweep.txt; PLANS="20:0:1,32:3:1 20:2:2,32:3:1 20:2:3,32:3:1 20:0:1,32:2:2 20:2:-3,32:3:1" SC1S="1 0" DPD_DW_TRIO=0 tools/bf16_bwd_sweep.sh $OUT/bf16_bwd_sweep.txt; tools/bf16_plan_sweep.sh $OUT/bf16_plan_sweep.txt; tools/trio_b32.sh $OUT/trio_other_configs.txt; tools/xcd_band_ab.sh $OUT/xcd_band_ab.txt ) > /dev/null 2>&1
python $R/tools/overlap_probe_bf16.py > $OUT/overlap_probe_bf16.txt 2>/dev/null
}
python $R/tools/ramp_probe.py > $OUT/ramp_probe.txt 2>/dev/null
{ python $R/tools/host_rate.py f32 32; python $R/tools/host_rate.py bf16 64; } > $OUT/host_rate.txt 2>/dev/null
hipcc --offload-arch=gfx950 -O3 $R/tools/ldsdma_bw.hip -o /tmp/ldsdma_bw 2>/dev/null && /tmp/ldsdma_bw > $OUT/ldsdma_bw.txt 2>&1
{ for e in "DPD_FORCE_DIST=0" "DPD_FORCE_DIST=1" "DPD_FORCE_DIST=1 DPD_DP_BACKEND=torch" "DPD_FORCE_DIST=1 DPD_DP_BUCKETS=3" "DPD_FORCE_DIST=1 DPD_DP_SCHEDULE=late" "DPD_FORCE_DIST=1 DPD_DP_MODE=rs_ag" "DPD_FORCE_DIST=1 DPD_DP_WIRE=bf16" "DPD_FORCE_DIST=1 DPD_DP_TWO_COMMS=1" "DPD_FORCE_DIST=1 DPD_DP_MODE=zero1" "DPD_FORCE_DIST=1 DPD_DP_MODE=zero1 DPD_DP_BACKEND=torch"; do
    env $e MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 $B --steps 200 --warmup 30 $NOCPU --no-other-dtypes --no-roofline 2>/dev/null | tail -1 | line "$e" ; done; } > $OUT/dp_single_rank.txt 2>&1
# the N > 1 skeleton on one GPU: watchdog -> fallback (injected hang), and the healthy forced-distributed line with its `dp` object
( cd $R; DPD_FORCE_DIST=1 DPD_WD_INJECT_HANG=timed DPD_WD_LIMITS=timed=8 MASTER_PORT=29531 $B --steps 20 --warmup 5 $NOCPU --no-other-dtypes 2> $OUT/watchdog_fallback.err | tail -1 > $OUT/watchdog_fallback.json
  DPD_FORCE_DIST=1 MASTER_PORT=29533 $B --cfg4 --steps 50 --warmup 10 $NOCPU 2>/dev/null | tail -1 > $OUT/bench_forced_dist_cfg4.json )
# row f2 / config 5 with the data-parallel plumbing on (single rank): DPDist trained, frozen, pose network registered
( cd $R; DPD_FORCE_DIST=1 MASTER_PORT=29535 timeout 900 python tools/registration_demo.py --loss ours 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > $OUT/registration_demo.txt )
python $R/tools/summarize_profiles.py $TAG
