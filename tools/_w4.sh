export DPD_TEST_SHARE_GPU=1 DPD_DP_SELECT=2,1,2 DPD_WD_DIR_BASE=/tmp/wd4
mkdir -p /tmp/wd4
( while true; do sleep 5; echo "$(date +%s) $(cat /tmp/wd4/rank0.a1 2>/dev/null | cut -d' ' -f1) | $(cat /tmp/wd4/rank2.a1 2>/dev/null | cut -d' ' -f1)"; done ) > gpurun_out/w4_phases.txt 2>&1 &
MON=$!
timeout 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/w4.out 2> gpurun_out/w4.err
echo rc=$?
kill $MON
uniq -c -f1 gpurun_out/w4_phases.txt | tail -40
