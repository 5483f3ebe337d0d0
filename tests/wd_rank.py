"""One rank of the watchdog -> fallback test (started by tests/test_ddp_gloo.py through bench.spawn_ranks).  Same skeleton as
bench.py's N > 1 run -- supervisor first, heartbeats per phase, make_reducer with its start-up cross-check, the JSON goes out before
the tear-down -- on gloo / CPU tensors so that it runs without a GPU.  DPD_WD_INJECT_HANG=<phase> (dpdist_amd/launch.py) makes the
worker of attempt 1 stop in that phase like a rank that never enters the next collective."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dpdist_amd import launch  # noqa: E402


def main(out_path):
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    launch.maybe_supervise(world)            # returns only in the worker
    hb = launch.Heartbeat(rank)
    hb.beat("start:import")
    import torch
    import torch.distributed as dist
    hb.beat("init:process group")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hb.beat("reducer:make + cross-check")
    from dpdist_amd.ddp import make_reducer
    flat = torch.zeros(4096)
    red = make_reducer(flat, [0, 1024, 2048, 4096])
    hb.beat("timed:all-reduce of 4096 floats")      # (phase texts may contain spaces)
    flat.fill_(rank + 1.0)
    red.reduce_async(1, upto=2)
    red.reduce_async(0)
    red.wait()
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"world": world, "sum": float(flat[0]), "all_equal": bool((flat == flat[0]).all()), "fallback": hb.fallback,
                       "attempt": hb.attempt, "backend_env": os.environ.get("DPD_DP_BACKEND"), "crosscheck": red.crosscheck,
                       "history": json.loads(os.environ.get("DPD_WD_HISTORY", "[]"))}, f)
    hb.beat("done")
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
