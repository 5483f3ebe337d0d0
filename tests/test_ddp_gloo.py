"""Data-parallel path on CPU: world_size 2 over gloo.  The bucketed all-reduce + 1/world scaling of ddp.py must
reproduce the full-batch gradient (mean of equal shard means), the role of average_gradients in the reference
(train_multi_gpu_pc_compare_dist.py:936-974) -- for the fp32 and the bf16 wire, as one all-reduce per bucket and as
reduce-scatter + all-gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dpdist_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat_grad(P, pcA, pcB, lab, mlp):
    """per-shard gradient of loss_samples from the oracle, laid out in the product's flat/bucket layout"""
    from oracle import restate as R
    W = R.as_torch_weights(synth.make_weights("wide", mlp=mlp), torch.float64, requires_grad=True)
    pred, _ = R.get_model(torch.tensor(pcA, dtype=torch.float64), torch.tensor(pcB, dtype=torch.float64), W)
    ls, _ = R.get_loss(pred, torch.tensor(lab, dtype=torch.float64))
    names = sorted(W)
    g = torch.autograd.grad(ls, [W[n] for n in names])
    P.load_tf_state_dict({n: gg.numpy() for n, gg in zip(names, g)})    # reuse the layout code: grads -> flat
    return P.flat.detach().clone()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from dpdist_amd.ddp import BucketReducer, shard_range
    from dpdist_amd.model import DPDistParams
    mlp = (64, 64, 64)
    GB = 4
    pcA, pcB, lab = synth.s2_modelnet_shaped(GB, 64, 100)
    lo, hi = shard_range(GB, rank, world)
    P = DPDistParams(k=5, mlp=mlp, device="cpu", init=None)
    shard = _flat_grad(P, pcA[lo:hi], pcB[lo:hi], lab[lo:hi], mlp).float()
    full = _flat_grad(P, pcA, pcB, lab, mlp).float() if rank == 0 else None
    res = {}
    for wire, mode in (("f32", "allreduce"), ("f32", "rs_ag"), ("bf16", "allreduce"), ("bf16", "rs_ag")):
        flat = shard.clone()
        red = BucketReducer(flat, P.bucket_bounds, wire=wire, mode=mode)
        assert len(P.bucket_bounds) == 4
        for b in (2, 1, 0):          # layers 3-4, layer 2, layer 1: the order trainer.backward produces them in
            red.reduce_async(b)
        red.wait()
        flat *= red.grad_scale
        if rank == 0:
            res[(wire, mode)] = (float((flat - full).abs().max()), float(full.abs().max()), red.world)
    if rank == 0:
        out.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_equals_full_batch_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for (wire, mode), (err, scale, world) in res.items():
        assert world == 2
        # fp32 wire: summation order only; bf16 wire: every addend and the sum are rounded to 8 bits of mantissa
        tol = 1e-6 * max(1.0, scale) if wire == "f32" else 2.0 ** -7 * scale
        assert err <= tol, (wire, mode, err, scale)


def test_shard_range():
    from dpdist_amd.ddp import shard_range
    assert [shard_range(512, r, 8) for r in (0, 7)] == [(0, 64), (448, 512)]
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)


_RANK_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
t = torch.tensor([float(os.environ["LOCAL_RANK"]) + 1.0])
dist.all_reduce(t)
if dist.get_rank() == 0:
    open(sys.argv[1], "w").write("%d %.1f" % (dist.get_world_size(), t.item()))
dist.destroy_process_group()
if os.environ["RANK"] == sys.argv[2]:
    sys.exit(7)
"""


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` with no launcher: bench.spawn_ranks starts N processes with the torchrun environment
    contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* on 127.0.0.1), waits for them and returns the first failure."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out = tmp_path / "r0.txt"
    assert bench.spawn_ranks(2, [sys.executable, "-c", _RANK_SCRIPT, str(out), "-1"]) == 0
    assert out.read_text() == "2 3.0"
    assert bench.spawn_ranks(2, [sys.executable, "-c", _RANK_SCRIPT, str(out), "1"]) == 7


def test_make_reducer_keeps_torch_distributed_off_the_gpu():
    """ddp.make_reducer: the direct librccl reducer needs an RCCL process group and a GPU buffer; anything else (gloo, CPU tensors,
    no process group) gets the torch.distributed BucketReducer."""
    import torch
    from dpdist_amd.ddp import BucketReducer, make_reducer
    r = make_reducer(torch.zeros(16), [0, 4, 8, 16])
    assert isinstance(r, BucketReducer) and not r.active


def _run_wd(tmp_path, monkeypatch, **env):
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out = tmp_path / "wd.json"
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("DPD_WD_LIMITS", "start=120,init=60,reducer=60,timed=4,done=10")
    monkeypatch.delenv("DPD_BENCH_CHILD", raising=False)
    rc = bench.spawn_ranks(2, [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "wd_rank.py"), str(out)], timeout_s=300)
    return rc, (json.loads(out.read_text()) if out.exists() else None)


def test_watchdog_passes_a_healthy_run_through(tmp_path, monkeypatch):
    """dpdist_amd/launch.py: every rank supervises its worker; a healthy run is attempt 1, no fallback, and the reducer's start-up
    cross-check (known integer pattern through the reducer and through a plain all-reduce, bitwise) is on the record."""
    rc, res = _run_wd(tmp_path, monkeypatch)
    assert rc == 0 and res is not None
    assert res["world"] == 2 and res["sum"] == 3.0 and res["all_equal"]
    assert res["fallback"] is False and res["attempt"] == 1 and res["history"] == []
    assert res["crosscheck"]["ok"] and res["crosscheck"]["reducer_bitwise"] and res["crosscheck"]["torch_all_reduce_bitwise"]


@pytest.mark.parametrize("who", ["1", "all"])
def test_watchdog_falls_back_after_a_hang(tmp_path, monkeypatch, who):
    """A rank that stops before a collective (injected: the worker of attempt 1 never leaves phase `timed`) hangs every other rank
    in it.  The supervisors notice the missing heartbeat, stop the workers by PID and run the launch once more with
    DPD_DP_BACKEND=torch on a fresh rendezvous; rank 0's line says so (`fallback`, the failure of attempt 1)."""
    env = {"DPD_WD_INJECT_HANG": "timed"}
    if who != "all":
        env["DPD_WD_INJECT_RANK"] = who
    rc, res = _run_wd(tmp_path, monkeypatch, **env)
    assert rc == 0 and res is not None
    assert res["world"] == 2 and res["sum"] == 3.0 and res["all_equal"]
    assert res["fallback"] is True and res["attempt"] == 2 and res["backend_env"] == "torch"
    assert len(res["history"]) == 1 and "no heartbeat" in res["history"][0]["failure"] and "timed" in res["history"][0]["failure"]


def test_watchdog_gives_up_after_the_second_failure(tmp_path, monkeypatch, capfd):
    """Both attempts fail (the worker dies at once): exit code 3 and rank 0's supervisor prints a JSON line with value null."""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    monkeypatch.delenv("DPD_BENCH_CHILD", raising=False)
    script = tmp_path / "dies.py"          # (a file: the supervisor re-runs sys.argv, which `python -c` does not preserve)
    script.write_text("import os, sys; sys.path.insert(0, %r); from dpdist_amd import launch; launch.maybe_supervise(2); sys.exit(5)\n"
                      % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    rc = bench.spawn_ranks(2, [sys.executable, str(script)], timeout_s=120)
    assert rc == 3
    line = [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["value"] is None and len(rec["watchdog"]) == 2 and "exited with code 5" in rec["watchdog"][0]["failure"]


def _zero_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dpdist_amd.ddp import BucketReducer, crosscheck
    n = 4 * 1000 + 12                      # not a multiple of 4 * world in every bucket: replicated remainders exist
    bounds = [0, 2004, 3008, n]
    b1, b2, eps, lr = 0.9, 0.999, 1e-8, 1e-3

    def grad(step, r):                     # this rank's gradient of step `step`
        g = torch.Generator().manual_seed(1000 * step + r)
        return torch.randn(n, generator=g)

    def adam(p, g, m, v, t, scale):        # TF-form Adam, elementwise (the role of dpd_adam_tf on a range)
        g = g * scale
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        lr_t = lr * (1 - b2 ** t) ** 0.5 / (1 - b1 ** t)
        p.sub_(lr_t * m / (v.sqrt() + eps))

    res = {}
    for mode in ("allreduce", "zero1"):
        p = torch.linspace(-1, 1, n)
        m, v = torch.zeros(n), torch.zeros(n)
        flat = torch.zeros(n)
        red = BucketReducer(flat, bounds, mode=mode)
        chk = crosscheck(red)
        for t in (1, 2, 3):
            flat.copy_(grad(t, rank))
            red.reduce_async(1, upto=2)
            red.reduce_async(0)
            red.wait()
            if mode == "zero1":
                own = red.owned_ranges()
                for lo, hi in own:
                    adam(p[lo:hi], flat[lo:hi], m[lo:hi], v[lo:hi], t, red.grad_scale)
                red.gather_params(p)
            else:
                adam(p, flat, m, v, t, red.grad_scale)
        if mode == "zero1":                # the Adam slots are gathered on demand (checkpoints)
            red.gather_params(m)
            red.gather_params(v)
        res[mode] = (p.clone(), m.clone(), v.clone(), chk, red.wire_bytes_per_step,
                     sum(hi - lo for lo, hi in red.owned_ranges()) if mode == "zero1" else n)
    if rank == 0:
        a, z = res["allreduce"], res["zero1"]
        out.put({"p": bool(torch.equal(a[0], z[0])), "m": bool(torch.equal(a[1], z[1])), "v": bool(torch.equal(a[2], z[2])),
                 "chk": [a[3], z[3]], "wire": [a[4], z[4]], "owned": z[5], "n": n})
    dist.barrier()
    dist.destroy_process_group()


def test_zero1_sharded_optimizer_is_bitwise_the_replicated_one():
    """DPD_DP_MODE=zero1 (ddp.py): reduce-scatter -> Adam on this rank's shards (+ the replicated remainders) -> all-gather of the
    fp32 parameters gives, after three steps, bit for bit the parameters AND Adam slots of the replicated optimizer behind an
    all-reduce (role of `average_gradients` + `apply_gradients`, train_multi_gpu_pc_compare_dist.py:936-990); each rank updates
    about half of the elements; both reducers pass the start-up cross-check; same bytes on the wire."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_zero_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res["p"] and res["m"] and res["v"]
    assert all(c["ok"] for c in res["chk"]) and res["chk"][1]["mode"] == "zero1"
    assert res["wire"][0] == res["wire"][1] == 4 * res["n"]          # 2 (P-1)/P x 4 B x n at P = 2
    assert res["n"] // 2 <= res["owned"] <= res["n"] // 2 + 3 * 8     # half of every bucket + remainders of < 4 * world elements


def test_zero_partition_and_wire_bytes():
    """ddp.zero_partition: `world` equal shards of a multiple of four elements + a replicated remainder of < 4 * world; the shards and
    remainders of the real bucket layout cover every parameter exactly once over the ranks; ddp._wire_bytes = 2 (P - 1) / P x payload."""
    from dpdist_amd.ddp import BucketReducer, _wire_bytes, zero_partition
    from dpdist_amd.model import DPDistParams
    for world in (1, 2, 3, 8):
        for lo, hi in ((0, 2589696), (2589696, 3639296), (3639296, 4692996), (0, 5), (8, 8 + 4 * world + 3)):
            main, shard = zero_partition(lo, hi, world)
            assert main % (4 * world) == 0 and shard * world == main and 0 <= (hi - lo) - main < 4 * world
    P = DPDistParams(device="cpu", init=None)
    world = 8
    seen = torch.zeros(P.numel, dtype=torch.int32)
    for rank in range(world):
        red = BucketReducer(torch.zeros(P.numel), P.bucket_bounds, mode="zero1")
        red.world, red.rank = world, rank                       # (no process group: the partition is pure arithmetic)
        red._calls = [(P.bucket_bounds[1], P.bucket_bounds[3]), (P.bucket_bounds[0], P.bucket_bounds[1])]
        for lo, hi in red.owned_ranges():
            seen[lo:hi] += 1
    lo2, hi2 = P.bucket_bounds[1], P.bucket_bounds[3]
    main2, _ = zero_partition(lo2, hi2, world)
    expect = torch.ones(P.numel, dtype=torch.int32)
    expect[lo2 + main2:hi2] = world                              # the replicated remainder is updated on every rank
    main0, _ = zero_partition(P.bucket_bounds[0], P.bucket_bounds[1], world)
    expect[P.bucket_bounds[0] + main0:P.bucket_bounds[1]] = world
    assert torch.equal(seen, expect)
    assert _wire_bytes([(0, 1000)], 8, "f32", "allreduce") == int(2 * 7 / 8 * 4000)
    assert _wire_bytes([(0, 1000)], 8, "bf16", "allreduce") == int(2 * 7 / 8 * 2000)
    assert _wire_bytes([(0, 1000)], 1, "f32", "zero1") == 0


def test_watchdog_limits_and_heartbeat_file(tmp_path, monkeypatch):
    """launch.limits honours DPD_WD_LIMITS; a heartbeat's phase text may contain spaces (the supervisor reads '<phase> <time>')."""
    from dpdist_amd import launch
    monkeypatch.setenv("DPD_WD_LIMITS", "timed=7, init=11")
    lim = launch.limits()
    assert lim["timed"] == 7.0 and lim["init"] == 11.0 and lim["start"] == launch.LIMITS["start"]
    monkeypatch.setenv("DPD_WD_DIR", str(tmp_path))
    monkeypatch.delenv("DPD_WD_INJECT_HANG", raising=False)
    hb = launch.Heartbeat(rank=3)
    hb.beat("timed:20 steps of the headline")
    phase, t = launch._read_beat(str(tmp_path / "rank3.a1"))
    assert phase == "timed:20_steps_of_the_headline" and abs(t - __import__("time").time()) < 5
    assert launch._read_beat(str(tmp_path / "missing")) == (None, None)


def _sched_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dpdist_amd.ddp import select_schedule
    calls = []
    # the ranks DISAGREE locally: rank 0 measures "early" fastest, rank 1 "grouped"; a step is as slow as its slowest rank
    local = {0: {"early": 0.30, "grouped": 0.33, "late": 0.40}, 1: {"early": 0.36, "grouped": 0.31, "late": 0.41}}[rank]

    def time_fn(name):
        calls.append(name)
        dist.barrier()                      # (a real candidate runs steps with collectives: every rank must be inside the same one)
        return local[name]

    choice, table = select_schedule(["early", "grouped", "late"], time_fn, torch.device("cpu"))
    # a candidate ONE rank cannot run is dropped on every rank BEFORE anything is timed (a rank that raised in the middle of a candidate
    # would leave its peers inside a collective)
    calls2 = []

    def timed2(name):
        calls2.append(name)
        dist.barrier()
        return local[name]
    choice2, table2 = select_schedule(["early", "grouped"], timed2, torch.device("cpu"), supported=lambda n: not (n == "grouped" and rank == 1))
    assert calls2 == ["early"]
    # an exception inside time_fn is not swallowed (here: raised on every rank at the same point, so nobody is left behind)
    try:
        select_schedule(["early"], lambda n: (_ for _ in ()).throw(RuntimeError("boom")), torch.device("cpu"))
        raised = False
    except RuntimeError as e:
        raised = "boom" in str(e)
    assert raised
    # a tie goes to the earlier candidate on every rank
    choice3, _ = select_schedule(["a", "b"], lambda n: (dist.barrier(), 1.0)[1], torch.device("cpu"))
    out.put((rank, choice, table, calls, choice2, table2, choice3))
    dist.barrier()
    dist.destroy_process_group()


def test_schedule_selection_is_the_same_on_every_rank():
    """ddp.select_schedule (the start-up choice of the data-parallel backward order, DPDistTrainer.select_dp_schedule): both ranks time
    every candidate in the same order, the MAX over ranks decides, and both ranks take the same winner although each one alone would
    have chosen differently."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sched_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, choice, table, calls, choice2, table2, choice3 in got:
        assert calls == ["early", "grouped", "late"]
        assert choice == "grouped"                      # max(0.33, 0.31) = 0.33 < max(0.30, 0.36) = 0.36
        assert table == {"early": 0.36, "grouped": 0.33, "late": 0.41}
        assert choice2 == "early" and table2 == {"early": 0.36, "grouped": None}
        assert choice3 == "a"
