"""Fail-safe launch of the data-parallel ranks: a per-phase watchdog around every rank and ONE retry on the conservative
collective backend.

Why: the reference's towers live in one process and cannot hang on each other (`train_multi_gpu_pc_compare_dist.py:237-302`,
`average_gradients` `:936-974` is a CPU-side stack + mean).  One process per GPU over RCCL can: a collective that one rank never
enters blocks every other rank for ever, and the first time the N > 1 path of this repository meets a second GPU is a driver run
nobody watches.  So every rank process is a *supervisor* that starts the real worker (same command line, `DPD_BENCH_CHILD=1`) and
reads the worker's heartbeat file:

    worker:      hb = Heartbeat(); hb.beat("init"); ...; hb.beat("timed"); ...; hb.beat("done")
    supervisor:  time since the last beat > limit of that phase  ->  create <dir>/fail.<attempt>, kill the worker BY PID
                 any supervisor that sees fail.<attempt>         ->  kill its worker, go to attempt 2
                 attempt 2 = the same command with DPD_DP_BACKEND=torch (torch.distributed collectives instead of the direct
                 librccl reducer), a fresh rendezvous port chosen by rank 0's supervisor, DPD_BENCH_FALLBACK=1

The supervisor imports neither torch nor the library (it must start in milliseconds and cannot hang in a GPU runtime).  No
process is ever stopped by pattern: only PIDs this module started.  Single node only (the heartbeat directory is in /tmp), which
is what `bench.py --gpus N` and `tools/registration_demo.py --gpus N` are specified for.

Knobs (environment): DPD_WD=0 turns the supervisor off; DPD_WD_LIMITS="init=60,timed=30" overrides limits (seconds);
DPD_WD_INJECT_HANG=<phase> makes the worker of attempt 1 stop beating (and sleep) when it reaches that phase (test hook:
tests/test_ddp_gloo.py drives watchdog -> fallback with it); DPD_WD_INJECT_RANK restricts the injected hang to one rank.
"""
import json
import os
import signal
import subprocess
import sys
import time

# seconds since the last heartbeat, by phase prefix (the part before ':').  "start" covers interpreter start + `import torch` on a
# fresh box (1-2 minutes while the image pages in); the others are generous multiples of what the phases take on one GPU.
LIMITS = {"start": 420.0, "init": 180.0, "reducer": 150.0, "crosscheck": 90.0, "aux": 200.0, "spinup": 60.0, "warmup": 90.0,
          "timed": 120.0, "profile": 150.0, "report": 200.0, "done": 30.0, "*": 180.0}


def limits():
    lim = dict(LIMITS)
    for item in filter(None, os.environ.get("DPD_WD_LIMITS", "").split(",")):
        k, v = item.split("=")
        lim[k.strip()] = float(v)
    return lim


def _write_atomic(path, text):
    tmp = "%s.%d.tmp" % (path, os.getpid())
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)


class Heartbeat:
    """Worker side.  Without a supervisor (DPD_WD_DIR unset) every call is a no-op."""

    def __init__(self, rank=None):
        self.dir = os.environ.get("DPD_WD_DIR")
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.attempt = int(os.environ.get("DPD_WD_ATTEMPT", "1"))
        self.fallback = os.environ.get("DPD_BENCH_FALLBACK") == "1"
        self._hang = os.environ.get("DPD_WD_INJECT_HANG") if self.attempt == 1 else None
        hr = os.environ.get("DPD_WD_INJECT_RANK")
        if hr is not None and int(hr) != self.rank:
            self._hang = None
        self.phase = None

    def beat(self, phase):
        self.phase = phase
        if self.dir:
            _write_atomic(os.path.join(self.dir, "rank%d.a%d" % (self.rank, self.attempt)), "%s %.3f\n" % ("_".join(phase.split()), time.time()))
        if self._hang and phase.split(":")[0] == self._hang:
            sys.stderr.write("launch.Heartbeat: injected hang in phase %r on rank %d (attempt %d)\n" % (phase, self.rank, self.attempt))
            sys.stderr.flush()
            while True:               # a rank that never enters the next collective: exactly what a lost RCCL rank looks like
                time.sleep(3600)


def _read_beat(path):
    try:
        txt = open(path).read().rsplit(None, 1)      # "<phase> <unix time>": the phase is everything before the last field
        return txt[0].strip(), float(txt[1])
    except (OSError, IndexError, ValueError):
        return None, None


def _stop(child, grace=3.0):
    """terminate -> kill, by PID (the worker may sit in a GPU runtime call that ignores SIGTERM)."""
    if child.poll() is not None:
        return
    try:
        child.terminate()
        t0 = time.time()
        while child.poll() is None and time.time() - t0 < grace:
            time.sleep(0.05)
        if child.poll() is None:
            child.kill()
        child.wait(timeout=10)
    except Exception:
        pass


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def wd_dir():
    """One directory per launch: every rank of a launch has the same parent (torchrun's agent: a fresh process per launch; bench.
    spawn_ranks passes a fresh temporary directory as DPD_WD_DIR_BASE, because one parent may launch several times)."""
    d = os.environ.get("DPD_WD_DIR_BASE")
    if d is None:
        d = os.path.join("/tmp", "dpd_wd_%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0")))
    os.makedirs(d, exist_ok=True)
    return d


def _watch(child, rank, beat_file, fail_file, lim):
    """0 = the worker finished (or hung in tear-down after printing its result); a string = why THIS supervisor declares the
    attempt failed; None = another rank's supervisor did."""
    while True:
        code = child.poll()
        phase, t_beat = _read_beat(beat_file)
        if code is not None:
            if code == 0 or phase == "done":   # "done": the result is out; a non-zero exit in tear-down does not un-print it
                return 0
            return "rank %d worker exited with code %d in phase %r" % (rank, code, phase)
        if os.path.exists(fail_file):
            return None
        if phase is not None:
            allowed = lim.get(phase.split(":")[0], lim["*"])
            if time.time() - t_beat > allowed:
                if phase == "done":
                    _stop(child)
                    return 0
                return "rank %d: no heartbeat for %.0f s in phase %r (limit %.0f s)" % (rank, time.time() - t_beat, phase, allowed)
        time.sleep(0.1)


def supervise(argv, rank, world, log=sys.stderr):
    """Run `argv` as this rank's worker under the watchdog; returns the exit code of the launch as this rank sees it.
    Attempt 1: the environment as given.  Attempt 2 (after any rank's supervisor declared attempt 1 failed): DPD_DP_BACKEND=torch
    on a fresh rendezvous.  After a failed attempt 2, rank 0 prints a JSON line that says so (`value` null) and the code is 3."""
    d = wd_dir()
    if world == 1:          # the only rank: nobody else reads this directory, so what an earlier launch of the same parent left is stale
        for f in os.listdir(d):
            try:
                os.unlink(os.path.join(d, f))
            except OSError:
                pass
    else:                   # a directory name can repeat (PID and port reuse): markers of a launch long gone must not decide this one.  The ranks
        for f in os.listdir(d):    # of one launch start within seconds of each other, so nothing of THIS launch is two minutes old yet
            if f.split(".")[0] in ("fail", "port", "ok", "reported"):
                try:
                    if os.path.getmtime(os.path.join(d, f)) < time.time() - 120.0:
                        os.unlink(os.path.join(d, f))
                except OSError:
                    pass
    lim = limits()
    t_launch = time.time()
    history = []
    for attempt in (1, 2):
        env = dict(os.environ, DPD_BENCH_CHILD="1", DPD_WD_DIR=d, DPD_WD_ATTEMPT=str(attempt))
        if attempt == 2:
            # a fresh store: keys of the failed attempt (unique ids, barrier counters) must not be read by the retry.  Rank 0's
            # supervisor picks the port, the others wait for it.
            pf = os.path.join(d, "port.2")
            if rank == 0:
                _write_atomic(pf, "%d\n" % _free_port())
            t0 = time.time()
            while not os.path.exists(pf):
                if time.time() - t0 > 60:
                    log.write("launch: rank %d never saw the retry port\n" % rank)
                    return 3
                time.sleep(0.05)
            env.update(DPD_DP_BACKEND="torch", DPD_BENCH_FALLBACK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=open(pf).read().strip(),
                       TORCHELASTIC_USE_AGENT_STORE="False")
            env.pop("DPD_WD_INJECT_HANG", None)
            env["DPD_WD_HISTORY"] = json.dumps(history)
        beat_file = os.path.join(d, "rank%d.a%d" % (rank, attempt))
        fail_file = os.path.join(d, "fail.%d" % attempt)
        _write_atomic(beat_file, "start %.3f\n" % time.time())
        child = subprocess.Popen(list(argv), env=env)
        try:
            why = _watch(child, rank, beat_file, fail_file, lim)
        except KeyboardInterrupt:            # the launcher is taking us down (SIGTERM): the worker goes first
            _stop(child)
            raise
        if why == 0:
            _write_atomic(os.path.join(d, "ok.%d.r%d" % (attempt, rank)), "%.3f\n" % time.time())
            return 0
        # Some rank's supervisor has already returned 0 for this attempt (its worker reached "done"): that rank will not take part in a
        # retry, which would then block in rendezvous until the init limit.  If rank 0 is among the finished, its JSON line is
        # out and a late failure elsewhere (report / profile / tear-down phases) cannot take it back: stop here.
        finished = [f for f in os.listdir(d) if f.startswith("ok.%d." % attempt)]
        if finished:
            _stop(child)
            log.write("launch watchdog (attempt %d): %s after rank(s) %s had finished -> no retry\n" %
                      (attempt, why if why is not None else "another rank failed", ",".join(sorted(f.rsplit("r", 1)[-1] for f in finished))))
            log.flush()
            if "ok.%d.r0" % attempt in finished:
                return 0
            history.append({"attempt": attempt, "failure": str(why), "after_s": round(time.time() - t_launch, 1), "no_retry": "ranks had finished"})
            break
        if why is not None:
            try:
                fd = os.open(fail_file, os.O_CREAT | os.O_EXCL | os.O_WRONLY)
                os.write(fd, (why + "\n").encode())
                os.close(fd)
            except FileExistsError:
                pass
            log.write("launch watchdog (attempt %d): %s -> stopping the workers%s\n" %
                      (attempt, why, ", retrying with DPD_DP_BACKEND=torch" if attempt == 1 else ""))
            log.flush()
        _stop(child)
        try:
            history.append({"attempt": attempt, "failure": open(fail_file).read().strip(), "after_s": round(time.time() - t_launch, 1)})
        except OSError:
            history.append({"attempt": attempt, "failure": "unknown"})
        if attempt == 1 and os.environ.get("DPD_DP_BACKEND", "rccl") == "torch" and os.environ.get("DPD_WD_RETRY_SAME", "0") != "1":
            # the conservative backend itself failed: a retry would run the same thing again
            break
    # Rank 0's line must be out before ANY supervisor exits with a failure code: the launcher (bench.spawn_ranks, torchrun) stops the
    # other ranks the moment one of them fails, and a rank-0 supervisor stopped before its print loses the only record of the launch
    # (seen under host load in tests/test_ddp_gloo.py: the line was missing once in three runs)
    reported = os.path.join(d, "reported")
    if rank == 0:
        print(json.dumps({"metric": "query-points/sec (DPDist fwd+bwd)", "value": None, "unit": "query-points/sec", "n_gpus": world,
                          "error": "data-parallel launch failed on both collective backends", "watchdog": history}), flush=True)
        _write_atomic(reported, "%.3f\n" % time.time())
    else:
        t0 = time.time()
        while not os.path.exists(reported) and time.time() - t0 < 15.0:
            time.sleep(0.05)
    return 3


def maybe_supervise(world=None):
    """Call first thing in a rank's `main`.  Returns None in the worker (or when there is nothing to supervise); otherwise runs
    the supervisor and exits the process with its code."""
    if os.environ.get("DPD_BENCH_CHILD") == "1" or os.environ.get("DPD_WD", "1") == "0":
        return None
    if world is None:
        world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and os.environ.get("DPD_FORCE_DIST") != "1":
        return None
    rank = int(os.environ.get("RANK", "0"))
    # children die with the supervisor: if the launcher kills us (torchrun on another rank's failure), stop the worker first
    raise SystemExit(_supervise_with_signals(rank, world))


def _supervise_with_signals(rank, world):
    def on_term(signum, frame):
        raise KeyboardInterrupt
    old = signal.signal(signal.SIGTERM, on_term)
    try:
        return supervise([sys.executable] + sys.argv, rank, world)
    except KeyboardInterrupt:
        return 130
    finally:
        signal.signal(signal.SIGTERM, old)
