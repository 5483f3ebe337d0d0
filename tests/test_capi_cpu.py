"""CPU-side checks: the C-ABI library builds/loads and exports every declared symbol, host logic (parameter layout,
TF interchange, schedules, argument errors).  No GPU compute calls."""
import os
import re

import numpy as np
import pytest
import torch

from dpdist_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from dpdist_amd import build, lib as L
    build.build(verbose=False)          # hipcc cross-compiles gfx950 without a GPU
    return L.load()


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "dpdist_capi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dpd_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(lib):
    from dpdist_amd import lib as L
    names = _header_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "libdpdist_hip.so does not export %s" % n
        assert n in L.SIGNATURES, "lib.py has no ctypes signature for %s" % n
    assert sorted(L.SIGNATURES) == names


@pytest.mark.parametrize("cc,std,ext", [("gcc", "-std=c11", ".c"), ("gcc", "-std=c99", ".c"), ("g++", "-std=c++17", ".cpp")])
def test_header_compiles_as_c_and_cxx(tmp_path, cc, std, ext):
    """include/dpdist_capi.h calls itself a C ABI: a translation unit that only includes it must pass a strict C and a
    C++ front end (round-2 defect: the `dpd_planes` typedef name was used before its definition)."""
    import shutil
    import subprocess
    if shutil.which(cc) is None:
        pytest.skip(cc + " not installed")
    tu = tmp_path / ("tu" + ext)
    tu.write_text('#include "dpdist_capi.h"\n'
                  'int use(const dpd_planes* p, const dpd_decoder_params* d, const dpd_small_grads* s, const dpd_asloss* g)\n'
                  '{ return p && d && s && g ? (int)DPD_BF16 + DPD_E_NULL + (int)sizeof(dpd_planes) : DPD_OK; }\n')
    r = subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                        "-I", os.path.join(ROOT, "include"), str(tu)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_host_only_entry_points(lib):
    assert lib.dpd_version().decode().startswith("dpdist_hip")
    assert lib.dpd_padded_width(5) == 2528 and lib.dpd_padded_width(3) == 544
    assert lib.dpd_workspace_bytes(4096, 2528, 1024, 0) >= 2 * 2528 * 1024 * 4
    assert lib.dpd_workspace_bytes(4096, 2528, 1024, 1) >= lib.dpd_workspace_bytes(4096, 2528, 1024, 0) + 3 * 2 * 4096 * 2528
    assert lib.dpd_set_gemm_plan(99, 0, 1) < 0          # argument errors are negative codes
    assert lib.dpd_set_gemm_plan(4, 0, 2) == 0


def test_null_and_dim_errors_without_gpu(lib):
    """Argument validation happens before any HIP call, so it can be exercised on the CPU box."""
    assert lib.dpd_mfv3d_fwd(None, 1, 64, 8, 0.125, None, None) == -1            # DPD_E_NULL
    assert lib.dpd_gemm_f32(0, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, None, 0, 1, 0, None, 0, None) == -1
    assert lib.dpd_l1_loss(None, None, 0, 0, 1.0, None, None, None) == -1
    assert lib.dpd_adam_tf(None, None, None, None, 4, 0.1, 0.9, 0.999, 1e-8, 1.0, None) == -1


def test_ops_refuse_cpu_tensors(lib):
    from dpdist_amd import ops
    with pytest.raises(RuntimeError, match="GPU"):
        ops.mfv3d_fwd(torch.zeros(1, 64, 3), 8, 0.125)
    with pytest.raises(RuntimeError):
        ops.adam_tf(torch.zeros(8), torch.zeros(8), torch.zeros(8), torch.zeros(8), 1e-3) if False else ops.patch_rows_fwd(
            torch.zeros(1, 64, 3), torch.zeros(1, 512, 20), 8, 5)


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from dpdist_amd import lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.load()


@pytest.mark.parametrize("mlp", [(64, 64, 64), (128, 128, 128)])
def test_param_layout_round_trip(mlp):
    from dpdist_amd.model import DPDistParams, TF_NAME
    W = synth.make_weights("wide", mlp=mlp)
    P = DPDistParams(k=5, mlp=mlp, device="cpu", init=None)
    P.load_tf_state_dict(W)
    back = P.tf_state_dict()
    assert sorted(back) == sorted(W)
    for n in W:
        assert back[n].shape == W[n].shape and np.array_equal(back[n], W[n]), n
    # internal layout: window rows first, xyz rows after, zero pad rows; flat buckets cover everything
    W1p = P.view("W1p").numpy()
    w1 = W[TF_NAME % (1, "weights")].reshape(2503, mlp[0])
    assert np.array_equal(W1p[:2500], w1[3:]) and np.array_equal(W1p[2500:2503], w1[:3]) and not W1p[2503:].any()
    assert P.KP == 2528 and P.bucket_bounds[0] == 0 and P.bucket_bounds[-1] == P.numel
    assert P.bucket_bounds[1] == 2528 * mlp[0] + mlp[0]
    assert all(off % 4 == 0 for off, _, _ in P._segments.values())


def test_xavier_init_matches_tf_limits():
    from dpdist_amd.model import DPDistParams, TF_NAME
    P = DPDistParams(k=5, mlp=(64, 64, 64), device="cpu")
    sd = P.tf_state_dict()
    lim1 = np.sqrt(6.0 / (2503 + 2503 * 64))     # utils/tf_util.py:90-91 on a [1,2503,1,64] kernel
    w1 = sd[TF_NAME % (1, "weights")]
    assert np.abs(w1).max() <= lim1 * (1 + 1e-6) and np.abs(w1).max() > 0.9 * lim1
    assert not sd[TF_NAME % (1, "biases")].any()


def test_learning_rate_schedule():
    from dpdist_amd.trainer import learning_rate
    from oracle import restate as R
    for step in (0, 1, 153599, 153600, 153601, 2 * 153600, 20 * 153600):
        assert learning_rate(step) == R.learning_rate(step)
    assert learning_rate(0) == 1e-4 and learning_rate(153600) == 5e-5 and learning_rate(10 ** 9) == 1e-7


def test_get_model_rejects_off_path_branches():
    from dpdist_amd import model as M
    a = torch.zeros(1, 64, 3)
    for kw in (dict(pn="pointnet", k=5), dict(k=0), dict(k=5, conv_version=2), dict(k=5, bn=1), dict(k=5, bn=0, full_fv=False)):
        with pytest.raises(NotImplementedError):
            M.get_model(a, a, True, **({"bn": 0, **kw}))


def test_synth_is_deterministic_and_shaped():
    a1, b1, l1 = synth.s2_modelnet_shaped(4, 64, 100)
    a2, b2, l2 = synth.s2_modelnet_shaped(4, 64, 100)
    assert np.array_equal(a1, a2) and np.array_equal(b1, b2) and np.array_equal(l1, l2)
    assert a1.shape == (4, 64, 3) and l1.shape == (4, 64)
    assert not l1[:, :32].any() and (l1[:, 32:48] > 0.001).all() and (l1[:, 32:48] < 0.1).all() and (l1[:, 48:] > 0.1).all()
    assert np.abs(a1).max() <= 1.0        # inside the Gaussian grid's cube
    pcA, pcB = synth.s1_random_patches(32, 64, 0)
    assert np.isin(pcB, synth.BOUNDARY_SET).mean() > 0.01


@pytest.mark.parametrize("B,N,H,dt", [(16, 64, 1024, 0), (16, 64, 1024, 1), (16, 64, 1024, 2), (3, 36, 192, 2), (32, 64, 64, 0)])
def test_asloss_engine_carve_is_a_partition_of_the_callers_buffer(lib, B, N, H, dt):
    """dpd_asloss_bytes / dpd_asloss_carve are host-side pointer arithmetic (no device call): every member lies inside the caller's
    allocation, 256-byte aligned, no two members overlap, the plane compute types get planes exactly for plane-shaped rows, and shapes the
    engine does not take are refused before anything is touched."""
    import ctypes
    from dpdist_amd import lib as L
    n = lib.dpd_asloss_bytes(B, N, 8, 5, H, dt)
    assert n > 0
    base = 1 << 30                                          # any 256-byte aligned "device address": nothing is dereferenced
    e = L.AsLoss()
    assert lib.dpd_asloss_carve(ctypes.c_void_p(base), n, B, N, 8, 5, H, dt, 0.125, e) == 0
    assert (e.B, e.N, e.m, e.k, e.KP, e.H, e.dtype) == (B, N, 8, 5, 2528, H, dt) and abs(e.sigma - 0.125) < 1e-7
    Q, KP = 2 * B * N, 2528
    planes = dt != 0 and Q % 32 == 0 and H % 64 == 0
    f4 = 4
    want = {"pts": 2 * B * N * 3 * f4, "q": 2 * B * N * 3 * f4, "fv": 2 * B * 512 * 20 * f4, "ssq": 2 * B * 4 * 20 * f4, "mask": Q * f4,
            "vox": Q * 4, "h3": Q * H * f4, "y": Q * 3 * f4, "pred": Q * 3 * f4, "dy": Q * 3 * f4, "g3": Q * H * f4, "dX": Q * KP * f4,
            "dfv": 2 * B * 512 * 20 * f4, "dpts": 2 * B * N * 3 * f4, "scratch": 8, "mfv_ws": e.mfv_ws_bytes, "ws": e.ws_bytes}
    if planes:
        np_ = 3 if dt == 1 else 1
        assert e.planes.np == np_ and e.planes.Q == Q and e.planes.Qb == Q and not e.X and not e.h1 and not e.W2T
        spans = dict(want)
        pl = e.planes
        for name, size in (("X_rc", np_ * 2 * Q * KP), ("h1_rc", np_ * 2 * Q * H), ("h2_rc", np_ * 2 * Q * H), ("g3_rc", np_ * 2 * Q * H),
                           ("g2_rc", np_ * 2 * Q * H), ("g1_rc", np_ * 2 * Q * H), ("W1_r8", np_ * 2 * KP * H), ("W1_rc", np_ * 2 * KP * H),
                           ("W2_r8", np_ * 2 * H * H), ("W3_r8", np_ * 2 * H * H), ("W2_rc", np_ * 2 * H * H), ("W3_rc", np_ * 2 * H * H)):
            spans["planes." + name] = size
        assert not pl.X_r8 and not pl.h1_r8 and not pl.g3_r8          # as-loss mode feeds no weight gradients: RC planes only
    else:
        assert e.planes.np == 0 and not e.planes.X_rc
        spans = dict(want, X=Q * KP * f4, h1=Q * H * f4, h2=Q * H * f4, g2=Q * H * f4, g1=Q * H * f4, W2T=H * H * f4, W3T=H * H * f4,
                     W1pT=H * KP * f4)
    iv = []
    for name, size in spans.items():
        ptr = getattr(e.planes, name[7:]) if name.startswith("planes.") else getattr(e, name)
        assert ptr and ptr % 256 == 0 and base <= ptr and ptr + size <= base + n, name
        iv.append((ptr, ptr + size, name))
    iv.sort()
    for (a0, a1, na), (b0, b1, nb) in zip(iv, iv[1:]):
        assert a1 <= b0, (na, nb)
    # refused: too small a buffer, a misaligned one, B * N beyond the fused output kernel, an even window, H not a multiple of 64
    assert lib.dpd_asloss_carve(ctypes.c_void_p(base), n - 1, B, N, 8, 5, H, dt, 0.125, e) == -4
    assert lib.dpd_asloss_carve(ctypes.c_void_p(base + 16), n, B, N, 8, 5, H, dt, 0.125, e) == -3
    assert lib.dpd_asloss_bytes(256, 64, 8, 5, H, dt) == 0 and lib.dpd_asloss_bytes(B, N, 8, 4, H, dt) == 0
    assert lib.dpd_asloss_bytes(B, N, 8, 5, H + 8, dt) == 0
    assert lib.dpd_asloss_forward(None, None, None, 0, None, None) == -1 and lib.dpd_asloss_backward(None, None, None, None, None) == -1
    e2 = L.AsLoss()                                            # carved but without weights: refused before any launch
    assert lib.dpd_asloss_carve(ctypes.c_void_p(base), n, B, N, 8, 5, H, dt, 0.125, e2) == 0
    assert lib.dpd_asloss_forward(e2, ctypes.c_void_p(base), ctypes.c_void_p(base), 0, ctypes.c_void_p(base), None) == -1


def test_schedule_selection_single_process():
    """ddp.select_schedule without a process group: the smallest time wins, ties go to the earlier candidate, an unsupported candidate is
    dropped before anything is timed, none supported is an error, and an exception inside a timed candidate is NOT swallowed."""
    from dpdist_amd.ddp import select_schedule
    t = {"early": 0.33, "grouped": 0.29, "late": 0.31}
    assert select_schedule(["early", "grouped", "late"], t.__getitem__, torch.device("cpu")) == ("grouped", t)
    assert select_schedule(["a", "b"], lambda n: 1.0, torch.device("cpu"))[0] == "a"
    timed = []

    def time_fn(n):
        timed.append(n)
        return t[n]
    assert select_schedule(["early", "grouped"], time_fn, torch.device("cpu"), supported=lambda n: n != "grouped") == ("early", {"early": 0.33, "grouped": None})
    assert timed == ["early"]
    with pytest.raises(RuntimeError, match="supported"):
        select_schedule(["grouped"], time_fn, torch.device("cpu"), supported=lambda n: False)

    def flaky(n):
        raise RuntimeError("boom")
    with pytest.raises(RuntimeError, match="boom"):
        select_schedule(["grouped"], flaky, torch.device("cpu"))


def test_scaling_model_of_the_data_parallel_step():
    """ddp.predict_scaling (the `dp.model` record of bench.py): a ring all-reduce over N-1 direct xGMI links -- two GPUs share ONE link and
    are the worst case per byte, collectives that fit under their hide window cost nothing, the bf16 wire halves the bytes."""
    from dpdist_amd import ddp
    assert ddp.allreduce_us(1e7, 1) == 0.0
    assert ddp.allreduce_us(1e7, 2) > ddp.allreduce_us(1e7, 4) > ddp.allreduce_us(1e7, 8)
    m = ddp.predict_scaling(0.56, [8.4e6, 10.3e6], [127.0, 30.0], plumbing_us=15.0)
    pw = m["per_world"]
    assert set(pw) == {"2", "4", "8"} and pw["8"]["links"] == 7 and pw["2"]["wire_bytes_per_gpu"] == int(18.7e6)
    assert pw["8"]["wire_bytes_per_gpu"] == int(2 * 7 / 8 * 18.7e6)
    assert pw["2"]["predicted_efficiency"] < pw["4"]["predicted_efficiency"] < pw["8"]["predicted_efficiency"] < 1.0
    for n in ("2", "4", "8"):
        exp = sum(max(0.0, c - w) for c, w in zip(pw[n]["collective_us"], (127.0, 30.0)))
        assert abs(pw[n]["predicted_exposed_us"] - exp) <= 0.11
        assert abs(pw[n]["predicted_ms_per_step"] - (0.56 + (15.0 + pw[n]["predicted_exposed_us"]) / 1e3)) <= 2e-4
    half = ddp.predict_scaling(0.56, [8.4e6, 10.3e6], [127.0, 30.0], plumbing_us=15.0, wire="bf16")["per_world"]
    assert half["8"]["bytes"] * 2 == pw["8"]["bytes"] and half["8"]["predicted_exposed_us"] < pw["8"]["predicted_exposed_us"]
    huge = ddp.predict_scaling(0.56, [8.4e6, 10.3e6], [1e6, 1e6])["per_world"]
    assert all(v["predicted_exposed_us"] == 0 and v["predicted_efficiency"] == 1.0 for v in huge.values())
