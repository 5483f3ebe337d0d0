"""The plain-C oracle restatement (oracle/cpu_ref.c) against the golden vectors and the torch restatement."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from dpdist_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cref():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libdpd_cpuref.so"))
    fp = ctypes.POINTER(ctypes.c_float)
    lib.cpuref_mfv3d.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, fp]
    lib.cpuref_forward.argtypes = [fp, fp, fp] + [ctypes.c_int] * 4 + [ctypes.c_float] + [fp] * 8 + [ctypes.c_int, fp, fp]
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


@pytest.mark.parametrize("m", [8, 5])
def test_c_mfv3d_golden(cref, golden_dir, m):
    d = np.load(os.path.join(golden_dir, "fv_cases.npz"))
    pts = np.ascontiguousarray(d["points"], np.float32)
    fv = np.zeros((4, m ** 3, 20), np.float32)
    cref.cpuref_mfv3d(_p(pts), 4, 64, m, 0.125, _p(fv))
    assert np.abs(fv - d["fv_m%d_f64" % m]).max() <= 3e-6


@pytest.mark.parametrize("case,wk", [("s1", "wide"), ("boundary", "wide"), ("s1", "xavier_tf")])
def test_c_forward_golden(cref, golden_dir, case, wk):
    d = np.load(os.path.join(golden_dir, "path_fwd_%s_%s.npz" % (case, wk)))
    W = synth.make_weights(wk)
    n = "pc_compare/dpdist_local/mapper_conv%d/%s"
    ws = [np.ascontiguousarray(W[n % (l, t)].reshape(-1, W[n % (l, t)].shape[-1]) if t == "weights" else W[n % (l, t)], np.float32)
          for l in (1, 2, 3, 4) for t in ("weights", "biases")]
    pcA, pcB = np.ascontiguousarray(d["pcA"]), np.ascontiguousarray(d["pcB"])
    ab, ba = np.zeros((2, 64, 3), np.float32), np.zeros((2, 64, 3), np.float32)
    cref.cpuref_forward(_p(pcA), _p(pcB), None, 2, 64, 8, 5, 0.125, *[_p(w) for w in ws], 1024, _p(ab), _p(ba))
    assert np.abs(ab - d["pred_listAB_f64"][:, :, 0]).max() <= 5e-5
    assert np.abs(ba - d["pred_listBA_f64"][:, :, 0]).max() <= 5e-5
