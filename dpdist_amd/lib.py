"""ctypes binding of libdpdist_hip.so (the C ABI of include/dpdist_capi.h).

There is NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
Thin typed wrappers (`ops_*`) take torch CUDA tensors, check dtype/contiguity and pass raw device
pointers + the current HIP stream.
"""
import ctypes
import os
from ctypes import c_long, POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdpdist_hip.so")

_ERR = {-1: "DPD_E_NULL (null pointer)", -2: "DPD_E_DIM (bad dimension)",
        -3: "DPD_E_UNSUPPORTED (unsupported size/configuration)", -4: "DPD_E_WORKSPACE (workspace too small)"}


class DecoderParams(Structure):
    _fields_ = [(n, c_void_p) for n in ("W1p", "b1", "W2", "b2", "W3", "b3", "W4", "b4", "W2T", "W3T", "W1pT")]


class SmallGrads(Structure):
    _fields_ = [(n, c_void_p) for n in ("db1", "db2", "db3", "dW4", "db4", "partials", "l1_pred", "l1_labels", "l1_loss")] + [
        ("l1_gscale", c_float), ("db_partials", c_void_p), ("fwd_y", c_void_p), ("fwd_pred", c_void_p)]


class AdamFuse(Structure):   # include/dpdist_capi.h: dpd_adam_fuse
    _fields_ = [("WT", c_void_p * 3), ("w_off", c_long * 3), ("w_rows", c_int * 3), ("w_cols", c_int * 3),
                ("W_rc", c_void_p * 3), ("W_r8", c_void_p * 3), ("np", c_int), ("partials", c_void_p),
                ("nparts", c_int), ("rec", c_int), ("H", c_int), ("Qb", c_int), ("tail_off", c_long), ("loss", c_void_p)]


class Planes(Structure):     # include/dpdist_capi.h: dpd_planes
    _fields_ = [("np", c_int), ("Q", c_int), ("Qb", c_int)] + [(n, c_void_p) for n in (
        "X_rc", "X_r8", "h1_rc", "h1_r8", "h2_rc", "h2_r8", "g3_rc", "g3_r8", "g2_rc", "g2_r8", "g1_rc", "g1_r8",
        "W1_r8", "W2_r8", "W3_r8", "W1_rc", "W2_rc", "W3_rc", "h3_rc")]


class PoseNetW(Structure):   # include/dpdist_capi.h: dpd_pose_net
    _fields_ = [("Wp", c_void_p * 5), ("bp", c_void_p * 5), ("Wh", c_void_p * 4), ("bh", c_void_p * 4), ("out_features", c_int)]


class AsLoss(Structure):     # include/dpdist_capi.h: dpd_asloss (the as-loss engine; driven by dpdist_amd/asloss.py)
    _fields_ = ([(n, c_int) for n in ("B", "N", "m", "k", "KP", "H", "dtype")] + [("sigma", c_float)] +
                [(n, c_void_p) for n in ("pts", "q", "fv", "ssq", "mask", "vox", "X", "h1", "h2", "h3", "y", "pred", "dy", "g3", "g2", "g1",
                                         "dX", "dfv", "dpts", "W2T", "W3T", "W1pT", "scratch")] +
                [("mfv_ws", c_void_p), ("mfv_ws_bytes", c_size_t), ("ws", c_void_p), ("ws_bytes", c_size_t),
                 ("planes", Planes), ("params", DecoderParams)])


# name -> (restype, argtypes); mirrors include/dpdist_capi.h one to one
SIGNATURES = {
    "dpd_version": (c_char_p, []),
    "dpd_padded_width": (c_int, [c_int]),
    "dpd_mfv3d_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "dpd_mfv3d_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dpd_mfv3d_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dpd_patch_rows_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                   c_void_p, POINTER(Planes), c_void_p]),
    "dpd_mfv3d_fwd_stacked": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p]),
    "dpd_patch_rows_fwd_scaled": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                          c_void_p, POINTER(Planes), c_void_p]),
    "dpd_decoder_out_asloss": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(DecoderParams), c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "dpd_decoder_out_asloss_planes": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(DecoderParams), c_float, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, POINTER(Planes), c_void_p, c_void_p]),
    "dpd_asloss_tail": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_size_t,
                                c_void_p, c_void_p, c_void_p]),
    "dpd_asloss_combine": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dpd_patch_rows_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                   c_void_p]),
    "dpd_decoder_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(DecoderParams), c_int, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, POINTER(Planes), c_void_p]),
    "dpd_decoder_bwd_data": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                     POINTER(DecoderParams), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     POINTER(SmallGrads), c_void_p, c_size_t, POINTER(Planes), c_int, c_void_p]),
    "dpd_stack_clouds": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dpd_decoder_bwd_weights": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                        c_void_p, c_void_p, c_size_t, POINTER(Planes), c_void_p, c_void_p]),
    "dpd_decoder_bwd_weights_pair": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p, c_size_t, POINTER(Planes), c_void_p, c_void_p, c_void_p]),
    "dpd_decoder_bwd_weights_trio": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, POINTER(Planes), c_void_p]),
    "dpd_crc32c": (ctypes.c_uint32, [c_void_p, c_size_t, ctypes.c_uint32]),
    "dpd_chamfer_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 6),
    "dpd_chamfer_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "dpd_pose_apply_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dpd_pose_apply_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p]),
    "dpd_pose_refine_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dpd_pose_refine": (c_int, [POINTER(PoseNetW), c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_size_t, c_void_p,
                                c_void_p, c_void_p, c_void_p]),
    "dpd_pose_point_bwd_workspace_bytes": (c_size_t, [c_int]),
    "dpd_pose_point_fwd_train": (c_int, [POINTER(PoseNetW), c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 7),
    "dpd_pose_point_bwd": (c_int, [POINTER(PoseNetW), c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 6 +
                           [POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_size_t, c_void_p]),
    "dpd_pose_head_bwd_workspace_bytes": (c_size_t, [c_int]),
    "dpd_pose_head_fwd_train": (c_int, [POINTER(PoseNetW), c_void_p, c_int] + [c_void_p] * 6),
    "dpd_pose_head_bwd": (c_int, [POINTER(PoseNetW), c_void_p, c_int] + [c_void_p] * 5 + [POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_void_p,
                                                                                           c_size_t, c_void_p]),
    "dpd_planes_bytes": (c_size_t, [c_int] * 6),
    "dpd_planes_carve": (c_int, [c_void_p, c_size_t] + [c_int] * 6 + [POINTER(Planes)]),
    "dpd_weights_to_planes": (c_int, [POINTER(DecoderParams), c_int, c_int, POINTER(Planes), c_void_p]),
    "dpd_asloss_bytes": (c_size_t, [c_int] * 6),
    "dpd_asloss_carve": (c_int, [c_void_p, c_size_t] + [c_int] * 6 + [c_float, POINTER(AsLoss)]),
    "dpd_asloss_init": (c_int, [POINTER(AsLoss), c_void_p]),
    "dpd_asloss_set_weights": (c_int, [POINTER(AsLoss), POINTER(DecoderParams), c_void_p]),
    "dpd_asloss_forward": (c_int, [POINTER(AsLoss), c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "dpd_asloss_backward": (c_int, [POINTER(AsLoss), c_void_p, c_void_p, c_void_p, c_void_p]),
    "dpd_asloss_forward_backward": (c_int, [POINTER(AsLoss), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dpd_weights_transpose": (c_int, [POINTER(DecoderParams), c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dpd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dpd_split_planes": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_long, c_void_p, c_long, c_void_p]),
    "dpd_gemm_planes": (c_int, [c_int] * 6 + [c_void_p, c_int, c_long, c_void_p, c_int, c_long, c_void_p, c_int, c_void_p,
                                c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "dpd_l1_loss": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "dpd_adam_tf": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float, c_float,
                            c_float, c_void_p]),
    "dpd_adam_tf_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float, c_float, c_float,
                                  POINTER(AdamFuse), c_void_p]),
    "dpd_adam_tf_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_float, c_float, c_float, c_float, c_void_p]),
    "dpd_gemm_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                             c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "dpd_set_gemm_plan": (c_int, [c_int, c_int, c_int]),
    "dpd_prof_enable": (c_int, [c_int]),
    "dpd_prof_collect_form": (c_int, [c_int, POINTER(ctypes.c_double), POINTER(ctypes.c_double)]),
    "dpd_prof_collect": (c_int, [POINTER(ctypes.c_double), POINTER(ctypes.c_double)]),
    "dpd_prof_collect_stage": (c_int, [c_int, POINTER(ctypes.c_double), POINTER(ctypes.c_double)]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libdpdist_hip.so not found at %s -- run `python -m dpdist_amd.build` "
                           "(there is no CPU fallback)" % LIB_PATH)
    # Load order matters: the PyTorch-ROCm wheel bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1.  Import
    # torch FIRST so that our DT_NEEDED entries resolve to the HIP runtime torch already initialised; loading this
    # library first would pull /opt/rocm's runtime next to torch's and every launch would fail with hipErrorNoDevice.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here == header/library mismatch
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


DTYPES = {"f32": 0, "f32x3": 1, "bf16": 2, 0: 0, 1: 1, 2: 2}   # include/dpdist_capi.h: enum dpd_dtype


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        raise RuntimeError("%s failed: %s" % (what, _ERR.get(rc, rc)))
    raise RuntimeError("%s failed: hipError_t %d" % (what, rc))


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def cur_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def req(t, dtype=None, name="tensor", shape=None, numel=None):
    """Boundary check of a tensor handed to the C ABI (which only sees a raw pointer): device, dtype, contiguity and --
    when given -- the exact shape / element count the entry point will read or write."""
    import torch
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError("%s must have shape %s, got %s" % (name, tuple(shape), tuple(t.shape)))
    if numel is not None and t.numel() != numel:
        raise RuntimeError("%s must have %d elements, got %d" % (name, numel, t.numel()))
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (dpdist_amd has no CPU path)" % name)
    if dtype is None:
        dtype = torch.float32
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    return t


def make_params(W1p, b1, W2, b2, W3, b3, W4, b4, W2T=None, W3T=None, W1pT=None):
    return DecoderParams(*[None if t is None else t.data_ptr() for t in (W1p, b1, W2, b2, W3, b3, W4, b4, W2T, W3T, W1pT)])


def make_small_grads(db1, db2, db3, dW4, db4, partials=None, l1_pred=None, l1_labels=None, l1_loss=None, l1_gscale=1.0,
                     db_partials=None, fwd_y=None, fwd_pred=None):
    return SmallGrads(*[None if t is None else t.data_ptr() for t in (db1, db2, db3, dW4, db4, partials, l1_pred, l1_labels, l1_loss)],
                      float(l1_gscale), *[None if t is None else t.data_ptr() for t in (db_partials, fwd_y, fwd_pred)])
