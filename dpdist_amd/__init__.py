"""dpdist_amd -- MI355X-native DPDist hot path (3DmFV encode + K^3 patch lookup + shared-MLP decoder).

The public surface mirrors the reference's model-module contract
(`models/dpdist_and_aue.py:23-86,203-204`): `get_model`, `get_loss`, plus an `nn.Module`
(`DPDistModel`) and an as-loss wrapper (`DPDistLoss`).  All compute runs in hand-written HIP
kernels behind the C ABI declared in `include/dpdist_capi.h`; there is no CPU fallback.
"""
from . import synth  # noqa: F401

__all__ = ["synth"]


def __getattr__(name):
    # model/functional import torch and the HIP library lazily so that `import dpdist_amd.synth`
    # stays numpy-only.
    if name in ("get_model", "get_loss", "DPDistModel", "DPDistLoss", "placeholder_inputs"):
        from . import model
        return getattr(model, name)
    raise AttributeError(name)
