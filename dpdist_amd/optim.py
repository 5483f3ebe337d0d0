"""`tf.train.AdamOptimizer` for the consumers' own small networks (rows f2 / f4), on the library's TF-form Adam kernel.

The as-loss consumers of the reference build `tf.train.AdamOptimizer(learning_rate)` over their own variables
(`pcrnet-registration/iterative_PCRNet_ours.py:239`, `train_multi_gpu_pc_compare_dist.py:216,457-463`).  TensorFlow's
update differs from `torch.optim.Adam` in where epsilon sits ("epsilon hat"):

    lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);   m, v as usual;   p -= lr_t * m / (sqrt(v) + eps)

`TFAdam` keeps every parameter of the network as a view into ONE flat fp32 buffer (so do the gradients and the two slots)
and runs the whole update as one launch of `dpd_adam_tf` (csrc/loss_adam.hip, the kernel the DPDist trainer uses).
There is no CPU form: parameters must live on the GPU, like everything else that goes through the C ABI.
"""
import math

import torch

from . import lib as L


class TFAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-8):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("TFAdam: no trainable parameters")
        super().__init__(params, dict(lr=lr, beta1=beta1, beta2=beta2, epsilon=epsilon))
        dev = params[0].device
        if dev.type != "cuda" or any(p.device != dev or p.dtype != torch.float32 for p in params):
            raise RuntimeError("TFAdam runs on the GPU library (dpd_adam_tf): float32 parameters on one GPU required")
        n = sum(p.numel() for p in params)
        n4 = (n + 3) // 4 * 4                      # the kernel strides in float4
        self.flat = torch.zeros(n4, device=dev)
        self.grad = torch.zeros(n4, device=dev)
        self.m = torch.zeros(n4, device=dev)
        self.v = torch.zeros(n4, device=dev)
        self.t = 0
        self.state_dev = torch.zeros(8, device=dev)   # [3] = lr_t for the captured (hipGraph) step: dpd_adam_tf_dev reads it from device memory
        self._params, off = params, 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.flat[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat[off:off + k].view_as(p)           # the module now computes on the flat buffer
                p.grad = self.grad[off:off + k].view_as(p)           # autograd accumulates in place into the flat gradient
                off += k

    def zero_grad(self, set_to_none=False):
        """Gradients stay views of the flat buffer (set_to_none would detach them from it)."""
        self.grad.zero_()
        for p in self._params:
            if p.grad is None or p.grad.data_ptr() < self.grad.data_ptr() or \
                    p.grad.data_ptr() >= self.grad.data_ptr() + self.grad.numel() * 4:
                self._rebind()
                break

    def _rebind(self):
        off = 0
        for p in self._params:
            k = p.numel()
            p.grad = self.grad[off:off + k].view_as(p)
            off += k

    def _inside(self, t, flat):
        return t is not None and flat.data_ptr() <= t.data_ptr() < flat.data_ptr() + flat.numel() * 4

    @torch.no_grad()
    def _check_views(self):
        """step() updates the FLAT buffers: a parameter or gradient that no longer lives in them (a backward with create_graph that
        replaced p.grad, zero_grad(set_to_none=True), net.to(...), an assignment to p.data) would otherwise be updated with a stale or
        zero gradient, silently.  Foreign gradients are copied in, foreign parameter storage is adopted and re-bound."""
        off = 0
        for p in self._params:
            k = p.numel()
            if not self._inside(p.data, self.flat):
                self.flat[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat[off:off + k].view_as(p)
            if p.grad is None:               # TF skips variables without a gradient; the fused kernel cannot: a zero gradient still
                self.grad[off:off + k].zero_()   # decays m / v and moves p by the remaining momentum (documented deviation)
                p.grad = self.grad[off:off + k].view_as(p)
            elif not self._inside(p.grad, self.grad):
                self.grad[off:off + k].copy_(p.grad.reshape(-1))
                p.grad = self.grad[off:off + k].view_as(p)
            off += k

    def state_dict(self):
        """torch's optimizer state_dict plus the flat slots and the step count (they do not live in `self.state`)."""
        sd = super().state_dict()
        sd["tf_adam"] = {"m": self.m.clone(), "v": self.v.clone(), "t": self.t}
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        extra = sd.pop("tf_adam", None)
        super().load_state_dict(sd)
        if extra is not None:
            self.m.copy_(extra["m"])
            self.v.copy_(extra["v"])
            self.t = int(extra["t"])

    def _lr_t(self):
        g = self.param_groups[0]
        return g["lr"] * math.sqrt(1.0 - g["beta2"] ** self.t) / (1.0 - g["beta1"] ** self.t)

    @torch.no_grad()
    def prepare_replay(self):
        """Before each replay of a hipGraph that captured `step()`: advance the step count on the host and put this step's lr_t (the
        same double-precision expression, rounded to fp32 like the eager call's argument) where the captured launch reads it."""
        self.t += 1
        self.state_dev[3:4].fill_(self._lr_t())

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("TFAdam: no closure form")
        self._check_views()
        g = self.param_groups[0]
        b1, b2 = g["beta1"], g["beta2"]
        if torch.cuda.is_current_stream_capturing():
            # captured form: same kernel arithmetic, lr_t comes from state_dev[3] (written by prepare_replay before every replay)
            L.check(L.load().dpd_adam_tf_dev(L.ptr(self.flat), L.ptr(self.grad), L.ptr(self.m), L.ptr(self.v), self.flat.numel(),
                                             L.ptr(self.state_dev), b1, b2, g["epsilon"], 1.0, L.cur_stream()), "dpd_adam_tf_dev")
            return
        self.t += 1
        lr_t = g["lr"] * math.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        L.check(L.load().dpd_adam_tf(L.ptr(self.flat), L.ptr(self.grad), L.ptr(self.m), L.ptr(self.v), self.flat.numel(),
                                     float(lr_t), b1, b2, g["epsilon"], 1.0, L.cur_stream()), "dpd_adam_tf")
