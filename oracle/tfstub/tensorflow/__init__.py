"""Minimal `tensorflow` 1.x stand-in used ONLY by oracle/gen_goldens.py.

TEST INFRASTRUCTURE -- not part of the product.  TensorFlow 1.x is not installable in the
authoring container, so the reference's own Python files (models/dpdist_and_aue.py,
utils/dpdist_util.py, utils/tf_util.py under /root/reference) are imported UNCHANGED with this
package first on sys.path.  Every symbol below restates the documented semantics of the TF 1.14
primitive of the same name (see SURVEY.md Appendix B for the list of assumptions), evaluated
eagerly on torch-CPU so that autograd also yields gradient goldens.

Only the ~50 symbols the DPDist hot path touches are provided.  Nothing here is copied from
TensorFlow or from the reference.
"""
import builtins as _b
import contextlib
import math as _pymath

import numpy as np
import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------------
# dtype handling: `tf.float32` maps to torch.float32 by default; gen_goldens.py can switch the
# whole stub to float64 to obtain a high-precision evaluation of the same reference graph.
# ---------------------------------------------------------------------------------------------
_REAL = torch.float32


def set_real_dtype(dt):
    global _REAL
    _REAL = dt


class _DT:
    def __init__(self, name):
        self.name = name

    def torch(self):
        return {"float32": _REAL, "float16": torch.float16, "int32": torch.int32,
                "int64": torch.int64, "bool": torch.bool}[self.name]

    def __repr__(self):
        return "tf." + self.name


float32, float16, int32, int64 = _DT("float32"), _DT("float16"), _DT("int32"), _DT("int64")
bool = _DT("bool")  # noqa: A001  (shadows the builtin inside this module on purpose)
AUTO_REUSE = "AUTO_REUSE"


class _Dim:
    def __init__(self, v):
        self.value = None if v is None else int(v)

    def __int__(self):
        return self.value

    def __index__(self):
        return self.value

    def __pow__(self, p):
        return self.value ** p

    def __mul__(self, o):
        return self.value * int(o)

    __rmul__ = __mul__

    def __eq__(self, o):
        return self.value == (o.value if isinstance(o, _Dim) else o)

    def __hash__(self):
        return hash(self.value)

    def __repr__(self):
        return "Dim(%s)" % self.value


class _Shape(list):
    def as_list(self):
        return [d.value for d in self]


def _unwrap(x):
    if isinstance(x, Tensor):
        return x.v
    if isinstance(x, _Dim):
        return x.value
    return x


def _t(x, like=None):
    """to torch tensor"""
    x = _unwrap(x)
    if isinstance(x, torch.Tensor):
        return x
    if isinstance(x, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(x))
        if t.is_floating_point():
            t = t.to(_REAL)
        return t
    if isinstance(x, (list, tuple)):
        return torch.stack([_t(e) for e in x])
    if isinstance(x, float):
        return torch.tensor(x, dtype=_REAL)
    return torch.tensor(x)


class Tensor:
    """Eager tensor with the slice of the tf.Tensor surface the reference uses."""

    def __init__(self, v, name=None):
        self.v = v
        self.name = name

    # -- shape protocol
    @property
    def shape(self):
        return _Shape(_Dim(s) for s in self.v.shape)

    def get_shape(self):
        return self.shape

    @property
    def dtype(self):
        return self.v.dtype

    def __repr__(self):
        return "<stub tf.Tensor %s shape=%s dtype=%s>" % (self.name, tuple(self.v.shape), self.v.dtype)

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        idx = tuple(_unwrap(i) if not isinstance(i, _b.slice) else
                    _b.slice(_unwrap(i.start), _unwrap(i.stop), _unwrap(i.step)) for i in idx)
        return Tensor(self.v[idx])

    # -- arithmetic
    def _bin(self, o, f, rev=False):
        a, b = self.v, _t(o)
        if b.is_floating_point() and a.is_floating_point() and b.dtype != a.dtype:
            b = b.to(a.dtype)
        return Tensor(f(b, a) if rev else f(a, b))

    def __add__(self, o): return self._bin(o, torch.add)
    def __radd__(self, o): return self._bin(o, torch.add, True)
    def __sub__(self, o): return self._bin(o, torch.sub)
    def __rsub__(self, o): return self._bin(o, torch.sub, True)
    def __mul__(self, o): return self._bin(o, torch.mul)
    def __rmul__(self, o): return self._bin(o, torch.mul, True)
    def __truediv__(self, o): return self._bin(o, torch.div)
    def __rtruediv__(self, o): return self._bin(o, torch.div, True)
    def __neg__(self): return Tensor(-self.v)
    def __gt__(self, o): return self._bin(o, torch.gt)
    def __ge__(self, o): return self._bin(o, torch.ge)
    def __lt__(self, o): return self._bin(o, torch.lt)
    def __le__(self, o): return self._bin(o, torch.le)

    def numpy(self):
        return self.v.detach().cpu().numpy()


# ---------------------------------------------------------------------------------------------
# graph-level state: variables, scopes, collections
# ---------------------------------------------------------------------------------------------
class _Graph:
    def __init__(self):
        self.variables = {}
        self.collections = {}
        self.scope = []          # list of (name, reuse)
        self.named = {}          # tf.identity(name=...) results
        self.overrides = {}      # variable name -> numpy array (set by the generator)
        self.rng = torch.Generator().manual_seed(0)


_G = _Graph()


def reset_default_graph():
    global _G
    ov = _G.overrides
    _G = _Graph()
    _G.overrides = ov


def set_variable_overrides(d):
    _G.overrides = dict(d)


def get_default_graph():
    return _G


def stub_variables():
    return _G.variables


def stub_named():
    return _G.named


class _VarScope:
    def __init__(self, name, reuse):
        self.name, self.reuse = name, reuse


@contextlib.contextmanager
def variable_scope(name_or_scope, reuse=None, **kw):
    if isinstance(name_or_scope, _VarScope):   # re-entering the current scope with reuse=True
        saved = _G.scope
        _G.scope = [(n, reuse if reuse is not None else r) for n, r in saved] or [("", reuse)]
        try:
            yield name_or_scope
        finally:
            _G.scope = saved
        return
    _G.scope.append((name_or_scope, reuse))
    try:
        yield _VarScope("/".join(n for n, _ in _G.scope if n), reuse)
    finally:
        _G.scope.pop()


@contextlib.contextmanager
def name_scope(name, *a, **k):
    yield name


@contextlib.contextmanager
def device(name):
    yield


def get_variable_scope():
    return _VarScope("/".join(n for n, _ in _G.scope if n), None)


def _reuse_now():
    r = None
    for _, rr in _G.scope:
        if rr:
            r = rr
    return r


def get_variable(name, shape=None, initializer=None, dtype=None, trainable=True):
    full = "/".join([n for n, _ in _G.scope if n] + [name])
    if full in _G.variables:
        return _G.variables[full]
    if full in _G.overrides:
        val = torch.from_numpy(np.ascontiguousarray(_G.overrides[full])).to(_REAL)
        assert list(val.shape) == [int(s) for s in shape], (full, val.shape, shape)
    else:
        val = initializer([int(s) for s in shape]).to(_REAL)
    val = val.clone().requires_grad_(trainable)
    var = Tensor(val, name=full)
    _G.variables[full] = var
    return var


def constant_initializer(value=0.0):
    return lambda shape: torch.full(shape, float(value), dtype=_REAL)


def truncated_normal_initializer(stddev=1.0):
    return lambda shape: torch.fmod(torch.randn(shape, generator=_G.rng), 2.0) * stddev


def add_to_collection(name, value):
    _G.collections.setdefault(name, []).append(value)


def get_collection(name, scope=None):
    return list(_G.collections.get(name, []))


# ---------------------------------------------------------------------------------------------
# tensor constructors / shape ops
# ---------------------------------------------------------------------------------------------
def constant(value, dtype=None, shape=None, name=None):
    t = _t(value)
    if dtype is not None:
        t = t.to(dtype.torch())
    return Tensor(t)


def ones(shape, dtype=float32):
    return Tensor(torch.ones([int(_unwrap(s)) for s in shape], dtype=dtype.torch()))


def zeros(shape, dtype=float32):
    return Tensor(torch.zeros([int(_unwrap(s)) for s in shape], dtype=dtype.torch()))


def shape(x):
    return [int(s) for s in _t(x).shape]


def range(n):  # noqa: A001
    return Tensor(torch.arange(int(_unwrap(n)), dtype=torch.int32))


def cast(x, dtype):
    return Tensor(_t(x).to(dtype.torch()))


def identity(x, name=None):
    out = Tensor(_t(x), name=name)
    if name is not None:
        full = "/".join([n for n, _ in _G.scope if n] + [name])
        _G.named[full] = out
    return out


def expand_dims(x, axis):
    return Tensor(_t(x).unsqueeze(axis))


def squeeze(x, axis=None):
    return Tensor(_t(x).squeeze() if axis is None else _t(x).squeeze(axis))


def tile(x, multiples):
    return Tensor(_t(x).repeat(*[int(_unwrap(m)) for m in multiples]))


def reshape(x, shp):
    return Tensor(_t(x).reshape([int(_unwrap(s)) for s in shp]))


def transpose(x, perm=None):
    t = _t(x)
    if perm is None:
        perm = list(reversed(list(np.arange(t.dim()))))
    return Tensor(t.permute(*[int(p) for p in perm]))


def concat(values, axis):
    ts = [_t(v) for v in values]
    return Tensor(torch.cat(ts, dim=axis))


def stack(values, axis=0):
    ts = [_t(v) for v in values]
    return Tensor(torch.stack(ts, dim=axis))


def split(value, num_or_size_splits, axis=0):
    t = _t(value)
    if isinstance(num_or_size_splits, int):
        return [Tensor(p) for p in torch.chunk(t, num_or_size_splits, dim=axis)]
    return [Tensor(p) for p in torch.split(t, [int(_unwrap(n)) for n in num_or_size_splits], dim=axis)]   # sized form


def slice(input_, begin, size, name=None):  # noqa: A001
    """tf.slice: size -1 = to the end of that axis."""
    t = _t(input_)
    idx = []
    for ax, (b, n) in enumerate(zip(begin, size)):
        b, n = int(_unwrap(b)), int(_unwrap(n))
        idx.append(_b.slice(b, t.shape[ax] if n == -1 else b + n))
    return Tensor(t[tuple(idx)])


def tensordot(a, b, axes, name=None):
    a_ax, b_ax = axes
    a_ax = [a_ax] if isinstance(a_ax, int) else list(a_ax)
    b_ax = [b_ax] if isinstance(b_ax, int) else list(b_ax)
    return Tensor(torch.tensordot(_t(a), _t(b), dims=(a_ax, b_ax)))


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    x, y = _t(a), _t(b)
    if transpose_a:
        x = x.transpose(-1, -2)
    if transpose_b:
        y = y.transpose(-1, -2)
    return Tensor(x @ y)


def norm(tensor, ord="euclidean", axis=None, keepdims=None, name=None):  # noqa: A002
    """tf.norm: ord 2 / 'euclidean' over `axis` = sqrt(sum x^2) (vector norm)."""
    assert ord in (2, "euclidean")
    t = _t(tensor)
    kd = _b.bool(keepdims)
    return Tensor(torch.sqrt((t * t).sum() if axis is None else (t * t).sum(dim=axis, keepdim=kd)))


def cond(pred, true_fn=None, false_fn=None, name=None):
    p = _unwrap(pred)
    p = _b.bool(p.item()) if isinstance(p, torch.Tensor) else _b.bool(p)
    return true_fn() if p else false_fn()


def gather_nd(params, indices):
    p, idx = _t(params), _t(indices).long()
    k = idx.shape[-1]
    return Tensor(p[tuple(idx[..., i] for i in builtins_range(k))])


import builtins as _b  # noqa: E402
builtins_range = _b.range


# ---------------------------------------------------------------------------------------------
# math
# ---------------------------------------------------------------------------------------------
def multiply(a, b, name=None):
    return Tensor(_t(a)) * b


def add_n(xs, name=None):
    out = _t(xs[0])
    for x in xs[1:]:
        out = out + _t(x)
    return Tensor(out)


def add(a, b, name=None): return Tensor(_t(a)) + b
def subtract(a, b, name=None): return Tensor(_t(a)) - b
def divide(a, b, name=None): return Tensor(_t(a)) / b
def tanh(x, name=None): return Tensor(torch.tanh(_t(x)))
def sin(x, name=None): return Tensor(torch.sin(_t(x)))
def cos(x, name=None): return Tensor(torch.cos(_t(x)))
def sqrt(x): return Tensor(torch.sqrt(_t(x)))
def abs(x): return Tensor(torch.abs(_t(x)))  # noqa: A001
def sign(x): return Tensor(torch.sign(_t(x)))
def square(x): return Tensor(_t(x) ** 2)
def exp(x): return Tensor(torch.exp(_t(x)))


def pow(x, y):  # noqa: A001
    return Tensor(torch.pow(_t(x), _unwrap(y)))


def maximum(a, b):
    a = _t(a)
    b = _t(b)
    if not b.is_floating_point() or b.dtype != a.dtype:
        b = b.to(a.dtype)
    # TF: gradient flows to `a` where a >= b, to `b` otherwise.
    return Tensor(torch.where(a >= b, a, b.expand_as(a) if b.dim() else b))


def minimum(a, b):
    a = _t(a)
    b = _t(b).to(a.dtype)
    return Tensor(torch.where(a <= b, a, b))


def _axes(axis):
    return axis


def reduce_sum(x, axis=None, keepdims=False, keep_dims=None):
    kd = keepdims if keep_dims is None else keep_dims
    t = _t(x)
    return Tensor(t.sum() if axis is None else t.sum(dim=axis, keepdim=kd))


def reduce_mean(x, axis=None, keepdims=False, keep_dims=None):
    kd = keepdims if keep_dims is None else keep_dims
    t = _t(x)
    if isinstance(x, (list, tuple)):
        t = torch.stack([_t(e) for e in x])
    return Tensor(t.mean() if axis is None else t.mean(dim=axis, keepdim=kd))


def reduce_max(x, axis=None, keepdims=False, keep_dims=None):
    kd = keepdims if keep_dims is None else keep_dims
    t = _t(x)
    # torch.amax distributes the gradient evenly among ties, like tf.reduce_max.
    return Tensor(t.amax() if axis is None else t.amax(dim=axis, keepdim=kd))


def reduce_min(x, axis=None, keepdims=False, keep_dims=None):
    kd = keepdims if keep_dims is None else keep_dims
    t = _t(x)
    return Tensor(t.amin() if axis is None else t.amin(dim=axis, keepdim=kd))


def extract_volume_patches(input, ksizes, strides, padding, name=None):  # noqa: A002
    """[B,D,H,W,C] -> [B,D',H',W', kd*kh*kw*C]; depth ordered (plane,row,col,channel), channel
    fastest; 'SAME' zero padding.  Only stride 1 / odd k is needed by the reference."""
    x = _t(input)
    kd, kh, kw = [int(k) for k in ksizes[1:4]]
    assert list(strides) == [1, 1, 1, 1, 1]
    if padding == "SAME":
        pd, ph, pw = (kd - 1) // 2, (kh - 1) // 2, (kw - 1) // 2
        x = F.pad(x, (0, 0, pw, kw - 1 - pw, ph, kh - 1 - ph, pd, kd - 1 - pd))
    p = x.unfold(1, kd, 1).unfold(2, kh, 1).unfold(3, kw, 1)   # [B,D',H',W',C,kd,kh,kw]
    p = p.permute(0, 1, 2, 3, 5, 6, 7, 4)                       # [...,kd,kh,kw,C]
    return Tensor(p.reshape(p.shape[0], p.shape[1], p.shape[2], p.shape[3], -1))


class _Math:
    @staticmethod
    def argmax(input, axis=None, name=None, output_type=None):  # noqa: A002
        return Tensor(torch.argmax(_t(input), dim=axis))


math = _Math()  # the tf.math namespace (python's math module is `_pymath` in this file)


class _NN:
    @staticmethod
    def relu(x, name=None):
        return Tensor(torch.relu(_t(x)))

    @staticmethod
    def relu6(x, name=None):
        return Tensor(torch.clamp(_t(x), 0.0, 6.0))

    @staticmethod
    def tanh(x, name=None):
        return Tensor(torch.tanh(_t(x)))

    @staticmethod
    def l2_loss(x):
        return Tensor((_t(x) ** 2).sum() / 2)

    @staticmethod
    def l2_normalize(x, axis=None, epsilon=1e-12, name=None, dim=None):
        ax = axis if dim is None else dim
        t = _t(x)
        ss = (t * t).sum(dim=ax, keepdim=True)
        return Tensor(t * torch.rsqrt(torch.clamp_min(ss, epsilon)))

    @staticmethod
    def conv2d(input, filter, strides, padding, data_format="NHWC", name=None):  # noqa: A002
        x, w = _t(input), _t(filter)
        assert data_format == "NHWC" and padding == "VALID"
        y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1),
                     stride=(int(strides[1]), int(strides[2])))
        return Tensor(y.permute(0, 2, 3, 1))

    @staticmethod
    def bias_add(value, bias, data_format="NHWC", name=None):
        return Tensor(_t(value) + _t(bias))

    @staticmethod
    def max_pool(value, ksize, strides, padding, data_format="NHWC", name=None):
        assert data_format == "NHWC" and padding == "VALID"
        y = F.max_pool2d(_t(value).permute(0, 3, 1, 2), (int(ksize[1]), int(ksize[2])), (int(strides[1]), int(strides[2])))
        return Tensor(y.permute(0, 2, 3, 1))

    @staticmethod
    def dropout(x, keep_prob, noise_shape=None, seed=None, name=None):
        raise NotImplementedError("stub: dropout is only reachable with is_training=True; goldens use the inference branch")


nn = _NN()


# ---------------------------------------------------------------------------------------------
# tf.contrib
# ---------------------------------------------------------------------------------------------
class _MVNDiag:
    def __init__(self, loc=None, scale_diag=None):
        self.loc, self.scale = _t(loc), _t(scale_diag)

    def prob(self, x):
        z = (_t(x) - self.loc) / self.scale
        k = z.shape[-1]
        log_unnorm = -0.5 * (z * z).sum(dim=-1)
        log_norm = 0.5 * k * _pymath.log(2.0 * _pymath.pi) + torch.log(self.scale).sum(dim=-1)
        return Tensor(torch.exp(log_unnorm - log_norm))


class _Distributions:
    MultivariateNormalDiag = _MVNDiag


def _xavier_initializer(uniform=True, seed=None, dtype=None):
    def init(shape):
        recept = 1
        for s in shape[:-2]:
            recept *= s
        fan_in, fan_out = shape[-2] * recept, shape[-1] * recept
        lim = _pymath.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(shape, generator=_G.rng) * 2 - 1) * lim
    return init


class _Layers:
    xavier_initializer = staticmethod(_xavier_initializer)

    @staticmethod
    def flatten(x):
        t = _t(x)
        return Tensor(t.reshape(t.shape[0], -1))


def _batch_norm(inputs, decay=0.999, center=True, scale=False, epsilon=0.001, is_training=True, updates_collections=None,
                scope=None, data_format="NHWC", reuse=None, **kw):
    """tf.contrib.layers.batch_norm (TF 1.14), the subset utils/tf_util.py:573-577 uses: variables <scope>/beta, gamma,
    moving_mean, moving_variance; training = batch statistics over all axes but the last (biased variance in the
    normalisation), moving averages updated in place with the UNBIASED variance (fused kernel, updates_collections=None);
    inference = the moving statistics."""
    assert data_format == "NHWC"
    x = _t(inputs)
    C = x.shape[-1]
    train = _unwrap(is_training)
    train = _b.bool(train.item()) if isinstance(train, torch.Tensor) else _b.bool(train)
    with variable_scope(scope or "BatchNorm", reuse=reuse):
        beta = get_variable("beta", [C], initializer=constant_initializer(0.0)) if center else None
        gamma = get_variable("gamma", [C], initializer=constant_initializer(1.0)) if scale else None
        mm = get_variable("moving_mean", [C], initializer=constant_initializer(0.0), trainable=False)
        mv = get_variable("moving_variance", [C], initializer=constant_initializer(1.0), trainable=False)
    axes = tuple(builtins_range(x.dim() - 1))
    if train:
        mean = x.mean(dim=axes)
        var = ((x - mean) ** 2).mean(dim=axes)
        n = x.numel() // C
        with torch.no_grad():
            mm.v.mul_(decay).add_(mean.detach() * (1 - decay))
            mv.v.mul_(decay).add_(var.detach() * (n / max(n - 1, 1)) * (1 - decay))
    else:
        mean, var = mm.v, mv.v
    y = (x - mean) * torch.rsqrt(var + epsilon)
    if gamma is not None:
        y = y * gamma.v
    if beta is not None:
        y = y + beta.v
    return Tensor(y)


_Layers.batch_norm = staticmethod(_batch_norm)


class _Contrib:
    distributions = _Distributions()
    layers = _Layers()


contrib = _Contrib()


class _Summary:
    @staticmethod
    def scalar(*a, **k):
        return None

    @staticmethod
    def histogram(*a, **k):
        return None


summary = _Summary()


# ---------------------------------------------------------------------------------------------
# tf.train: what the trainer's optimizer assembly touches (train_multi_gpu_pc_compare_dist.py:216,274-277,301,976-990)
# ---------------------------------------------------------------------------------------------
class _AdamOptimizer:
    """tf.train.AdamOptimizer (TF 1.14, 'epsilon hat' form):
        lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  w -= lr_t m / (sqrt(v) + eps)
    `learning_rate` may be a python float or a callable evaluated at every apply (the stub has no graph, so a learning-rate
    TENSOR that depends on the global step is passed as a thunk by the generator)."""

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, name="Adam"):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon
        self.t = 0
        self.m, self.v = {}, {}

    def compute_gradients(self, loss, var_list=None):
        vs = list(var_list)
        gs = torch.autograd.grad(_t(loss), [v.v for v in vs], retain_graph=True, allow_unused=True)
        return [(None if g is None else Tensor(g), v) for g, v in zip(gs, vs)]

    def apply_gradients(self, grads_and_vars, global_step=None):
        lr = self.lr() if callable(self.lr) else self.lr
        lr = float(_t(lr)) if not isinstance(lr, float) else lr
        self.t += 1
        lr_t = lr * _pymath.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        with torch.no_grad():
            for g, var in grads_and_vars:
                if g is None:
                    continue
                g = _t(g).to(var.v.dtype)
                m = self.m.setdefault(var.name, torch.zeros_like(var.v))
                v = self.v.setdefault(var.name, torch.zeros_like(var.v))
                m.mul_(self.b1).add_(g * (1 - self.b1))
                v.mul_(self.b2).add_(g * g * (1 - self.b2))
                var.v.sub_(lr_t * m / (v.sqrt() + self.eps))
            if global_step is not None:
                global_step.v.add_(1)


class _Train:
    AdamOptimizer = _AdamOptimizer

    @staticmethod
    def exponential_decay(learning_rate, global_step, decay_steps, decay_rate, staircase=False, name=None):
        p = _t(global_step).to(torch.float64) / float(_unwrap(decay_steps))
        if staircase:
            p = torch.floor(p)
        return Tensor((float(learning_rate) * float(decay_rate) ** p).to(_REAL))


train = _Train()
