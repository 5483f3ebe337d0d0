"""CPU restatement of the DPDist hot path (torch-CPU, differentiable, compact dataflow).

TEST INFRASTRUCTURE -- the oracle.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this file; the product (dpdist_amd/) never does.

Pinned (tests/test_oracle.py) against tests/golden/*.npz, which were produced by executing the
reference's own Python under oracle/tfstub (oracle/gen_goldens.py).  Residual assumption: the
TF 1.14 primitive semantics restated in the stub; the reference ships no
tests or golden vectors of its own.

Each function cites the reference lines (relative to /root/reference) it follows.
"""
import math

import numpy as np
import torch

F_PER_G = 20  # d_pi{mean,max} + d_mu{mean,max,min}x3 + d_sigma{mean,max,min}x3


def grid_axis(m):
    """utils/dpdist_util.py:42 (`linspace(-1,1,m,False)+1/m`) == :987-988 (`arange(-1,1,2/m)+1/m`)."""
    return np.linspace(-1, 1, m, False) + (1.0 / m)


def grid_centers(m):
    """[m^3,3] float64.  utils/dpdist_util.py:47-48: np.meshgrid(l,l,l) default 'xy' indexing,
    flattened row-major => index g = i*m*m + j*m + t has centre (x,y,z) = (l[j], l[i], l[t])."""
    l = grid_axis(m)
    x, y, z = np.meshgrid(l, l, l)
    return np.stack([x.flatten(), y.flatten(), z.flatten()]).T


def mfv3d(points, m=8, sigma=0.125):
    """3DmFV of each cloud: [C,N,3] -> [C,m^3,20].  utils/dpdist_util.py:22-141 (full_fv, normalize)."""
    dt = points.dtype
    Cn, N, D = points.shape
    G = m ** 3
    mu = torch.tensor(grid_centers(m), dtype=dt)                       # :50
    w = 1.0 / G                                                         # :49
    z = (points[:, :, None, :] - mu[None, None]) / sigma                # [C,N,G,3]
    # :69-71 MultivariateNormalDiag.prob
    logp = -0.5 * (z * z).sum(-1) - (0.5 * D * math.log(2 * math.pi) + D * math.log(sigma))
    p = torch.exp(logp)
    wp = p * w                                                          # :73
    Q = wp / wp.sum(-1, keepdim=True)                                   # :74
    d_pi_all = (Q - w) / (math.sqrt(w) * N)                             # :78
    d_pi = torch.stack([d_pi_all.mean(1), d_pi_all.amax(1)], -1)        # :80-83  [C,G,2]
    d_mu_all = Q[..., None] * z                                         # :87
    d_mu = torch.cat([d_mu_all.mean(1), d_mu_all.amax(1), d_mu_all.amin(1)], -1) * (1.0 / math.sqrt(w))   # :89-98
    d_sig_all = Q[..., None] * (z * z - 1)                              # :100
    d_sig = torch.cat([d_sig_all.mean(1), d_sig_all.amax(1), d_sig_all.amin(1)], -1) * (1.0 / math.sqrt(2 * w))  # :102-109

    def norm(x):
        x = torch.sign(x) * torch.sqrt(torch.clamp_min(torch.abs(x), 1e-12))      # :119-121
        ss = (x * x).sum(1, keepdim=True)                                         # :124-126, over the Gaussian axis
        return x * torch.rsqrt(torch.clamp_min(ss, 1e-12))

    return torch.cat([norm(d_pi), norm(d_mu), norm(d_sig)], -1)        # :134-137


def local_window(fv, m=8, k=5):
    """[C,m^3,20] -> [C,m^3,k^3*20].  utils/dpdist_util.py:911-930 (extract_volume_patches, SAME)."""
    Cn = fv.shape[0]
    g = fv.reshape(Cn, m, m, m, -1)
    h = (k - 1) // 2
    g = torch.nn.functional.pad(g, (0, 0, h, h, h, h, h, h))
    p = g.unfold(1, k, 1).unfold(2, k, 1).unfold(3, k, 1)       # [C,m,m,m,F,k,k,k]
    p = p.permute(0, 1, 2, 3, 5, 6, 7, 4)
    return p.reshape(Cn, m ** 3, -1)


def voxel_lookup(q, m=8):
    """q [C,N,3] -> (v int64 [C,N], mask [C,N], local [C,N,3]).
    utils/dpdist_util.py:459-492: half-open cells (lo, hi] tested against every centre in float32
    exactly as the reference does (`pc > C - g`, `pc <= C + g`), then argmax."""
    dt = q.dtype
    Cc = torch.tensor(grid_centers(m), dtype=dt)                       # :925-929 cast to float32
    g = torch.abs(Cc[0][2] - Cc[1][2]) / 2                             # :468
    inside = ((q[:, :, None, :] > Cc[None, None] - g) & (q[:, :, None, :] <= Cc[None, None] + g)).all(-1)
    insf = inside.to(dt)
    v = torch.argmax(insf, dim=2)                                      # :490
    mask = torch.gather(insf, 2, v[..., None])[..., 0]                 # :436-440
    local = q - Cc[v]                                                  # :491, :443-447
    return v, mask, local


def decoder(x, W, out_act=True):
    """x [R,2503] -> [R,3].  utils/dpdist_util.py:513-544 + utils/tf_util.py:213-228 (1xW VALID conv ==
    dense layer), relu6/3 at :691."""
    hs = []
    h = x
    for l in (1, 2, 3):
        w = W["pc_compare/dpdist_local/mapper_conv%d/weights" % l]
        b = W["pc_compare/dpdist_local/mapper_conv%d/biases" % l]
        w2 = w.reshape(-1, w.shape[-1])
        h = torch.relu(h @ w2 + b)
        hs.append(h)
    w = W["pc_compare/dpdist_local/mapper_conv4/weights"]
    b = W["pc_compare/dpdist_local/mapper_conv4/biases"]
    y = h @ w.reshape(-1, w.shape[-1]) + b
    if out_act:
        y = torch.clamp(y, 0.0, 6.0) / 3.0
    return y, hs


def as_torch_weights(W, dtype=torch.float32, requires_grad=False):
    return {n: torch.tensor(np.asarray(a), dtype=dtype).requires_grad_(requires_grad) for n, a in W.items()}


def get_model(pcA, pcB, W, add_noise=None, m=8, k=5, sigma=0.125):
    """models/dpdist_and_aue.py:31-86.  Returns (pred_set, aux)."""
    B, N, _ = pcA.shape
    pcA_noise = pcA if add_noise is None else pcA + add_noise           # :45
    fvA = mfv3d(pcA_noise, m, sigma)                                    # :56-58
    fvB = mfv3d(pcB, m, sigma)                                          # :59-61
    embA, embB = local_window(fvA, m, k), local_window(fvB, m, k)       # :63-65
    # DPDist (:494-511): AB half = points of B against surface A; BA half = points of A (UN-noised, :69) vs B
    vB, maskB, locB = voxel_lookup(pcB, m)
    vA, maskA, locA = voxel_lookup(pcA, m)
    rowsAB = torch.cat([locB, torch.gather(embA, 1, vB[..., None].expand(-1, -1, embA.shape[-1]))], -1)
    rowsBA = torch.cat([locA, torch.gather(embB, 1, vA[..., None].expand(-1, -1, embB.shape[-1]))], -1)
    x = torch.cat([rowsAB, rowsBA], 0).reshape(2 * B * N, -1)           # :511
    y, hs = decoder(x, W)
    y = y.reshape(2, B, N, 1, 3)                                        # :695 split
    predAB = y[0] * maskB[..., None, None]                              # :697
    predBA = y[1] * maskA[..., None, None]                              # :698
    aux = {"fvA": fvA, "fvB": fvB, "x": x, "vB": vB, "vA": vA, "maskB": maskB, "maskA": maskA, "hs": hs,
           "embA": embA, "embB": embB}
    return {"pred_listAB": predAB, "pred_listBA": predBA}, aux


def get_loss(pred_set, labels):
    """utils/dpdist_util.py:962-980 -> (loss_samples scalar, loss_pred scalar)."""
    ab = pred_set["pred_listAB"][:, :, :, 0].squeeze(-1)
    ba = pred_set["pred_listBA"][:, :, :, 0]
    loss_samples = (ab - labels).abs().mean()                           # :972
    loss_pred = (pred_set["pred_listAB"][:, :, :, 0].mean() + ba.mean()) / 2   # :976-977
    return loss_samples, loss_pred


def adam_tf_step(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer update (epsilon-hat form), used at train_multi_gpu_pc_compare_dist.py:216.
    numpy, in place; t is the 1-based step."""
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m *= b1
    m += (1 - b1) * g
    v *= b2
    v += (1 - b2) * g * g
    p -= lr_t * m / (np.sqrt(v) + eps)


def learning_rate(step, base=1e-4, decay_step=300 * 512, decay_rate=0.5, floor=1e-7):
    """train_multi_gpu_pc_compare_dist.py:976-990: `exponential_decay(base, batch, DECAY_STEP=300*512,
    DECAY_RATE=0.5, staircase=True)` on the raw step counter, clipped below at 1e-7."""
    return max(base * decay_rate ** math.floor(step / decay_step), floor)
