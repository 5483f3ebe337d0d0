"""Deterministic synthetic inputs and decoder weight sets (numpy only).

These are the workloads of SURVEY.md section 8(d):

* S1 "random patches"  -- parity config 2 (B=32 pairs of 64 points, 2 % of pcB snapped onto voxel
  boundaries / just outside the cube).
* S2 "ModelNet-shaped" -- training configs 3-4: restates the reference trainer's batch recipe
  (`train_multi_gpu_pc_compare_dist.py:747-766`: pcA = 64 surface samples, pcB = 32 surface +
  16 near-surface + 16 far points, labels_AB = [0]*32 + distances) on analytic surfaces, because
  the ModelNet `*_dist_c_scaled.txt` files are not in the reference tree.
* weight sets `xavier_tf` (what `tf.contrib.layers.xavier_initializer` gives the four "1xW conv"
  kernels, `utils/tf_util.py:90-91`) and `wide` (outputs spread over [0, 2], both relu6
  saturations exercised).

Weights are returned in the reference's TF variable layout (`[1,2503,1,1024]`, `[1,1,1024,1024]`
x2, `[1,1,1024,3]` + biases) keyed by the TF variable names, so the same dict feeds the oracle
(reference under the stub), the restatement and `DPDistModel.load_tf_state_dict`.
"""
import math

import numpy as np

TF_SCOPE = "pc_compare/dpdist_local/mapper_conv%d/%s"


def tf_shapes(E_plus_D=2503, mlp=(1024, 1024, 1024), out=3):
    return [[1, E_plus_D, 1, mlp[0]], [1, 1, mlp[0], mlp[1]], [1, 1, mlp[1], mlp[2]],
            [1, 1, mlp[2], out]]


def _dense_fan_in(shape):
    return shape[1] * shape[2]


def make_weights(kind="xavier_tf", E_plus_D=2503, mlp=(1024, 1024, 1024), out=3):
    """dict TF-variable-name -> float32 array.  Recipes fixed by SURVEY.md 8(d)."""
    shapes = tf_shapes(E_plus_D, mlp, out)
    w = {}
    if kind == "xavier_tf":
        rng = np.random.default_rng(2)
        for l, shp in enumerate(shapes, 1):
            recept = shp[0] * shp[1]
            fan_in, fan_out = shp[2] * recept, shp[3] * recept
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            w[TF_SCOPE % (l, "weights")] = rng.uniform(-lim, lim, shp).astype(np.float32)
            w[TF_SCOPE % (l, "biases")] = np.zeros(shp[3], np.float32)
    elif kind == "wide":
        rng = np.random.default_rng(3)
        gains = [16.0, 1.0, 1.0, 4.0]
        for l, shp in enumerate(shapes, 1):
            fan_in = _dense_fan_in(shp)
            w[TF_SCOPE % (l, "weights")] = (rng.standard_normal(shp).astype(np.float32)
                                            * np.float32(math.sqrt(2.0 / fan_in) * gains[l - 1]))
            w[TF_SCOPE % (l, "biases")] = np.zeros(shp[3], np.float32)
        w[TF_SCOPE % (4, "biases")][:] = 3.0
    else:
        raise ValueError(kind)
    return w


BOUNDARY_SET = np.array([-1.0, -0.75, -0.5, -0.25, 0.0, 0.25, 0.5, 0.75, 1.0, 1.05, -1.05],
                        dtype=np.float32)


def s1_random_patches(B=32, N=64, seed=0):
    """S1: uniform(-0.9, 0.9) clouds; 2 % of pcB entries overwritten with cell-boundary values."""
    rng = np.random.default_rng(seed)
    pcA = rng.uniform(-0.9, 0.9, (B, N, 3)).astype(np.float32)
    pcB = rng.uniform(-0.9, 0.9, (B, N, 3)).astype(np.float32)
    r1 = np.random.default_rng(seed + 1)
    hit = r1.random((B, N, 3)) < 0.02
    vals = BOUNDARY_SET[r1.integers(0, len(BOUNDARY_SET), (B, N, 3))]
    pcB = np.where(hit, vals, pcB).astype(np.float32)
    return pcA, pcB


def boundary_cloud(B=2, N=64, seed=7):
    """Every coordinate of pcB sits exactly on a voxel face or just outside the cube."""
    rng = np.random.default_rng(seed)
    pcA = rng.uniform(-0.9, 0.9, (B, N, 3)).astype(np.float32)
    pcB = BOUNDARY_SET[rng.integers(0, len(BOUNDARY_SET), (B, N, 3))].astype(np.float32)
    return pcA, pcB


# ------------------------------------------------------------------------------------------
# S2: analytic surfaces with exact point-to-surface distance
# ------------------------------------------------------------------------------------------
def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)


def _sample_sphere(rng, n, r):
    v = rng.standard_normal((n, 3))
    return r * v / np.linalg.norm(v, axis=1, keepdims=True)


def _sample_box(rng, n, h):
    """uniform-ish samples on the surface of the axis-aligned box with half extents h"""
    p = rng.uniform(-1, 1, (n, 3)) * h
    ax = rng.integers(0, 3, n)
    sg = rng.choice([-1.0, 1.0], n)
    p[np.arange(n), ax] = sg * h[ax]
    return p


def _dist_sphere(p, r):
    return np.abs(np.linalg.norm(p, axis=1) - r)


def _dist_box(p, h):
    q = np.abs(p) - h
    outside = np.linalg.norm(np.maximum(q, 0.0), axis=1)
    inside = np.minimum(np.max(q, axis=1), 0.0)
    return np.abs(outside + inside)


# ------------------------------------------------------------------------------------------
# "chair-like" surfaces: unions of axis-aligned boxes (seat + back + four legs + optional arm rests).  ModelNet40 'chair'
# (BASELINE config 5, pcrnet-registration/run_train_and_eval_PCRNet.bash:43) is not in the tree; spheres, ellipsoids and
# single boxes have unobservable or ambiguous rotations, a chair does not (no symmetry except left/right mirroring, which
# is not a rotation).  y is up, the back rest sits at -z.
# ------------------------------------------------------------------------------------------
class BoxUnion:
    """Surface of a union of axis-aligned boxes: area-weighted surface samples (points buried inside another box are
    rejected) and the exact distance to the surface for points outside every box."""

    def __init__(self, centers, halves):
        self.c = np.asarray(centers, np.float64)
        self.h = np.asarray(halves, np.float64)

    def sdf_each(self, p):
        q = np.abs(p[:, None, :] - self.c[None]) - self.h[None]               # [n, K, 3]
        return np.linalg.norm(np.maximum(q, 0.0), axis=2) + np.minimum(q.max(axis=2), 0.0)

    def outside(self, p):
        return (self.sdf_each(p) > 0).all(axis=1)

    def dist(self, p):
        """distance to the union's surface; exact where `outside(p)`"""
        return np.abs(self.sdf_each(p)).min(axis=1)

    def sample(self, rng, n):
        hx, hy, hz = self.h[:, 0], self.h[:, 1], self.h[:, 2]
        area = np.stack([hy * hz, hx * hz, hx * hy], 1)                        # face-pair areas per box (x4, irrelevant)
        pr = (area / area.sum()).ravel()
        out = np.zeros((0, 3))
        while len(out) < n:
            m = 2 * (n - len(out)) + 8
            f = rng.choice(len(pr), m, p=pr)
            bi, ax = f // 3, f % 3
            p = rng.uniform(-1, 1, (m, 3)) * self.h[bi]
            p[np.arange(m), ax] = rng.choice([-1.0, 1.0], m) * self.h[bi, ax]
            p = p + self.c[bi]
            sd = self.sdf_each(p)
            sd[np.arange(m), bi] = 0.0
            out = np.concatenate([out, p[(sd > -1e-9).all(axis=1)]])
        return out[:n]

    def scaled(self, shift, s):
        return BoxUnion((self.c - shift) * s, self.h * s)


def make_chair(rng):
    """A random chair inside the ball of radius 0.8 (clouds are normalised to the unit sphere and scaled by 0.8,
    dataset_sample_with_gt.py:82), centred on its bounding box."""
    w, d = rng.uniform(0.40, 0.60), rng.uniform(0.40, 0.60)          # seat width (x) / depth (z)
    ts, tb, tl = rng.uniform(0.04, 0.09), rng.uniform(0.04, 0.09), rng.uniform(0.04, 0.08)
    hb, ll = rng.uniform(0.40, 0.75), rng.uniform(0.30, 0.55)        # back height above the seat, leg length
    c = [[0.0, 0.0, 0.0]]
    h = [[w / 2, ts / 2, d / 2]]
    c.append([0.0, ts / 2 + hb / 2, -d / 2 + tb / 2])
    h.append([w / 2, hb / 2, tb / 2])
    for sx in (-1, 1):
        for sz in (-1, 1):
            c.append([sx * (w / 2 - tl / 2), -ts / 2 - ll / 2, sz * (d / 2 - tl / 2)])
            h.append([tl / 2, ll / 2, tl / 2])
    if rng.random() < 0.35:                                           # arm rests
        ha, ta = rng.uniform(0.15, 0.25), rng.uniform(0.03, 0.06)
        for sx in (-1, 1):
            c.append([sx * (w / 2 - ta / 2), ts / 2 + ha, 0.05 * d])
            h.append([ta / 2, ta / 2, 0.45 * d])
            c.append([sx * (w / 2 - ta / 2), ts / 2 + ha / 2, 0.45 * d])
            h.append([ta / 2, ha / 2, ta / 2])
    u = BoxUnion(c, h)
    lo, hi = (u.c - u.h).min(0), (u.c + u.h).max(0)
    mid = (lo + hi) / 2
    rad = np.linalg.norm(np.abs(u.c - mid) + u.h, axis=1).max()           # farthest box corner from the centre
    return u.scaled(mid, 0.8 / rad)


def euler_rotation(rx, ry, rz):
    """R = Rx Ry Rz: what helper.apply_transformation (pcrnet-registration/helper.py:229-258) applies (z first, then y, x)."""
    cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def registration_pairs(B=16, N=64, seed=0, max_deg=45.0, t_clip=0.01, rng=None):
    """(source, template, gt_pose) as the reference's registration trainer composes them with its shipped flags
    (SPARSE_SAMPLING=1, s_random_points=1.0, centroid_sub=0, Noise=0: run_train_and_eval_PCRNet.bash:16-40):
    helper.split_template_source (helper.py:925-961): template and source are DIFFERENT N-point samples of the same surface,
    source = apply_transformation(source, pose); poses as utils/create_dataset/generate_poses_ours.py:4-18 draws them:
    gt_pose [B,6] = (t ~ U(-t_clip, t_clip)^3, (rx, ry, rz) ~ U(-max_deg, max_deg)^3 in radians)."""
    rng = np.random.default_rng(seed) if rng is None else rng
    src = np.zeros((B, N, 3), np.float32)
    tmpl = np.zeros((B, N, 3), np.float32)
    t = rng.uniform(-t_clip, t_clip, (B, 3))
    e = np.radians(rng.uniform(-max_deg, max_deg, (B, 3)))
    for b in range(B):
        ch = make_chair(rng)
        pts = ch.sample(rng, 2 * N)
        tmpl[b] = pts[:N]
        src[b] = pts[N:] @ euler_rotation(*e[b]).T + t[b]
    return src, tmpl, np.concatenate([t, e], 1)


def s2_modelnet_shaped(B=64, N=64, seed=100, shapes="analytic", tilt_deg=0.0):
    """(pcA, pcB, labels_AB) following the trainer's recipe on spheres/boxes of extent <= 0.8.

    Per pair: a shape (sphere radius 0.3-0.7 or box half-extents 0.2-0.55), a random y-rotation
    and a shift U(-0.1,0.1)^3 (`modelnet_dataset.py:87,91`); pcA = N surface samples; pcB = N/2
    other surface samples + N/4 near-surface points (distance in (0.001, 0.1)) + N/4 points
    uniform in the unit ball with distance > 0.1; labels = exact distance to the surface.

    shapes="chair": the same recipe on `make_chair` surfaces (what the config-5 / row-f2 demo trains DPDist on); tilt_deg > 0
    adds a rotation by U(-tilt_deg, tilt_deg) about each of x and z before the y-rotation (NOT in the reference's augmentation:
    its DPDist only sees upright chairs; the registration demo states which one it used).  The default "analytic" stream is
    unchanged (benchmarks and fixtures depend on it bit for bit).
    """
    rng = np.random.default_rng(seed)
    H, Qn = N // 2, N // 4
    pcA = np.zeros((B, N, 3), np.float32)
    pcB = np.zeros((B, N, 3), np.float32)
    lab = np.zeros((B, N), np.float32)
    for b in range(B):
        if shapes == "chair":
            _chair_pair(rng, N, pcA, pcB, lab, b, tilt_deg)
            continue
        is_sphere = rng.random() < 0.5
        if is_sphere:
            r = rng.uniform(0.3, 0.7)
            samp = lambda n: _sample_sphere(rng, n, r)          # noqa: E731
            dist = lambda p: _dist_sphere(p, r)                 # noqa: E731
        else:
            h = rng.uniform(0.2, 0.55, 3)
            samp = lambda n: _sample_box(rng, n, h)             # noqa: E731
            dist = lambda p: _dist_box(p, h)                    # noqa: E731
        R = _rot_y(rng.uniform(0, 2 * math.pi))
        shift = rng.uniform(-0.1, 0.1, 3)
        surfA, surfB = samp(N), samp(H)
        # near-surface: push surface samples along a random direction until 0.001 < d < 0.1
        near = np.zeros((0, 3))
        while len(near) < Qn:
            c = samp(4 * Qn) + rng.standard_normal((4 * Qn, 3)) * 0.04
            d = dist(c)
            near = np.concatenate([near, c[(d > 0.001) & (d < 0.1)]])
        near = near[:Qn]
        far = np.zeros((0, 3))
        while len(far) < Qn:
            c = rng.standard_normal((8 * Qn, 3))
            c = c / np.linalg.norm(c, axis=1, keepdims=True) * rng.random((8 * Qn, 1)) ** (1 / 3)
            c = c * 0.85
            d = dist(c)
            far = np.concatenate([far, c[d > 0.1]])
        far = far[:Qn]
        B_local = np.concatenate([surfB, near, far])
        lab[b] = np.concatenate([np.zeros(H), dist(near), dist(far)]).astype(np.float32)
        pcA[b] = (surfA @ R.T + shift).astype(np.float32)
        pcB[b] = (B_local @ R.T + shift).astype(np.float32)
    return pcA, pcB, lab


def _chair_pair(rng, N, pcA, pcB, lab, b, tilt_deg):
    H, Qn = N // 2, N // 4
    ch = make_chair(rng)
    R = _rot_y(rng.uniform(0, 2 * math.pi))
    if tilt_deg > 0:
        tx, tz = np.radians(rng.uniform(-tilt_deg, tilt_deg, 2))
        R = R @ euler_rotation(tx, 0.0, tz)
    shift = rng.uniform(-0.1, 0.1, 3)
    surf = ch.sample(rng, N + H)
    near = np.zeros((0, 3))
    while len(near) < Qn:
        c = ch.sample(rng, 4 * Qn) + rng.standard_normal((4 * Qn, 3)) * 0.04
        d = ch.dist(c)
        near = np.concatenate([near, c[(d > 0.001) & (d < 0.1) & ch.outside(c)]])
    near = near[:Qn]
    far = np.zeros((0, 3))
    while len(far) < Qn:
        c = rng.standard_normal((8 * Qn, 3))
        c = c / np.linalg.norm(c, axis=1, keepdims=True) * rng.random((8 * Qn, 1)) ** (1 / 3) * 0.85
        d = ch.dist(c)
        far = np.concatenate([far, c[(d > 0.1) & ch.outside(c)]])
    far = far[:Qn]
    B_local = np.concatenate([surf[N:], near, far])
    lab[b] = np.concatenate([np.zeros(H), ch.dist(near), ch.dist(far)]).astype(np.float32)
    pcA[b] = (surf[:N] @ R.T + shift).astype(np.float32)
    pcB[b] = (B_local @ R.T + shift).astype(np.float32)


# ----------------------------------------------------------------------------------------------------------------
# weights of the consumers' small networks by seed (fixtures store seeds, never weight blobs)
# ----------------------------------------------------------------------------------------------------------------
def make_named_weights(spec, seed, scale=1.0):
    """spec: ordered list of (tf variable name, shape).  Kernels ('weights'): N(0, 2/fan_in) * scale; 'biases' / 'beta':
    N(0, 0.05); 'gamma': 1 + N(0, 0.1); 'moving_mean': N(0, 0.1); 'moving_variance': U(0.5, 1.5)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in spec:
        leaf = name.split("/")[-1]
        if leaf == "weights":
            fan_in = int(np.prod(shape[:-1]))
            w = rng.standard_normal(shape) * np.sqrt(2.0 / fan_in) * scale
        elif leaf in ("biases", "beta"):
            w = rng.standard_normal(shape) * 0.05
        elif leaf == "gamma":
            w = 1.0 + rng.standard_normal(shape) * 0.1
        elif leaf == "moving_mean":
            w = rng.standard_normal(shape) * 0.1
        elif leaf == "moving_variance":
            w = rng.uniform(0.5, 1.5, shape)
        else:
            raise ValueError(name)
        out[name] = w.astype(np.float32)
    return out


def pose_net_spec(out_features=1024):
    """Variables of pcrnet-registration/models/ipcr_model.py: pointnet (:198-233, conv1..5) + get_pose (:273-284, fc1..4)."""
    dims = [3, 64, 64, 64, 128, out_features]
    spec = []
    for i in range(5):
        spec += [("conv%d/weights" % (i + 1), (1, 3 if i == 0 else 1, 1 if i == 0 else dims[i], dims[i + 1])),
                 ("conv%d/biases" % (i + 1), (dims[i + 1],))]
    fdims = [2 * out_features, 1024, 512, 256, 7]
    for i in range(4):
        spec += [("fc%d/weights" % (i + 1), (fdims[i], fdims[i + 1])), ("fc%d/biases" % (i + 1), (fdims[i + 1],))]
    return spec


def aue_pn_spec(num_point=64):
    """Variables of models/dpdist_and_aue.py:get_model_aue_pn (:88-145) under scope 'aue': conv1..5 + fc1, fc2 with batch norm
    (beta, gamma, moving_mean, moving_variance in '<layer>/bn'), fc3 plain."""
    dims = [3, 64, 64, 64, 128, 1024]
    spec = []

    def bn(scope, c):
        return [("%s/bn/%s" % (scope, n), (c,)) for n in ("beta", "gamma", "moving_mean", "moving_variance")]

    for i in range(5):
        sc = "aue/conv%d" % (i + 1)
        spec += [(sc + "/weights", (1, 3 if i == 0 else 1, 1 if i == 0 else dims[i], dims[i + 1])), (sc + "/biases", (dims[i + 1],))]
        spec += bn(sc, dims[i + 1])
    for i, (a, b) in enumerate(((1024, 1024), (1024, 1024))):
        sc = "aue/fc%d" % (i + 1)
        spec += [(sc + "/weights", (a, b)), (sc + "/biases", (b,))] + bn(sc, b)
    spec += [("aue/fc3/weights", (1024, num_point * 3)), ("aue/fc3/biases", (num_point * 3,))]
    return spec
