#!/usr/bin/env python3
"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN PYTHON under oracle/tfstub.

TEST INFRASTRUCTURE.  Runs only in the authoring container (needs /root/reference); the GPU box
and the test-suite only ever see the committed .npz files (inputs by seed + expected outputs).

What is executed, unchanged, from /root/reference:
    models/dpdist_and_aue.py:get_model / get_loss      (module contract, :31-86, :203-204)
    utils/dpdist_util.py:get_3dmfv_tf / local_z / DPDist / get_loss   (:22-141, :850-960, :412-700, :962-980)
    utils/tf_util.py:conv2d                            (:161-228)
with `import tensorflow` resolved to oracle/tfstub/tensorflow (our restatement of the TF 1.14
primitives).  Each fixture stores the float32 evaluation (what TF would
compute in) and a float64 evaluation of the same graph (a roundoff-free target).

Usage:  python oracle/gen_goldens.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from dpdist_amd import synth  # noqa: E402


def load_reference(ref):
    sys.path.insert(0, os.path.join(HERE, "tfstub"))
    sys.path.insert(0, os.path.join(ref, "utils"))
    sys.path.insert(0, os.path.join(ref, "models"))
    import tensorflow as tf            # the stub
    import dpdist_and_aue as MODEL     # the reference, unchanged
    import dpdist_util
    assert os.path.realpath(MODEL.__file__).startswith(os.path.realpath(ref))
    return tf, MODEL, dpdist_util


def run_model(tf, MODEL, pcA, pcB, noise, labels, weights, dtype, m3=512, k=5, mlp=(1024, 1024, 1024),
              want_grads=None):
    """One eager evaluation of the reference graph.  Returns dict of numpy arrays."""
    tf.set_real_dtype(dtype)
    tf.reset_default_graph()
    tf.set_variable_overrides(weights)
    tA = tf.Tensor(torch.tensor(pcA, dtype=dtype, requires_grad=True))
    tB = tf.Tensor(torch.tensor(pcB, dtype=dtype, requires_grad=True))
    tN = tf.Tensor(torch.tensor(noise, dtype=dtype, requires_grad=True))
    tL = tf.Tensor(torch.tensor(labels, dtype=dtype))
    with contextlib.redirect_stdout(io.StringIO()):   # the reference prints tensors while building
        # flags exactly as train_multi_gpu_pc_compare_dist.py:224-229 passes them
        pred, end_points, emb = MODEL.get_model(
            tA, tB, True, bn_decay=None, wd=0.0, bn=0, sig=False, Embedding_Size=m3, pn="3dmfv",
            k=k, localSNmlp=list(mlp), overlap=True, full_fv=True, conv_version=1,
            sigma3dmfv=2.0 * 0.0625, add_noise=tN)
        MODEL.get_loss(pred, end_points, tL, loss_type="l1_dist")
    loss_s = tf.get_collection("loss_samples")[0].v
    loss_p = tf.get_collection("loss_pred")[0].v
    out = {
        "pred_listAB": pred["pred_listAB"].numpy(), "pred_listBA": pred["pred_listBA"].numpy(),
        "loss_samples": loss_s.detach().numpy(), "loss_pred": loss_p.detach().numpy(),
        "_emb": emb, "_vars": tf.stub_variables(), "_named": tf.stub_named(),
    }
    if want_grads == "as_loss":
        g = torch.autograd.grad(loss_p, [tA.v, tB.v, tN.v], allow_unused=True)
        for n, gg in zip(("d_pcA", "d_pcB", "d_noise"), g):
            out[n] = (torch.zeros(pcA.shape, dtype=dtype) if gg is None else gg).numpy()
    elif want_grads == "train":
        names = sorted(tf.stub_variables())
        g = torch.autograd.grad(loss_s, [tf.stub_variables()[n].v for n in names])
        out["_wgrads"] = {n: gg.numpy() for n, gg in zip(names, g)}
    return out


def fx_fv(tf, dpdist_util, out_dir):
    """F1: 3DmFV of four clouds."""
    rng = np.random.default_rng(11)
    N = 64
    c0 = rng.uniform(-1, 1, (N, 3))
    v = rng.standard_normal((N, 3))
    c1 = 0.6 * v / np.linalg.norm(v, axis=1, keepdims=True)
    c2 = np.tile(np.array([[0.3, -0.2, 0.55]]), (N, 1))
    c3 = rng.uniform(-1.3, 1.3, (N, 3))
    pts = np.stack([c0, c1, c2, c3]).astype(np.float32)
    res = {"points": pts}
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        tf.set_real_dtype(dt)
        for m in (8, 5):
            with contextlib.redirect_stdout(io.StringIO()):
                fv = dpdist_util.get_3dmfv_tf(tf.Tensor(torch.tensor(pts, dtype=dt)), n_gaussians=m ** 3,
                                              flatten=False, full_fv=True, normalize=True, sigma=0.125)
            res["fv_m%d_%s" % (m, tag)] = fv.numpy()
    np.savez_compressed(os.path.join(out_dir, "fv_cases.npz"), **res)
    print("fv_cases:", {k: v.shape for k, v in res.items()})


def fx_path(tf, MODEL, out_dir):
    """F2/F5: forward through the whole module contract, B=2 slice of S1, both weight sets;
    plus an all-boundary query cloud."""
    pcA32, pcB32 = synth.s1_random_patches(32, 64, seed=0)
    cases = {"s1": (pcA32[:2], pcB32[:2]), "boundary": synth.boundary_cloud(2, 64, seed=7)}
    for cname, (pcA, pcB) in cases.items():
        noise = np.zeros_like(pcA)
        labels = np.zeros(pcA.shape[:2], np.float32)
        for wk in ("xavier_tf", "wide"):
            W = synth.make_weights(wk)
            res = {"pcA": pcA, "pcB": pcB}
            for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
                o = run_model(tf, MODEL, pcA, pcB, noise, labels, W, dt)
                res["pred_listAB_" + tag] = o["pred_listAB"]
                res["pred_listBA_" + tag] = o["pred_listBA"]
                res["loss_pred_" + tag] = o["loss_pred"]
                if tag == "f32" and wk == "wide":
                    # 8 selected rows of the [B,512,2500] local-window embedding (local_z_3d output)
                    sel = np.array([[0, 0], [0, 73], [0, 219], [0, 511], [1, 7], [1, 64], [1, 300], [1, 448]])
                    eA = o["_emb"]["embedding_A"].numpy()
                    eB = o["_emb"]["embedding_B"].numpy()
                    res["emb_sel"] = sel
                    res["embA_rows"] = eA[sel[:, 0], sel[:, 1]]
                    res["embB_rows"] = eB[sel[:, 0], sel[:, 1]]
                    res["var_names"] = np.array(sorted(o["_vars"]))
                    res["var_shapes"] = np.array([list(o["_vars"][n].v.shape) + [0] * (4 - o["_vars"][n].v.dim())
                                                  for n in sorted(o["_vars"])])
                    res["named_outputs"] = np.array(sorted(o["_named"]))
            fn = "path_fwd_%s_%s.npz" % (cname, wk)
            np.savez_compressed(os.path.join(out_dir, fn), **res)
            print(fn, "AB ch0 mean %.4f max %.4f  zeros %.2f" % (
                res["pred_listAB_f32"][..., 0].mean(), res["pred_listAB_f32"].max(),
                (res["pred_listAB_f32"][..., 0] == 0).mean()))


def fx_bwd(tf, MODEL, out_dir):
    """F3: losses + gradients (as-loss mode: d loss_pred / d inputs; training mode: d loss_samples / d weights)."""
    pcA, pcB, lab = synth.s2_modelnet_shaped(2, 64, seed=100)
    rng = np.random.default_rng(5)
    noise = (rng.standard_normal(pcA.shape) * 0.01).astype(np.float32)
    W = synth.make_weights("wide")
    res = {"pcA": pcA, "pcB": pcB, "labels": lab, "noise": noise}
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        o = run_model(tf, MODEL, pcA, pcB, noise, lab, W, dt, want_grads="as_loss")
        for n in ("pred_listAB", "pred_listBA", "loss_samples", "loss_pred", "d_pcA", "d_pcB", "d_noise"):
            res[n + "_" + tag] = o[n]
        o = run_model(tf, MODEL, pcA, pcB, noise, lab, W, dt, want_grads="train")
        for n, g in o["_wgrads"].items():
            short = n.split("/")[-2][-1] + ("w" if n.endswith("weights") else "b")   # e.g. '1w'
            g2 = g.reshape(-1, g.shape[-1]) if g.ndim == 4 else g
            res["g%s_norm_%s" % (short, tag)] = np.array(np.sqrt((g2.astype(np.float64) ** 2).sum()))
            if g.ndim == 4:
                res["g%s_corner_%s" % (short, tag)] = g2[:16, :16].copy()
                res["g%s_tail_%s" % (short, tag)] = g2[-16:, -16:].copy()
                res["g%s_colsum_%s" % (short, tag)] = g2.sum(0)
                res["g%s_rowsum_%s" % (short, tag)] = g2.sum(1)
            else:
                res["g%s_%s" % (short, tag)] = g2.copy()
    np.savez_compressed(os.path.join(out_dir, "path_bwd_s2_wide.npz"), **res)
    print("path_bwd: loss_samples %.6f loss_pred %.6f |d_pcA| %.4e |d_pcB| %.4e" % (
        res["loss_samples_f32"], res["loss_pred_f32"], np.abs(res["d_pcA_f32"]).max(), np.abs(res["d_pcB_f32"]).max()))


def fx_small_mlp(tf, MODEL, out_dir):
    """Module contract with localSNmlp=[64,64,64] and m=5 (125 Gaussians, non-dyadic cell edges)."""
    pcA, pcB = synth.s1_random_patches(2, 64, seed=21)
    noise = np.zeros_like(pcA)
    labels = np.zeros(pcA.shape[:2], np.float32)
    for m in (8, 5):
        W = synth.make_weights("wide", mlp=(64, 64, 64))
        res = {"pcA": pcA, "pcB": pcB}
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            o = run_model(tf, MODEL, pcA, pcB, noise, labels, W, dt, m3=m ** 3, mlp=(64, 64, 64))
            res["pred_listAB_" + tag] = o["pred_listAB"]
            res["pred_listBA_" + tag] = o["pred_listBA"]
        np.savez_compressed(os.path.join(out_dir, "path_fwd_mlp64_m%d.npz" % m), **res)
        print("mlp64 m=%d AB mean %.4f" % (m, res["pred_listAB_f32"][..., 0].mean()))


# ------------------------------------------------------------------------------------------------------------------
# f-rows: the consumers of the boundary.  Functions that live in modules which cannot be imported (the
# trainer and the evaluation script run argparse / dataset / log-directory code at import time and need h5py / trimesh)
# are taken from the reference FILE by name with `ast` and executed unchanged in a namespace holding the stub.
# ------------------------------------------------------------------------------------------------------------------
def extract_functions(path, names, namespace):
    import ast
    src = open(path).read()
    tree = ast.parse(src)
    found = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
            exec(code, namespace)
            found[node.name] = namespace[node.name]
    missing = set(names) - set(found)
    assert not missing, missing
    return found


def load_registration(ref):
    sys.path.insert(0, os.path.join(ref, "pcrnet-registration"))
    sys.path.insert(0, os.path.join(ref, "pcrnet-registration", "models"))
    import importlib
    # the registration code has its OWN utils/tf_util.py: import it under the name the model expects
    for m in ("tf_util",):
        sys.modules.pop(m, None)
    sys.path.insert(0, os.path.join(ref, "pcrnet-registration", "utils"))
    with contextlib.redirect_stdout(io.StringIO()):
        helper = importlib.import_module("helper")
        ipcr = importlib.import_module("ipcr_model")
    assert os.path.realpath(helper.__file__).startswith(os.path.realpath(ref))
    return helper, ipcr


def fx_pose(tf, ref, out_dir):
    """F-f2: quaternion transform, pose normalisation, the pose network (inference branch) and the error metric."""
    helper, ipcr = load_registration(ref)
    import numpy as _np
    import transforms3d
    import transforms3d.euler as t3d
    ns = {"np": _np, "t3d": t3d, "transforms3d": transforms3d}
    fe = extract_functions(os.path.join(ref, "pcrnet-registration", "results_itrPCRNet_no_stop.py"), ["find_errors"], ns)["find_errors"]
    rng = np.random.default_rng(31)
    B, N = 4, 64
    data = rng.uniform(-0.8, 0.8, (B, N, 3)).astype(np.float32)
    quat = rng.standard_normal((B, 4)).astype(np.float32)
    quat[:2] /= np.linalg.norm(quat[:2], axis=1, keepdims=True)       # two unit quaternions, two un-normalised ones
    trans = rng.uniform(-0.2, 0.2, (B, 3)).astype(np.float32)
    raw7 = (rng.standard_normal((B, 7)) * 1.5).astype(np.float32)
    src = rng.uniform(-0.8, 0.8, (B, N, 3)).astype(np.float32)
    tmpl = rng.uniform(-0.8, 0.8, (B, N, 3)).astype(np.float32)
    from dpdist_amd import synth
    W = synth.make_named_weights(synth.pose_net_spec(1024), seed=41)
    res = {"data": data, "quat": quat, "trans": trans, "raw7": raw7, "source": src, "template": tmpl, "weights_seed": np.array(41)}
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        tf.set_real_dtype(dt)
        tf.reset_default_graph()
        tf.set_variable_overrides(W)
        T = lambda a: tf.Tensor(torch.tensor(a, dtype=dt))   # noqa: E731
        with contextlib.redirect_stdout(io.StringIO()):
            res["transformed_" + tag] = helper.transformation_quat_tensor(T(data), T(quat), T(trans)).numpy()
            res["quat_normalize45_" + tag] = ipcr.quat_normalize(T(raw7), rot_lim=45.0).numpy()
            res["quat_normalize10_" + tag] = ipcr.quat_normalize(T(raw7), rot_lim=10.0).numpy()
            is_training = tf.constant(False)
            fs, ft = ipcr.get_model(T(src), T(tmpl), is_training, bn_decay=None, PN=True, POOL="max", out_features=1024)
            res["feat_source_" + tag], res["feat_template_" + tag] = fs.numpy(), ft.numpy()
            res["pose_raw_" + tag] = ipcr.get_pose(fs, ft, is_training, bn_decay=None, lim_rot=False).numpy()
            tf.reset_default_graph()
            tf.set_variable_overrides(W)
            fs, ft = ipcr.get_model(T(src), T(tmpl), is_training, bn_decay=None, PN=True, POOL="max", out_features=1024)
            res["pose_lim45_" + tag] = ipcr.get_pose(fs, ft, is_training, bn_decay=None, lim_rot=45.0).numpy()
    # the evaluation metric (results_itrPCRNet_no_stop.py:112-133): poses are (x,y,z, rx,ry,rz) in radians
    gt = np.concatenate([rng.uniform(-0.3, 0.3, (8, 3)), rng.uniform(-0.8, 0.8, (8, 3))], 1)
    fin = gt + np.concatenate([rng.normal(0, 0.02, (8, 3)), rng.normal(0, 0.15, (8, 3))], 1)
    errs = np.array([fe(gt[i], fin[i]) for i in range(8)])
    res.update({"err_gt_pose": gt, "err_final_pose": fin, "err_translation": errs[:, 0], "err_rotation_deg": errs[:, 1]})
    # helper.transformation_quat2mat (:309-329) and find_final_pose (:331-345): numpy + transforms3d
    poses = np.concatenate([trans, quat], 1).astype(np.float64)
    TR = np.tile(np.eye(4), (B, 1, 1))
    TR, moved = helper.transformation_quat2mat(poses.reshape(1, B, 7), TR, data.astype(np.float64).copy())
    res.update({"quat2mat_T": TR, "quat2mat_data": moved, "final_pose": helper.find_final_pose(TR)})
    np.savez_compressed(os.path.join(out_dir, "pose_cases.npz"), **res)
    print("pose_cases:", {k: np.asarray(v).shape for k, v in res.items()})


def fx_chamfer(tf, ref, out_dir):
    """F-f4: pairwise_diff / chmafer_dist of the trainer (train_multi_gpu_pc_compare_dist.py:891-916) + autograd gradients."""
    ns = {"tf": tf, "np": np}
    fns = extract_functions(os.path.join(ref, "train_multi_gpu_pc_compare_dist.py"), ["pairwise_diff", "chmafer_dist"], ns)
    rng = np.random.default_rng(51)
    pc = rng.uniform(-0.8, 0.8, (3, 64, 3)).astype(np.float32)
    rec = (pc[:, :48] + rng.normal(0, 0.05, (3, 48, 3))).astype(np.float32)
    rec[0, 5] = rec[0, 6]                         # duplicate points: ties in the minima
    res = {"pc": pc, "rec_pc": rec}
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        tf.set_real_dtype(dt)
        a = tf.Tensor(torch.tensor(pc, dtype=dt, requires_grad=True))
        b = tf.Tensor(torch.tensor(rec, dtype=dt, requires_grad=True))
        with contextlib.redirect_stdout(io.StringIO()):
            loss = fns["chmafer_dist"](a, b)
            sq = fns["pairwise_diff"](b, a)
        ga, gb = torch.autograd.grad(loss.v, [a.v, b.v])
        res.update({"loss_" + tag: loss.numpy(), "d_pc_" + tag: ga.numpy(), "d_rec_" + tag: gb.numpy(),
                    "sqdist_rec_pc_" + tag: sq.numpy()})
    np.savez_compressed(os.path.join(out_dir, "chamfer_cases.npz"), **res)
    print("chamfer_cases: loss %.6f" % res["loss_f32"])


def fx_aue_pn(tf, MODEL, out_dir):
    """F-f4: the PointNet autoencoder get_model_aue_pn (models/dpdist_and_aue.py:88-145), training and inference branches of
    its batch norm, weights by seed (synth.aue_pn_spec)."""
    from dpdist_amd import synth
    rng = np.random.default_rng(61)
    pts = rng.uniform(-0.8, 0.8, (4, 64, 3)).astype(np.float32)
    W = synth.make_named_weights(synth.aue_pn_spec(64), seed=62, scale=1.0)
    res = {"points": pts, "weights_seed": np.array(62)}
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        for train in (False, True):
            tf.set_real_dtype(dt)
            tf.reset_default_graph()
            tf.set_variable_overrides(W)
            x = tf.Tensor(torch.tensor(pts, dtype=dt, requires_grad=True))
            with contextlib.redirect_stdout(io.StringIO()):
                out = MODEL.get_model_aue_pn(x, tf.constant(train), bn_decay=0.9)
            key = "train" if train else "eval"
            res["out_%s_%s" % (key, tag)] = out.numpy()
            if train:   # moving statistics after ONE training-mode forward (updates_collections=None: updated in place)
                v = tf.stub_variables()
                res["mm_conv1_" + tag] = v["aue/conv1/bn/moving_mean"].numpy()
                res["mv_conv1_" + tag] = v["aue/conv1/bn/moving_variance"].numpy()
                res["mm_fc2_" + tag] = v["aue/fc2/bn/moving_mean"].numpy()
                res["mv_fc2_" + tag] = v["aue/fc2/bn/moving_variance"].numpy()
                g, = torch.autograd.grad(out.v.sum(), [x.v])
                res["d_points_train_" + tag] = g.numpy()
    res["var_names"] = np.array(sorted(tf.stub_variables()))
    np.savez_compressed(os.path.join(out_dir, "aue_pn_cases.npz"), **res)
    print("aue_pn_cases: |out| %.4f, %d variables" % (np.abs(res["out_eval_f32"]).mean(), len(res["var_names"])))


def fx_step(tf, MODEL, ref, out_dir):
    """F4: 3 optimizer steps with the trainer's own assembly -- two towers on batch slices, compute_gradients per
    tower, average_gradients (:936-974), get_learning_rate (:976-990: staircase decay + floor), AdamOptimizer.apply_gradients
    with the global step (:216,301) -- on a fixed batch, decoder 64-64-64 so that whole weight tensors fit the fixture."""
    mlp = (64, 64, 64)
    GB, towers = 4, 2
    pcA, pcB, lab = synth.s2_modelnet_shaped(GB, 64, seed=100)
    W0 = synth.make_weights("wide", mlp=mlp)
    consts = {"BASE_LEARNING_RATE": 1e-3, "DECAY_STEP": 2, "DECAY_RATE": 0.5}     # decays after step 2: the staircase is exercised
    res = {"pcA": pcA, "pcB": pcB, "labels": lab, "base_lr": np.array(1e-3), "decay_step": np.array(2), "decay_rate": np.array(0.5)}
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        ns = {"tf": tf, "np": np, **consts}
        fns = extract_functions(os.path.join(ref, "train_multi_gpu_pc_compare_dist.py"), ["average_gradients", "get_learning_rate"], ns)
        tf.set_real_dtype(dt)
        tf.reset_default_graph()
        W = {k: v.copy() for k, v in W0.items()}
        batch = tf.Tensor(torch.zeros((), dtype=torch.float64), name="batch")                  # global step (:205-207)
        with contextlib.redirect_stdout(io.StringIO()):
            opt = tf.train.AdamOptimizer(lambda: fns["get_learning_rate"](batch, consts["BASE_LEARNING_RATE"]))
        losses, lrs = [], []
        for step in range(3):
            tf.reset_default_graph()
            tf.set_variable_overrides(W)
            tower_grads, tower_loss = [], []
            per = GB // towers
            for t in range(towers):                                                        # tf.slice per tower (:241-251)
                sl = slice(t * per, (t + 1) * per)
                with tf.variable_scope(tf.get_variable_scope(), reuse=(t > 0) or None):
                    with contextlib.redirect_stdout(io.StringIO()):
                        pred, end_points, _ = MODEL.get_model(
                            tf.Tensor(torch.tensor(pcA[sl], dtype=dt)), tf.Tensor(torch.tensor(pcB[sl], dtype=dt)), True, bn_decay=None,
                            wd=0.0, bn=0, sig=False, Embedding_Size=512, pn="3dmfv", k=5, localSNmlp=list(mlp), overlap=True,
                            full_fv=True, conv_version=1, sigma3dmfv=2.0 * 0.0625, add_noise=tf.Tensor(torch.zeros(per, 64, 3, dtype=dt)))
                        MODEL.get_loss(pred, end_points, tf.Tensor(torch.tensor(lab[sl], dtype=dt)), loss_type="l1_dist")
                loss_t = tf.get_collection("loss_samples")[-1]
                names = sorted(tf.stub_variables())
                tower_grads.append(opt.compute_gradients(loss_t, [tf.stub_variables()[n] for n in names]))
                tower_loss.append(loss_t)
            with contextlib.redirect_stdout(io.StringIO()):
                grads = fns["average_gradients"](tower_grads)
                lrs.append(float(fns["get_learning_rate"](batch, consts["BASE_LEARNING_RATE"]).numpy()))
                opt.apply_gradients(grads, global_step=batch)
            losses.append(float(tf.reduce_mean(tower_loss).numpy()))
            W = {n: tf.stub_variables()[n].numpy().copy() for n in names}
        res["loss_samples_" + tag] = np.array(losses)
        res["lr_" + tag] = np.array(lrs)
        for n in sorted(W):
            short = n.split("/")[-2][-1] + ("w" if n.endswith("weights") else "b")
            res["final_%s_%s" % (short, tag)] = W[n]
    np.savez_compressed(os.path.join(out_dir, "step_adam.npz"), **res)
    print("step_adam: losses", res["loss_samples_f32"], "lr", res["lr_f32"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default="", help="comma list of fixtures: fv,path,bwd,mlp64,chamfer,aue,step,pose")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(8)
    tf, MODEL, dpdist_util = load_reference(a.ref)
    todo = set(a.only.split(",")) if a.only else None
    want = lambda n: todo is None or n in todo   # noqa: E731
    if want("fv"): fx_fv(tf, dpdist_util, a.out)
    if want("path"): fx_path(tf, MODEL, a.out)
    if want("bwd"): fx_bwd(tf, MODEL, a.out)
    if want("mlp64"): fx_small_mlp(tf, MODEL, a.out)
    if want("chamfer"): fx_chamfer(tf, a.ref, a.out)
    if want("aue"): fx_aue_pn(tf, MODEL, a.out)
    if want("step"): fx_step(tf, MODEL, a.ref, a.out)
    if want("pose"): fx_pose(tf, a.ref, a.out)      # last: it re-binds the module name `tf_util` to the registration code's copy


if __name__ == "__main__":
    main()
