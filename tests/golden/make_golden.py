#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz by running the REFERENCE ITSELF (imported unmodified from
/root/reference) on seeded inputs.  Runs only in the build container; the fixtures it writes are
data (inputs + the reference's outputs) and are committed so that the GPU box -- where
/root/reference does not exist -- can check both the oracle and the HIP path against them.

How the reference is made importable (nothing is copied from it):
  * sys.path gets tests/golden/_absl_shim (a stand-in ``absl`` written for this repo: absl-py is
    not installed and there is no network) and /root/reference; bytecode writing is disabled
    because the reference tree is read-only.
  * ``config.config`` then defines every flag with its default; ``network.fs_net_repo.*`` import
    as they are.
  * The Chamfer CUDA extension cannot be built here (no CUDA).  The reference's own CPU
    statement of Chamfer, ``losses/metrics/CD/chamfer_python.py:distChamfer`` (declared
    equivalent by the reference's unit test losses/metrics/CD/unit_test.py:22-33), is loaded by
    file path and used to produce the Chamfer vectors.

Weights: the reference ships no checkpoint, so ``tgpose_amd.init_weights.seeded_state_dict`` is
loaded into the reference with ``load_state_dict(strict=True)``; fixtures store only the seed.

Usage:  python tests/golden/make_golden.py   (from the repo root)
"""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "_absl_shim"), REF]
sys.path.append(ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(8)


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


from config.config import *  # noqa: E402,F401,F403  (defines the flags on the shim's FLAGS)
import absl.flags as flags  # noqa: E402
import network.fs_net_repo.gcn3d as ref_gcn  # noqa: E402
from network.fs_net_repo.PoseNet9D import PoseNet9D as RefPoseNet9D  # noqa: E402

FLAGS = flags.FLAGS
ref_chamfer = _load_by_path("ref_chamfer_python", os.path.join(REF, "losses/metrics/CD/chamfer_python.py"))
iw = _load_by_path("tgp_init_weights", os.path.join(ROOT, "tg-pose_amd", "init_weights.py"))


def synth_points(B, N, seed):
    """SURVEY.md 8(d) synthetic clouds: 0.1*randn + per-object camera-frame offset."""
    g = torch.Generator().manual_seed(seed)
    pts = 0.1 * torch.randn(B, N, 3, generator=g)
    off = torch.rand(B, 1, 3, generator=g) * torch.tensor([0.4, 0.4, 1.0]) + torch.tensor([-0.2, -0.2, 0.5])
    obj = torch.randint(0, 6, (B, 1), generator=g).float()
    return (pts + off).contiguous(), obj


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print("wrote %-28s %8.1f KB" % (name, os.path.getsize(path) / 1024.0))


def small_idx(t):
    t = t.detach().cpu().numpy()
    assert t.max() < 32767
    return t.astype(np.int16)


# ----------------------------------------------------------------------------- operators
def gen_knn():
    g = torch.Generator().manual_seed(11)
    xyz = 0.1 * torch.randn(2, 300, 3, generator=g)
    src = xyz[:, torch.randperm(300, generator=g)[:75], :].contiguous()
    bottle = torch.from_numpy(np.load(os.path.join(REF, "obj_model/points_bottle.npy"))).float()[None]
    bottle = bottle - bottle.mean(dim=1, keepdim=True)
    feat = torch.relu(torch.randn(2, 200, 128, generator=g) * 0.7 + 0.2)      # post-ReLU like features
    feat256 = torch.relu(torch.randn(1, 96, 256, generator=g) * 0.5 + 0.1)
    dup = torch.cat([xyz[:1, :100], xyz[:1, :100], xyz[:1, :100]], dim=1)       # 3x tiled cloud: exact ties
    save("knn_ops.npz",
         xyz=xyz, xyz_k20=small_idx(ref_gcn.get_neighbor_index(xyz, 20)), xyz_k4=small_idx(ref_gcn.get_neighbor_index(xyz, 4)),
         src=src, nearest=small_idx(ref_gcn.get_nearest_index(xyz, src)),
         bottle=bottle, bottle_k20=small_idx(ref_gcn.get_neighbor_index(bottle, 20)),
         feat=feat, feat_k20=small_idx(ref_gcn.get_neighbor_index(feat, 20)),
         feat256=feat256, feat256_k8=small_idx(ref_gcn.get_neighbor_index(feat256, 8)),
         dup=dup, dup_k8=small_idx(ref_gcn.get_neighbor_index(dup, 8)))


def gen_layers():
    """Single layers of gcn3d.py run stand-alone on a small cloud (B=2, n=160)."""
    sd = iw.seeded_state_dict(3)
    g = torch.Generator().manual_seed(5)
    xyz = 0.1 * torch.randn(2, 160, 3, generator=g)
    S = 7
    conv0 = ref_gcn.HSlayer_surface(kernel_num=128, support_num=S)
    conv1 = ref_gcn.HS_layer(128, 128, support_num=S)
    pre = "face_all.encoder."
    conv0.load_state_dict({k[len(pre + "conv_0."):]: v for k, v in sd.items() if k.startswith(pre + "conv_0.")})
    conv1.load_state_dict({k[len(pre + "conv_1."):]: v for k, v in sd.items() if k.startswith(pre + "conv_1.")})
    with torch.no_grad():
        f0 = conv0(xyz, 20)
        fin = torch.relu(f0)
        idx_f = ref_gcn.get_neighbor_index(fin, 20)
        f1 = conv1(xyz, fin, 20)
        torch.manual_seed(77)
        pool = ref_gcn.Pool_layer(4, 4)
        vp, fp = pool(xyz, fin)
        torch.manual_seed(77)
        sample = torch.randperm(160)[:40]
    save("layers.npz", seed=np.int64(3), xyz=xyz, conv0_out=f0, conv1_in=fin, conv1_rf_idx=small_idx(idx_f),
         conv1_out=f1, pool_sample=small_idx(sample), pool_v=vp, pool_f=fp)


# ----------------------------------------------------------------------------- full forward
def run_reference(net, pts, obj, seed, train, enable_proj=False, grad=False):
    """Reference forward with every get_neighbor_index / get_nearest_index result recorded."""
    knn_rec, nn_rec = [], []
    o_knn, o_nn = ref_gcn.get_neighbor_index, ref_gcn.get_nearest_index

    def w_knn(v, k):
        r = o_knn(v, k)
        knn_rec.append(r)
        return r

    def w_nn(t, s):
        r = o_nn(t, s)
        nn_rec.append(r)
        return r

    ref_gcn.get_neighbor_index, ref_gcn.get_nearest_index = w_knn, w_nn
    try:
        FLAGS.train = train
        torch.manual_seed(seed)
        with torch.set_grad_enabled(grad):
            out = net(pts, obj, enable_proj=True) if enable_proj else net(pts, obj)
    finally:
        ref_gcn.get_neighbor_index, ref_gcn.get_nearest_index = o_knn, o_nn
    names = ["conv_0.rf", "conv_0.orl_xyz", "conv_1.rf", "conv_1.orl_xyz", "pool_1.xyz", "conv_2.rf",
             "conv_2.orl_xyz", "conv_3.rf", "conv_3.orl_xyz", "pool_2.xyz", "conv_4.rf", "conv_4.orl_xyz"]
    assert len(knn_rec) == len(names) and len(nn_rec) == 2
    idx = {"face_all.encoder." + n: r for n, r in zip(names, knn_rec)}
    idx["face_all.encoder.up_1"], idx["face_all.encoder.up_2"] = nn_rec
    return out, idx


def sample_indices(N, seed):
    """The two randperm draws Face_Enc.forward makes (gcn3d.py:242) after torch.manual_seed(seed)."""
    torch.manual_seed(seed)
    i1 = torch.randperm(N)[: int(N / 4)]
    i2 = torch.randperm(i1.numel())[: int(i1.numel() / 4)]
    return i1, i2


def gen_forward(name, B, N, wseed, pseed, fseed, pts=None, obj=None, keep_feat_rows=64):
    sd = iw.seeded_state_dict(wseed)
    net = RefPoseNet9D().eval()
    net.load_state_dict(sd, strict=True)
    if pts is None:
        pts, obj = synth_points(B, N, pseed)
    out_test, idx = run_reference(net, pts, obj, fseed, train=0)
    out_train, idx2 = run_reference(net, pts, obj, fseed, train=1)
    for k in idx:
        assert torch.equal(idx[k], idx2[k])
    i1, i2 = sample_indices(N, fseed)
    arrays = dict(weight_seed=np.int64(wseed), forward_seed=np.int64(fseed), points=pts, obj_id=obj,
                  sample_idx_1=small_idx(i1), sample_idx_2=small_idx(i2))
    for k, v in out_test.items():
        arrays["test." + k] = v
    for k, v in out_train.items():
        if k == "feat":                       # (B,N,1286) is large: keep a row slice + per-row sums
            arrays["train.feat_rows"] = v[:, :keep_feat_rows].contiguous()
            arrays["train.feat_rowsum"] = v.double().sum(dim=2).float()
            arrays["train.feat_colsum"] = v.double().sum(dim=1).float()
        else:
            arrays["train." + k] = v
    for k, v in idx.items():
        arrays["idx." + k] = small_idx(v)
    save(name, **arrays)


def gen_forward_train(name, B, N, wseed, pseed, fseed, steps=2):
    """The reference in TRAINING mode (net.train(): batch-statistics BatchNorm that moves its running statistics), as
    the trainer runs it (trainer/RL_TDA.py:116-120 never calls .eval()).  Dropout is set to p = 0: its mask comes from
    the host generator and cannot be reproduced on another device.  `steps` consecutive forwards on the same batch;
    the fixture holds the last step's outputs and every BatchNorm buffer after the last step."""
    sd = iw.seeded_state_dict(wseed)
    net = RefPoseNet9D().train()
    net.load_state_dict(sd, strict=True)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    pts, obj = synth_points(B, N, pseed)
    for _ in range(steps):
        out, idx = run_reference(net, pts, obj, fseed, train=1)
    i1, i2 = sample_indices(N, fseed)
    arrays = dict(weight_seed=np.int64(wseed), forward_seed=np.int64(fseed), steps=np.int64(steps), points=pts, obj_id=obj,
                  sample_idx_1=small_idx(i1), sample_idx_2=small_idx(i2))
    for k, v in out.items():
        if k == "feat":
            arrays["train.feat_rows"] = v[:, :32].contiguous()
            arrays["train.feat_rowsum"] = v.double().sum(dim=2).float()
        else:
            arrays["train." + k] = v
    for k, v in idx.items():            # the graphs are the same at every step: training-mode BN ignores the running statistics
        arrays["idx." + k] = small_idx(v)
    for k, v in net.state_dict().items():
        if "running_" in k or "num_batches" in k:
            arrays["bn." + k] = v
    save(name, **arrays)


def gen_proj(name, B, N, wseed, pseed, fseed):
    """enable_proj=True (FaceRecon.py:32-35,80-84; PoseNet9D.py:33,39,49): feat_global = max over points of proj_layer(feat).
    The reference in eval mode (running statistics), and in training mode (batch statistics; dropout p = 0) with the gradients of
    sum(feat_global ** 2) with respect to the projection head's own parameters and the moved BatchNorm buffers."""
    sd = iw.seeded_state_dict(wseed)
    pts, obj = synth_points(B, N, pseed)
    net = RefPoseNet9D().eval()
    net.load_state_dict(sd, strict=True)
    out_eval, idx = run_reference(net, pts, obj, fseed, train=1, enable_proj=True)
    plain, _ = run_reference(net, pts, obj, fseed, train=1)
    assert not torch.equal(plain["feat_global"], out_eval["feat_global"])
    net = RefPoseNet9D().train()
    net.load_state_dict(sd, strict=True)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    out_tr, idx2 = run_reference(net, pts, obj, fseed, train=1, enable_proj=True, grad=True)
    (out_tr["feat_global"] ** 2).sum().backward()
    i1, i2 = sample_indices(N, fseed)
    pl = "face_all.encoder.proj_layer."
    arrays = dict(weight_seed=np.int64(wseed), forward_seed=np.int64(fseed), points=pts, obj_id=obj,
                  sample_idx_1=small_idx(i1), sample_idx_2=small_idx(i2))
    arrays["eval.feat_global"] = out_eval["feat_global"]
    arrays["train.feat_global"] = out_tr["feat_global"].detach()
    for k, v in idx.items():
        arrays["idx." + k] = small_idx(v)
    for k, v in idx2.items():
        arrays["idx_train." + k] = small_idx(v)
    params = dict(net.named_parameters())
    for k in ("0.weight", "1.weight", "1.bias", "3.weight"):
        g = params[pl + k].grad
        arrays["grad." + k] = g if g.numel() < 4096 else g.reshape(g.shape[0], -1)[:, :16].contiguous()     # (a column slice of the big ones)
        arrays["gradnorm." + k] = g.double().norm().float()
    for k, v in net.state_dict().items():
        if k.startswith(pl) and ("running_" in k or "num_batches" in k):
            arrays["bn." + k] = v
    save(name, **arrays)


def gen_backward(name, B, N, wseed, pseed, fseed):
    """loss.backward() through the reference in training mode (dropout p = 0): loss = sum_k <out_k, w_k> with seeded random
    weights w_k on every output tensor.  The fixture keeps, per parameter, the gradient's L2 norm, its sum and 16 evenly
    spaced entries (the full gradients are 110 MB), plus the graphs and the subsample that produced them."""
    sd = iw.seeded_state_dict(wseed)
    net = RefPoseNet9D().train()
    net.load_state_dict(sd, strict=True)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    pts, obj = synth_points(B, N, pseed)
    knn_rec, nn_rec = [], []
    o_knn, o_nn = ref_gcn.get_neighbor_index, ref_gcn.get_nearest_index
    ref_gcn.get_neighbor_index = lambda v, k: (knn_rec.append(o_knn(v, k)), knn_rec[-1])[1]
    ref_gcn.get_nearest_index = lambda t, s_: (nn_rec.append(o_nn(t, s_)), nn_rec[-1])[1]
    try:
        FLAGS.train = 1
        torch.manual_seed(fseed)
        out = net(pts, obj)
    finally:
        ref_gcn.get_neighbor_index, ref_gcn.get_nearest_index = o_knn, o_nn
    gen = torch.Generator().manual_seed(fseed)
    scale = dict(feat=1e-3, recon=1e-2, h1=1e-2, h2=1e-2, feat_global=1e-2)
    keys = ["recon", "p_green_R", "p_red_R", "f_green_R", "f_red_R", "Pred_T", "Pred_s", "h1", "h2", "feat", "feat_global"]
    weights = {k: torch.randn(out[k].shape, generator=gen) * scale.get(k, 1.0) for k in keys}
    sum((out[k] * weights[k]).sum() for k in keys).backward()
    names = ["conv_0.rf", "conv_0.orl_xyz", "conv_1.rf", "conv_1.orl_xyz", "pool_1.xyz", "conv_2.rf",
             "conv_2.orl_xyz", "conv_3.rf", "conv_3.orl_xyz", "pool_2.xyz", "conv_4.rf", "conv_4.orl_xyz"]
    i1, i2 = sample_indices(N, fseed)
    arrays = dict(weight_seed=np.int64(wseed), forward_seed=np.int64(fseed), points=pts, obj_id=obj,
                  sample_idx_1=small_idx(i1), sample_idx_2=small_idx(i2))
    for n_, r in zip(names, knn_rec):
        arrays["idx.face_all.encoder." + n_] = small_idx(r)
    arrays["idx.face_all.encoder.up_1"], arrays["idx.face_all.encoder.up_2"] = small_idx(nn_rec[0]), small_idx(nn_rec[1])
    for k, p in net.named_parameters():
        if p.grad is None:
            continue
        gflat = p.grad.reshape(-1)
        pick = torch.linspace(0, gflat.numel() - 1, 16).long()
        arrays["grad." + k] = torch.cat([gflat.norm().view(1), gflat.double().sum().float().view(1), gflat[pick]])
    save(name, **arrays)


# ----------------------------------------------------------------------------- Chamfer
def gen_chamfer():
    g = torch.Generator().manual_seed(21)
    a = torch.rand(4, 100, 3, generator=g)           # the reference's own unit-test shapes,
    b = torch.rand(4, 200, 3, generator=g)           # losses/metrics/CD/unit_test.py:15-16
    d1, d2, i1, i2 = ref_chamfer.distChamfer(a, b)
    cats = ["bottle", "bowl", "camera", "can", "laptop", "mug"]
    prior = torch.stack([torch.from_numpy(np.load(os.path.join(REF, "obj_model/points_%s.npy" % c))).float() for c in cats])
    noisy = prior[:, torch.randperm(1024, generator=g)[:1000]] + 0.01 * torch.randn(6, 1000, 3, generator=g)
    e1, e2, j1, j2 = ref_chamfer.distChamfer(noisy, prior)
    save("chamfer.npz", a=a, b=b, dist1=d1, dist2=d2, idx1=small_idx(i1), idx2=small_idx(i2),
         noisy=noisy, prior=prior, p_dist1=e1, p_dist2=e2, p_idx1=small_idx(j1), p_idx2=small_idx(j2))


# ----------------------------------------------------------------------------- density-aware Chamfer loss
def load_reference_loss():
    """Import losses/TDA_loss_sym_recon.py unmodified.  Its two import-time dependencies that cannot exist here
    are pre-seeded in sys.modules: `losses.chamfer3D.dist_chamfer_3D` (JIT-builds CUDA; replaced by a module whose
    chamfer_3DDist delegates to the reference's own losses/metrics/CD/chamfer_python.distChamfer) and
    `tools.geom_utils` (source missing from the reference tree, only py3.8 bytecode; `batch_dot` restated from
    the disassembly recorded in SURVEY.md 8c)."""
    import types
    cham_pkg = types.ModuleType("losses.chamfer3D")
    cham_pkg.__path__ = []
    cham_mod = types.ModuleType("losses.chamfer3D.dist_chamfer_3D")

    class chamfer_3DDist(torch.nn.Module):
        def forward(self, a, b):
            return ref_chamfer.distChamfer(a, b)

    cham_mod.chamfer_3DDist = chamfer_3DDist
    cham_pkg.dist_chamfer_3D = cham_mod
    geom = types.ModuleType("tools.geom_utils")

    def batch_dot(a, b, keepdim=False):
        r = torch.matmul(a.unsqueeze(-2), b.unsqueeze(-1)).squeeze(-1)
        return r if keepdim else r.squeeze(-1)

    geom.batch_dot = batch_dot
    sys.modules["losses.chamfer3D"] = cham_pkg
    sys.modules["losses.chamfer3D.dist_chamfer_3D"] = cham_mod
    sys.modules.setdefault("tools.geom_utils", geom)
    import losses.TDA_loss_sym_recon as ref_loss
    return ref_loss


def gen_dcd():
    ref_loss = load_reference_loss()
    g = torch.Generator().manual_seed(31)
    cats = ["bottle", "bowl", "camera", "can", "laptop", "mug"]
    prior = torch.stack([torch.from_numpy(np.load(os.path.join(REF, "obj_model/points_%s.npy" % c))).float() for c in cats])
    B = 6
    # a plausible network state: noisy, posed copies of the priors as "recon", predicted axes near the true ones
    def rand_rot(n):
        q = torch.randn(n, 4, generator=g)
        q = q / q.norm(dim=1, keepdim=True)
        w, x, y, z = q.unbind(1)
        return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                            2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                            2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).view(n, 3, 3)
    gR = rand_rot(B)
    t = torch.randn(B, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 1.0])
    s = torch.rand(B, 3, generator=g) * 0.5 + 0.75
    sel = torch.randperm(1024, generator=g)[:1028 - 1024 + 1000]
    src = prior[:, torch.randint(0, 1024, (1028,), generator=g)]
    recon = torch.matmul(src / s.unsqueeze(1), gR.transpose(1, 2)) + t.unsqueeze(1) + 0.004 * torch.randn(B, 1028, 3, generator=g)
    p_g = gR[:, :, 1] + 0.05 * torch.randn(B, 3, generator=g)
    p_r = gR[:, :, 0] + 0.05 * torch.randn(B, 3, generator=g)
    p_g, p_r = p_g / p_g.norm(dim=1, keepdim=True), p_r / p_r.norm(dim=1, keepdim=True)
    f_g, f_r = torch.rand(B, generator=g) * 0.5 + 0.4, torch.rand(B, generator=g) * 0.5 + 0.4
    sym = torch.zeros(B, 4)
    sym[0, 0] = sym[1, 0] = sym[3, 0] = 1                     # bottle, bowl, can are symmetric about y
    loss_mod = ref_loss.TDA_loss()
    with torch.no_grad():
        dcd = ref_loss.calc_dcd(recon, prior, alpha=70, n_lambda=0.3)
        cd_p, cd_t = ref_loss.calc_cd(recon, prior)
        r_dcd = loss_mod.R_DCD(prior, recon, gR, p_g, f_g, p_r, f_r, t, s, sym)
        ny, nx = ref_loss.get_vertical_rot_vec_in_batch(f_g, f_r, p_g, p_r)
        pR = ref_loss.get_rot_mat_y_first(ny, nx)
    # gradient of mean(calc_dcd) w.r.t. the reconstruction (through exp(-alpha d) only: the weights are detached)
    rr = recon.clone().requires_grad_(True)
    ref_loss.calc_dcd(rr, prior, alpha=70, n_lambda=0.3).mean().backward()
    save("dcd.npz", recon=recon, prior=prior, gR=gR, t=t, s=s, p_g=p_g, p_r=p_r, f_g=f_g, f_r=f_r, sym=sym,
         dcd=dcd, cd_p=cd_p, cd_t=cd_t, r_dcd=r_dcd, new_y=ny, new_x=nx, p_R=pR, dcd_grad=rr.grad)


def gen_recon_completion():
    """recon_completion (losses/TDA_loss_sym_recon.py:453-490) and TDA_loss.recon_completion_loss (:344-348): value and gradient
    w.r.t. both clouds, on the dcd fixture's clouds (the reference's chamfer_python stands in for the CUDA extension)."""
    ref_loss = load_reference_loss()
    gd = np.load(os.path.join(HERE, "dcd.npz"))
    recon, prior = torch.from_numpy(gd["recon"]), torch.from_numpy(gd["prior"])
    a, b = recon.clone().requires_grad_(True), prior.clone().requires_grad_(True)
    val = ref_loss.recon_completion(a, b, alpha=70, n_lambda=0.3)
    val.backward()
    loss_mod = ref_loss.TDA_loss()
    with torch.no_grad():
        via_module = loss_mod.recon_completion_loss(recon, prior)
        plain = ref_loss.recon_completion(recon[:, :700], prior, alpha=0.1, n_lambda=0.3, non_reg=True)
    save("recon_completion.npz", value=val.detach(), grad_a=a.grad, grad_b=b.grad, via_module=via_module, non_reg_700=plain)


def gen_tda_loss():
    """TDA_loss.forward with the trainer's 'TDA' name list minus R_DCD (covered by dcd.npz), and the two functions of
    losses/consistency_loss.py, on a seeded batch that holds every symmetry pattern; values and gradients w.r.t. every prediction."""
    ref_loss = load_reference_loss()
    import losses.consistency_loss as ref_con
    from tests.util import synth_loss_batch
    out = {}
    for kind in ("l1", "smoothl1"):
        FLAGS.fsnet_loss_type = kind
        mod = ref_loss.TDA_loss()
        pred, gt, sym, extra = synth_loss_batch(seed=41)
        names = ['Rot1', 'Rot2', 'Rot1_cos', 'Rot2_cos', 'Rot_regular', 'Tran', 'Size', 'R_con', 'TDA_h1', 'TDA_h2', 'TDA_h1_cate', 'TDA_h2_cate']
        if kind == "l1":
            names.append('Prop_sym')                       # TDA_loss defines loss_func for 'l1' only
        for v in pred.values():
            v.requires_grad_(True)
        res = mod(names, pred, gt, sym)
        sum(v.sum() for v in res.values()).backward()
        for k, v in res.items():
            out["%s.loss.%s" % (kind, k)] = v.detach().reshape(-1)
        for k, v in pred.items():
            out["%s.grad.%s" % (kind, k)] = v.grad if v.grad is not None else torch.zeros_like(v)
    FLAGS.fsnet_loss_type = "l1"
    pred, gt, sym, extra = synth_loss_batch(seed=41)
    for name, d in (("pred", pred), ("gt", gt), ("in", extra)):
        for k, v in d.items():
            out["%s.%s" % (name, k)] = v
    out["in.sym"] = sym
    # consistency_loss.py: both clouds and both feature sets as leaves
    x1, x2 = extra["feat1"].clone().requires_grad_(True), extra["feat2"].clone().requires_grad_(True)
    l = ref_con.feat_consistency_loss(x1, x2)
    l.backward()
    out["con.feat"], out["con.feat.g1"], out["con.feat.g2"] = l.detach().reshape(1), x1.grad, x2.grad
    a, b = pred["Recon"].detach().clone().requires_grad_(True), extra["recon2"].clone().requires_grad_(True)
    l = ref_con.prop_sym_matching_loss(a, b, gt["R"], gt["Tran"], sym)
    l.backward()
    out["con.sym"], out["con.sym.gPC"], out["con.sym.gRe"] = l.detach().reshape(1), a.grad, b.grad
    # the NaN / Inf branches of ph_loss_fn
    mod = ref_loss.TDA_loss()
    bad_pred, bad_gt = pred["TDA_h1"].detach().clone(), gt["h1"].clone()
    bad_pred[1, 3] = float("inf")
    bad_gt[2, 5] = float("nan")
    out["ph.bad_pred"] = mod.ph_loss_fn(bad_pred, gt["h1"]).reshape(1)
    out["ph.bad_gt"] = mod.ph_loss_fn(pred["TDA_h1"].detach(), bad_gt).reshape(1)
    save("tda_loss.npz", **out)


def gen_pose_assembly():
    """to_R_matrices (tools/rot_utils.py:95-98) is importable; generate_RT itself exists only as py3.8 bytecode
    (tools/geom_utils), its recorded semantics are: zero f_red where sym[:,0]==1, R = to_R_matrices,
    RT = [[R, T], [0, 1]].  The fixture stores the rotations the reference computes for both confidence settings."""
    import tools.rot_utils as ru
    g = torch.Generator().manual_seed(41)
    B = 16
    p_g = torch.randn(B, 3, generator=g)
    p_r = torch.randn(B, 3, generator=g)
    p_g, p_r = p_g / p_g.norm(dim=1, keepdim=True), p_r / p_r.norm(dim=1, keepdim=True)
    f_g, f_r = torch.rand(B, generator=g), torch.rand(B, generator=g)
    T = torch.randn(B, 3, generator=g)
    sym = torch.zeros(B, 4)
    sym[::3, 0] = 1
    with torch.no_grad():
        R_plain = ru.to_R_matrices(f_g, f_r, p_g, p_r)
        R_sym = ru.to_R_matrices(f_g, torch.where(sym[:, 0] == 1, torch.zeros_like(f_r), f_r), p_g, p_r)
    save("pose_assembly.npz", p_g=p_g, p_r=p_r, f_g=f_g, f_r=f_r, T=T, sym=sym, R_plain=R_plain, R_sym=R_sym)


# ----------------------------------------------------------------------------- the trainer's step
CATS = ["bottle", "bowl", "camera", "can", "laptop", "mug"]
SYM_INFO = [[1, 1, 0, 1], [1, 1, 0, 1], [0, 0, 0, 0], [1, 1, 1, 1], [0, 1, 0, 0], [0, 1, 0, 0]]   # datasets/load_data.py:521-566 get_sym_info


def category_tables():
    """the six obj_model clouds and their persistence-image priors as the loader reads them (datasets/load_data.py:303-311, 327-329)"""
    pts = torch.stack([torch.from_numpy(np.load(os.path.join(REF, "obj_model/points_%s.npy" % c)).astype(np.float32)) for c in CATS])
    h1 = torch.stack([torch.from_numpy(np.load(os.path.join(REF, "obj_model/pdh1_%s.npy" % c)).astype(np.float32)) for c in CATS])
    h2 = torch.stack([torch.from_numpy(np.load(os.path.join(REF, "obj_model/pdh2_%s.npy" % c)).astype(np.float32)) for c in CATS])
    return pts, h1, h2


def load_reference_trainer():
    """Import trainer/RL_TDA.py unmodified.  Beyond load_reference_loss()'s two stand-ins it needs `tools.training_utils`, which the
    reference ships as py3.8 bytecode only: get_gt_v is restated from its disassembly (bmm(R, [[0,0,1],[0,1,0],[0,0,0]])
    -> green = R[:, :, 1], red = R[:, :, 0]); build_optimizer / build_lr_rate are not reached by RL_TDA_train_step."""
    import types
    load_reference_loss()
    tu = types.ModuleType("tools.training_utils")

    def get_gt_v(Rs, axis=2):
        bs = Rs.shape[0]
        corners = torch.tensor([[0, 0, 1], [0, 1, 0], [0, 0, 0]], dtype=Rs.dtype).to(Rs.device).view(1, 3, 3).repeat(bs, 1, 1)
        v = torch.bmm(Rs, corners).transpose(2, 1).reshape(bs, -1)
        return v[:, 3:6], v[:, 6:9]

    tu.get_gt_v = get_gt_v
    tu.build_lr_rate = tu.build_optimizer = None
    sys.modules["tools.training_utils"] = tu
    import trainer.RL_TDA as ref_tr
    return ref_tr


def synth_train_batch(cat_ids, N, seed):
    """A batch in the train loader's format (datasets/load_data.py:313-349): clouds = posed, scaled, noisy samples of the
    categories' obj_model clouds; aug_pcl_in = the same object under the loader's kind of perturbation (jitter + a small rigid
    motion); pdh1 / pdh2 = the category priors with noise."""
    from tests.util import synth_train_db
    pts, h1, h2 = category_tables()
    return synth_train_db(pts, h1, h2, SYM_INFO, cat_ids, N, seed)


def gen_train_step(name="train_step_b4_n256.npz", cat_ids=(0, 2, 3, 4), N=256, wseed=4, dseed=8, fseed=35):
    """The reference's own RL_TDA_train_step (trainer/RL_TDA.py:110-200) + the loop body's total loss and backward (:205-222) on a
    seeded batch: net1 = PoseNet9D(), net2 = PoseNet9D(only_encoder=True) under no_grad on the augmented cloud, the three
    consistency terms, the fourteen control_loss('TDA') terms.  Dropout p = 0 (host-generator masks cannot be reproduced on the
    device).  Recorded: every loss term, the total, per net1 parameter the gradient's norm / sum / 16 samples, the BatchNorm
    buffers of both nets after the step, both nets' neighbour graphs and subsamples."""
    ref_tr = load_reference_trainer()
    FLAGS.train = 1
    FLAGS.fsnet_loss_type = "l1"
    tr = ref_tr.RT_TDA_Trainer(logger=None)
    tr.device = torch.device("cpu")
    tr.init_network("RL_TDA")
    tr.init_loss()
    assert tr.name_TDA_list == ['Rot1', 'Rot2', 'Rot1_cos', 'Rot2_cos', 'Rot_regular', 'Tran', 'Size', 'R_con', 'TDA_h1', 'TDA_h2',
                                'TDA_h1_cate', 'TDA_h2_cate', 'Prop_sym', 'R_DCD_cate_pred']
    tr.net1.load_state_dict(iw.seeded_state_dict(wseed), strict=True)
    tr.net2.load_state_dict(iw.seeded_state_dict(wseed + 1, only_encoder=True), strict=True)
    for net in (tr.net1, tr.net2):
        net.train()
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    db = synth_train_batch(list(cat_ids), N, dseed)
    knn_rec, nn_rec = [], []
    o_knn, o_nn = ref_gcn.get_neighbor_index, ref_gcn.get_nearest_index
    ref_gcn.get_neighbor_index = lambda v, k: (knn_rec.append(o_knn(v, k)), knn_rec[-1])[1]
    ref_gcn.get_nearest_index = lambda t, s_: (nn_rec.append(o_nn(t, s_)), nn_rec[-1])[1]
    try:
        torch.manual_seed(fseed)
        _, loss_dict = tr.RL_TDA_train_step(db)
    finally:
        ref_gcn.get_neighbor_index, ref_gcn.get_nearest_index = o_knn, o_nn
    tda = loss_dict['TDA_loss']
    total = 0.1 * loss_dict['RL_loss'] + 0.1 * loss_dict['recon_1_loss'] + 0.1 * loss_dict['recon_consistency_loss'] + 0.9 * sum(tda.values())
    total.backward()                                                                    # trainer/RL_TDA.py:214,222
    names = ["conv_0.rf", "conv_0.orl_xyz", "conv_1.rf", "conv_1.orl_xyz", "pool_1.xyz", "conv_2.rf",
             "conv_2.orl_xyz", "conv_3.rf", "conv_3.orl_xyz", "pool_2.xyz", "conv_4.rf", "conv_4.orl_xyz"]
    assert len(knn_rec) == 2 * len(names) and len(nn_rec) == 4
    torch.manual_seed(fseed)
    draws = [torch.randperm(N)[: N // 4], None, None, None]
    draws[1] = torch.randperm(N // 4)[: N // 16]
    draws[2] = torch.randperm(N)[: N // 4]
    draws[3] = torch.randperm(N // 4)[: N // 16]
    arrays = dict(weight_seed=np.int64(wseed), data_seed=np.int64(dseed), forward_seed=np.int64(fseed), cat_ids=np.array(cat_ids),
                  n_points=np.int64(N), total=total.detach().reshape(1))
    for k, v in db.items():
        arrays["db." + k] = v
    for i, d in enumerate(draws):
        arrays["sample.%d" % i] = small_idx(d)
    for j, pre in enumerate(("face_all.encoder.", "face_enc.encoder.")):
        for n_, r in zip(names, knn_rec[j * len(names):(j + 1) * len(names)]):
            arrays["idx." + pre + n_] = small_idx(r)
        arrays["idx." + pre + "up_1"], arrays["idx." + pre + "up_2"] = small_idx(nn_rec[2 * j]), small_idx(nn_rec[2 * j + 1])
    for k in ("RL_loss", "recon_1_loss", "recon_consistency_loss"):
        arrays["loss." + k] = loss_dict[k].detach().reshape(1)
    for k, v in tda.items():
        arrays["loss.TDA." + k] = v.detach().reshape(1)
    for k, p_ in tr.net1.named_parameters():
        if p_.grad is None:
            continue
        gflat = p_.grad.reshape(-1)
        pick = torch.linspace(0, gflat.numel() - 1, 16).long()
        arrays["grad." + k] = torch.cat([gflat.norm().view(1), gflat.double().sum().float().view(1), gflat[pick]])
    assert all(p_.grad is None for p_ in tr.net2.parameters())
    for tag, net in (("net1", tr.net1), ("net2", tr.net2)):
        for k, v in net.state_dict().items():
            if "running_" in k or "num_batches" in k:
                arrays["bn.%s.%s" % (tag, k)] = v
    save(name, **arrays)


def gen_category_clouds():
    """BASELINE config 3's workload as data: the six obj_model clouds (float64 on disk -> float32 as the loader casts them,
    datasets/load_data.py:329) with their pdh1 / pdh2 priors, and -- on a B = 6 batch of them under seeded random rotations --
    the reference's TRAINING-mode forward (dropout p = 0) and its R_DCD loss against the category clouds."""
    ref_loss = load_reference_loss()
    pts, h1, h2 = category_tables()
    from tests.util import rand_rotations
    B, wseed, fseed = 6, 5, 37
    R = rand_rotations(B, 51)
    g = torch.Generator().manual_seed(52)
    t = torch.randn(B, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 1.0])
    s = torch.rand(B, 3, generator=g) * 0.1 + 0.25
    clouds = torch.matmul(pts * s.unsqueeze(1), R.transpose(1, 2)) + t.unsqueeze(1)       # (6, 1024, 3) camera frame
    obj = torch.arange(6).float().view(6, 1)
    net = RefPoseNet9D().train()
    net.load_state_dict(iw.seeded_state_dict(wseed), strict=True)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    out, idx = run_reference(net, clouds, obj, fseed, train=1)
    i1, i2 = sample_indices(1024, fseed)
    sym = torch.tensor(SYM_INFO, dtype=torch.float32)
    mod = ref_loss.TDA_loss()
    with torch.no_grad():
        r_dcd = mod.R_DCD(pts, out["recon"], R, out["p_green_R"], out["f_green_R"], out["p_red_R"], out["f_red_R"], out["Pred_T"],
                          out["Pred_s"], sym)
    arrays = dict(weight_seed=np.int64(wseed), forward_seed=np.int64(fseed), points_category=pts, pdh1_category=h1, pdh2_category=h2,
                  sym=sym, gt_R=R, gt_t=t, gt_s=s, points=clouds, obj_id=obj, sample_idx_1=small_idx(i1), sample_idx_2=small_idx(i2),
                  r_dcd=r_dcd.reshape(1))
    for k, v in out.items():
        if k == "feat":
            arrays["train.feat_rowsum"] = v.double().sum(dim=2).float()
        else:
            arrays["train." + k] = v
    for k, v in idx.items():
        arrays["idx." + k] = small_idx(v)
    save("category_clouds.npz", **arrays)


# ----------------------------------------------------------------------------- evaluation metrics (mAP)
from tests.util import synth_eval_results  # noqa: E402  (pure numpy; shared with the GPU tests)


def load_reference_eval():
    import types
    for m in ("cv2", "skimage", "skimage.color", "scipy.misc"):       # imported at module level, unused by the mAP path
        sys.modules.setdefault(m, types.ModuleType(m))
    return _load_by_path("ref_eval_utils", os.path.join(REF, "evaluation/eval_utils_v1.py"))


def gen_eval_map():
    """compute_degree_cm_mAP of the reference (evaluation/eval_utils_v1.py:1227) on a synthetic result list, with the
    evaluater's threshold grids (RT_TDA_Evaluater.py:111-124) coarsened, in both pose-matching modes; plus raw pair metrics."""
    import tempfile
    ev = load_reference_eval()
    synset = ['BG', 'bottle', 'bowl', 'camera', 'can', 'laptop', 'mug']
    res = synth_eval_results(4)
    deg, shift, iou = list(range(0, 61, 5)), [i / 2 for i in range(0, 21, 2)], [i / 100 for i in range(0, 101, 5)]
    arrays = dict(seed=np.int64(4), degree=np.array(deg), shift=np.array(shift), iou=np.array(iou))
    with tempfile.TemporaryDirectory() as tmp:
        for tag, use in (("pose_only", True), ("pose_det", False)):
            a, b = ev.compute_degree_cm_mAP(res, synset, tmp, deg, shift, iou, iou_pose_thres=0.1, use_matches_for_pose=use,
                                            plot_figure=False)
            arrays[tag + ".iou_aps"], arrays[tag + ".pose_aps"] = a, b
    rng = np.random.RandomState(11)
    RT1, RT2, S1, S2, sym, mode, out_iou, out_err = [], [], [], [], [], [], [], []
    names = {0: ("camera", 1), 1: ("bottle", 1), 2: ("phone", 1)}
    insts = [(r['pred_RTs'][i], r['pred_scales'][i], r['gt_RTs'][j], r['gt_scales'][j])
             for r in res for i in range(len(r['pred_RTs'])) for j in range(len(r['gt_RTs']))][:120]
    for k, (a, sa, b, sb) in enumerate(insts):
        m = k % 3
        cname, hv = names[m]
        out_iou.append(ev.compute_3d_iou_new(a, b, sa, sb, hv, cname, cname))
        out_err.append(ev.compute_RT_degree_cm_symmetry(a, b, 1, hv, ['BG', cname]))
        RT1.append(a), RT2.append(b), S1.append(sa), S2.append(sb), sym.append(int(cname == "bottle")), mode.append(m)
    arrays.update(pair_RT1=np.stack(RT1), pair_RT2=np.stack(RT2), pair_S1=np.stack(S1), pair_S2=np.stack(S2),
                  pair_sym=np.array(sym, dtype=np.int32), pair_mode=np.array(mode, dtype=np.int32),
                  pair_iou=np.array(out_iou, dtype=np.float64), pair_err=np.stack(out_err).astype(np.float64))
    np.savez_compressed(os.path.join(HERE, "eval_map.npz"), **arrays)
    print("wrote eval_map.npz")


# ----------------------------------------------------------------------------- input side (depth image -> cloud)
def gen_input_side():
    """The reference's own ``PoseDataset.__getitem__`` (evaluation/load_data_eval.py:239-400) run on synthetic frames written to
    a temporary dataset directory in its on-disk layout (label / detection pickles written by this script).  Stand-ins, all
    written for this repo: ``cv2`` (absent from the image) = imread from .npy twins + getAffineTransform / warpAffine as
    restated in oracle/input_ref.py (so THOSE TWO functions are not pinned by this fixture; everything else -- box window,
    affine set-up, back-projection, validity tests, outlier cut, np.random resampling -- is the reference executing);
    ``tools.eval_utils`` (bytecode only in the reference) = load_depth / get_bbox of its source twin
    network/point_sample/pc_sample_sphere.py, imported from there."""
    import pickle
    import tempfile
    import types
    from tests.util import synth_depth_scene
    from oracle import input_ref as ir

    store = {}
    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST, cv2.INTER_LINEAR = 0, 1
    cv2.getAffineTransform = lambda src, dst: ir.get_affine_transform_cv(src, dst)

    def warp_affine(img, M, dsize, flags=1):
        assert flags == cv2.INTER_NEAREST
        return ir.warp_affine_nearest(img, M, dsize)

    def imread(path, flag=1):
        return np.load(path + ".npy") if os.path.exists(path + ".npy") else None
    cv2.warpAffine, cv2.imread = warp_affine, imread
    sys.modules["cv2"] = cv2
    for m in ("skimage", "skimage.color", "scipy.misc"):
        sys.modules.setdefault(m, types.ModuleType(m))
    twin = _load_by_path("ref_pc_sample_sphere", os.path.join(REF, "network/point_sample/pc_sample_sphere.py"))
    eu = types.ModuleType("tools.eval_utils")
    eu.load_depth, eu.get_bbox = twin.load_depth, twin.get_bbox
    sys.modules["tools.eval_utils"] = eu
    lde = _load_by_path("ref_load_data_eval", os.path.join(REF, "evaluation/load_data_eval.py"))

    frames = [synth_depth_scene(21, 4), synth_depth_scene(22, 5, edge_cases=True), synth_depth_scene(23, 1)]
    arrays = dict(n_frames=np.int64(len(frames)), scene_seeds=np.array([21, 22, 23]), scene_dets=np.array([4, 5, 1]),
                  scene_edge=np.array([0, 1, 0]), np_seed=np.int64(77))
    with tempfile.TemporaryDirectory() as tmp:
        det_dir = os.path.join(tmp, "det")
        os.makedirs(os.path.join(tmp, "Real", "test", "scene_1"))
        os.makedirs(os.path.join(det_dir, "REAL275"))
        ds = lde.PoseDataset.__new__(lde.PoseDataset)
        ds.data_dir, ds.detection_dir, ds.per_obj_id, ds.invaild_list = tmp, det_dir, None, []
        ds.camera_intrinsics = np.array([[577.5, 0, 319.5], [0, 577.5, 239.5], [0, 0, 1]], dtype=np.float32)
        ds.real_intrinsics = np.array([[591.0125, 0, 322.525], [0, 590.16775, 244.11084], [0, 0, 1]], dtype=np.float32)
        ds.id2cat_name = {'1': 'bottle', '2': 'bowl', '3': 'camera', '4': 'can', '5': 'laptop', '6': 'mug'}
        ds.img_list = []
        for i, fr in enumerate(frames):
            stem = os.path.join("Real", "test", "scene_1", "%04d" % i)
            ds.img_list.append(stem)
            with open(os.path.join(tmp, stem + "_label.pkl"), "wb") as f:
                pickle.dump(dict(instance_ids=[1, 2], class_ids=[1, 2]), f)
            np.save(os.path.join(tmp, stem + "_color.png.npy"), np.zeros(fr["depth"].shape + (3,), np.uint8))
            np.save(os.path.join(tmp, stem + "_depth.png.npy"), fr["depth"])
            open(os.path.join(tmp, stem + "_depth.png"), "wb").close()          # __getitem__ tests os.path.exists on it
            with open(os.path.join(det_dir, "REAL275", "results_test_scene_1_%04d.pkl" % i), "wb") as f:
                pickle.dump({k: fr[k] for k in ("pred_masks", "pred_bboxes", "pred_class_ids", "pred_scores")}, f)
        ds.length = len(ds.img_list)
        np.random.seed(77)
        for i in range(len(frames)):
            data, det, _ = ds[i]
            arrays["pcl_in.%d" % i] = data["pcl_in"].numpy()
            arrays["cat_id.%d" % i] = data["cat_id"].numpy()
            assert "pred_masks" not in det
    np.savez_compressed(os.path.join(HERE, "input_side.npz"), **arrays)
    print("wrote input_side.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "input_side.npz")) / 1024.0))


def gen_train_loader():
    """The reference's own TRAINING ``PoseDataset.__getitem__`` (datasets/load_data.py:170-351) run on synthetic frames written to a
    temporary dataset directory in its on-disk layout.  Augmentation is switched off through the reference's own FLAGS where it
    is optional (roi_mask_pro = 0: defor_2D returns the mask; aug_*_pro = 0: PC_BasicAugment leaves the cloud alone); the box
    augmentation aug_bbox_DZI runs as configured per item -- 'uniform' (the reference's default: its drawn window is recorded,
    the build takes windows as an input) or switched off.  Stand-ins written for this repo, none of them on the pinned path
    except cv2's two functions (as for the evaluation loader: oracle/input_ref.py's restatement of getAffineTransform /
    warpAffine): ``cv2`` (imread from .npy twins), ``mmengine`` (load = pickle.load), ``tools.eval_utils`` (load_depth / get_bbox
    of the source twin network/point_sample/pc_sample_sphere.py), ``datasets.compute_pd`` (gudhi + persim: returns zeros).
    Recorded per item: the window aug_bbox_DZI returned, NumPy's generator state at the first _sample_points call (the build
    replays the two permutations from there), pcl_in."""
    import pickle
    import tempfile
    import types
    from tests.util import synth_depth_scene
    from oracle import input_ref as ir

    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST, cv2.INTER_LINEAR = 0, 1
    cv2.getAffineTransform = lambda src, dst: ir.get_affine_transform_cv(src, dst)

    def warp_affine(img, M, dsize, flags=1):
        assert flags == cv2.INTER_NEAREST
        return ir.warp_affine_nearest(img, M, dsize)

    def imread(path, flag=1):
        return np.load(path + ".npy") if os.path.exists(path + ".npy") else None
    cv2.warpAffine, cv2.imread = warp_affine, imread
    sys.modules["cv2"] = cv2
    mm = types.ModuleType("mmengine")
    mm.load = lambda path: pickle.load(open(path, "rb"))
    sys.modules["mmengine"] = mm
    twin = _load_by_path("ref_pc_sample_sphere", os.path.join(REF, "network/point_sample/pc_sample_sphere.py"))
    eu = types.ModuleType("tools.eval_utils")
    eu.load_depth, eu.get_bbox = twin.load_depth, twin.get_bbox
    sys.modules["tools.eval_utils"] = eu
    import datasets as ref_datasets                      # the reference's package (REF precedes site-packages on sys.path)
    assert ref_datasets.__file__.startswith(REF), ref_datasets.__file__
    cpd = types.ModuleType("datasets.compute_pd")
    cpd.compute_pd = lambda pts: (torch.zeros(2500), torch.zeros(2500))
    sys.modules["datasets.compute_pd"] = cpd
    ld = _load_by_path("ref_load_data_train", os.path.join(REF, "datasets/load_data.py"))
    F = ld.FLAGS
    F.train, F.roi_mask_pro, F.aug_pc_pro, F.aug_rt_pro, F.aug_bb_pro, F.aug_bc_pro = 1, 0.0, 0.0, 0.0, 0.0, 0.0
    windows = []
    real_dzi = ld.aug_bbox_DZI

    def dzi_spy(flags_, bbox_xyxy, im_H, im_W):
        c, sc = real_dzi(flags_, bbox_xyxy, im_H, im_W)
        windows.append((np.asarray(c, dtype=np.float64).copy(), float(sc)))
        return c, sc
    ld.aug_bbox_DZI = dzi_spy

    # items: (scene seed, detection index inside the scene, DZI type, np.random seed)
    items = [(31, 0, "uniform", 5), (31, 2, "uniform", 6), (32, 1, "none", 7), (33, 0, "uniform", 8), (33, 3, "none", 9),
             (34, 1, "uniform", 10)]
    arrays = dict(n_items=np.int64(len(items)), item_scene=np.array([i[0] for i in items]), item_det=np.array([i[1] for i in items]),
                  item_dzi=np.array([i[2] == "uniform" for i in items]), scene_dets=np.int64(4))
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "Real", "train", "scene_1"))
        try:
            os.chdir(REF)                                  # __getitem__ reads ./obj_model/points_{cat}.npy (:303-311)
            for n, (sseed, j, dzi, npseed) in enumerate(items):
                fr = synth_depth_scene(sseed, 4)
                inst_mask = np.zeros(fr["depth"].shape, np.uint8)
                for q in range(4):
                    inst_mask[fr["pred_masks"][:, :, q]] = q + 1                      # instance ids 1..4 (later blobs on top)
                stem = os.path.join("Real", "train", "scene_1", "%04d" % n)
                cls = int(fr["pred_class_ids"][j])
                if cls == 6:
                    cls = 5                                                           # (mug needs the handle-visibility table)
                rot = np.eye(3, dtype=np.float32)
                with open(os.path.join(tmp, stem + "_label.pkl"), "wb") as f:
                    pickle.dump(dict(class_ids=[cls], instance_ids=[j + 1], bboxes=[fr["pred_bboxes"][j]], model_list=["m0"],
                                     scales=[0.3], rotations=[rot], translations=[np.array([0.0, 0.0, 0.8], np.float32)]), f)
                np.save(os.path.join(tmp, stem + "_color.png.npy"), np.zeros(fr["depth"].shape + (3,), np.uint8))
                np.save(os.path.join(tmp, stem + "_depth.png.npy"), fr["depth"])
                open(os.path.join(tmp, stem + "_depth.png"), "wb").close()
                np.save(os.path.join(tmp, stem + "_mask.png.npy"), np.repeat(inst_mask[:, :, None], 3, axis=2))
                ds = ld.PoseDataset.__new__(ld.PoseDataset)
                ds.source, ds.mode, ds.data_dir, ds.per_obj, ds.per_obj_id = "Real", "train", tmp, "", None
                ds.img_list, ds.length, ds.invaild_list = [stem], 1, []
                ds.camera_intrinsics = np.array([[577.5, 0, 319.5], [0, 577.5, 239.5], [0, 0, 1]], dtype=np.float32)
                ds.real_intrinsics = np.array([[591.0125, 0, 322.525], [0, 590.16775, 244.11084], [0, 0, 1]], dtype=np.float32)
                ds.cat_names = ['bottle', 'bowl', 'camera', 'can', 'laptop', 'mug']
                ds.id2cat_name = {'1': 'bottle', '2': 'bowl', '3': 'camera', '4': 'can', '5': 'laptop', '6': 'mug'}
                ds.models = {"m0": np.random.RandomState(1).rand(64, 3).astype(np.float32) - 0.5}
                ds.mug_sym = {}
                ds.base_aug = ld.PC_BasicAugment()
                ds.operator_name = ['Jitter', 'RandomCutout', 'RandomCrop', 'RandomDropout']
                ds.custom_aug_operator = [ld.PcJitter(std=0.005, clip=0.05, p=0.6), ld.PcRandomCutout(p=0.9, min_num_points=1024),
                                          ld.PcRandomCrop(p=0.9, min_num_points=1024), ld.PcRandomDropout(p=0.9, max_dropout_ratio=0.5)]
                F.DZI_TYPE = dzi
                states = []
                real_sample = ds._sample_points

                def sample_spy(pcl, n_pts, _s=states, _r=real_sample):
                    if not _s:
                        _s.append(np.random.get_state())
                    return _r(pcl, n_pts)
                ds._sample_points = sample_spy
                del windows[:]
                np.random.seed(npseed)
                torch.manual_seed(npseed)
                import random as _random
                _random.seed(npseed)
                data = ds[0]
                assert len(windows) == 1 and len(states) == 1
                st = states[0]
                arrays["pcl_in.%d" % n] = data["pcl_in"].numpy()
                arrays["window.%d" % n] = np.array([windows[0][0][0], windows[0][0][1], windows[0][1]], dtype=np.float64)
                arrays["rng_keys.%d" % n] = np.asarray(st[1], dtype=np.uint32)
                arrays["rng_pos.%d" % n] = np.array([st[2], st[3]], dtype=np.int64)
                arrays["rng_gauss.%d" % n] = np.float64(st[4])
                arrays["cat_id.%d" % n] = data["cat_id"].numpy()
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "train_loader.npz"), **arrays)
    print("wrote train_loader.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "train_loader.npz")) / 1024.0))


def main():
    if sys.argv[1:] == ["train_loader"]:
        return gen_train_loader()
    if sys.argv[1:] == ["recon_completion"]:
        return gen_recon_completion()
    if sys.argv[1:] == ["input_side"]:
        return gen_input_side()
    if sys.argv[1:] == ["proj"]:
        return gen_proj("proj_b2_n256.npz", 2, 256, wseed=5, pseed=7, fseed=37)
    if sys.argv[1:] == ["train_step"]:
        gen_train_step()
        return gen_category_clouds()
    gen_train_step()
    gen_category_clouds()
    gen_tda_loss()
    gen_eval_map()
    gen_pose_assembly()
    gen_dcd()
    gen_recon_completion()
    gen_knn()
    gen_layers()
    bottle = torch.from_numpy(np.load(os.path.join(REF, "obj_model/points_bottle.npy"))).float()[None]
    gen_forward("forward_bottle.npz", 1, 1024, wseed=0, pseed=0, fseed=7, pts=bottle, obj=torch.zeros(1, 1))
    gen_forward("forward_b2_n1028.npz", 2, 1028, wseed=0, pseed=1, fseed=123)
    gen_forward("forward_b3_n256.npz", 3, 256, wseed=1, pseed=2, fseed=9, keep_feat_rows=96)
    gen_forward_train("forward_train_b4_n256.npz", 4, 256, wseed=2, pseed=5, fseed=31)
    gen_backward("backward_b3_n256.npz", 3, 256, wseed=3, pseed=6, fseed=33)
    gen_proj("proj_b2_n256.npz", 2, 256, wseed=5, pseed=7, fseed=37)
    gen_chamfer()
    gen_input_side()        # last: it installs a cv2 stand-in
    gen_train_loader()


if __name__ == "__main__":
    main()
