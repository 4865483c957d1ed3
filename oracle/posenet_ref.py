"""Oracle (TEST INFRASTRUCTURE): CPU restatement of ``PoseNet9D.forward`` and the sub-networks it
calls, as one stateless function over a reference-format state dict.

Follows, in eval mode (BatchNorm running statistics, dropout off) and -- with bn_train=True -- in training
mode with dropout disabled (batch statistics; the running statistics after the step are returned too):
    network/fs_net_repo/PoseNet9D.py:33-91   top level, output dict
    network/fs_net_repo/FaceRecon.py:39-86   Face_Enc.forward
    network/fs_net_repo/FaceRecon.py:139-167 PH_Predictor.forward
    network/fs_net_repo/FaceRecon.py:112-117 Face_Dec.forward
    network/fs_net_repo/PoseR.py:26-39       Rot_green / Rot_red
    network/fs_net_repo/PoseTs.py:31-45      Pose_Ts
"""
import torch
import torch.nn.functional as F

from . import gcn_ref as G

BN_EPS = 1e-5  # nn.BatchNorm1d default, used by every BN in the reference


def _bn(P, name, x):
    """BatchNorm1d over (B,C,L) or (B,C): running statistics (eval) or, when P["_bn_train"] is set, batch statistics
    with the momentum-0.1 running-statistics update nn.BatchNorm1d makes, collected in P["_bn_new"]."""
    if P.get("_bn_train"):
        rm, rv = P[name + ".running_mean"].clone(), P[name + ".running_var"].clone()
        y = F.batch_norm(x, rm, rv, P[name + ".weight"], P[name + ".bias"], True, 0.1, BN_EPS)
        P["_bn_new"][name + ".running_mean"], P["_bn_new"][name + ".running_var"] = rm, rv
        P["_bn_new"][name + ".num_batches_tracked"] = P[name + ".num_batches_tracked"] + 1
        return y
    return F.batch_norm(x, P[name + ".running_mean"], P[name + ".running_var"], P[name + ".weight"],
                        P[name + ".bias"], False, 0.0, BN_EPS)


def _act(P, z, slope=0.0, cf=False):
    """ReLU / LeakyReLU.  With forced decisions (P["_force"], tests): activation number i of the forward takes its mask -- which
    elements pass -- and its value from the recorded output "act.i" of the run under test; the gradient is this run's, through the
    recorded mask.  (The forward's activations are numbered in call order, the same on both sides.)"""
    f = P.get("_force")
    if f is None:
        y = F.leaky_relu(z, slope) if slope else torch.relu(z)
        if P.get("_record") is not None:            # (self-check of the forcing machinery: channel-last, like the run under test)
            i = P["_act_i"] = P.get("_act_i", -1) + 1
            P["_record"]["act.%d" % i] = (y.transpose(1, 2) if (cf and y.dim() == 3) else y).detach().contiguous().clone()
        return y
    i = P["_act_i"] = P.get("_act_i", -1) + 1
    y_rec = f["act.%d" % i].to(z.dtype)
    if cf and z.dim() == 3:                         # recorded channel-last (B, n, C), computed channel-first (B, C, n)
        y_rec = y_rec.reshape(z.shape[0], -1, z.shape[1]).transpose(1, 2)
    y_rec = y_rec.reshape(z.shape)
    on = (y_rec > 0).to(z.dtype)
    y = z * (on + slope * (1 - on))
    return y + (y_rec - y).detach()


def _act_maxpool(P, z, slope=0.0):
    """max over the points (last dim) of act(z), z (B, C, n) -> (B, C).  Forced: pooled output number i takes the recorded winner per
    (object, channel) and the recorded sign."""
    f = P.get("_force")
    if f is None:
        y, arg = torch.max(F.leaky_relu(z, slope) if slope else torch.relu(z), 2)
        if P.get("_record") is not None:
            i = P["_pool_i"] = P.get("_pool_i", -1) + 1
            P["_record"]["pool.%d" % i] = (arg.int(), y.detach().clone())
        return y
    i = P["_pool_i"] = P.get("_pool_i", -1) + 1
    arg, pooled = f["pool.%d" % i]
    zz = z.gather(2, arg.long().unsqueeze(2)).squeeze(2)
    on = (pooled > 0).to(z.dtype)
    y = zz * (on + slope * (1 - on))
    return y + (pooled.to(z.dtype) - y).detach()


def _bn_rows(P, name, x):
    """BatchNorm1d applied to channel-last rows (B,n,C), as FaceRecon.py:58-65 does via transposes."""
    return _bn(P, name, x.transpose(1, 2)).transpose(1, 2)


def _conv(P, name, x):
    return F.conv1d(x, P[name + ".weight"], P.get(name + ".bias"))


def encoder(P, pre, xyz, cat_id, sample_idx, cache, flags):
    """Face_Enc.forward -> feat (B,N,1286).  sample_idx = (idx_pool1, idx_pool2)."""
    enc = pre + "encoder."
    B, N, _ = xyz.shape
    kmax = flags["gcn_n_num"]
    one_hot = torch.zeros(B, flags["obj_c"]).scatter_(1, cat_id.view(-1, 1).long(), 1)

    fm0 = _act(P, G.surface_conv(P, enc + "conv_0", xyz, kmax, cache))
    fm1 = _act(P, _bn_rows(P, enc + "bn1", G.hs_conv(P, enc + "conv_1", xyz, fm0, kmax, cache)))
    v1, fp1 = G.pool(xyz, fm1, sample_idx[0], cache, enc + "pool_1")
    k1 = min(kmax, v1.shape[1] // 8)
    fm2 = _act(P, _bn_rows(P, enc + "bn2", G.hs_conv(P, enc + "conv_2", v1, fp1, k1, cache)))
    fm3 = _act(P, _bn_rows(P, enc + "bn3", G.hs_conv(P, enc + "conv_3", v1, fm2, k1, cache)))
    v2, fp2 = G.pool(v1, fm3, sample_idx[1], cache, enc + "pool_2")
    k2 = min(kmax, v2.shape[1] // 8)
    fm4 = G.hs_conv(P, enc + "conv_4", v2, fp2, k2, cache)

    near1 = cache.nn1(enc + "up_1", xyz, v1)
    near2 = cache.nn1(enc + "up_2", xyz, v2)
    up = lambda f, i: G.gather_rows(f, i).squeeze(2)
    feat = torch.cat([fm0, fm1, up(fm2, near1), up(fm3, near1), up(fm4, near2),
                      one_hot.unsqueeze(1).repeat(1, N, 1)], dim=2)
    inter = dict(fm_0=fm0, fm_1=fm1, fm_2=fm2, fm_3=fm3, fm_4=fm4, v_pool_1=v1, v_pool_2=v2)
    return feat, inter


def ph_predictor(P, pre, feat):
    """PH_Predictor.forward: feat (B,N,1286) -> feat_ph (B,1286,N), h1, h2 (B,2500)."""
    ph = pre + "ph_pred."
    B, N, _ = feat.shape
    if P.get("_force") is None and P.get("_record") is None:
        x = F.leaky_relu(_bn(P, ph + "conv_5.1", _conv(P, ph + "conv_5.0", feat.permute(0, 2, 1))), 0.2)
        g = F.adaptive_max_pool1d(x, 1).view(B, -1)
    else:
        g = _act_maxpool(P, _bn(P, ph + "conv_5.1", _conv(P, ph + "conv_5.0", feat.permute(0, 2, 1))), 0.2)
    g = torch.cat((g, g), 1)
    g = _act(P, _bn(P, ph + "bn5", F.linear(g, P[ph + "linear1.weight"])), 0.2)
    pi1 = F.linear(g, P[ph + "linear2.weight"], P[ph + "linear2.bias"])
    pi2 = F.linear(g, P[ph + "linear3.weight"], P[ph + "linear3.bias"])
    back1 = F.linear(pi1, P[ph + "linear4.weight"], P[ph + "linear4.bias"])
    back2 = F.linear(pi2, P[ph + "linear5.weight"], P[ph + "linear5.bias"])
    feat_ph = feat.permute(0, 2, 1) + back1.unsqueeze(-1) + back2.unsqueeze(-1)
    return feat_ph, torch.sigmoid(pi1), torch.sigmoid(pi2)


def decoder(P, pre, x):
    """Face_Dec.forward: (B,1286,N) -> recon (B,N,3)."""
    d = pre + "decoder."
    for conv, bn in (("0", "1"), ("3", "4"), ("6", "7")):
        x = _act(P, _bn(P, d + "conv1d_block." + bn, _conv(P, d + "conv1d_block." + conv, x)), cf=True)
    x = _act(P, _bn(P, d + "recon_head.1", _conv(P, d + "recon_head.0", x)), cf=True)
    return _conv(P, d + "recon_head.3", x).permute(0, 2, 1)


def point_head(P, name, x):
    """Rot_green / Rot_red / Pose_Ts body: (B,C,N) -> (B,out)."""
    x = _act(P, _bn(P, name + ".bn1", _conv(P, name + ".conv1", x)), cf=True)
    if P.get("_force") is None and P.get("_record") is None:
        x = torch.relu(_bn(P, name + ".bn2", _conv(P, name + ".conv2", x)))
        x = torch.max(x, 2, keepdim=True)[0]
    else:
        x = _act_maxpool(P, _bn(P, name + ".bn2", _conv(P, name + ".conv2", x))).unsqueeze(2)
    x = _act(P, _bn(P, name + ".bn3", _conv(P, name + ".conv3", x)), cf=True)
    return _conv(P, name + ".conv4", x).squeeze(2)


DEFAULT_FLAGS = dict(gcn_n_num=20, gcn_sup_num=7, obj_c=6)


def proj_layer(P, face, feat_t):
    """Face_Enc.proj_layer (FaceRecon.py:32-35, applied at :80-84 when enable_proj): Conv1d 1286 -> 1286 (no bias) + BatchNorm1d +
    LeakyReLU(0.2) + Conv1d 1286 -> 1286 (no bias) on feat_global (B, 1286, N)."""
    pl = face + "encoder.proj_layer."
    x = F.conv1d(feat_t, P[pl + "0.weight"])
    x = F.leaky_relu(_bn(P, pl + "1", x), 0.2)
    return F.conv1d(x, P[pl + "3.weight"])


def posenet_forward(P, points, obj_id, sample_idx=None, train_keys=False, mode="exact", inject=None,
                    flags=None, want_intermediates=False, bn_train=False, force=None, record=None, enable_proj=False):
    """PoseNet9D(only_encoder=False).forward in eval mode, or (bn_train=True) in training mode with dropout
    disabled; then out["_bn_new"] holds every BatchNorm buffer after the step.

    P           reference-format state dict (CPU fp32 tensors), keys as trainer/RL_TDA.py saves them
    sample_idx  (idx1, idx2) for the two Pool_layers; None -> drawn from the global CPU generator
                in the reference's order (pool_1 then pool_2, gcn3d.py:242)
    train_keys  FLAGS.train != 0 -> the 11-key dict of PoseNet9D.py:69-82, else the 6-key one
    """
    flags = dict(DEFAULT_FLAGS, **(flags or {}))
    P = dict(P)
    P["_support_num"] = flags["gcn_sup_num"]
    P["_bn_train"], P["_bn_new"] = bool(bn_train), {}
    if force is not None:
        # tests (decision forcing): intermediates and decisions recorded from the run under test -- layer outputs by name, activation
        # outputs and pooled winners in call order -- replace this run's values / ReLU masks / max-over-points winners, so that its
        # autograd differentiates the SAME piecewise-linear branch of the network as the run under test
        P["_force"] = force
    if record is not None:
        P["_record"] = record                       # the same intermediates / decisions of THIS run, under the same names
    cache = G.GraphCache(mode=mode, inject=inject)
    B, N, _ = points.shape
    if sample_idx is None:
        i1 = G.draw_sample_idx(N)
        i2 = G.draw_sample_idx(i1.numel())
        sample_idx = (i1, i2)

    mean = points.mean(dim=1, keepdim=True)
    xyz = points - mean
    feat, inter = encoder(P, "face_all.", xyz, obj_id, sample_idx, cache, flags)
    feat_ph, h1, h2 = ph_predictor(P, "face_all.", feat)
    recon = decoder(P, "face_all.", feat_ph)
    feat_t = feat.permute(0, 2, 1)

    green = point_head(P, "rot_green", feat_t)
    red = point_head(P, "rot_red", feat_t)
    ts = point_head(P, "ts", torch.cat([feat, xyz], dim=2).permute(0, 2, 1))

    out = dict()
    if train_keys:
        out["recon"] = recon + mean
    out["p_green_R"] = green[:, 1:] / (torch.norm(green[:, 1:], dim=1, keepdim=True) + 1e-6)
    out["p_red_R"] = red[:, 1:] / (torch.norm(red[:, 1:], dim=1, keepdim=True) + 1e-6)
    out["f_green_R"] = torch.sigmoid(green[:, 0])
    out["f_red_R"] = torch.sigmoid(red[:, 0])
    out["Pred_T"] = ts[:, 0:3] + points.mean(dim=1)
    out["Pred_s"] = ts[:, 3:6]
    if train_keys:
        out["h1"], out["h2"] = h1, h2
        out["feat"] = feat
        out["feat_global"] = (proj_layer(P, "face_all.", feat_t) if enable_proj else feat_t).max(2)[0]     # PoseNet9D.py:49-50
    if bn_train:
        out["_bn_new"] = P["_bn_new"]
    if want_intermediates:
        inter.update(indices=cache.record, sample_idx=sample_idx, xyz=xyz, feat=feat,
                     green=green, red=red, ts=ts, recon=recon, h1=h1, h2=h2)
        return out, inter
    return out


def encoder_only_forward(P, points, obj_id, sample_idx=None, mode="exact", inject=None, flags=None, bn_train=False,
                         want_intermediates=False, enable_proj=False):
    """PoseNet9D(only_encoder=True).forward (PoseNet9D.py:35-45; the trainer's net2): encoder + decoder on the plain
    feature (pred_PH=False, FaceRecon.py:190-199), keys prefixed face_enc."""
    flags = dict(DEFAULT_FLAGS, **(flags or {}))
    P = dict(P)
    P["_support_num"] = flags["gcn_sup_num"]
    P["_bn_train"], P["_bn_new"] = bool(bn_train), {}
    cache = G.GraphCache(mode=mode, inject=inject)
    B, N, _ = points.shape
    if sample_idx is None:
        i1 = G.draw_sample_idx(N)
        sample_idx = (i1, G.draw_sample_idx(i1.numel()))
    xyz = points - points.mean(dim=1, keepdim=True)
    feat, inter = encoder(P, "face_enc.", xyz, obj_id, sample_idx, cache, flags)
    recon = decoder(P, "face_enc.", feat.permute(0, 2, 1))
    feat_t = feat.permute(0, 2, 1)
    out = dict(feat_global=(proj_layer(P, "face_enc.", feat_t) if enable_proj else feat_t).max(2)[0], recon=recon)     # PoseNet9D.py:39-41
    if bn_train:
        out["_bn_new"] = P["_bn_new"]
    if want_intermediates:
        inter.update(indices=cache.record, sample_idx=sample_idx, feat=feat)
        return out, inter
    return out
