"""State-dict contract of PoseNet9D and a seeded weight generator.

The reference ships no checkpoint, so benchmarks and parity tests run on
random weights of the reference architecture.  ``state_spec()`` lists every tensor a reference
checkpoint's ``net1_state_dict`` holds (names/shapes measured by instantiating the reference:
164 tensors, 27,430,569 parameters; loaded strictly at evaluater/RT_TDA_Evaluater.py:39);
``seeded_state_dict(seed)`` fills them deterministically on the CPU so the build container, the
GPU box, the oracle and the HIP path all see bit-identical weights.
"""
import zlib

import torch

SUPPORT_NUM = 7      # config/config.py:44 gcn_sup_num
OUT_CHANNELS = 2500  # config/config.py:150 output_channels
FEAT_C = 1286        # 128+128+256+256+512+6, FaceRecon.py:123


_KIND = {}  # name -> one of conv_w conv_b bn_w bn_b bn_mean bn_var bn_count hs_w hs_b hs_dir


def _put(spec, name, shape, kind):
    spec[name] = tuple(shape)
    _KIND[name] = kind


def _bn(spec, name, c):
    _put(spec, name + ".weight", (c,), "bn_w")
    _put(spec, name + ".bias", (c,), "bn_b")
    _put(spec, name + ".running_mean", (c,), "bn_mean")
    _put(spec, name + ".running_var", (c,), "bn_var")
    _put(spec, name + ".num_batches_tracked", (), "bn_count")


def _conv(spec, name, cout, cin, bias=True):
    _put(spec, name + ".weight", (cout, cin, 1), "conv_w")
    if bias:
        _put(spec, name + ".bias", (cout,), "conv_b")


def _linear(spec, name, cout, cin, bias=True):
    _put(spec, name + ".weight", (cout, cin), "conv_w")
    if bias:
        _put(spec, name + ".bias", (cout,), "conv_b")


def state_spec(only_encoder=False):
    """Ordered {name: shape} of the reference's state dict (PoseNet9D.py:24-31)."""
    S = SUPPORT_NUM
    spec = {}
    face = "face_enc." if only_encoder else "face_all."
    e = face + "encoder."
    _put(spec, e + "conv_0.directions", (3, S * 128), "hs_dir")
    _conv(spec, e + "conv_0.STE_layer", 128, 3, bias=False)
    _conv(spec, e + "conv_0.conv2", 128, 256, bias=False)
    for i, (cin, cout) in zip((1, 2, 3, 4), ((128, 128), (128, 256), (256, 256), (256, 512))):
        p = e + "conv_%d." % i
        _put(spec, p + "weights", (cin, (S + 1) * cout), "hs_w")
        _put(spec, p + "bias", ((S + 1) * cout,), "hs_b")
        _put(spec, p + "directions", (3, S * cout), "hs_dir")
        _conv(spec, p + "STE_layer", cout, cin, bias=False)
        _conv(spec, p + "conv2", cout, 2 * cout, bias=False)
    _bn(spec, e + "bn1", 128)
    _bn(spec, e + "bn2", 256)
    _bn(spec, e + "bn3", 256)
    _conv(spec, e + "proj_layer.0", FEAT_C, FEAT_C, bias=False)
    _bn(spec, e + "proj_layer.1", FEAT_C)
    _conv(spec, e + "proj_layer.3", FEAT_C, FEAT_C, bias=False)
    d = face + "decoder."
    for conv, bn, cout, cin in (("0", "1", 512, FEAT_C), ("3", "4", 512, 512), ("6", "7", 256, 512)):
        _conv(spec, d + "conv1d_block." + conv, cout, cin)
        _bn(spec, d + "conv1d_block." + bn, cout)
    _conv(spec, d + "recon_head.0", 128, 256)
    _bn(spec, d + "recon_head.1", 128)
    _conv(spec, d + "recon_head.3", 3, 128)
    p = face + "ph_pred."
    _conv(spec, p + "conv_5.0", 1024, FEAT_C, bias=False)
    _bn(spec, p + "conv_5.1", 1024)
    _linear(spec, p + "linear1", 1024, 2048, bias=False)
    _bn(spec, p + "bn5", 1024)
    for n in ("linear2", "linear3"):
        _linear(spec, p + n, OUT_CHANNELS, 1024)
    for n in ("linear4", "linear5"):
        _linear(spec, p + n, FEAT_C, OUT_CHANNELS)
    if only_encoder:
        return spec
    for head, cin, cout in (("rot_green", 1286, 4), ("rot_red", 1286, 4), ("ts", 1289, 6)):
        _conv(spec, head + ".conv1", 1024, cin)
        _conv(spec, head + ".conv2", 256, 1024)
        _conv(spec, head + ".conv3", 256, 256)
        _conv(spec, head + ".conv4", cout, 256)
        _bn(spec, head + ".bn1", 1024)
        _bn(spec, head + ".bn2", 256)
        _bn(spec, head + ".bn3", 256)
    return spec


def seeded_state_dict(seed=0, only_encoder=False, dtype=torch.float32):
    """Deterministic random weights with magnitudes like the reference's initialisers.

    * Conv1d / Linear: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (torch default, kaiming-uniform a=sqrt(5))
    * HS layers: U(-stdv, stdv), stdv = 1/sqrt(out*(S+1)) resp. 1/sqrt(S*C) (gcn3d.py:73-75,137-140)
    * BatchNorm: non-trivial running stats and affine so that the eval-mode fold is exercised
      (weight U(.5,1.5), bias U(-.2,.2), mean U(-.2,.2), var U(.5,1.5))
    Every tensor has its own generator seeded from (seed, crc32(name)) so tensors are independent
    of iteration order.
    """
    spec = state_spec(only_encoder)
    out = {}
    for name, shape in spec.items():
        g = torch.Generator().manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))

        def U(lo, hi, shape=shape):
            return (torch.rand(shape, generator=g, dtype=torch.float64) * (hi - lo) + lo).to(dtype)

        kind = _KIND[name]
        if kind == "bn_count":
            out[name] = torch.zeros((), dtype=torch.long)
        elif kind in ("bn_mean", "bn_b"):
            out[name] = U(-0.2, 0.2)
        elif kind in ("bn_var", "bn_w"):
            out[name] = U(0.5, 1.5)
        elif kind in ("hs_dir", "hs_w"):
            stdv = 1.0 / shape[1] ** 0.5             # S*C resp. (S+1)*out columns
            out[name] = U(-stdv, stdv)
        elif kind == "hs_b":
            stdv = 1.0 / shape[0] ** 0.5
            out[name] = U(-stdv, stdv)
        elif kind == "conv_w":
            bound = 1.0 / shape[1] ** 0.5
            out[name] = U(-bound, bound)
        else:                                        # conv_b: bound from the sibling weight's fan-in
            bound = 1.0 / spec[name[: -len("bias")] + "weight"][1] ** 0.5
            out[name] = U(-bound, bound)
    return out


def param_count(spec=None):
    spec = spec or state_spec()
    n = 0
    for name, shape in spec.items():
        if name.endswith(("running_mean", "running_var", "num_batches_tracked")):
            continue
        c = 1
        for s in shape:
            c *= s
        n += c
    return n
