"""CPU tests of the host side: C-ABI surface, state-dict contract, flags, RNG order, sharding.
No compute call into the HIP library happens here (there is no GPU in the build container)."""
import os
import re

import numpy as np
import pytest
import torch

from tests.util import ROOT, golden


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tgpose.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tgp_[a-z0-9_]+)\s*\(", text)))


def test_c_abi_exports_every_declared_symbol():
    from tgpose_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 24
    assert sorted(_lib.SIGNATURES) == names            # the Python binding covers the header exactly
    handle = _lib.lib()                                # dlopen + symbol lookup for each; raises if one is missing
    for n in names:
        assert hasattr(handle, n)
    assert handle.tgp_version() == _lib.ABI_VERSION
    assert handle.tgp_knn_max_points() >= 1028 and handle.tgp_knn_max_k() >= 20


def test_c_abi_argument_errors_do_not_launch():
    """Size/shape validation happens on the host before any kernel launch, so it is safe without a GPU."""
    from tgpose_amd import _lib
    h = _lib.lib()
    assert h.tgp_knn_xyz(None, 1, 64, 8, None, None) == -1
    # fused (matrix-free) shapes need the transposed features + the squared norms; the others the (B, n, n) matrix + the norms
    assert h.tgp_knn_feat_workspace_bytes(2, 100, 128) == (2 * 128 * 128 + 2 * 100) * 4
    assert h.tgp_knn_feat_workspace_bytes(2, 100, 64) == (2 * 100 * 100 + 2 * 100) * 4
    assert h.tgp_knn_feat_workspace_bytes(32, 1028, 128) < 32 * 1028 * 1028
    assert h.tgp_orl_partial_floats(2, 1028, 128) == 2 * 17 * 128
    a = _lib.GemmArgs()
    assert h.tgp_gemm_f32(a, None) == -1
    assert h.tgp_heads_fused(_lib.HeadsFusedArgs(), None) == -1 and h.tgp_conv_max_fused(_lib.ConvMaxFusedArgs(), None) == -1
    assert h.tgp_heads_pack_w2(None, None, None, None, 3, None, None) == -1 and h.tgp_heads_w2_bytes(3) == 3 * 32 * 33 * 1024
    # (ABI 5) blocked fp16 planes: 2 KB per (32 rows, 16 columns) chunk; a split without buffers is refused
    assert h.tgp_planes_bytes(32896, 268) == 1028 * 17 * 2048 and h.tgp_planes_bytes(1, 1) == 2048 and h.tgp_planes_bytes(0, 16) == 0
    assert h.tgp_planes_split(None, 4, 16, 16, None, 1, None, None) == -1


def test_gemm_args_struct_matches_header_layout():
    """ctypes mirror of struct tgp_gemm_args: field order and natural alignment as a C compiler lays it out."""
    import ctypes
    import subprocess
    import tempfile
    from tgpose_amd import _lib
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "tgpose.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",' \
          'sizeof(tgp_gemm_args),offsetof(tgp_gemm_args,M),offsetof(tgp_gemm_args,rowbias),' \
          'offsetof(tgp_gemm_args,slope),offsetof(tgp_gemm_args,ldcm),offsetof(tgp_gemm_args,c_col0),' \
          'offsetof(tgp_gemm_args,batch_stride_colmax),offsetof(tgp_gemm_args,A_planes),offsetof(tgp_gemm_args,a_amax),' \
          'offsetof(tgp_gemm_args,cp_col0),offsetof(tgp_gemm_args,pp_config));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    G = _lib.GemmArgs
    assert [int(x) for x in out] == [ctypes.sizeof(G), G.M.offset, G.rowbias.offset, G.slope.offset, G.ldcm.offset,
                                     G.c_col0.offset, G.batch_stride_colmax.offset, G.A_planes.offset, G.a_amax.offset,
                                     G.cp_col0.offset, G.pp_config.offset]


def test_fused_kernel_arg_structs_match_header_layout():
    """ctypes mirrors of tgp_heads_fused_args / tgp_conv_max_fused_args against the C compiler's layout of the header's structs."""
    import ctypes
    import subprocess
    import tempfile
    from tgpose_amd import _lib
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "tgpose.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",' \
          'sizeof(tgp_heads_fused_args),offsetof(tgp_heads_fused_args,idx2),offsetof(tgp_heads_fused_args,keys),' \
          'offsetof(tgp_heads_fused_args,overflow),sizeof(tgp_conv_max_fused_args),offsetof(tgp_conv_max_fused_args,idx2),' \
          'offsetof(tgp_conv_max_fused_args,slope),offsetof(tgp_conv_max_fused_args,overflow));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = [int(x) for x in subprocess.check_output([os.path.join(d, "t")]).decode().split()]
    H, C = _lib.HeadsFusedArgs, _lib.ConvMaxFusedArgs
    assert out == [ctypes.sizeof(H), H.idx2.offset, H.keys.offset, H.overflow.offset,
                   ctypes.sizeof(C), C.idx2.offset, C.slope.offset, C.overflow.offset]
    # (ABI 7) the decoder's and the chained layer tails' structs: every field's offset
    names = (("tgp_dec_fused_args", _lib.DecFusedArgs), ("tgp_dec_l1_args", _lib.DecL1Args), ("tgp_hs_chain_args", _lib.HsChainArgs),
             ("tgp_proj_planes_args", _lib.ProjPlanesArgs))
    fields = [(cn, f[0]) for cn, cls in names for f in cls._fields_]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "tgpose.h"\nint main(){' + "".join(
        'printf("%%zu ", offsetof(%s, %s));' % cf for cf in fields) + "".join('printf("%%zu ", sizeof(%s));' % cn for cn, _ in names) + "return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = [int(x) for x in subprocess.check_output([os.path.join(d, "t")]).decode().split()]
    want = [getattr(cls, f[0]).offset for _, cls in names for f in cls._fields_] + [ctypes.sizeof(cls) for _, cls in names]
    assert out == want


def test_state_dict_contract_matches_reference_checkpoint_names():
    from tgpose_amd import PoseNet9D, seeded_state_dict, state_spec
    net = PoseNet9D()
    sd = net.state_dict()
    spec = state_spec()
    assert list(sd) == list(spec) and all(tuple(sd[k].shape) == spec[k] for k in spec)
    assert sum(p.numel() for p in net.parameters()) == 27430569
    net.load_state_dict(seeded_state_dict(0), strict=True)
    enc = PoseNet9D(only_encoder=True)                  # net2 of trainer/RL_TDA.py: prefix face_enc.
    assert list(enc.state_dict()) == list(state_spec(only_encoder=True))
    assert all(k.startswith("face_enc.") for k in enc.state_dict())


def test_seeded_weights_are_deterministic_and_independent_of_order():
    from tgpose_amd import seeded_state_dict
    a, b = seeded_state_dict(3), seeded_state_dict(3)
    assert all(torch.equal(a[k], b[k]) for k in a)
    c = seeded_state_dict(4)
    assert not torch.equal(a["rot_green.conv1.weight"], c["rot_green.conv1.weight"])
    assert (a["ts.bn1.running_var"] > 0).all()


def test_flags_defaults_and_override():
    from tgpose_amd import FLAGS
    assert (FLAGS.gcn_n_num, FLAGS.gcn_sup_num, FLAGS.obj_c, FLAGS.output_channels) == (20, 7, 6, 2500)
    assert (FLAGS.feat_c_R, FLAGS.R_c, FLAGS.feat_c_ts, FLAGS.Ts_c) == (1286, 4, 1289, 6)
    old = FLAGS.train
    FLAGS.train = 0
    assert FLAGS.train == 0
    FLAGS.train = old
    with pytest.raises(AttributeError):
        FLAGS.no_such_flag


def test_random_subsample_order_matches_reference_draws():
    """Face_Enc.forward draws randperm(N) for pool_1 then randperm(N//4) for pool_2 from the global CPU
    generator (gcn3d.py:242); the golden files hold what the reference drew after manual_seed."""
    from tgpose_amd import engine
    for name in ("forward_b2_n1028.npz", "forward_b3_n256.npz", "forward_bottle.npz"):
        gd = golden(name)
        torch.manual_seed(int(gd["forward_seed"]))
        i1, i2 = engine.draw_sample_idx(gd["points"].shape[1])
        assert np.array_equal(i1.numpy(), gd["sample_idx_1"]) and np.array_equal(i2.numpy(), gd["sample_idx_2"])


def test_no_fallback_when_library_missing(monkeypatch, tmp_path):
    from tgpose_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.TgpError):
        _lib.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tg-pose_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "/root/reference" not in text, f


def _shard_worker(rank, world, port, q, done):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tgpose_amd import shard
    lo, hi = shard.object_range(67, rank, world)
    t = shard.max_over_ranks(0.5 + rank, device="cpu")
    gathered = shard.gather_rows_to_rank0(torch.arange(lo, hi, dtype=torch.float32).view(-1, 1), 67, device="cpu")
    q.put((rank, lo, hi, t, None if gathered is None else gathered.view(-1).tolist()))      # plain Python values only
    done.wait(timeout=120)                                  # the parent has received every rank's result
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(worker, port):
    """two spawned ranks; results travel as plain lists / floats (no tensor fd-passing, which races with the sender's exit) and
    the workers stay alive until the parent has read both"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, done = ctx.Queue(), ctx.Event()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q, done)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    finally:
        done.set()
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_object_sharding_two_ranks_gloo():
    """Eval objects are independent: rank r takes a contiguous slice, no data-path collective;
    timing is the max over ranks; rank 0 can collect the small per-object outputs."""
    res = _run_two_ranks(_shard_worker, 29500 + os.getpid() % 2000)
    (r0, lo0, hi0, t0, g0), (r1, lo1, hi1, t1, g1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 34, 34, 67)
    assert t0 == t1 == 1.5
    assert g0 == [float(i) for i in range(67)] and g1 is None


def _grad_worker(rank, world, port, q, done):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tgpose_amd import shard
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in ((300, 7), (5,), (64, 64), (1000,), (3, 3))]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1)) * (i + 1) + torch.arange(p.numel(), dtype=torch.float32).view_as(p) * 1e-3
    params[1].grad = None                                  # an unused parameter is skipped, not sent
    n = shard.allreduce_gradients(params, bucket_bytes=8192)
    q.put((rank, n, [None if p.grad is None else p.grad.reshape(-1).tolist() for p in params]))
    done.wait(timeout=120)
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_two_ranks_gloo():
    """Data-parallel training step: after the exchange every rank holds the mean of the ranks' gradients, bucketed."""
    (_, n0, g0), (_, n1, g1) = _run_two_ranks(_grad_worker, 31500 + os.getpid() % 2000)
    assert n0 == n1 and n0 >= 2                            # 8 KB buckets: several messages
    for i, (a, b) in enumerate(zip(g0, g1)):
        if i == 1:
            assert a is None and b is None
            continue
        a, b = torch.tensor(a), torch.tensor(b)
        want = torch.full_like(a, 1.5) * (i + 1) + torch.arange(a.numel(), dtype=torch.float32) * 1e-3
        assert torch.allclose(a, want) and torch.equal(a, b)


def _bucket_worker(rank, world, port, q, done):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tgpose_amd import shard
    names = ["face_all.encoder.conv_1.weights", "face_all.encoder.proj_layer.0.weight", "face_all.ph_pred.linear1.weight",
             "rot_green.conv1.weight", "face_all.encoder.bn1.bias", "ts.conv4.bias", "face_all.decoder.recon_head.3.weight"]
    shapes = [(128, 1024), (40, 40), (64, 257), (33, 7), (128,), (6,), (3, 128)]
    params = [(n, torch.nn.Parameter(torch.zeros(s))) for n, s in zip(names, shapes)]
    late = ("face_all.ph_pred.", "face_all.decoder.", "rot_green.", "rot_red.", "ts.")
    b = shard.GradBuckets(params, late)
    assert all(p.grad is not None for _, p in params)
    for i, (n, p) in enumerate(params):                      # what a backward pass does: accumulate into the existing .grad (a view)
        p.grad += torch.full_like(p, float(rank + 1)) * (i + 1) + torch.arange(p.numel(), dtype=torch.float32).view_as(p) * 1e-3
    h0 = b.reduce(0)                                         # the late layers' bucket travels ...
    dummy = sum(float(p.sum()) for _, p in params[:1])       # ... while this rank keeps computing
    h1 = b.reduce(1)
    b.wait(h0), b.wait(h1)
    views_ok = all(p.grad.data_ptr() >= f.data_ptr() and p.grad.data_ptr() < f.data_ptr() + f.numel() * 4
                   for f, ps in zip(b.flat, b.params) for p in ps)
    q.put((rank, [p.grad.reshape(-1).tolist() for _, p in params], views_ok, [f.numel() for f in b.flat], dummy))
    done.wait(timeout=120)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_overlapped_gradient_exchange_two_ranks_gloo():
    """shard.GradBuckets as the overlapped step uses it: gradients accumulate into views of two flat buckets (late layers /
    encoder), bucket 0's exchange is started before bucket 1's "backward" is over, both ranks end with the same averaged
    gradients; the never-used proj_layer is not sent (each rank keeps its own)."""
    res = _run_two_ranks(_bucket_worker, 33500 + os.getpid() % 2000)
    (_, g0, ok0, n0, _), (_, g1, ok1, n1, _) = res
    assert ok0 and ok1 and n0 == n1 and all(n % 8 == 0 for n in n0)
    for i, (a, b) in enumerate(zip(g0, g1)):
        a, b = torch.tensor(a), torch.tensor(b)
        base = torch.arange(a.numel(), dtype=torch.float32) * 1e-3
        if i == 1:                                            # proj_layer: not exchanged
            assert torch.allclose(a, base + 2.0) and torch.allclose(b, base + 4.0)
            continue
        assert torch.allclose(a, base + 1.5 * (i + 1)) and torch.equal(a, b), i


def test_bench_self_launch_builds_rank_environments_and_propagates_failure(capfd):
    """`python bench.py --gpus N` without a launcher starts its ranks as child processes (bench.self_launch): each child gets the
    environment torch.distributed.run would give it, rank 0's output is relayed, and one failing rank fails the call (and stops the
    others).  CPU-only: the children here are one-line Python programs, not the bench."""
    import sys
    import time
    sys.path.insert(0, ROOT)
    import bench
    env = bench.rank_env(3, 8, 29511, base={"PATH": "/bin"})
    assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"], env["MASTER_PORT"]) == ("3", "3", "8", "127.0.0.1", "29511")
    assert env["PATH"] == "/bin" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    ok = [sys.executable, "-c", "import os, json; r = int(os.environ['RANK']); w = int(os.environ['WORLD_SIZE']);"
          "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0;"
          "print(json.dumps({'rank': r, 'world': w}))"]
    assert bench.self_launch(3, cmd=ok) == 0
    out = capfd.readouterr().out.strip().splitlines()
    assert out == ['{"rank": 0, "world": 3}']                    # only rank 0's line is relayed
    # rank 1 fails at once; ranks 0 and 2 would run for a minute: the call returns rank 1's code quickly, the others are stopped
    bad = [sys.executable, "-c", "import os, sys, time; r = int(os.environ['RANK']);\n"
           "if r == 1: sys.exit(7)\ntime.sleep(60)"]
    t0 = time.time()
    assert bench.self_launch(3, cmd=bad) == 7
    assert time.time() - t0 < 30
    assert capfd.readouterr().out == ""                          # no JSON line from a failed job


def test_input_side_host_packing_and_category_tables():
    """Host logic of the input side that needs no GPU: window descriptors / mask offsets of a batch of frames, the refusal of
    inconsistent frames, and the evaluater's per-category tables against the reference's values (load_data_eval.py:477-566)."""
    import numpy as np
    from tgpose_amd.evaluation import load_data_eval as lde
    from tgpose_amd.evaluater.RT_TDA_Evaluater import MEAN_SHAPE_MM, SYM_INFO, SYNSET_NAMES
    from tests.util import synth_depth_scene
    frames = [synth_depth_scene(1, 3), synth_depth_scene(2, 0), synth_depth_scene(3, 2, edge_cases=True)]
    frames[1]["pred_masks"] = np.zeros((480, 640, 0), bool)
    frames[1]["pred_bboxes"] = np.zeros((0, 4), np.int32)
    win, det_img, off, stride = lde._windows(frames)
    assert det_img == [0, 0, 0, 2, 2] and stride == [3, 3, 3, 2, 2]
    assert off == [0, 1, 2, 480 * 640 * 3, 480 * 640 * 3 + 1]
    for (sumc, sumr, s), fr_i, j in zip(win, det_img, [0, 1, 2, 0, 1]):
        rmin, rmax, cmin, cmax = lde.get_bbox(frames[fr_i]["pred_bboxes"][j])
        assert (sumc, sumr, s) == (cmin + cmax, rmin + rmax, min(max(rmax - rmin, cmax - cmin), 640))
        assert 0 <= rmin < rmax <= 480 and 0 <= cmin < cmax <= 640 and s % 40 == 0 and s <= 440
    bad = synth_depth_scene(4, 2)
    bad["pred_bboxes"] = bad["pred_bboxes"][:1]
    import pytest
    with pytest.raises(ValueError):
        lde._windows([bad])
    assert SYNSET_NAMES == ['BG', 'bottle', 'bowl', 'camera', 'can', 'laptop', 'mug']
    assert MEAN_SHAPE_MM[1] == (87, 220, 89) and MEAN_SHAPE_MM[5] == (346, 200, 335) and MEAN_SHAPE_MM[6] == (146, 83, 114)
    assert SYM_INFO[1] == SYM_INFO[2] == (1, 1, 0, 1) and SYM_INFO[4] == (1, 1, 1, 1) and SYM_INFO[3] == (0, 0, 0, 0)
    assert SYM_INFO[5] == SYM_INFO[6] == (0, 1, 0, 0)
