"""Development (GPU box): full-network gradient error against the CPU oracle's autograd in exact-fp32 GEMM mode, seed by seed -- which
inputs have NO flipped ReLU / max decision between the two sides (their gradients then agree to rounding and a tight bar applies)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tests.test_gpu_parity as T
from tgpose_amd import FLAGS, seeded_state_dict, ops

ops.GEMM_MODE = "fp32"
_, _, PR = T._oracle()
B, N = int(sys.argv[1]), int(sys.argv[2])
for seed in range(int(sys.argv[3]), int(sys.argv[4])):
    sd = seeded_state_dict(seed)
    pts, obj = T.synth_points(B, N, seed)
    torch.manual_seed(seed)
    i1 = torch.randperm(N)[: N // 4]
    sample = (i1, torch.randperm(i1.numel())[: i1.numel() // 4])
    with torch.no_grad():
        probe = PR.posenet_forward(sd, pts, obj, sample_idx=sample, train_keys=True, mode="exact", bn_train=True)
    probe.pop("_bn_new")
    weights = T._loss_weights(probe, seed)
    want_out, inter, want = T._oracle_grads(PR, sd, pts, obj, sample, weights)
    net = T._train_net(seed)
    FLAGS.train = 1
    try:
        out = net(T.g(pts), T.g(obj), sample_idx=sample, inject=inter["indices"])
    finally:
        FLAGS.train = 0
    loss = sum((out[k] * T.g(weights[k])).sum() for k in weights)
    loss.backward()
    rel = {}
    for k, w in want.items():
        gp = dict(net.named_parameters())[k].grad
        rel[k] = (gp.cpu() - w).norm().item() / (w.norm().item() + 1e-2)
    big = {k: v for k, v in rel.items() if want[k].norm().item() > 1e-2}          # (biases in front of a BatchNorm: zero gradient, noise)
    worst = sorted(big, key=big.get, reverse=True)[:3]
    print("seed %3d  B %d N %d: max rel %.2e (%s)  then %.2e (%s), %.2e; median %.1e" % (
        seed, B, N, big[worst[0]], worst[0], big[worst[1]], worst[1], big[worst[2]], sorted(big.values())[len(big) // 2]), flush=True)
