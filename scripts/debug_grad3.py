import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from tgpose_amd import PoseNet9D, seeded_state_dict, autograd as AG, ops
dev = "cuda:0"
B, N, seed = 3, 256, 41
net = PoseNet9D(); net.load_state_dict(seeded_state_dict(seed)); net = net.to(dev).train()
dec = net.face_all.decoder
gen = torch.Generator().manual_seed(0)
feat_c = torch.randn(B, N, 1286, generator=gen).abs()
three_d = len(sys.argv) > 1
x0 = feat_c.to(dev) if three_d else feat_c.view(B * N, 1286).to(dev)
dyo = torch.randn(*x0.shape[:-1], 3, generator=torch.Generator().manual_seed(1)).to(dev)
layers = [(dec.conv1d_block[0], dec.conv1d_block[1]), (dec.conv1d_block[3], dec.conv1d_block[4]), (dec.conv1d_block[6], dec.conv1d_block[7]),
          (dec.recon_head[0], dec.recon_head[1])]
def chain(mine, dt):
    x = F.pad(x0, (0, 2)).to(dt).requires_grad_()
    keep = [x]
    h = x
    for conv, bn in layers:
        W = conv.weight[:, :, 0].detach().to(dt); W = F.pad(W, (0, h.shape[-1] - W.shape[1])).requires_grad_()
        b = conv.bias.detach().to(dt).requires_grad_(); g = bn.weight.detach().to(dt).requires_grad_(); be = bn.bias.detach().to(dt).requires_grad_()
        if mine:
            z = AG.linear(h, W, b); z.retain_grad()
            h = AG._BNAct.apply(z, g, be, None, 1, 0.0); h.retain_grad()
        else:
            z = F.linear(h, W, b); z.retain_grad()
            zz = z.reshape(-1, z.shape[-1])
            h = torch.relu(F.batch_norm(zz, None, None, g, be, True, 0.0, 1e-5)).view(z.shape); h.retain_grad()
        keep += [W, b, g, be, z, h]
    W3 = dec.recon_head[3].weight[:, :, 0].detach().to(dt).requires_grad_(); b3 = dec.recon_head[3].bias.detach().to(dt).requires_grad_()
    out = (AG.linear(h, W3, b3) if mine else F.linear(h, W3, b3))
    out.backward(dyo.to(dt))
    global fwd
    fwd = [k.detach().double() for k in keep[5::6]] + [k.detach().double() for k in keep[6::6]]
    return [k.grad for k in keep] + [W3.grad]
a = chain(True, torch.float32); fa = fwd
r = chain(False, torch.float64); fr = fwd
for i, (u, v) in enumerate(zip(fa, fr)):
    print("forward", "z" if i < 4 else "h", i % 4, "max abs diff %.2e (max %.2e)" % ((u - v).abs().max().item(), v.abs().max().item()))
t32 = chain(False, torch.float32)
names = ["x"] + [n + str(i) for i in range(4) for n in ("W", "b", "gamma", "beta", "dz", "dh")] + ["W3"]
for n, u, v in zip(names, a, r):
    print("%-8s rel err %.2e  (max %.2e)" % (n, (u.double() - v).abs().max().item() / (v.abs().max().item() + 1e-30), v.abs().max().item()))
for n, u, v in zip(names, t32, r):
    print("torch32 %-8s rel err %.2e" % (n, (u.double() - v).abs().max().item() / (v.abs().max().item() + 1e-30)))

# ---- isolate the BatchNorm backward of layer 2 on its real data
conv, bn = layers[2]
with torch.no_grad():
    h = F.pad(x0, (0, 2))
    for i, (cv, b_) in enumerate(layers[:2]):
        W = F.pad(cv.weight[:, :, 0], (0, h.shape[-1] - cv.weight.shape[1]))
        h = torch.relu(F.batch_norm(F.linear(h, W, cv.bias).reshape(-1, W.shape[0]), None, None, b_.weight, b_.bias, True, 0.0, 1e-5))
    z2 = F.linear(h, conv.weight[:, :, 0], conv.bias).reshape(-1, 256).contiguous()
dh2 = r[1 + 6 * 2 + 5].float().reshape(-1, 256).contiguous()
zz = z2.double().requires_grad_(); g = bn.weight.detach().double().requires_grad_(); be = bn.bias.detach().double().requires_grad_()
torch.relu(F.batch_norm(zz, None, None, g, be, True, 0.0, 1e-5)).backward(dh2.double())
_, mean, var = ops.bn_train(z2, bn.weight.detach(), bn.bias.detach(), 1e-5, 1, 0.0, out=torch.empty_like(z2))
print("stats err", (mean.double() - z2.double().mean(0)).abs().max().item(), (var.double() - z2.double().var(0, unbiased=False)).abs().max().item(),
      "min var", var.min().item())
dx, dg, db = ops.bn_bwd(dh2.clone(), z2, mean, var, bn.weight.detach(), bn.bias.detach(), 1e-5, 1, 0.0, dx=torch.empty_like(z2))
for n, u, v in (("dx", dx, zz.grad), ("dgamma", dg, g.grad), ("dbeta", db, be.grad)):
    e = (u.double() - v).abs()
    print(n, "rel err %.2e" % (e.max().item() / v.abs().max().item()), "worst channel", int(e.reshape(-1, 256).max(0)[0].argmax()) if e.dim() > 1 else int(e.argmax()))
ch = int((dx.double() - zz.grad).abs().max(0)[0].argmax())
print("channel", ch, "var", var[ch].item(), "gamma", bn.weight[ch].item(), "beta", bn.bias[ch].item(), "mean", mean[ch].item())
zc = (z2[:, ch].double() - mean[ch].double()) / torch.sqrt(var[ch].double() + 1e-5) * bn.weight[ch].double() + bn.bias[ch].double()
print("z near zero:", (zc.abs() < 1e-4).sum().item(), "of", zc.numel(), " positive:", (zc > 0).sum().item())
