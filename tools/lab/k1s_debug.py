import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jspsr_amd import ops
from oracle import jspsr_ref as R
g_ = torch.Generator().manual_seed(11)
B, H, W = 3, 45, 200
aff = torch.rand(B, 9, H, W, generator=g_)
off = (2.0 * torch.randn(B, 18, H, W, generator=g_)).clamp(-6.5, 6.5)
off[:, 8:10] = 0
wk = 1 + 0.3 * torch.randn(9, generator=g_)
v = torch.randn(B, 1, H, W, generator=g_)
g = torch.randn(B, 1, H, W, generator=g_)
ws = ops._step_workspace(B, H, W, "cuda")
gd = torch.zeros_like(v).cuda()
ops._step_backward(g.cuda(), v.cuda(), aff.cuda(), off.cuda(), wk.cuda(), 0.7, 0, 0, torch.empty_like(aff).cuda(), torch.empty_like(off).cuda(), gd, ws)
jv = ops._step_forward(v.cuda(), aff.cuda(), off.cuda(), wk.cuda(), torch.zeros(1).cuda(), 0.7, 0, torch.empty_like(v).cuda())
vd = v.double().requires_grad_()
S = R.sample_taps(vd, off.double())
out = (wk.double().view(1, 9, 1, 1) * aff.double() * S).sum(1, keepdim=True) + 0.7 * vd
out.backward(g.double())
print("forward HIP vs oracle", (jv.cpu().double() - out.detach()).abs().max().item())
d = (gd.cpu().double() - vd.grad).abs()
print("backward HIP vs oracle max", d.max().item(), "at", divmod(int(d.argmax()), W), "mean", d.mean().item())
print("adjoint", (g.double() * jv.cpu().double()).sum().item(), (gd.cpu().double() * v.double()).sum().item())
print("per image max err", [d[b].max().item() for b in range(B)])
print("rows max err", [round(d[0, 0, y].max().item(), 3) for y in range(0, H, 4)])
