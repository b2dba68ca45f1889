import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jspsr_amd.JSPSR import Model
from jspsr_amd import engine as E, ops
from oracle import jspsr_ref as R

nf, B, H, W, seed = 8, 1, 64, 64, 5
ic = {"lr_dem": 1, "image": 3}
shapes = R.jspsr_param_shapes(ic, nf)
sd = R.make_state_dict(shapes, seed)
inputs, gt = R.synthetic_batch(B, H, W, False, seed=seed + 1)
m = Model(dict(ic, COP30=1), num_feature=nf)
m.load_state_dict(sd)
m = m.cuda().train()
cap = {}
orig = m.postprocessor.forward
def hooked(dem, weight, offset):
    weight.retain_grad(); offset.retain_grad()
    cap["w"], cap["o"], cap["dem"] = weight, offset, dem
    return orig(dem, weight, offset)
m.postprocessor.forward = hooked
pred = m(*[t.cuda() for t in inputs])
((pred - gt.cuda()) ** 2).mean().backward()
sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in R.make_state_dict(shapes, seed).items()}
ref, aux = R.jspsr_forward(sd64, [t.double() for t in inputs], True, return_aux=True)
aux["offset"].requires_grad_(); aux["weight"].requires_grad_()
# recompute the propagation alone on the oracle's own operands with grads
w_o = aux["weight"].detach().requires_grad_(); o_o = aux["offset"].detach().requires_grad_()
out = R.propagate(inputs[0].double(), w_o, o_o, sd64["postprocessor.w"], sd64["postprocessor.b"])
gout = (2 * (out - gt.double()) / out.numel()).detach()
out.backward(gout)
off18 = torch.cat((cap["o"][:, :8], torch.zeros_like(cap["o"][:, :2]), cap["o"][:, 8:]), 1).detach().cpu().double()
print("offset fwd diff", (off18 - o_o.detach()).abs().max().item(), "weight fwd diff", (cap["w"].detach().cpu().double() - w_o.detach()).abs().max().item())
go_hip = cap["o"].grad.cpu().double()
go_ref = torch.cat((o_o.grad[:, :8], o_o.grad[:, 10:]), 1)
d = (go_hip - go_ref).abs()
print("grad_offset rel", (d.norm() / go_ref.norm()).item(), "max", d.max().item(), "ref max", go_ref.abs().max().item())
print("grad_weight rel", ((cap["w"].grad.cpu().double() - w_o.grad).norm() / w_o.grad.norm()).item())
idx = (d > 0.2 * d.max()).nonzero()
print("n bad", len(idx))
for i in idx[:12]:
    i = tuple(i.tolist())
    ch = i[1] if i[1] < 8 else i[1] + 2
    k = ch // 2
    print(i, "hip", go_hip[i].item(), "ref", go_ref[i].item(), "off(dy,dx)", off18[0, 2*k, i[2], i[3]].item(), off18[0, 2*k+1, i[2], i[3]].item())
# analytic closed form on the oracle operands
ga = R.propagate_analytic_backward(inputs[0].double(), w_o.detach(), o_o.detach(), sd64["postprocessor.w"], sd64["postprocessor.b"], gout)[1]
ga16 = torch.cat((ga[:, :8], ga[:, 10:]), 1)
print("analytic vs autograd rel", ((ga16 - go_ref).norm() / go_ref.norm()).item(), " hip vs analytic rel", ((go_hip - ga16).norm() / ga16.norm()).item())
# K1 alone on the oracle's exact operands (fp32-rounded)
wt = w_o.detach().float().cuda().requires_grad_(); of = torch.cat((o_o.detach()[:, :8], o_o.detach()[:, 10:]), 1).float().cuda().requires_grad_()
o2 = ops.propagate(inputs[0].cuda(), wt, of, sd["postprocessor.w"].cuda(), sd["postprocessor.b"].cuda())
o2.backward(gout.float().cuda())
print("K1 alone vs analytic rel", ((of.grad.cpu().double() - ga16).norm() / ga16.norm()).item())
