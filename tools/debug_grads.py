"""Per-parameter gradient error of the HIP model vs the oracle (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jspsr_amd.JSPSR import Model
from oracle import jspsr_ref as R

nf, B, H, W, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
with_mask = len(sys.argv) > 6 and sys.argv[6] == "msk"
dtype = torch.bfloat16 if (len(sys.argv) > 7 and sys.argv[7] == "bf16") else torch.float32
ic = {"lr_dem": 1, "image": 3}
if with_mask:
    ic["mask"] = 15
shapes = R.jspsr_param_shapes(ic, nf)
sd = R.make_state_dict(shapes, seed)
inputs, gt = R.synthetic_batch(B, H, W, with_mask, seed=seed + 1)
m = Model(dict(ic, COP30=1), num_feature=nf)
m.load_state_dict(sd)
m = m.cuda().train()
m.compute_dtype = dtype
pred = m(*[t.cuda() for t in inputs])
((pred - gt.cuda()) ** 2).mean().backward()
sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in R.make_state_dict(shapes, seed).items()}
for k, v in sd64.items():
    if v.is_floating_point() and "running" not in k:
        v.requires_grad_()
ref, aux = R.jspsr_forward(sd64, [t.double() for t in inputs], True, return_aux=True)
for t in aux.values():
    t.retain_grad()
((ref - gt.double()) ** 2).mean().backward()
print("pred err", (pred.detach().cpu().double() - ref.detach()).abs().max().item())
for k, p in m.named_parameters():
    g, r = p.grad.cpu().double(), sd64[k].grad
    e = ((g - r).norm() / r.norm().clamp_min(1e-30)).item()
    flag = " <<<<" if e > 1e-3 else ""
    print(f"{k:50s} {e:.3e} |ref|={r.norm().item():.3e}{flag}")
