"""Lab: dump the head-fed propagation step's results (forward, head gradient, parameter gradients) for fixed seeded
inputs, to compare the two kernel families bit for bit across processes: JSPSR_PROP_HEAD_DMA=0|1 python tools/lab/k1h_ab_dump.py out.pt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jspsr_amd import ops  # noqa: E402

res = {}
for (B, H, W, sig) in ((2, 128, 128, 1.5), (8, 512, 512, 1.5), (2, 128, 128, 6.0), (3, 36, 200, 3.0)):
    g = torch.Generator().manual_seed(B * 1000 + H + W)
    dem = torch.rand(B, 1, H, W, generator=g).cuda()
    head = torch.randn(B, H, W, 32, generator=g)
    head.view(B, H, W, 8, 4)[..., 1:3] *= sig
    head = head.bfloat16().cuda().requires_grad_()
    w = (1 + 0.3 * torch.randn(1, 1, 3, 3, generator=g)).cuda().requires_grad_()
    b = (0.1 * torch.randn(1, generator=g)).cuda().requires_grad_()
    gout = torch.randn(B, 1, H, W, generator=g).cuda()
    for rep in range(3):
        head.grad = w.grad = b.grad = None
        out = ops.propagate_head(dem, head, w, b, 1.0)
        out.backward(gout)
        res[(B, H, W, sig, rep)] = (out.detach().cpu(), head.grad.cpu(), w.grad.cpu(), b.grad.cpu())
torch.save(res, sys.argv[1])
print("saved", len(res))
