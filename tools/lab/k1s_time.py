"""Lab: K1s backward (jspsr_prop_step_backward_f32) at 8 x 512 x 512, with / without accumulate and grad_dem."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jspsr_amd import ops as O  # noqa: E402

B, H, W = 8, 512, 512
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
dem = torch.rand(B, 1, H, W, device=dev, generator=g)
sets = [(torch.sigmoid(torch.randn(B, 9, H, W, device=dev, generator=g)), sigma * torch.randn(B, 16, H, W, device=dev, generator=g)) for _ in range(4)]
gs = [(torch.zeros(B, 9, H, W, device=dev), torch.zeros(B, 16, H, W, device=dev)) for _ in range(4)]
ones9 = torch.ones(9, device=dev)
ws = O._step_workspace(B, H, W, dev)
gdem = torch.zeros_like(dem)
gout = torch.randn(B, 1, H, W, device=dev, generator=g)
for acc in (1, 0):
    for gd in (gdem, None):
        f = lambda i: O._step_backward(gout, dem, sets[i % 4][0], sets[i % 4][1], ones9, 0.0, 0, acc, gs[i % 4][0], gs[i % 4][1], gd, ws)
        for i in range(50):
            f(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            f(i)
        e1.record()
        e1.synchronize()
        print(f"[{os.environ.get('JSPSR_LAB_LIB', 'product')[-24:]:24s}] sigma {sigma}: K1s backward accumulate {acc} grad_dem {gd is not None}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)
