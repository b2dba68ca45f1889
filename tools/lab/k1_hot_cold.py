"""Lab: does the chip's state after MFMA-dense work change what the HBM-bound K1 kernels reach?  Times the planar K1 pair
cold, then right after N seconds of back-to-back bf16 GEMMs (what the benchmark's training loop leaves behind), then again
after an idle second.  python tools/lab/k1_hot_cold.py [seconds=3]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jspsr_amd import ops  # noqa: E402

B, H, W = 8, 512, 512
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
dem = torch.rand(B, 1, H, W, device=dev, generator=g)
sets = [(torch.sigmoid(torch.randn(B, 9, H, W, device=dev, generator=g)), 1.5 * torch.randn(B, 16, H, W, device=dev, generator=g)) for _ in range(4)]
gsets = [(torch.empty(B, 9, H, W, device=dev), torch.empty(B, 16, H, W, device=dev)) for _ in range(4)]
w, b = torch.ones(1, 1, 3, 3, device=dev), torch.zeros(1, device=dev)
gout = torch.randn(B, 1, H, W, device=dev, generator=g)
out = torch.empty_like(dem)
ws = ops.prop_backward_workspace(B, H, W, dev)
fwd = lambda i: ops.prop_forward_raw(dem, sets[i % 4][0], sets[i % 4][1], w, b, 1.0, out)
bwd = lambda i: ops.prop_backward_raw(gout, dem, sets[i % 4][0], sets[i % 4][1], w, gsets[i % 4][0], gsets[i % 4][1], None, None, ws)


def t(fn, n=20):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def show(tag):
    f, bw = t(fwd), t(bwd)
    px = B * H * W
    print(f"{tag:34s} fwd {f:6.1f} us ({108.0 * px / f / 8e6:.3f})  bwd {bw:6.1f} us ({208.0 * px / bw / 8e6:.3f})", flush=True)


secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
show("cold")
show("cold again")
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
c = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
t0 = time.time()
while time.time() - t0 < secs:
    for _ in range(20):
        a @ c
    torch.cuda.synchronize()
show(f"right after {secs:.0f} s of bf16 GEMMs")
show("next")
time.sleep(1.0)
show("after 1 s idle")
big = [torch.empty(1 << 30, dtype=torch.uint8, device=dev) for _ in range(40)]      # 40 GiB held by the caching allocator, like the model's activations
del big
show("with 40 GiB cached in the allocator")
