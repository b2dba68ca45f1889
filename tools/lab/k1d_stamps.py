"""Lab: per-phase cycle counts of the persistent LDS-DMA backward kernel (prop_dma.hip, -DK1D_STAMPS build).
JSPSR_LAB_LIB=jspsr_amd/lib_lab/libjspsr_k1d_stamps.so python tools/lab/k1d_stamps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jspsr_amd import ops  # noqa: E402

B, H, W = 8, 512, 512
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
dem = torch.rand(B, 1, H, W, device=dev, generator=g)
wt = torch.sigmoid(torch.randn(B, 9, H, W, device=dev, generator=g))
off = 1.5 * torch.randn(B, 16, H, W, device=dev, generator=g)
w = torch.ones(1, 1, 3, 3, device=dev)
gout = torch.randn(B, 1, H, W, device=dev, generator=g)
gw, go = torch.empty_like(wt), torch.empty_like(off)
ws = ops.prop_backward_workspace(B, H, W, dev)
for _ in range(3):
    ops.prop_backward_raw(gout, dem, wt, off, w, gw, go, None, None, ws)
torch.cuda.synchronize()
raw = ws.cpu().numpy()
import numpy as np
rows = int(np.frombuffer(raw[:4].tobytes(), dtype=np.int32)[0])
nw = int(os.environ.get("JSPSR_PROP_NW", "8"))
st = np.frombuffer(raw[16 + 4096 * 40:16 + 4096 * 40 + rows * 2 * nw * 64].tobytes(), dtype=np.uint64).reshape(rows, 2, nw, 8).astype(np.float64)
names = ["wait", "barrier", "lift+issue", "compute", "store", "loop"]
for role, rn in ((0, "compute waves"), (1, "mover waves")):
    r = st[:, role]
    tiles = r[..., 7]
    print(f"{rn}: grid {rows} x {nw}, tiles per workgroup {tiles.mean():.1f}; cycles per tile per wave (mean / median / max over waves):")
    for i, n in enumerate(names):
        v = r[..., i] / tiles
        print(f"  {n:10s} {v.mean():8.0f} {np.median(v):8.0f} {v.max():8.0f}")
    tot = r[..., 6]
    print(f"  whole kernel per wave: mean {tot.mean():.0f} cycles, max {tot.max():.0f}  (s_memtime ticks)")
