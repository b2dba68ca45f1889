"""Lab: does a padded PLANE PITCH of the (B,25,H,W) head tensor help the in-model propagation kernel?  At 512 x 512 a plane is
exactly 1 MiB: the 27 streams a workgroup reads (and the 25 it writes) sit at power-of-two strides.
JSPSR_LAB_PLANE_PAD=<floats> python tools/lab/k1_plane_pad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jspsr_amd import _lib, ops  # noqa: E402

pad = int(os.environ.get("JSPSR_LAB_PLANE_PAD", "0"))
B, H, W = 8, 512, 512
dev = "cuda"
lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(1)
dem = torch.rand(B, 1, H, W, device=dev, generator=g)
gout = torch.randn(B, 1, H, W, device=dev, generator=g)
out = torch.empty_like(dem)
w, b = torch.ones(9, device=dev), torch.zeros(1, device=dev)
ps = H * W + pad
nset = 8
shift = int(os.environ.get("K1_LAB_SHIFT_KB", "0"))
dummy = torch.empty(max(shift, 1) * 256, device=dev) if shift else None       # moves every later allocation by `shift` KiB
nset = int(os.environ.get("K1_LAB_NSET", str(nset)))
heads = [1.5 * torch.randn(B * 25 * ps, device=dev, generator=g) for _ in range(nset)]
print("head addresses mod 2 MiB (KiB):", [h.data_ptr() % (2 << 20) // 1024 for h in heads[:4]], "mod 1 GiB (MiB):", [h.data_ptr() % (1 << 30) >> 20 for h in heads[:4]], flush=True)
gheads = heads if os.environ.get("K1_LAB_INPLACE") == "1" else [torch.empty_like(t) for t in heads]
ws = ops.prop_backward_workspace(B, H, W, dev)
st = lambda: torch.cuda.current_stream().cuda_stream
lf = lambda i: _lib.check(lib.jspsr_prop_logits_forward_f32(dem.data_ptr(), heads[i % nset].data_ptr(), w.data_ptr(), b.data_ptr(), 1.0, out.data_ptr(), B, H, W, st()), "lf")
lb = lambda i: _lib.check(lib.jspsr_prop_logits_backward_f32(gout.data_ptr(), dem.data_ptr(), heads[i % nset].data_ptr(), w.data_ptr(), gheads[i % nset].data_ptr(), None, None, ws.data_ptr(), B, H, W, st()), "lb")
px = B * H * W
res = []
for name, fn, nbytes in (("fwd", lf, 108.0 * px), ("bwd", lb, 208.0 * px)):
    ts = []
    for rep in range(5):
        for i in range(300):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            fn(i)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    res.append(f"{name} " + " ".join(f"{t:.1f}" for t in ts) + f" us (best {nbytes / min(ts) / 8e6:.3f})")
print(f"plane pad {pad:6d} floats: " + " | ".join(res), flush=True)
