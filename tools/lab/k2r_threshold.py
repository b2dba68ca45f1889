"""Lab: K2r against the patch kernel on small rasters (where does the persistent kernel stop paying?).
Run twice: JSPSR_CONV_RESIDENT_MIN=1 (K2r everywhere) and JSPSR_CONV_RESIDENT=0."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jspsr_amd import kernels as K

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for B, H, W in [(1, 128, 128), (1, 256, 256), (2, 256, 256), (1, 512, 512), (2, 512, 512), (4, 512, 512), (8, 512, 512), (16, 512, 512)]:
    x = torch.randn(B, H, W, 64, device="cuda").bfloat16()
    w = torch.randn(64, 64, 3, 3, device="cuda") / 24
    wp, wpt = K.pack_weight(w, 0, 64, torch.bfloat16), K.pack_weight(w, 1, 64, torch.bfloat16)
    tf = timeit(lambda: K.conv2d_forward(x, wp, None, 1, 1, stats=True))
    td = timeit(lambda: K.conv2d_dgrad(x, wpt, (H, W), 1, 1))
    print(f"{B}x{H}x{W}: tiles {B*H*W//256:6d}  fwd+stats {tf:8.1f} us  dgrad {td:8.1f} us", flush=True)
