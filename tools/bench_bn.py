"""K4 micro-benchmark: the BatchNorm backward (reduce + finalize + apply; 5 tensor passes: dy, x read twice, dx written) and
forward apply as the models call them, effective GB/s over the bytes they must move."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jspsr_amd import kernels as K  # noqa: E402

SHAPES = [(8, 512, 512, 64), (8, 256, 256, 128), (8, 128, 128, 256), (8, 64, 64, 512), (8, 512, 512, 32)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for dtype in (torch.bfloat16,):
    for B, H, W, C in SHAPES:
        x = torch.randn(B, H, W, C, device="cuda").to(dtype)
        dy = torch.randn(B, H, W, C, device="cuda").to(dtype)
        gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        y, mean, invstd = K.bn_forward(x, gamma, beta, rm, rv, 0.1, 1e-5, True, relu=True)
        nbytes = x.numel() * x.element_size()
        t_f = timeit(lambda: K.bn_forward(x, gamma, beta, rm, rv, 0.1, 1e-5, True, relu=True))
        t_b = timeit(lambda: K.bn_backward(dy, None, x, gamma, mean, invstd, True, 2, beta=beta))
        t_by = timeit(lambda: K.bn_backward(dy, y, x, gamma, mean, invstd, True, 1))
        print(f"{str(dtype)[6:]} B{B} {H}x{W}x{C}: forward (stats + apply, 3 passes) {t_f*1e6:7.1f} us {3*nbytes/t_f/1e9:6.0f} GB/s | "
              f"backward, mask from x (5 passes) {t_b*1e6:7.1f} us {5*nbytes/t_b/1e9:6.0f} GB/s | mask from y (6 passes) {t_by*1e6:7.1f} us {6*nbytes/t_by/1e9:6.0f} GB/s", flush=True)
