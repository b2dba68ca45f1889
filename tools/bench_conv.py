"""K2 micro-benchmark: TFLOP/s of the implicit-GEMM conv kernels vs torch/MIOpen on the same shapes."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jspsr_amd import kernels as K  # noqa: E402

SHAPES = [  # B, H, W, Cin, Cout, k, stride
    (8, 512, 512, 64, 64, 3, 1),
    (8, 512, 512, 128, 128, 3, 1),
    (8, 256, 256, 128, 128, 3, 1),
    (8, 128, 128, 256, 256, 3, 1),
    (8, 64, 64, 512, 512, 3, 1),
    (8, 64, 64, 1536, 256, 3, 1),
    (8, 512, 512, 256, 64, 3, 1),
    (8, 512, 512, 192, 128, 3, 2),
    (8, 512, 512, 64, 64, 1, 1),     # 8: K sweep at N = 64 (fixed per-tile cost vs per-stage cost)
    (8, 512, 512, 64, 64, 5, 1),     # 9
    (8, 512, 512, 128, 128, 1, 1),   # 10
    (8, 512, 512, 128, 128, 5, 1),   # 11
]


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
    only = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) >= 0 else None      # shape index (-1: all)
    dts = (torch.bfloat16,) if len(sys.argv) > 3 and sys.argv[3] == "bf16" else (torch.float32, torch.bfloat16)
    ours_only = len(sys.argv) > 4
    with_stats = len(sys.argv) > 5 and sys.argv[5] == "stats"     # forward: also the BatchNorm partial statistics from the epilogue
    with_bias = len(sys.argv) > 5 and sys.argv[5] == "bias"       # forward: bias + ReLU epilogue (the BN-free units)
    for dtype in dts:
        for si, (B, H, W, Ci, Co, k, s) in enumerate(SHAPES):
            if only is not None and si != only:
                continue
            pad = k // 2
            x = torch.randn(B, H, W, Ci, device="cuda").to(dtype)
            w = torch.randn(Co, Ci, k, k, device="cuda") / (Ci * k * k) ** 0.5
            OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
            flops = 2.0 * B * OH * OW * Co * Ci * k * k
            if which == "fwd":
                wp = K.pack_weight(w, 0, Ci, dtype)
                b = torch.randn(Co, device="cuda") if with_bias else None
                t = timeit(lambda: K.conv2d_forward(x, wp, b, s, pad, relu=with_bias, stats=with_stats))
                xc = x.permute(0, 3, 1, 2)  # channels_last view
                wc = w.to(dtype).contiguous(memory_format=torch.channels_last)
                tm = float("nan") if ours_only else timeit(lambda: F.conv2d(xc, wc, None, s, pad))
            elif which == "wgrad":
                go = torch.randn(B, OH, OW, Co, device="cuda").to(dtype)
                t = timeit(lambda: K.conv2d_wgrad(go, x, Co, Ci, k, k, s, pad))
                tm = float("nan")
            else:
                go = torch.randn(B, OH, OW, Co, device="cuda").to(dtype)
                wpt = K.pack_weight(w, 1, Co, dtype)
                t = timeit(lambda: K.conv2d_dgrad(go, wpt, (H, W), s, pad))
                tm = float("nan")
            print(f"{which} {str(dtype)[6:]:9s} B{B} {H}x{W} {Ci}->{Co} k{k} s{s}: ours {t*1e3:8.3f} ms {flops/t/1e12:7.1f} TF | miopen {tm*1e3:8.3f} ms {flops/tm/1e12:7.1f} TF", flush=True)


if __name__ == "__main__":
    main()
