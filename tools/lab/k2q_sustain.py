"""Lab: is the in-step slowdown of K2q (717 us in the step against 523 us in a 40 ms micro-benchmark, 8 x 512^2 forward +
statistics) the chip's sustained power state?  The same launch timed in blocks of 50 over ~3 s of back-to-back launches,
then the patch kernel the same way (JSPSR_CONV_RESIDENT128=0 in a child)."""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jspsr_amd import kernels as K  # noqa: E402


def child():
    B, H, W = 8, 512, 512
    x = torch.randn(B, H, W, 128, device="cuda").to(torch.bfloat16)
    xr = torch.relu(torch.randn(B, H, W, 128, device="cuda")).to(torch.bfloat16)      # half zeros, as a post-ReLU activation
    w = torch.randn(128, 128, 3, 3, device="cuda") / (128 * 9) ** 0.5
    wp = K.pack_weight(w, 0, 128, torch.bfloat16)
    for name, inp in (("randn input", x), ("post-ReLU input", xr)):
        ts = []
        for blk in range(60):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                K.conv2d_forward(inp, wp, None, 1, 1, stats=True)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 50 * 1e3)
        print(f"  {name}: us per launch, blocks of 50: first {ts[0]:.0f} {ts[1]:.0f} {ts[2]:.0f} ... mid {ts[28]:.0f} {ts[29]:.0f} {ts[30]:.0f} ... last {ts[-3]:.0f} {ts[-2]:.0f} {ts[-1]:.0f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for tag, env in (("K2q", {"JSPSR_CONV_RESIDENT128": "1"}), ("patch kernel", {"JSPSR_CONV_RESIDENT128": "0"})):
            print(f"== {tag}", flush=True)
            subprocess.run([sys.executable, __file__, "child"], env={**os.environ, **env}, timeout=600, check=True)
