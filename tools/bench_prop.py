"""K1 micro-benchmark: achieved algorithmic GB/s of the fused propagation kernel.
Usage: python tools/bench_prop.py [B H W sigma oc]   (JSPSR_PROP_PX=1|2|4 selects the variant)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jspsr_amd import ops  # noqa: E402


def main():
    a = sys.argv[1:]
    B, H, W = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (8, 512, 512)
    sigma = float(a[3]) if len(a) > 3 else 1.5
    oc = int(a[4]) if len(a) > 4 else 18
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    dem = torch.rand(B, 1, H, W, device=dev, generator=g)
    weight = torch.sigmoid(torch.randn(B, 9, H, W, device=dev, generator=g)).requires_grad_()
    offset = (sigma * torch.randn(B, oc, H, W, device=dev, generator=g)).requires_grad_()
    w = torch.ones(1, 1, 3, 3, device=dev).requires_grad_()
    b = torch.zeros(1, device=dev).requires_grad_()
    gout = torch.randn(B, 1, H, W, device=dev, generator=g)
    # rotate over several operand sets so the 256 MiB Infinity Cache cannot hold the working set
    nset = max(1, int(600e6 // (B * H * W * 4 * (1 + 9 + oc))) + 1)
    sets = [(weight.detach().clone().requires_grad_(), offset.detach().clone().requires_grad_()) for _ in range(nset)]
    px = os.environ.get("JSPSR_PROP_PX", "default")
    npx = B * H * W
    fwd_b = (1 + 9 + oc + 1) * 4 * npx
    bwd_b = (1 + 1 + 9 + oc + 9 + oc) * 4 * npx
    for name, nbytes in (("fwd", fwd_b), ("bwd", bwd_b)):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        outs = []
        for it in range(3 + 20):
            if it == 3:
                torch.cuda.synchronize()
                ev[0].record()
            wt, of = sets[it % nset]
            if name == "fwd":
                with torch.no_grad():
                    ops.propagate(dem, wt, of, w, b)
            else:
                o = ops._Propagate.apply(dem, wt, of, w, b, 1.0)
                torch.autograd.grad(o, (wt, of, w, b), gout)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 20
        if name == "bwd":
            ms_b = ms - fwd_ms
            print(f"PX={px} {B}x{H}x{W} oc={oc} sigma={sigma} bwd(only) {ms_b*1e3:.1f} us  {bwd_b/ms_b/1e6:.0f} GB/s (fwd+bwd {ms*1e3:.1f} us)")
        else:
            fwd_ms = ms
            print(f"PX={px} {B}x{H}x{W} oc={oc} sigma={sigma} fwd {ms*1e3:.1f} us  {nbytes/ms/1e6:.0f} GB/s  sets={nset}")


if __name__ == "__main__":
    main()
