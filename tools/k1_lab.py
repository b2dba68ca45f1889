"""K1 tile-shape lab: the planar kernels alone, back-to-back launches through the C ABI between events on the launch
stream (what bench.py's roofline leg does), for the shape the JSPSR_PROP_* environment selects.
Usage: JSPSR_PROP_TW=.. JSPSR_PROP_TH=.. JSPSR_PROP_PX=.. python tools/k1_lab.py [B H W sigma]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jspsr_amd import ops  # noqa: E402


WARM = int(os.environ.get("K1_LAB_WARM", "300"))    # warm-up launches in front of every timed block


def main():
    a = sys.argv[1:]
    B, H, W = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (8, 512, 512)
    sigma = float(a[3]) if len(a) > 3 else 1.5
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    dem = torch.rand(B, 1, H, W, device=dev, generator=g)
    nset = max(2, int(700e6 // (B * H * W * 4 * 26)) + 1)
    sets = [(torch.sigmoid(torch.randn(B, 9, H, W, device=dev, generator=g)),
             sigma * torch.randn(B, 16, H, W, device=dev, generator=g)) for _ in range(nset)]
    gsets = [(torch.empty(B, 9, H, W, device=dev), torch.empty(B, 16, H, W, device=dev)) for _ in range(nset)]
    w = torch.ones(1, 1, 3, 3, device=dev)
    b = torch.zeros(1, device=dev)
    gout = torch.randn(B, 1, H, W, device=dev, generator=g)
    out = torch.empty_like(dem)
    ws = ops.prop_backward_workspace(B, H, W, dev)
    fwd = lambda i: ops.prop_forward_raw(dem, sets[i % nset][0], sets[i % nset][1], w, b, 1.0, out)
    bwd = lambda i: ops.prop_backward_raw(gout, dem, sets[i % nset][0], sets[i % nset][1], w, gsets[i % nset][0],
                                          gsets[i % nset][1], None, None, ws)
    px = B * H * W
    tag = " ".join(f"{k[11:]}={os.environ[k]}" for k in sorted(os.environ) if k.startswith("JSPSR_PROP_")) or "default"
    res = []
    for name, fn, nbytes in (("fwd", fwd, 108.0 * px), ("bwd", bwd, 208.0 * px)):
        best = 1e9
        for rep in range(3):
            for i in range(WARM):      # steady state: see bench.py time_k1 / profiles/r03_k1_warmup_transient.txt
                fn(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(20):
                fn(i)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e-3)
        res.append(f"{name} {best * 1e6:6.1f} us {nbytes / best / 1e12:.3f} TB/s ({nbytes / best / 8e12:.3f})")
    print(f"[{tag:28s}] {B}x{H}x{W} s={sigma}: " + " | ".join(res), flush=True)
    # the in-model entry (round 4): logits + offsets as planes of one (B,25,H,W) tensor, sigmoid inside the kernel
    from jspsr_amd import _lib, kernels as K
    lib = _lib.load()
    st = lambda: torch.cuda.current_stream().cuda_stream
    lsets = [torch.cat((1.5 * torch.randn(B, 9, H, W, device=dev, generator=g), sigma * torch.randn(B, 16, H, W, device=dev, generator=g)), 1)
             for _ in range(nset)]
    glsets = [torch.empty_like(t) for t in lsets]
    lf = lambda i: _lib.check(lib.jspsr_prop_logits_forward_f32(dem.data_ptr(), lsets[i % nset].data_ptr(), w.data_ptr(), b.data_ptr(), 1.0,
                                                                out.data_ptr(), B, H, W, st()), "lf")
    lb = lambda i: _lib.check(lib.jspsr_prop_logits_backward_f32(gout.data_ptr(), dem.data_ptr(), lsets[i % nset].data_ptr(), w.data_ptr(),
                                                                 glsets[i % nset].data_ptr(), None, None, ws.data_ptr(), B, H, W, st()), "lb")
    res = []
    for name, fn, nbytes in (("fwd", lf, 108.0 * px), ("bwd", lb, 208.0 * px)):
        best = 1e9
        for rep in range(3):
            for i in range(WARM):
                fn(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(20):
                fn(i)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e-3)
        res.append(f"{name} {best * 1e6:6.1f} us {nbytes / best / 1e12:.3f} TB/s ({nbytes / best / 8e12:.3f})")
    print(f"   in-model logits entry    : " + " | ".join(res), flush=True)
    del lsets, glsets
    if os.environ.get("K1_LAB_LEGACY", "1") == "0":
        return
    # rounds 2-3's head-fed entry (K1h): actual bytes moved per pixel = 32 head channels (+ their gradient) + dem + out/gout
    for dt, es in ((torch.float32, 4), (torch.bfloat16, 2)):
        heads = [(sigma * torch.randn(B, H, W, 32, device=dev, generator=g)).to(dt) for _ in range(max(2, nset // 2))]
        gheads = [torch.empty_like(h) for h in heads]
        hws = torch.empty(max(lib.jspsr_prop_head_backward_workspace_bytes(B, H, W), 16), dtype=torch.uint8, device=dev)
        hf = lambda i: _lib.check(lib.jspsr_prop_head_forward(K._dt(heads[0]), dem.data_ptr(), heads[i % len(heads)].data_ptr(),
                                                              w.data_ptr(), b.data_ptr(), 1.0, out.data_ptr(), B, H, W, st()), "hf")
        hb = lambda i: _lib.check(lib.jspsr_prop_head_backward(K._dt(heads[0]), gout.data_ptr(), dem.data_ptr(),
                                                               heads[i % len(heads)].data_ptr(), w.data_ptr(),
                                                               gheads[i % len(heads)].data_ptr(), None, None, hws.data_ptr(),
                                                               B, H, W, st()), "hb")
        res = []
        for name, fn, moved, algo in (("fwd", hf, (32 * es + 8.0) * px, (25 * es + 8.0) * px),
                                      ("bwd", hb, (64 * es + 8.0) * px, (50 * es + 8.0) * px)):
            best = 1e9
            for rep in range(3):
                for i in range(WARM):
                    fn(i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(20):
                    fn(i)
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20 * 1e-3)
            res.append(f"{name} {best * 1e6:6.1f} us moved {moved / best / 1e12:.3f} TB/s algorithmic {algo / best / 1e12:.3f} TB/s")
        print(f"   head-fed {str(dt)[6:]:9s}: " + " | ".join(res), flush=True)


if __name__ == "__main__":
    main()
