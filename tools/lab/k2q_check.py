"""Lab: K2q (conv128.hip, the register-resident 128 -> 128 kernel) against the patch kernel on the same inputs, and its timing.

  python tools/lab/k2q_check.py            parity on small / ragged / sliced shapes + timing at 8 x 512^2 and 8 x 256^2
The reference of every case is the SAME library with JSPSR_CONV_RESIDENT128=0 (the patch kernel, itself tested against
torch fp64 in tests/test_conv_gpu.py), run in a child process (the switch is read once per process)."""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jspsr_amd import kernels as K  # noqa: E402

CASES = [  # B, H, W, in pitch, in coff, addend, relu
    (1, 8, 16, 128, 0, False, False),
    (2, 64, 128, 128, 0, False, False),
    (1, 40, 50, 128, 0, True, True),        # ragged both ways
    (3, 21, 130, 192, 64, True, False),     # channel slice of a wider tensor, ragged
    (2, 128, 128, 128, 0, False, True),
]


def run_cases(out_path):
    res = {}
    g = torch.Generator(device="cuda").manual_seed(7)
    for ci, (B, H, W, pitch, coff, add, relu) in enumerate(CASES):
        xw = torch.randn(B, H, W, pitch, device="cuda", generator=g).to(torch.bfloat16)
        w = torch.randn(128, 128, 3, 3, device="cuda", generator=g) / (128 * 9) ** 0.5
        wp = K.pack_weight(w, 0, 128, torch.bfloat16)
        wpt = K.pack_weight(w, 1, 128, torch.bfloat16)
        addend = torch.randn(B, H, W, 128, device="cuda", generator=g).to(torch.bfloat16) if add else None
        y, st = K.conv2d_forward(xw, wp, None, 1, 1, stats=True, cin=128, in_coff=coff)
        y2 = K.conv2d_forward(xw, wp, None, 1, 1, relu=relu, cin=128, in_coff=coff)
        go = torch.randn(B, H, W, 128, device="cuda", generator=g).to(torch.bfloat16)
        dx = K.conv2d_dgrad(go, wpt, (H, W), 1, 1, addend=addend, relu=relu)
        dx = dx[0] if isinstance(dx, tuple) else dx
        res[f"y{ci}"], res[f"st{ci}"], res[f"y2{ci}"], res[f"dx{ci}"] = y.float().cpu(), st.cpu(), y2.float().cpu(), dx.float().cpu()
    torch.save(res, out_path)


def timing():
    for (B, H, W) in ((8, 512, 512), (8, 256, 256)):
        x = torch.randn(B, H, W, 128, device="cuda").to(torch.bfloat16)
        w = torch.randn(128, 128, 3, 3, device="cuda") / (128 * 9) ** 0.5
        wp = K.pack_weight(w, 0, 128, torch.bfloat16)
        wpt = K.pack_weight(w, 1, 128, torch.bfloat16)
        add = torch.randn(B, H, W, 128, device="cuda").to(torch.bfloat16)
        flops = 2.0 * B * H * W * 128 * 128 * 9
        # the size the dynamic tile queue runs at: forward (+ statistics) and data gradient against torch's fp32 convolution
        import torch.nn.functional as F
        xr = x.float().permute(0, 3, 1, 2).requires_grad_()
        wr = w.to(torch.bfloat16).float()
        ref = F.conv2d(xr, wr, None, 1, 1)
        y, st = K.conv2d_forward(x, wp, None, 1, 1, stats=True)
        rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
        e_y = rel(y.float().permute(0, 3, 1, 2), ref.detach())
        e_s = rel(st.sum(0)[0], ref.detach().sum((0, 2, 3)))
        e_q = rel(st.sum(0)[1], (ref.detach() ** 2).sum((0, 2, 3)))
        ref.backward(x.float().permute(0, 3, 1, 2))
        dx = K.conv2d_dgrad(x, wpt, (H, W), 1, 1, addend=add)
        e_d = rel(dx.float().permute(0, 3, 1, 2), xr.grad.to(torch.bfloat16).float() + add.float().permute(0, 3, 1, 2))
        print(f"  {B}x{H}x{W} vs torch fp32: fwd {e_y:.2e} sum {e_s:.2e} sumsq {e_q:.2e} dgrad+addend {e_d:.2e}", flush=True)
        assert e_y < 6e-3 and e_s < 1e-3 and e_q < 1e-4 and e_d < 6e-3
        del xr, ref, y, st, dx
        for name, fn in (("fwd", lambda: K.conv2d_forward(x, wp, None, 1, 1)),
                         ("fwd+stats", lambda: K.conv2d_forward(x, wp, None, 1, 1, stats=True)),
                         ("dgrad", lambda: K.conv2d_dgrad(x, wpt, (H, W), 1, 1)),
                         ("dgrad+addend", lambda: K.conv2d_dgrad(x, wpt, (H, W), 1, 1, addend=add))):
            for _ in range(60):      # run the timed block straight out of ~40 ms of the same launches (clock transient after idle)
                fn()
            n = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / n * 1e-3
            print(f"  {B}x{H}x{W} {name:13s} {t * 1e6:8.1f} us {flops / t / 1e12:7.1f} TF/s", flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        run_cases(sys.argv[2])
        if len(sys.argv) > 3:
            timing()
        return
    os.makedirs("gpurun_out", exist_ok=True)
    outs = {}
    for tag, env in (("k2q", {"JSPSR_CONV_RESIDENT128": "1", "JSPSR_CONV_RESIDENT128_MIN": "1"}), ("patch", {"JSPSR_CONV_RESIDENT128": "0"})):
        path = f"/tmp/k2q_{tag}.pt"
        print(f"== {tag}", flush=True)
        r = subprocess.run([sys.executable, __file__, "--child", path, "time"], env={**os.environ, **env}, timeout=900)
        if r.returncode:
            print(f"{tag}: child failed rc={r.returncode}")
            sys.exit(1)
        outs[tag] = torch.load(path)
    bad = 0
    for k in outs["k2q"]:
        a, b = outs["k2q"][k], outs["patch"][k]
        if a.shape != b.shape:
            print(f"{k}: shape {tuple(a.shape)} vs {tuple(b.shape)}")
            bad += 1
            continue
        d = (a - b).abs()
        scale = b.abs().max().item() + 1e-30
        # bf16 outputs of two summation orders: a last-place difference (2^-8 relative) on a few percent of the elements;
        # fp32 statistics: rounding of the order of the sums
        tol = 2e-5 if k.startswith("st") else 1.6e-2
        rel = d.max().item() / scale
        frac = (d > 0).float().mean().item()
        print(f"{k}: max|diff| / max|ref| = {rel:.3e}  (differing elements {frac:.3%})  {'ok' if rel <= tol else 'BAD'}")
        bad += rel > tol or not torch.isfinite(a).all()
    print("PARITY OK" if not bad else f"PARITY FAILED ({bad})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
