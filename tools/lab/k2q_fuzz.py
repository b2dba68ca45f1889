"""Lab: K2q (conv128.hip) on random geometries against torch's fp32 convolution on the GPU -- ragged rasters down to less than
one tile, batch sizes, channel slices on both sides, addend, ReLU, per-channel scale / bias, statistics rows, the dynamic tile queue switched on and off.
JSPSR_CONV_RESIDENT128_MIN=1 so that every size takes the kernel.  python tools/lab/k2q_fuzz.py [cases] [seed]"""
import os
import random
import sys

os.environ.setdefault("JSPSR_CONV_RESIDENT128_MIN", "1")
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jspsr_amd import _lib  # noqa: E402
from jspsr_amd import kernels as K  # noqa: E402


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rnd = random.Random(seed)
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(seed)
    worst = 0.0
    for case in range(n):
        B = rnd.choice((1, 1, 2, 3, 5))
        H = rnd.choice((1, 3, 7, 8, 9, 16, 23, 40, 64, 100, 129, 200, 256))
        W = rnd.choice((1, 5, 15, 16, 17, 31, 48, 100, 130, 257, 300, 512))
        ip = rnd.choice((128, 128, 160, 256))
        ic = rnd.choice([c for c in (0, 8, 32, 64, 128) if c + 128 <= ip])
        op = rnd.choice((128, 128, 192, 256))
        oc = rnd.choice([c for c in (0, 8, 64, 128) if c + 128 <= op])
        add, relu, dynq = rnd.random() < 0.5, rnd.random() < 0.5, rnd.random() < 0.3
        xw = torch.randn(B, H, W, ip, device="cuda", generator=g).to(torch.bfloat16)
        x = xw[..., ic:ic + 128]
        w = (torch.randn(128, 128, 3, 3, device="cuda", generator=g) / 34.0).to(torch.bfloat16).float()
        wp, wpt = K.pack_weight(w, 0, 128, torch.bfloat16), K.pack_weight(w, 1, 128, torch.bfloat16)
        xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_()
        ref = F.conv2d(xr, w, None, 1, 1)
        n0 = lib.jspsr_launch_count(b"conv128_resident")
        lib.jspsr_conv_dynamic_queue(1 if dynq else 0)
        try:
            y, st = K.conv2d_forward(xw, wp, None, 1, 1, stats=True, cin=128, in_coff=ic)
            out = torch.full((B, H, W, op), 3.0, dtype=torch.bfloat16, device="cuda")
            K.conv2d_forward(xw, wp, None, 1, 1, relu=relu, out=out, out_coff=oc, cin=128, in_coff=ic)
            # affine form: [relu]((acc * scale + bias) rounded to bf16 + addend)
            aff = rnd.random() < 0.5
            scale = (torch.rand(128, device="cuda", generator=g) + 0.5) if aff and rnd.random() < 0.7 else None
            bias = torch.randn(128, device="cuda", generator=g) if aff and (scale is None or rnd.random() < 0.7) else None
            res = torch.randn(B, H, W, 128, device="cuda", generator=g).to(torch.bfloat16) if aff and rnd.random() < 0.5 else None
            n_aff = 0
            if aff:
                ya = K.conv2d_forward(xw, wp, bias, 1, 1, relu=relu, cin=128, in_coff=ic, scale=scale, addend=res)
                n_aff = 1
            go = torch.randn(B, H, W, 128, device="cuda", generator=g).to(torch.bfloat16)
            addend = torch.randn(B, H, W, 128, device="cuda", generator=g).to(torch.bfloat16) if add else None
            dx = K.conv2d_dgrad(go, wpt, (H, W), 1, 1, addend=addend, relu=relu)
            torch.cuda.synchronize()
        finally:
            lib.jspsr_conv_dynamic_queue(-1)
        assert lib.jspsr_launch_count(b"conv128_resident") == n0 + 3 + n_aff, "not K2q"
        ref.backward(go.float().permute(0, 3, 1, 2))
        rd = ref.detach()
        dref = xr.grad.to(torch.bfloat16).float()
        if add:
            dref = dref + addend.float().permute(0, 3, 1, 2)
        if relu:
            dref = torch.relu(dref)
        errs = {
            "fwd": rel(y.float().permute(0, 3, 1, 2), rd),
            "fwd max": ((y.float().permute(0, 3, 1, 2) - rd).abs().max() / rd.abs().max().clamp_min(1e-30)).item(),
            "sum": ((st.sum(0)[0].double() - rd.double().sum((0, 2, 3))).abs().max() / rd.double().abs().sum((0, 2, 3)).max().clamp_min(1e-30)).item(),
            "sumsq": rel(st.sum(0)[1], (rd * rd).sum((0, 2, 3))),
            "slice": rel(out[..., oc:oc + 128].float().permute(0, 3, 1, 2), torch.relu(rd) if relu else rd),
            "dgrad": rel(dx.float().permute(0, 3, 1, 2), dref),
            "dgrad max": ((dx.float().permute(0, 3, 1, 2) - dref).abs().max() / dref.abs().max().clamp_min(1e-30)).item(),
        }
        if aff:
            ra = rd
            if scale is not None:
                ra = ra * scale.view(1, -1, 1, 1)
            if bias is not None:
                ra = ra + bias.view(1, -1, 1, 1)
            if res is not None:
                ra = ra.to(torch.bfloat16).float() + res.float().permute(0, 3, 1, 2)
            if relu:
                ra = torch.relu(ra)
            errs["affine"] = rel(ya.float().permute(0, 3, 1, 2), ra)
        untouched = bool((out[..., :oc] == 3.0).all() and (out[..., oc + 128:] == 3.0).all())
        # statistics rows: each 8x16 tile's own pixels
        rows = st.double().reshape(B, (H + 7) // 8, (W + 15) // 16, 2, 128)
        r8 = F.pad(rd.double(), (0, -W % 16, 0, -H % 8)).reshape(B, 128, (H + 7) // 8, 8, (W + 15) // 16, 16).sum((3, 5)).permute(0, 2, 3, 1)
        a8 = F.pad(rd.double().abs(), (0, -W % 16, 0, -H % 8)).reshape(B, 128, (H + 7) // 8, 8, (W + 15) // 16, 16).sum((3, 5)).permute(0, 2, 3, 1)
        errs["rows"] = ((rows[..., 0, :] - r8).abs().max() / a8.max().clamp_min(1e-30)).item()
        tol = {"fwd": 6e-3, "fwd max": 2e-2, "sum": 1e-4, "sumsq": 2e-4, "slice": 6e-3, "dgrad": 6e-3, "dgrad max": 2e-2, "rows": 1e-4, "affine": 6e-3}
        bad = [k for k in errs if not errs[k] <= tol[k]] + ([] if untouched else ["outside the slice"])
        worst = max(worst, errs["fwd"], errs["dgrad"])
        print(f"{case:3d} B{B} {H}x{W} in {ip}@{ic} out {op}@{oc} addend {int(add)} relu {int(relu)} dynq {int(dynq)}: "
              + " ".join(f"{k} {v:.1e}" for k, v in errs.items()) + ("  BAD " + ",".join(bad) if bad else ""), flush=True)
        if bad:
            sys.exit(1)
    print(f"FUZZ OK: {n} cases, worst relative L2 {worst:.2e}")


if __name__ == "__main__":
    main()
