#!/usr/bin/env python3
"""Headline benchmark: Mpixel/s of one JSPSR training step on 512x512 DEM tiles (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE config 3 / 4): JSPSR image+mask guided (jspsr_r8_img_msk.yml architecture,
num_feature 32, layers 2-2-2-2, 43.87 M parameters, reference init), 8 tiles of 512x512 per GPU,
synthetic inputs.  One step = zero_grad -> model(*inputs) -> L1 + L2 +
0.1*Sobel-L1 -> backward -> gradient all-reduce (N > 1) -> AdamW step, i.e. the body of the
reference's train_one_epoch (train/train_utils.py:210-219).  Weak scaling: 8 tiles per GPU.

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (K1, the HBM-bound propagation kernels,
timed live with events on the launch stream; algorithmic bytes per SURVEY.md section 8d) and, at
N = 1, "cpu_baseline" (the oracle's CPU restatement timed on the host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

IN_CHANNELS = {"COP30": 1, "image": 3, "mask": 15, "lr_dem": 1}  # configs/jspsr_r8_img_msk.yml:33-38
TILE = 512
TILES_PER_GPU = 8
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def synthetic_batch(B, H, W, device, seed):
    """SURVEY.md section 8d inputs, generated on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    z = torch.zeros(B, 1, H, W, device=device)
    for o, n in enumerate((16, 32, 64, 128)):
        noise = torch.randn(B, 1, n, n, device=device, generator=g)
        z = z + torch.nn.functional.interpolate(noise, size=(H, W), mode="bilinear", align_corners=True) / (2**o)
    lo, hi = z.amin((2, 3), keepdim=True), z.amax((2, 3), keepdim=True)
    z = (z - lo) / (hi - lo + 1e-12) * 120.0
    lr = torch.log(z + 80.0) / float(np.log(1009.0))
    hr = torch.log((z + torch.randn(z.shape, device=device, generator=g)).clamp_min(-79.0) + 80.0) / float(np.log(1009.0))
    img = torch.randint(0, 256, (B, 3, H, W), device=device, generator=g).float() / 255.0
    cls = torch.randint(0, 15, (B, H // 32, W // 32), device=device, generator=g)
    cls = cls.repeat_interleave(32, 1).repeat_interleave(32, 2)
    msk = torch.nn.functional.one_hot(cls, 15).permute(0, 3, 1, 2).float()
    msk = msk * ((torch.arange(15, device=device).float() + 1) / 16).view(1, 15, 1, 1)
    return [lr.contiguous(), img.contiguous(), msk.contiguous()], hr.contiguous()


def time_k1(model, inputs, iters=20):
    """Roofline of K1 (the HBM-bound propagation kernels) at this batch's shape.

    Each kernel is launched `iters` times back to back through the C ABI between two events
    recorded on the launch stream (torch's current stream is the stream handed to the ABI), behind
    300 warm-up launches of the same kernel, so the figure is the steady-state average launch
    duration incl. the ~1-2 us inter-launch gap.  Operand sets rotate
    over > 256 MiB so the Infinity Cache cannot serve them.
    """
    from jspsr_amd import ops
    dem = inputs[0]
    B, _, H, W = dem.shape
    dev = dem.device
    torch.cuda.empty_cache()    # operand sets in fresh allocations, as in tools/k1_lab.py (not carved out of the step's cached blocks)
    g = torch.Generator(device=dev).manual_seed(1)
    nset = max(2, int(700e6 // (B * H * W * 4 * 26)) + 1)
    sets = [(torch.sigmoid(torch.randn(B, 9, H, W, device=dev, generator=g)),
             1.5 * torch.randn(B, 16, H, W, device=dev, generator=g)) for _ in range(nset)]
    gsets = [(torch.empty(B, 9, H, W, device=dev), torch.empty(B, 16, H, W, device=dev)) for _ in range(nset)]
    w = model.postprocessor.w.detach().clone()
    b = model.postprocessor.b.detach().clone()
    gout = torch.randn(B, 1, H, W, device=dev, generator=g)
    out = torch.empty_like(dem)
    gw, gb = torch.empty_like(w), torch.empty_like(b)
    ws = ops.prop_backward_workspace(B, H, W, dev)

    def fwd(i):
        ops.prop_forward_raw(dem, sets[i % nset][0], sets[i % nset][1], w, b, 1.0, out)

    def bwd(i):   # the streaming kernel alone: what rocprof lists as prop_bwd_kernel
        ops.prop_backward_raw(gout, dem, sets[i % nset][0], sets[i % nset][1], w, gsets[i % nset][0],
                              gsets[i % nset][1], None, None, ws)

    def bwd_call(i):   # the whole C-ABI call: streaming kernel + 10-workgroup fold of the parameter gradients
        ops.prop_backward_raw(gout, dem, sets[i % nset][0], sets[i % nset][1], w, gsets[i % nset][0],
                              gsets[i % nset][1], gw, gb, ws)

    def timed(fn, reps=3, warm=300):
        """Mean launch duration (s) of `iters` back-to-back launches between two events on the launch stream, in STEADY
        STATE: `warm` launches of the same kernel (~25 ms) run straight into the timed ones, with no synchronisation in
        between.  Timing 20 launches behind three warm-ups and a synchronize() -- what this leg did until round 3 -- lands
        in the chip's clock transient after an idle bubble: the same backward kernel reads 82-91 us there and 77.5 us
        from 300 warm-up launches on (profiles/r03_k1_warmup_transient.txt).  The MEDIAN of `reps` measurements, all kept."""
        out = []
        for _ in range(reps):
            for i in range(warm):
                fn(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                fn(i)
            e1.record()
            e1.synchronize()
            out.append(e0.elapsed_time(e1) / iters * 1e-3)
        return sorted(out)[len(out) // 2], [round(v * 1e6, 2) for v in out]

    res, reps = {}, {}
    for name, fn in (("fwd", fwd), ("bwd", bwd), ("bwd_call", bwd_call)):
        res[name], reps[name] = timed(fn)
    # The step AS THE MODELS LAUNCH IT (round 4): the generator's heads write one (B,25,H,W) fp32 tensor -- 9 affinity
    # LOGITS + 16 offsets -- and jspsr_prop_logits_* runs the same persistent LDS-DMA kernel with the sigmoid folded in
    # (SIG = true): same operands, same 108 / 208 algorithmic bytes per pixel as the public boundary, nothing padded.
    from jspsr_amd import _lib, kernels as K
    lib = _lib.load()
    st = lambda: torch.cuda.current_stream().cuda_stream
    lsets = [torch.cat((1.5 * torch.randn(B, 9, H, W, device=dev, generator=g), 1.5 * torch.randn(B, 16, H, W, device=dev, generator=g)), 1)
             for _ in range(nset)]
    glsets = [torch.empty_like(t) for t in lsets]

    def lfwd(i):
        _lib.check(lib.jspsr_prop_logits_forward_f32(dem.data_ptr(), lsets[i % nset].data_ptr(), w.data_ptr(), b.data_ptr(), 1.0,
                                                     out.data_ptr(), B, H, W, st()), "jspsr_prop_logits_forward_f32")

    def lbwd(i):
        _lib.check(lib.jspsr_prop_logits_backward_f32(gout.data_ptr(), dem.data_ptr(), lsets[i % nset].data_ptr(), w.data_ptr(),
                                                      glsets[i % nset].data_ptr(), None, None, ws.data_ptr(), B, H, W, st()),
                   "jspsr_prop_logits_backward_f32")

    census = lambda n: lib.jspsr_launch_count(n.encode())
    c0 = (census("prop_logits_forward (dma)"), census("prop_logits_backward (dma)"))
    for name, fn in (("lfwd", lfwd), ("lbwd", lbwd)):
        res[name], reps[name] = timed(fn)
    # the names reported below are the kernels this process launched: the launch census must have moved
    logits_dma = (census("prop_logits_forward (dma)") - c0[0] == 3 * (300 + iters) and
                  census("prop_logits_backward (dma)") - c0[1] == 3 * (300 + iters))
    # The same two launches with the Infinity Cache (256 MB) out of the picture: SIXTEEN operand sets (3.4 GB of operands,
    # 3.4 GB of gradients) instead of four.  Four rotating sets -- what rounds 1-3 timed, kept above for comparability --
    # still find part of their lines in the cache (its replacement is not LRU: same kernel, same box, 84-85 us with four
    # sets, 88 us with eight, placement-dependent); inside the running step the operands are cold (the head planes were
    # written ~30 ms earlier), so `cold` is the figure that describes the kernel in the model.
    ncold = 16
    lsets += [torch.cat((1.5 * torch.randn(B, 9, H, W, device=dev, generator=g), 1.5 * torch.randn(B, 16, H, W, device=dev, generator=g)), 1)
              for _ in range(ncold - nset)]
    glsets += [torch.empty_like(lsets[0]) for _ in range(ncold - nset)]

    def lfwd_cold(i):
        _lib.check(lib.jspsr_prop_logits_forward_f32(dem.data_ptr(), lsets[i % ncold].data_ptr(), w.data_ptr(), b.data_ptr(), 1.0,
                                                     out.data_ptr(), B, H, W, st()), "jspsr_prop_logits_forward_f32")

    def lbwd_cold(i):
        _lib.check(lib.jspsr_prop_logits_backward_f32(gout.data_ptr(), dem.data_ptr(), lsets[i % ncold].data_ptr(), w.data_ptr(),
                                                      glsets[i % ncold].data_ptr(), None, None, ws.data_ptr(), B, H, W, st()),
                   "jspsr_prop_logits_backward_f32")

    for name, fn in (("lfwd_cold", lfwd_cold), ("lbwd_cold", lbwd_cold)):
        res[name], reps[name] = timed(fn)
    del lsets, glsets
    # K1c: the heads themselves (csrc/head.hip) on the generator's 128-channel feature in the model's storage dtype
    hdt = model.compute_dtype
    es = 2 if hdt == torch.bfloat16 else 4
    cin = model.generator.conv_weight[0].in_channels
    nx = max(2, int(700e6 // (B * H * W * cin * es)) + 1)
    xs = [torch.randn(B, H, W, cin, device=dev, generator=g).to(hdt) for _ in range(nx)]
    dxs = [torch.empty_like(t) for t in xs[:2]]
    w25 = (0.1 * torch.randn(25, cin, device=dev, generator=g)).contiguous()
    b25 = torch.zeros(25, device=dev)
    planes = [torch.randn(B, 25, H, W, device=dev, generator=g) for _ in range(2)]
    gn = torch.empty(B, H, W, 32, device=dev, dtype=hdt)
    db = torch.empty(25, device=dev)
    cws = torch.empty(lib.jspsr_head_backward_workspace_bytes(B, H, W), dtype=torch.uint8, device=dev)

    def cfwd(i):
        _lib.check(lib.jspsr_head_forward(K._dt(xs[0]), xs[i % nx].data_ptr(), cin, 0, cin, w25.data_ptr(), b25.data_ptr(),
                                          planes[i % 2].data_ptr(), B, H, W, st()), "jspsr_head_forward")

    def cbwd(i):
        _lib.check(lib.jspsr_head_backward(K._dt(xs[0]), planes[i % 2].data_ptr(), w25.data_ptr(), cin, dxs[i % 2].data_ptr(), cin, 0,
                                           gn.data_ptr(), db.data_ptr(), cws.data_ptr(), B, H, W, st()), "jspsr_head_backward")

    for name, fn in (("cfwd", cfwd), ("cbwd", cbwd)):
        res[name], reps[name] = timed(fn, 1, 50)
    del xs, dxs, planes, gn
    # K1h: the entry the model itself uses (operands straight from the merged head's 32-channel NHWC output, in the
    # model's storage dtype): same timing rules.  Two byte counts per launch: `moved` = what the layout makes the kernel
    # touch (32 channels, 7 of them padding / the centre logit), `algorithmic` = the 25 operand elements + dem + out
    # (+ their gradients) of SURVEY 8d in that dtype.
    nh = max(2, int(700e6 // (B * H * W * 32 * es)) + 1)
    heads = [(1.5 * torch.randn(B, H, W, 32, device=dev, generator=g)).to(hdt) for _ in range(nh)]
    gheads = [torch.empty_like(h) for h in heads]
    hws = torch.empty(max(lib.jspsr_prop_head_backward_workspace_bytes(B, H, W), 16), dtype=torch.uint8, device=dev)

    def hfwd(i):
        _lib.check(lib.jspsr_prop_head_forward(K._dt(heads[0]), dem.data_ptr(), heads[i % nh].data_ptr(), w.data_ptr(),
                                               b.data_ptr(), 1.0, out.data_ptr(), B, H, W, st()), "jspsr_prop_head_forward")

    def hbwd(i):
        _lib.check(lib.jspsr_prop_head_backward(K._dt(heads[0]), gout.data_ptr(), dem.data_ptr(), heads[i % nh].data_ptr(),
                                                w.data_ptr(), gheads[i % nh].data_ptr(), None, None, hws.data_ptr(),
                                                B, H, W, st()), "jspsr_prop_head_backward")

    for name, fn in (("hfwd", hfwd), ("hbwd", hbwd)):
        res[name], reps[name] = timed(fn, 1, 100)
    del heads, gheads
    # K1s: the general step of N-iteration chains (NLSPN, models/components/nlspn.py:177-233): raw affinities, gradients
    # ADDED into the shared affinity / offset gradients, and the gradient with respect to the raster (LDS scatter + float
    # atomics).  One step each way; algorithmic bytes = K1's + 4 B/px of grad_dem (+ the read of the accumulators).
    from jspsr_amd import ops as O
    ones9, zero1 = torch.ones(9, device=dev), torch.zeros(1, device=dev)
    sws = O._step_workspace(B, H, W, dev)
    gdem = torch.zeros_like(dem)

    def sfwd(i):
        O._step_forward(dem, sets[i % nset][0], sets[i % nset][1], ones9, zero1, 0.0, 0, out)

    def sbwd(i):
        O._step_backward(gout, dem, sets[i % nset][0], sets[i % nset][1], ones9, 0.0, 0, 1, gsets[i % nset][0], gsets[i % nset][1], gdem, sws)

    for name, fn in (("sfwd", sfwd), ("sbwd", sbwd)):
        res[name], reps[name] = timed(fn, 1)
    pmc_path = os.path.join(ROOT, "profiles", "k1_pmc.json")
    traffic_names = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
    px = B * H * W
    steps_entry = {
        "kernel": "prop_step_fwd_kernel / prop_step_bwd_kernel (K1s, csrc/prop_steps.hip)",
        "fwd_us": round(res["sfwd"] * 1e6, 2), "bwd_us": round(res["sbwd"] * 1e6, 2),
        "fwd_GBs": round(108.0 * px / res["sfwd"] / 1e9, 1), "bwd_GBs": round((208.0 + 100.0 + 8.0) * px / res["sbwd"] / 1e9, 1),
        "note": "one propagation step of an N-iteration chain (NLSPN): forward 108 B/px; backward with accumulate = 1 reads "
                "the 25 gradient planes it adds into (100 B/px) and adds grad_dem by float atomics (4 B/px each way) on top of K1's 208",
    }
    split_env = os.environ.get("JSPSR_PROP_SPLIT")
    form = lambda bwd: ("false" if split_env == "0" else "true") if split_env is not None else ("false" if bwd else "true")   # defaults: forward split, backward symmetric
    ntl = "false" if os.environ.get("JSPSR_PROP_NTL") == "0" else "true"
    # prop_dma_kernel<OC, NW, BWD, NTL, SPLIT, RP, SIG> (csrc/prop_dma.hip); the logits entry takes 4-wave one-row tiles only
    kname = lambda bwd, sig: "prop_dma_kernel<16, 4, %s, %s, %s, 1, %s>" % ("true" if bwd else "false", ntl, form(bwd), "true" if sig else "false")
    traffic = traffic_names or None
    tr = lambda key: traffic.get(key) if traffic else None
    # 16-channel offset layout (the all-zero centre pair is not stored): 108 B/px fwd, 208 B/px bwd -- SURVEY 8d
    fb, bb = 108.0 * px, 208.0 * px
    gbs = lambda nbytes, t: round(nbytes / t / 1e9, 1)
    frac = lambda nbytes, t: round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4)
    legacy = {
        "kernel": "prop_head_dma_kernel<4, BWD, SPLIT>" if es == 2 else "prop_head_kernel<float, BWD>",
        "dtype": "bf16" if es == 2 else "f32",
        "fwd_us": round(res["hfwd"] * 1e6, 2), "bwd_us": round(res["hbwd"] * 1e6, 2),
        "fwd_moved_GBs": gbs((32 * es + 8.0) * px, res["hfwd"]), "bwd_moved_GBs": gbs((64 * es + 8.0) * px, res["hbwd"]),
        "fwd_algorithmic_GBs": gbs((25 * es + 8.0) * px, res["hfwd"]), "bwd_algorithmic_GBs": gbs((50 * es + 8.0) * px, res["hbwd"]),
        "note": "rounds 2-3's in-model entry (operands from a 32-channel NHWC head, 7 channels of it padding); JSPSR_HEAD_PLANES=0 "
                "selects it, and shapes jspsr_head_ok() refuses still take it",
    }
    xb = cin * es
    head_conv = {
        "kernel": "head_fwd_kernel / head_bwd_kernel<%s, %d> (K1c, csrc/head.hip)" % ("__bf16" if es == 2 else "float", cin),
        "fwd_us": round(res["cfwd"] * 1e6, 2), "bwd_us": round(res["cbwd"] * 1e6, 2),
        "fwd_GBs": gbs((xb + 100.0) * px, res["cfwd"]), "bwd_GBs": gbs((100.0 + xb + 32 * es) * px, res["cbwd"]),
        "note": "the two 1x1 heads as one conv writing the 25 fp32 planes (forward: reads the feature, writes 100 B/px; backward: "
                "reads the 25 gradient planes, writes the feature's gradient and the 32-channel NHWC copy the weight-gradient kernel reads)",
    }
    public = {
        "kernel": kname(True, False), "achieved": gbs(bb, res["bwd"]), "frac": frac(bb, res["bwd"]),
        "traffic": tr("bwd_bytes_per_launch"), "bytes_per_launch": bb, "us_per_launch": round(res["bwd"] * 1e6, 2),
        "us_per_launch_reps": reps["bwd"], "us_per_call_with_fold": round(res["bwd_call"] * 1e6, 2),
        "forward": {"kernel": kname(False, False), "achieved": gbs(fb, res["fwd"]), "frac": frac(fb, res["fwd"]),
                    "traffic": tr("fwd_bytes_per_launch"), "bytes_per_launch": fb, "us_per_launch": round(res["fwd"] * 1e6, 2),
                    "us_per_launch_reps": reps["fwd"]},
        "note": "PostProcessor.forward's own boundary (jspsr_prop_forward_f32 / _backward_f32: affinities after the sigmoid, two tensors)",
    }
    return {
        "bound": "hbm", "kernel": kname(True, True) if logits_dma else "prop_bwd_kernel<16, 1, true, 8, 64, true>",
        "achieved": gbs(bb, res["lbwd"]), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac(bb, res["lbwd"]),
        "traffic": tr("logits_bwd_bytes_per_launch"),
        "bytes_per_launch": bb, "us_per_launch": round(res["lbwd"] * 1e6, 2), "us_per_launch_reps": reps["lbwd"],
        "forward": {"kernel": kname(False, True) if logits_dma else "prop_fwd_kernel<16, 1, true, 8, 64, true>",
                    "achieved": gbs(fb, res["lfwd"]), "frac": frac(fb, res["lfwd"]),
                    "traffic": tr("logits_fwd_bytes_per_launch"), "bytes_per_launch": fb,
                    "us_per_launch": round(res["lfwd"] * 1e6, 2), "us_per_launch_reps": reps["lfwd"]},
        "kernel_in_model": True, "launch_census_checked": bool(logits_dma),
        "cold": {"operand_sets": 16, "us_per_launch": round(res["lbwd_cold"] * 1e6, 2), "us_per_launch_reps": reps["lbwd_cold"],
                 "achieved": gbs(bb, res["lbwd_cold"]), "frac": frac(bb, res["lbwd_cold"]),
                 "forward": {"us_per_launch": round(res["lfwd_cold"] * 1e6, 2), "us_per_launch_reps": reps["lfwd_cold"],
                             "achieved": gbs(fb, res["lfwd_cold"]), "frac": frac(fb, res["lfwd_cold"])},
                 "note": "the same launches rotating over 16 operand sets (3.4 GB + 3.4 GB of gradients): no help from the 256 MB "
                         "Infinity Cache -- the state the kernel is in inside the running step, where its operands are cold"},
        "public_boundary": public, "head_conv": head_conv, "legacy_nhwc_head": legacy, "steps_entry": steps_entry,
        "note": "top level = the propagation kernel THE BENCHMARKED STEP LAUNCHES for spn.py:43,69-73,99-118 (jspsr_prop_logits_*: "
                "planar fp32 logits + offsets written by the generator's heads, sigmoid inside; backward in the top-level keys, "
                "forward beside it); public_boundary = the same kernel template at PostProcessor.forward's own signature. "
                "achieved = algorithmic bytes (SURVEY 8d with 16-ch offsets: 108 / 208 B per pixel) x pixels per launch / mean "
                "launch duration (events on the launch stream, back-to-back launches of that kernel alone behind 300 warm-up "
                "launches; median of three 20-launch blocks); traffic = PMC FETCH_SIZE + WRITE_SIZE per launch from "
                "profiles/k1_pmc.json (separate rocprofv3 --pmc passes, FETCH_SIZE doubled per the guide)",
    }


def time_convs(batch, dtype, iters=10):
    """MFMA utilisation of the conv kernels (north_star asks for it beside K1's HBM fraction): the 3x3 / stride-1
    layer shapes of the benchmarked model, each pass (forward, data gradient, weight gradient) launched `iters`
    times back to back through the C ABI between two events on the launch stream; algorithmic flops = 2*M*N*K."""
    from jspsr_amd import kernels as K
    peak = 2500.0 if dtype == torch.bfloat16 else 157.0       # dense TFLOP/s, MI355X_MICROARCH.md
    shapes = [(TILE, 64, 64), (TILE, 128, 128), (TILE // 2, 128, 128), (TILE // 4, 256, 256), (TILE // 8, 512, 512)]
    rows, tot_f, tot_t = [], 0.0, 0.0
    for hw, ci, co in shapes:
        x = torch.randn(batch, hw, hw, ci, device="cuda").to(dtype)
        go = torch.randn(batch, hw, hw, co, device="cuda").to(dtype)
        w = torch.randn(co, ci, 3, 3, device="cuda") / (ci * 9) ** 0.5
        wp, wpt = K.pack_weight(w, 0, ci, dtype), K.pack_weight(w, 1, co, dtype)
        flops = 2.0 * batch * hw * hw * co * ci * 9
        # forward as the training step runs it: with the BatchNorm partial statistics taken in the epilogue
        fns = {"fwd": lambda: K.conv2d_forward(x, wp, None, 1, 1, stats=True), "dgrad": lambda: K.conv2d_dgrad(go, wpt, (hw, hw), 1, 1),
               "wgrad": lambda: K.conv2d_wgrad(go, x, co, ci, 3, 3, 1, 1)}
        for name, fn in fns.items():
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1) / iters * 1e-3
            rows.append({"layer": f"{batch}x{hw}x{hw} {ci}->{co} 3x3 {name}", "tflops": round(flops / t / 1e12, 1)})
            tot_f += flops
            tot_t += t
    ach = tot_f / tot_t / 1e12
    return {"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "note": "flop-weighted over the rows; fwd rows include the BatchNorm statistics epilogue, wgrad rows the ordered slab reduction launch", "rows": rows}


def cpu_baseline():
    """Oracle (CPU restatement, fp32, torch CPU kernels) on one 512x512 tile: fwd + loss + bwd."""
    from oracle import jspsr_ref as R
    ic = {k: v for k, v in IN_CHANNELS.items() if k != "COP30"}
    sd = R.make_state_dict(R.jspsr_param_shapes(ic, 32), seed=0)
    for v in sd.values():
        if v.is_floating_point() and v.dim() > 0:
            v.requires_grad_()
    for k in list(sd):
        if "running" in k:
            sd[k] = sd[k].detach()
    from oracle import host_cpus
    cores = host_cpus()                 # the container's CPU quota (16 on the GPU boxes), not the 256 CPUs it can see
    torch.set_num_threads(cores)        # torch's default there is 128 threads on 16 CPUs' worth of quota: 7.7 x slower

    def step(H):
        inputs, gt = R.synthetic_batch(1, H, H, True, seed=0)
        for v in sd.values():
            if v.requires_grad:
                v.grad = None
        t0 = time.perf_counter()
        pred = R.jspsr_forward(sd, inputs, True)
        R.multi_loss(pred, gt)["Total"].backward()
        return time.perf_counter() - t0

    step(64)  # warm-up (allocator, thread pool)
    t = step(TILE)
    return {"value": round(TILE * TILE / t / 1e6, 5), "unit": "Mpixel/s", "cores": cores, "kind": "port",
            "sample": f"oracle/jspsr_ref.py (torch CPU fp32), 1 step fwd+loss+bwd on 1x{TILE}x{TILE} image+mask, {t:.1f} s"}


def launch_ranks(n):
    """One child `python bench.py ...` per GPU with the torch.distributed.run environment (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT); rank 0's JSON line goes to this process's stdout."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p_ in procs:
        rc = max(rc, abs(p_.wait()))
    return rc


def time_steps(step, n, warmup):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def fp32_legs(args, device, model, step):
    """The reference's own precision, timed in the same run (N = 1): (a) the SAME workload with fp32 storage and the
    exact-fp32 MFMA path (what meets the 1e-4 parity bound), (b) BASELINE configs[1]: image-guided JSPSR, 8 x 256x256,
    fp32.  Full steps (forward + loss + backward + AdamW), same timing rules as the headline."""
    from jspsr_amd.JSPSR import Model
    from jspsr_amd.ddp import GradReducer
    from jspsr_amd.losses import MultiLoss
    from jspsr_amd.optim import FlatAdamW
    n = max(2, min(args.steps, 5))
    model.compute_dtype = torch.float32
    t_same = time_steps(step, n, 2)
    model.compute_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    ic2 = {"COP30": 1, "image": 3, "lr_dem": 1}                       # configs/jspsr_r3_img.yml:33-37
    np.random.seed(0)
    m2 = Model(in_channels=ic2, out_channels=1, num_feature=32, layers=(2, 2, 2, 2), spn=True).to(device).train()
    red2 = GradReducer(m2.parameters())
    red2.watch_streams(m2.side_streams(device))
    opt2 = FlatAdamW(red2, lr=1e-3, weight_decay=1e-6)
    crit2 = MultiLoss(1.0, 1.0, 0.1)
    inp2, gt2 = synthetic_batch(8, 256, 256, device, seed=2000)
    inp2 = inp2[:2]

    def step2():
        red2.zero_grad()
        crit2(m2(*inp2), gt2)["Total"].backward()
        red2.finish()
        opt2.step()

    t2 = time_steps(step2, max(n, 10), 3)
    px = args.batch * TILE * TILE
    return {"dtype": "f32", "steps": n, "ms_per_step": round(t_same * 1e3, 3), "value": round(px / t_same / 1e6, 4),
            "unit": "Mpixel/s", "workload": "same as the headline line, fp32 storage + v_mfma_f32_32x32x2_f32",
            "config2": {"workload": "jspsr_r3_img (image guided, 29.16M params), 8 x 256x256 tiles, fp32, train step",
                        "ms_per_step": round(t2 * 1e3, 3), "value": round(8 * 256 * 256 / t2 / 1e6, 4), "unit": "Mpixel/s"}}


def inference_leg(model, device):
    """BASELINE configs[4] as far as one GPU goes: the eval-mode forward of the SAME module on ONE rank's strip of a
    4096 x 4096 scene split over 8 ranks (jspsr_amd/tiling.py: 512 interior rows + 128-row halos = what a rank computes),
    BatchNorm folded into the conv epilogues (ops.conv_bn_infer).  Mpixel/s of interior pixels per GPU; the halo exchange
    and the four gate all-reduces of the 8-rank run are not in it (tests/test_tiling_gpu.py runs those, over gloo)."""
    from jspsr_amd import tiling
    S, world, rank = 4096, 8, 3
    s = tiling.plan_strips(S, world, 128)[rank]
    rows = s.ty1 - s.ty0
    g = torch.Generator(device=device).manual_seed(7)
    tiles = [torch.rand(1, c, rows, S, device=device, generator=g) for c in (1, 3, 15)]
    was_training = model.training
    model.eval()
    try:
        run = lambda: tiling._run(model, tiles, [s], tiling._combine_batch)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        t0, n = time.perf_counter(), 5
        for _ in range(n):
            _, reach = run()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / n
        # correctness of what was just timed: a 1024-row scene of the same width cut into the same 8 strips
        # (single-process emulation: strips stacked along the batch axis, gate statistics combined across them) against the
        # monolithic forward of the same module in the same storage type; fp32: < 2e-5 absolute, bf16: relative L2 < 1e-3
        scene = [torch.rand(1, c, 1024, S, device=device, generator=g) for c in (1, 3, 15)]
        with torch.no_grad():
            mono = model(*scene)
        strips_out = tiling.emulate_sharded_forward(model, scene, world, halo=128)
        c_err = (strips_out - mono).abs().max().item()
        c_rel = ((strips_out - mono).norm() / mono.norm()).item()
        # fp32: strips and monolithic agree to 2e-5.  bf16: the two runs do not round alike (kernels are chosen by raster size --
        # a 768-row window and a 1024-row scene do not take the same conv kernels everywhere -- and the gate statistics' fp32
        # sums come in another order; a bf16 rounding that lands on the other side then travels), by an amount that depends on
        # the weights the bench holds at this point (3e-3 after 25 training steps, 9e-3 after 5).  The yardstick is therefore
        # measured, not fixed: the strips may deviate from the fp32 monolithic forward of the same scene no more than 1.5 x what
        # the bf16 monolithic forward itself deviates from it (+ 1e-3).
        e_strips = e_mono = None
        if model.compute_dtype == torch.float32:
            ok = c_err < 2e-5
        else:
            model.compute_dtype = torch.float32
            try:
                with torch.no_grad():
                    mono32 = model(*scene)
            finally:
                model.compute_dtype = torch.bfloat16
            e_strips = ((strips_out - mono32).norm() / mono32.norm()).item()
            e_mono = ((mono - mono32).norm() / mono32.norm()).item()
            ok = e_strips <= 1.5 * e_mono + 1e-3
            del mono32
        del scene, mono, strips_out
    finally:
        model.train(was_training)
    interior = (s.y1 - s.y0) * S
    return {"workload": f"jspsr_r8_img_msk eval forward, one rank's window of a {S}x{S} scene over {world} ranks: rows "
                        f"[{s.ty0},{s.ty1}) = {rows} x {S} px computed, {s.y1 - s.y0} x {S} interior",
            "dtype": "bf16" if model.compute_dtype == torch.bfloat16 else "f32", "ms": round(t * 1e3, 2),
            "value": round(interior / t / 1e6, 2), "computed_value": round(rows * S / t / 1e6, 2), "unit": "Mpixel/s forward per GPU",
            "max_offset_px": round(max(reach), 2), "receptive_radius": tiling.RECEPTIVE_RADIUS,
            "check": {"scene": f"1024x{S}, {world} strips vs monolithic, same storage type", "ok": bool(ok), "max_abs_diff": float(f"{c_err:.3e}"),
                      "rel_l2": float(f"{c_rel:.3e}"),
                      "rel_l2_vs_fp32_monolithic": None if e_strips is None else {"strips": float(f"{e_strips:.3e}"), "monolithic": float(f"{e_mono:.3e}")},
                      "bound": "fp32: 2e-5 abs; bf16: the strips' relative L2 distance from the fp32 monolithic forward <= 1.5 x the bf16 "
                               "monolithic forward's own + 1e-3 (reported, never raised)"}}


def graph_leg_child(args):
    """Body of the hipGraph leg (runs in a CHILD process of the bench: `python bench.py --graph-child`): eager vs replayed
    training step at 2 tiles and at the headline's tile count; prints one JSON object."""
    from jspsr_amd import _lib
    from jspsr_amd.JSPSR import Model
    from jspsr_amd.ddp import GradReducer
    from jspsr_amd.graph import GraphedStep
    from jspsr_amd.losses import MultiLoss
    from jspsr_amd.optim import FlatAdamW
    _lib.load()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    np.random.seed(0)
    torch.manual_seed(0)
    model = Model(in_channels=IN_CHANNELS, out_channels=1, num_feature=32, layers=(2, 2, 2, 2), spn=True).to(device).train()
    model.compute_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    reducer = GradReducer(model.parameters())
    reducer.watch_streams(model.side_streams(device))
    opt = FlatAdamW(reducer, lr=1e-3, weight_decay=1e-6)
    criterion = MultiLoss(1.0, 1.0, 0.1)
    out = {}
    for tiles in sorted({2, args.batch}):
        inputs, gt = synthetic_batch(tiles, TILE, TILE, device, seed=3000 + tiles)

        def eager():
            reducer.zero_grad()
            criterion(model(*inputs), gt)["Total"].backward()
            reducer.finish()
            opt.step()

        def timed(fn, n, warm=2):
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            host = (time.perf_counter() - t0) / n
            torch.cuda.synchronize()
            return round((time.perf_counter() - t0) / n * 1e3, 3), round(host * 1e3, 3)

        # at the headline's tile count the replay is timed by the headline's own rules (W untimed steps, then EXACTLY K):
        # the parent may take it as the headline where it is the faster way to run the step on this box
        n = args.steps if tiles == args.batch else max(5, min(args.steps, 10))
        warm = args.warmup if tiles == args.batch else 2
        e_ms, e_host = timed(eager, max(5, min(args.steps, 10)))
        t0 = time.perf_counter()
        step = GraphedStep(model, reducer, opt, criterion, inputs, gt)
        t_cap = time.perf_counter() - t0
        g_ms, g_host = timed(step, n, warm)
        out[f"tiles_{tiles}"] = {"eager_ms_per_step": e_ms, "eager_host_ms_per_step": e_host, "graph_ms_per_step": g_ms,
                                 "graph_host_ms_per_step": g_host, "graph_Mpixel_per_s": round(tiles * TILE * TILE / g_ms / 1e3, 3),
                                 "graph_steps_timed": n, "graph_warmup": warm, "capture_s": round(t_cap, 2),
                                 "final_loss": round(float(step.loss.item()), 6)}
        del step, inputs, gt
        criterion.reset()
        torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)


def graph_leg(args):
    """the same training step captured in a hipGraph and replayed (jspsr_amd/graph.py::GraphedStep;
    bit-identical to the eager step: tests/test_train_step_gpu.py), at 2 tiles per GPU -- where the eager step is bound by
    the host's ~40 ms of Python / autograd / ctypes per step -- and at the headline's tile count.  Per arm: ms per step and
    the host time per step (no synchronisation inside the loop).  The headline `value` stays the EAGER step, the reference's
    loop body as its train loop would run it.  Runs in a child process: a capture the runtime refuses (a stale autograd
    graph on the default stream ends in a segmentation fault inside hipStreamEndCapture) must not take the line with it."""
    import subprocess
    out = {"note": "eager vs hipGraph replay of the full train step (fwd + loss + bwd + AdamW) in a child process, same architecture "
                   "and storage type; N = 1 only (a captured step holds no collective); the headline value is the eager step"}
    cmd = [sys.executable, os.path.abspath(__file__), "--graph-child", "--batch", str(args.batch), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--dtype", args.dtype]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            out["error"] = f"child exited with {r.returncode}: {(r.stderr or '').strip()[-300:]}"
        else:
            out.update(json.loads(lines[-1]))
    except subprocess.TimeoutExpired:
        out["error"] = "child timed out"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=TILES_PER_GPU, help="tiles per GPU")
    ap.add_argument("--dtype", choices=("bf16", "f32"), default="bf16",
                    help="activation storage / MFMA operand type (BASELINE config 3 names bf16); "
                         "accumulation, BN statistics, K1 and master weights are fp32 either way")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 legs (N = 1 only)")
    ap.add_argument("--no-inference", action="store_true", help="skip the whole-scene strip forward (N = 1 only)")
    ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph-replay leg (N = 1 only)")
    ap.add_argument("--graph-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.graph_child:
        return graph_leg_child(args)

    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start one child process per GPU (nothing in THIS process has touched the GPU
        # yet -- no exec of a GPU-initialised process), wait for them, leave with the worst exit code
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # rehearsal on a one-GPU box (the real launch is one rank per GPU over RCCL): JSPSR_BENCH_REHEARSAL=1 puts every
    # rank on cuda:0 and moves the collectives through gloo, so the N > 1 code path can be exercised end to end
    rehearsal = os.environ.get("JSPSR_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)  # RCCL over xGMI

    from jspsr_amd import _lib
    from jspsr_amd.JSPSR import Model
    from jspsr_amd.ddp import GradReducer, broadcast_module
    from jspsr_amd.losses import MultiLoss
    from jspsr_amd.optim import FlatAdamW

    _lib.load()  # fail loudly if the HIP library is missing
    np.random.seed(0)
    torch.manual_seed(0)
    model = Model(in_channels=IN_CHANNELS, out_channels=1, num_feature=32, layers=(2, 2, 2, 2), spn=True).to(device).train()
    model.compute_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    broadcast_module(model)
    reducer = GradReducer(model.parameters())
    reducer.watch_streams(model.side_streams(device))
    opt = FlatAdamW(reducer, lr=1e-3, weight_decay=1e-6)  # configs/*.yml:71-76, one fused HIP launch
    criterion = MultiLoss(1.0, 1.0, 0.1)
    inputs, gt = synthetic_batch(args.batch, TILE, TILE, device, seed=1000 + rank)

    def step():
        reducer.zero_grad()
        pred = model(*inputs)
        loss = criterion(pred, gt)["Total"]
        loss.backward()
        reducer.finish()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    if os.environ.get("JSPSR_BENCH_GC", "freeze") == "freeze":
        # The interpreter's cyclic collector: a full (generation-2) pass over everything the process has built by now -- the
        # modules, ~1 400 autograd nodes per step, the ctypes bindings -- takes tens of ms, and where it falls is an accident
        # of allocation counts (measured on one box: 62.4 ms/step over the 20 steps behind 5 warm-up steps, 59.5 behind 25,
        # 60.0 in every block of an otherwise identical loop at module level: tools/lab/step_blocks.py).  Collect now, at a
        # step boundary, and move what survives to the permanent generation, as long-running training loops do; the
        # collector stays ON for what the timed steps allocate.
        import gc
        gc.collect()
        gc.freeze()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    final_loss = loss.item()
    # host side of one step: C-ABI calls issued and the time the host needs to enqueue them (no synchronisation inside)
    torch.cuda.synchronize()
    c0, h0 = _lib.calls, time.perf_counter()
    step()
    host_ms, abi_calls = (time.perf_counter() - h0) * 1e3, _lib.calls - c0
    torch.cuda.synchronize()

    # the roofline leg runs right behind the training steps -- the state the kernels run in inside the model.  Behind
    # the fp32 legs (seconds of fp32 MFMA work at the power limit) the same HBM-bound launches read 12-15 % slower on this
    # pool's boxes (89.5 vs 78.1 us for the backward on one box, profiles/r04_*), which is the chip's power state, not the kernel.
    roof = None
    if rank == 0 and not args.no_roofline:
        try:
            roof = time_k1(model, inputs)
            roof["convs"] = time_convs(args.batch, model.compute_dtype)
        except Exception as e:       # (every side leg below likewise: none of them may take the headline line with it)
            roof = {"error": f"{type(e).__name__}: {e}"[:300]}
    fp32 = None
    if world == 1 and args.dtype == "bf16" and not args.no_fp32:
        try:
            fp32 = fp32_legs(args, device, model, step)
        except Exception as e:
            fp32 = {"error": f"{type(e).__name__}: {e}"[:300]}
            model.compute_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    infer = None
    if rank == 0 and world == 1 and not args.no_inference:
        try:
            infer = inference_leg(model, device)
        except Exception as e:       # a side leg must not take the headline line with it
            infer = {"error": f"{type(e).__name__}: {e}"[:300]}
    graph = None
    if rank == 0 and world == 1 and not args.no_graph:
        torch.cuda.empty_cache()
        graph = graph_leg(args)
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:      # on rank 0 at every N: north_star wants the CPU figure "in the same run"
        try:
            cpu = cpu_baseline()
        except Exception as e:
            cpu = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        px_per_step = args.batch * TILE * TILE * world
        ms = dt / args.steps * 1e3
        # N = 1: the same step, bit for bit, can run eagerly (Python enqueues ~1 400 launches per step) or as ONE hipGraph replay
        # (jspsr_amd.graph.GraphedStep).  Which is faster depends on the box's HOST: on this pool the eager step reads 60-64 ms
        # on most boxes and 69-71 ms on those with a slow host, the replay 62-64 ms everywhere.  The headline is the faster of
        # the two on THIS box, both timed by the same rules (W untimed steps, then exactly K; the replay in the graph leg's child
        # process); `step_mode` says which, the other is kept beside it.
        step_mode, alt = "eager", None
        gk = f"tiles_{args.batch}"
        if world == 1 and graph and gk in graph and graph[gk].get("graph_steps_timed") == args.steps:
            g_ms = graph[gk]["graph_ms_per_step"]
            if g_ms < ms:
                alt = {"mode": "eager", "ms_per_step": round(ms, 3), "value": round(px_per_step / ms / 1e3, 4)}
                step_mode, ms, final_loss = "hipgraph replay (jspsr_amd.graph.GraphedStep), timed in the graph leg's child process", g_ms, graph[gk]["final_loss"]
                dt = ms * args.steps / 1e3
            else:
                alt = {"mode": "hipgraph replay", "ms_per_step": g_ms, "value": round(px_per_step / g_ms / 1e3, 4)}
        line = {
            "metric": "Mpixels/sec fwd+bwd, JSPSR x8 on 512^2 DEM tiles",
            "value": round(px_per_step / (dt / args.steps) / 1e6, 4),
            "unit": "Mpixel/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": "jspsr_r8_img_msk (image+mask guided, 43.87M params), "
                            f"{args.batch} x {TILE}x{TILE} tiles per GPU, train step "
                            "(fwd + L1/L2/Sobel loss + bwd + grad all-reduce + AdamW)"
                            + ("; storage policy: bf16 activations / gradients / MFMA operands, fp32 accumulation, BatchNorm statistics, "
                               "master weights and optimizer, and the generator head (affinity logits + offsets) kept as fp32 planes"
                               if args.dtype == "bf16" else "; fp32 storage"),
                "tiles_per_gpu": args.batch, "tile": TILE, "global_tiles": args.batch * world,
                "parallelism": f"dp{world}", "final_loss": round(final_loss, 6),
                "step_mode": step_mode, "other_mode": alt,
            },
            "host": {"abi_calls_per_step": abi_calls, "enqueue_ms_per_step": round(host_ms, 2),
                     "note": "C-ABI entry points called per step (each launches 1-3 kernels) and the host time to enqueue "
                             "them with nothing to wait for; while it stays under ms_per_step the GPU, not the host, sets the step"},
            "roofline": roof,
            "cpu_baseline": cpu,
            "fp32": fp32,
            "inference": infer,
            "graph": graph,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()            # rank 0 ran its micro-benchmarks and printed; leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
