"""TEST INFRASTRUCTURE (not collected by pytest): which tensors must stay fp32 for bf16 storage to train like fp32?

VERDICT r3 item 1.  The experiment of tests/test_model_scale_gpu.py::test_bf16_trains_like_fp32 (image+mask JSPSR,
num_feature 8, 2 x 128 x 128, four batches cycled, AdamW lr 1e-3 wd 1e-6, MultiLoss(1, 1, 0.1), held-out scores after
steps 70 / 80 / 90 / 100 averaged; reference loop: train/train_utils.py:205-219, scores: evaluation/metrics.py:338-420)
run on the CPU in the oracle, with the storage roundings of a bf16 activation pipeline inserted SELECTIVELY.

Rounding points (what the HIP path stores in its compute dtype; accumulation is fp32 everywhere):
  * forward:  the output of every convolution / transposed convolution, BatchNorm and ReLU; the packed conv weights
    (fp32 masters, bf16 copies used by the MFMA kernels: straight-through); the network inputs (engine.from_nchw);
  * backward: the gradient flowing back through the same points (data-gradient outputs, BatchNorm-backward outputs).
A *policy* names the points that stay fp32.  Every point knows the state-dict name of the layer it belongs to (the conv's
or BatchNorm's own parameter tensor identifies it; a ReLU belongs to the layer evaluated last before it).

usage: python tests/ablation_bf16_policy.py [--seeds 4] [--workers 4] [--steps 100] [--variants a,b,...] [--out file]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import jspsr_ref as R          # noqa: E402
from oracle import metrics_ref as MR       # noqa: E402

MSK = {"lr_dem": 1, "image": 3, "mask": 15}
HEAD = ("generator.conv_weight.0", "generator.conv_offset.conv.0")


class _Quant(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, qf, qb):
        ctx.qb = qb
        return y.to(torch.bfloat16).to(y.dtype) if qf else y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).to(g.dtype) if ctx.qb else g), None, None


class PolicyF:
    """Stand-in for torch.nn.functional inside the oracle.  policy(layer_name, kind) -> (round forward, round gradient),
    kind in {"conv", "bn", "relu", "weight"}."""

    def __init__(self, real, names, policy):
        self._real, self._names, self._policy, self._last = real, names, policy, "input"

    def __getattr__(self, name):
        return getattr(self._real, name)

    def _q(self, y, kind):
        qf, qb = self._policy(self._last, kind)
        return _Quant.apply(y, qf, qb) if (qf or qb) else y

    def _layer(self, t):
        n = self._names.get(id(t))
        if n is not None:
            self._last = n.rsplit(".", 1)[0]

    def _w(self, w):
        self._layer(w)
        if w.dim() == 4 and self._policy(self._last, "weight")[0] and ".camb." not in self._last:
            return w + (w.detach().to(torch.bfloat16).to(w.dtype) - w.detach())       # straight-through: fp32 master
        return w

    def conv2d(self, x, w, *a, **k):
        wq = self._w(w)
        if ".camb" in self._last:                     # the gate's MLP on pooled vectors runs in fp32 (gate_mlp_*)
            return self._real.conv2d(x, w, *a, **k)
        return self._q(self._real.conv2d(x, wq, *a, **k), "conv")

    def conv_transpose2d(self, x, w, *a, **k):
        return self._q(self._real.conv_transpose2d(x, self._w(w), *a, **k), "conv")

    def batch_norm(self, x, rm, rv, weight, *a, **k):
        self._layer(weight)
        return self._q(self._real.batch_norm(x, rm, rv, weight, *a, **k), "bn")

    def relu(self, x, *a, **k):
        if ".camb" in self._last:
            return self._real.relu(x, *a, **k)
        return self._q(self._real.relu(x, *a, **k), "relu")


def _in(name, prefixes):
    return any(name.startswith(p) for p in prefixes)


def make_policy(tag):
    """-> (policy, round the inputs?)"""
    keep = set(tag.split("+")) if tag not in ("fp32", "bf16") else set()
    if tag == "fp32":
        return (lambda n, k: (False, False)), False

    def policy(n, kind):
        qf = qb = True
        if kind == "weight":
            return ("w32" not in keep), False
        if "head" in keep and _in(n, HEAD):                              # (a) the 25-channel head: logits + offsets
            qf = qb = False
        if "headfwd" in keep and _in(n, HEAD):                           #     ... forward only
            qf = False
        if "bngrad" in keep and kind in ("bn", "relu"):                  # (b) gradients entering BatchNorm backward
            qb = False
        if "gen" in keep and n.startswith("generator."):                 # (c) the generator (all at full resolution)
            qf = qb = False
        if "genblock" in keep and _in(n, ("generator.block", "generator.conv.")):   # its last three convs only
            qf = qb = False
        if "dem" in keep and (n.startswith("conv_dem") or "_dem." in n):  # (d) the DEM branch and its stem
            qf = qb = False
        if "stems" in keep and _in(n, ("conv_dem", "conv_img", "conv_aux")):
            qf = qb = False
        if "dec" in keep and _in(n, ("layer3d", "layer2d", "layer1d", "conv0")):
            qf = qb = False
        if "conv0" in keep and n.startswith("conv0"):
            qf = qb = False
        if "allgrad" in keep:                                            # diagnostic: every gradient fp32
            qb = False
        if "allfwd" in keep:                                             # diagnostic: every activation fp32
            qf = False
        return qf, qb

    return policy, ("in32" not in keep)


def run(tag, seed, steps=100, decay_from=0, nf=8, B=2, HW=128, threads=2):
    torch.set_num_threads(threads)
    policy, round_inputs = make_policy(tag)
    sd0 = R.make_state_dict(R.jspsr_param_shapes(MSK, nf), 991, torch.float64)
    rs = np.random.RandomState(5000 + seed)
    jig = (lambda v: v * (1 + 2.0 ** -23 * torch.from_numpy(rs.uniform(-1, 1, tuple(v.shape))))) if seed else (lambda v: v)
    sd = {k: (jig(v).float() if v.is_floating_point() and v.dim() > 0 and "running" not in k else (v.float() if v.is_floating_point() else v.clone()))
          for k, v in sd0.items()}
    params = {k: v.requires_grad_() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    names = {id(v): k for k, v in sd.items()}
    q = (lambda t: t.to(torch.bfloat16).float()) if round_inputs else (lambda t: t)
    batches = []
    for s in range(5):
        i32, g32 = R.synthetic_batch(B, HW, HW, True, seed=1000 + s, dtype=torch.float32)
        batches.append(([i32[0]] + [q(t) for t in i32[1:]], g32, q(i32[0])))
    held = batches.pop()
    opt = torch.optim.AdamW(list(params.values()), lr=1e-3, weight_decay=1e-6)
    real = R.F
    losses, evals = [], []

    def fwd(inputs, dem_q, training):
        # the propagation step reads the fp32 raster (K1h: dem and out are fp32); the conv stems read the rounded one
        R.F = PolicyF(real, names, policy)
        try:
            return _forward_split_dem(sd, inputs, dem_q, training)
        finally:
            R.F = real

    for i in range(steps):
        inputs, gt, dem_q = batches[i % 4]
        if decay_from and i == decay_from:          # the reference's StepLR (common_config.py:339-358), compressed: one drop
            for grp in opt.param_groups:
                grp["lr"] *= 0.1
        opt.zero_grad()
        loss = R.multi_loss(fwd(inputs, dem_q, True), gt)["Total"]
        loss.backward()
        opt.step()
        losses.append(loss.item())
        if i + 1 in (steps - 30, steps - 20, steps - 10, steps):
            with torch.no_grad():
                pred = fwd(held[0], held[2], False)
            evals.append(MR.mean_scores(pred.numpy(), held[1].numpy(), -80.0, 929.0, 0.05, True))
    win = np.array(losses).reshape(-1, 10).mean(1)
    return {"tag": tag, "seed": seed, "decay_from": decay_from, "loss_windows": [float(v) for v in win],
            "RMSE": float(np.mean([e["RMSE"] for e in evals])), "PSNR": float(np.mean([e["PSNR"] for e in evals]))}


def _forward_split_dem(sd, inputs, dem_q, training):
    """R.jspsr_forward, except that the conv stem sees the raster rounded to the storage type while the generator's
    detached raster and the propagation step keep the fp32 one -- as jspsr_amd/JSPSR.py does (K1h reads fp32 dem)."""
    if dem_q is inputs[0]:
        return R.jspsr_forward(sd, inputs, training)
    # jspsr_forward uses inputs[0] both for the stem and (detached) for generator + propagate: swap it around the stem
    orig_basic2d = R.basic2d

    def basic2d(c, x, p, *a, **k):
        if p in ("conv_dem", "generator.convd1"):      # both read engine.from_nchw(dem): the rounded raster
            x = dem_q
        return orig_basic2d(c, x, p, *a, **k)

    R.basic2d = basic2d
    try:
        return R.jspsr_forward(sd, inputs, training)
    finally:
        R.basic2d = orig_basic2d


def _job(args):
    t = time.time()
    r = run(*args)
    r["seconds"] = round(time.time() - t, 1)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=4)
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--variants", default="fp32,bf16,head,bngrad,gen,dem,allgrad,allfwd,in32,w32")
    ap.add_argument("--out", default="gpurun_out/bf16_policy_ablation.jsonl")
    ap.add_argument("--decay-from", type=int, default=0, help="multiply the learning rate by 0.1 from this step on (0: never)")
    a = ap.parse_args()
    jobs = [(v, s, a.steps, a.decay_from) for s in range(a.seeds) for v in a.variants.split(",")]
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(a.workers) as pool, open(a.out, "a") as f:
        for r in pool.imap_unordered(_job, jobs):
            f.write(json.dumps(r) + "\n")
            f.flush()
            print(f"{r['tag']:16s} seed {r['seed']} RMSE {r['RMSE']:.3f} PSNR {r['PSNR']:.2f} last window {r['loss_windows'][-1]:.5f} ({r['seconds']} s)", flush=True)


if __name__ == "__main__":
    main()
