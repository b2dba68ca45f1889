"""Shared fixture plumbing for the parity tests (test infrastructure).

The fixtures under tests/golden/ were produced by the reference's own modules (oracle/gen_golden.py) on
parameters and inputs drawn from numpy's frozen legacy RandomState stream; a fixture stores the seed and
an order-sensitive checksum of what that seed must regenerate.  A mismatch is a hard FAILURE (never a skip):
a silently skipped fixture test would read as green while pinning nothing.
"""
import os

import numpy as np
import torch

from oracle import jspsr_ref as R

IMG = {"lr_dem": 1, "image": 3}
MSK = {"lr_dem": 1, "image": 3, "mask": 15}


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def regen(z, shapes, with_mask):
    """(state_dict fp64, inputs fp64, target fp64) of fixture `z`; fails loudly if the stream moved."""
    seed = int(z["seed"])
    B, H, W = (int(v) for v in z["BHW"])
    sd = R.make_state_dict(shapes, seed, torch.float64)
    inputs, gt = R.synthetic_batch(B, H, W, with_mask, seed=seed + 1, dtype=torch.float64)
    c1, c2 = R.checksum(sd.values()), R.checksum(list(inputs) + [gt])
    assert abs(c1 - float(z["param_checksum"])) <= 1e-9 * abs(c1), \
        "fixture parameters do not regenerate (numpy RandomState stream / erfinv changed?): rerun oracle/gen_golden.py"
    assert abs(c2 - float(z["input_checksum"])) <= 1e-9 * abs(c2), \
        "fixture inputs do not regenerate: rerun oracle/gen_golden.py"
    return sd, inputs, gt


def regen_jspsr(z, in_channels):
    return regen(z, R.jspsr_param_shapes(in_channels, int(z["nf"])), "mask" in in_channels)


def as_f32(sd):
    return {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def oracle_gradients(forward, sd, inputs, probe, dtype=torch.float64):
    """Parameter gradients of mean(pred * probe) from the CPU oracle in `dtype` -> (pred, {name: grad})."""
    cast = lambda v: v.detach().to(dtype) if v.is_floating_point() else v.clone()
    sd = {k: cast(v) for k, v in sd.items()}
    params = {k: v.requires_grad_() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    sd.update(params)
    pred = forward(sd, [cast(t) for t in inputs])
    (pred * probe.to(dtype)).mean().backward()
    return pred.detach(), {k: v.grad.detach().double() for k, v in params.items() if v.grad is not None}


class _RoundingF:
    """Stand-in for `torch.nn.functional` inside the oracle while a noise-floor draw runs: every convolution output
    is multiplied by (1 + eps n), n ~ N(0,1), eps = 2^-24 sqrt(K) with K = taps x input channels -- the rounding
    error a K-term fp32 accumulation carries, whatever the order of its additions.  Everything else passes through."""

    def __init__(self, real, rs, scale=1.0):
        self._real, self._rs, self._scale = real, rs, scale

    def __getattr__(self, name):
        return getattr(self._real, name)

    def _jitter(self, y, K):
        eps = self._scale * 2.0 ** -24 * float(K) ** 0.5
        return y * (1 + eps * torch.from_numpy(self._rs.standard_normal(tuple(y.shape))).to(y.dtype))

    def conv2d(self, x, w, *a, **k):
        return self._jitter(self._real.conv2d(x, w, *a, **k), w.shape[1] * w.shape[2] * w.shape[3])

    def conv_transpose2d(self, x, w, *a, **k):
        return self._jitter(self._real.conv_transpose2d(x, w, *a, **k), w.shape[0] * w.shape[2] * w.shape[3] / 4.0)


def gradient_noise_floor(forward, sd, inputs, probe, g_ref, n_trials=2, n_rounding=8, forward_dev=None, pred_ref=None):
    """Per-parameter relative gradient change the ORACLE ITSELF shows under fp32-sized disturbances -- the
    floor below which a gradient comparison against an fp32 implementation carries no information (every
    ReLU whose pre-activation sits within rounding of zero may flip, and a flipped mask changes gradient
    entries by O(1)).  Two measurements, max of both per parameter:
      (a) fp64 oracle with every floating input and parameter multiplied by (1 + 2^-23 u), u ~ U(-1,1)
          (an fp32 rounding of the operands), `n_trials` draws;
      (b) the oracle evaluated in fp32 end to end (torch CPU kernels: another fp32 implementation of the
          same formulae, rounding at every layer as any fp32 implementation must);
      (c) fp64 oracle with every convolution output carrying the rounding error of a K-term fp32 accumulation
          (_RoundingF), `n_rounding` draws.  (a) and (b) are single draws of a heavy-tailed quantity -- most
          draws flip no mask at all, one flip moves every gradient upstream of it by 1e-3..1e-2 -- so on their
          own they can read 2e-6 on a fixture where the next fp32 implementation (the HIP one) does hit a flip;
          (c) samples the same mechanism often enough to see it.  The sequential-accumulation estimate 2^-24 sqrt(K)
          is pessimistic for tree / MFMA accumulation; with `forward_dev` (the max |prediction - fp64 reference| the
          implementation under test actually shows) and `pred_ref` the noise is scaled so that the oracle's
          prediction deviates by just that much -- the disturbance is sized by the implementation's own measured
          forward error (scale clamped to [0.05, 1]).
    Returns {name: (norm_floor, tensor_floor)}: relative change of the gradient norm, and relative L2 change
    of the gradient tensor."""
    floors = {k: [0.0, 0.0] for k in g_ref}

    def fold(g):
        for k, ref in g_ref.items():
            n = ref.norm().item()
            floors[k][0] = max(floors[k][0], abs(g[k].norm().item() - n) / max(n, 1e-30))
            floors[k][1] = max(floors[k][1], (g[k] - ref).norm().item() / max(n, 1e-30))

    for t in range(n_trials):
        rs = np.random.RandomState(1000 + t)
        jig = lambda v: v * (1 + 2.0 ** -23 * torch.from_numpy(rs.uniform(-1, 1, tuple(v.shape)))) if v.is_floating_point() else v
        fold(oracle_gradients(forward, {k: jig(v) for k, v in sd.items()}, [jig(x) for x in inputs], probe)[1])
    fold(oracle_gradients(forward, sd, inputs, probe, torch.float32)[1])
    real = R.F
    try:
        scale = 1.0
        if forward_dev is not None and pred_ref is not None:
            R.F = _RoundingF(real, np.random.RandomState(1999))
            with torch.no_grad():
                dev1 = (forward(sd, inputs) - pred_ref).abs().max().item()
            scale = min(1.0, max(0.05, forward_dev / max(dev1, 1e-30)))
        for t in range(n_rounding):
            R.F = _RoundingF(real, np.random.RandomState(2000 + t), scale)
            fold(oracle_gradients(forward, sd, inputs, probe)[1])
    finally:
        R.F = real
    return {k: tuple(v) for k, v in floors.items()}


# ---- where a gradient may legitimately jump: a census of the network's kinks ------------------------------------------
def kink_census(forward, sd, inputs, forward_dev=None, pred_ref=None, factor=10.0, patience=4):
    """Every point of the oracle's graph where the gradient is discontinuous in the activations, with a count of the
    elements that sit close enough to the discontinuity for an fp32 implementation to land on the other side:
      * ReLU (every F.relu of the oracle): pre-activation z, at risk if |z| < factor x dev;
      * the channel gate's global max-pool (resnet_cbam.py:44: the gradient goes to the arg-max pixel): at risk if the
        two largest values of a channel are closer than factor x dev;
      * the bilinear sampler (spn.py:105): a learned tap (not the constant centre one) at risk if its coordinate is
        within factor x dev of an integer, where d/d(offset) jumps.
    dev = max |fp32 evaluation - fp64 evaluation| of THAT tensor in the oracle (the deviation an fp32 implementation
    shows at that point), scaled up by (forward deviation of the implementation under test / the fp32 oracle's own) when
    that ratio exceeds 1.  For every at-risk kink the set of parameters UPSTREAM of it (the ones whose gradient passes
    through it) is found structurally (autograd.grad of the kink tensor, allow_unused).  -> list of dicts
    (index, kind, shape, dev, n_risk, upstream)."""
    real_F, real_ca, real_st = R.F, R.channel_attention, R.sample_taps

    def run(dtype, grad):
        rec = []

        class Rec:
            def __getattr__(self, name):
                return getattr(real_F, name)

            def relu(self, x, *a, **k):
                rec.append(("relu", x))
                return real_F.relu(x, *a, **k)

        def ca(c, x, p_):
            rec.append(("gate-max " + p_, x))
            return real_ca(c, x, p_)

        def st(dem, offset):
            rec.append(("sampler", offset))
            return real_st(dem, offset)

        cast = lambda v: v.detach().to(dtype) if v.is_floating_point() else v.clone()
        sd_ = {k: cast(v) for k, v in sd.items()}
        params = {k: v.requires_grad_() for k, v in sd_.items() if grad and v.is_floating_point() and "running" not in k}
        sd_.update(params)
        try:
            R.F, R.channel_attention, R.sample_taps = Rec(), ca, st
            with torch.enable_grad() if grad else torch.no_grad():
                pred = forward(sd_, [cast(t) for t in inputs])
        finally:
            R.F, R.channel_attention, R.sample_taps = real_F, real_ca, real_st
        return pred, rec, params

    p64, r64, params = run(torch.float64, True)
    p32, r32, _ = run(torch.float32, False)
    assert len(r64) == len(r32) and all(a[0] == b[0] for a, b in zip(r64, r32))
    scale = 1.0
    if forward_dev is not None and pred_ref is not None:
        own = (p32.double() - pred_ref).abs().max().item()
        scale = max(1.0, forward_dev / max(own, 1e-30))
    names, plist = list(params), list(params.values())
    out = []
    for i, ((kind, z), (_, z32)) in enumerate(zip(r64, r32)):
        zd = z.detach()
        dev = scale * (z32.double() - zd).abs().max().item()
        if kind == "relu":
            n = int((zd.abs() < factor * dev).sum())
        elif kind.startswith("gate-max"):
            top = zd.flatten(2).topk(2, dim=2).values
            n = int(((top[..., 0] - top[..., 1]) < factor * dev).sum())
        else:
            B, C, H, W = zd.shape
            ys = torch.arange(H, dtype=zd.dtype).view(1, 1, H, 1)
            xs = torch.arange(W, dtype=zd.dtype).view(1, 1, 1, W)
            pos = zd.clone()
            pos[:, 0::2] += ys
            pos[:, 1::2] += xs
            d = (pos - pos.round()).abs()
            d[:, 8:10] = 1.0                                   # the centre tap's offset is the constant 0
            n = int((d < factor * dev).sum())
        out.append({"index": i, "kind": kind, "shape": tuple(zd.shape), "dev": dev, "n_risk": n, "upstream": None})
    # who is upstream of an at-risk kink: one backward pass per kink, from the output end of the network (the late kinks
    # have nearly every parameter upstream); stop once `patience` further kinks add no parameter that was not already
    # covered -- the listing per parameter is then a subset of its at-risk kinks, the tolerance tiers are exact
    covered, idle = set(), 0
    for kk, (kind, z) in reversed(list(zip(out, r64))):
        if kk["n_risk"] == 0 or not z.requires_grad or idle >= patience:
            continue
        gs = torch.autograd.grad(z.sum(), plist, retain_graph=True, allow_unused=True)
        kk["upstream"] = {k for k, g in zip(names, gs) if g is not None}
        idle = 0 if kk["upstream"] - covered else idle + 1
        covered |= kk["upstream"]
    return out


def gradient_tolerances(floor, census, factor=2.0):
    """Per-parameter tolerance from the measured floors and the kink census.  A parameter with NO at-risk kink between it
    and the output must meet factor x its OWN measured floor (1e-6 .. 1e-5: nothing discrete can happen to its
    gradient).  A parameter upstream of at least one at-risk kink may see a flip -- a discrete event: a given draw of
    the noise floor either hits one inside a sub-network or does not, and the handful of oracle draws cannot visit every
    sub-network; what the draws DO measure is what a flip costs in this network, the median floor over all
    parameters -- and gets factor x max(own floor, that median).  -> ({name: tolerance}, {name: [at-risk kink indices
    downstream of it]})."""
    med = float(np.median([v[1] for v in floor.values()]))
    risky = {}
    for kk in census:
        if kk["n_risk"] > 0 and kk["upstream"]:
            for name in kk["upstream"]:
                risky.setdefault(name, []).append(kk["index"])
    tol = {k: factor * (max(v[1], med) if k in risky else v[1]) + 1e-5 for k, v in floor.items()}
    return tol, risky


def describe_census(census):
    risk = [k for k in census if k["n_risk"] > 0]
    head = f"{len(census)} kinks ({sum(k['kind'] == 'relu' for k in census)} ReLU), {len(risk)} with elements at risk"
    rows = [f"  #{k['index']:3d} {k['kind']:28s} {str(k['shape']):22s} dev {k['dev']:.1e} at risk {k['n_risk']:5d} upstream params {len(k['upstream']) if k['upstream'] is not None else '(not traced)'}"
            for k in risk]
    return head + ("\n" + "\n".join(rows) if rows else "")


# ---- bf16-storage emulation of the oracle: what ANY implementation that stores activations (and their gradients) in
#      bf16 between layers does to the numbers -- the yardstick for the bf16 path's stated tolerances ----------------
class _QuantBoth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y):
        return y.to(torch.bfloat16).to(y.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _Bf16F:
    """Stand-in for `torch.nn.functional` inside the oracle: the result of every convolution, BatchNorm and ReLU is
    rounded to bf16 (round-to-nearest-even) on the way forward, and the gradient flowing back through the same
    points is rounded too -- the storage roundings of a bf16 activation pipeline with fp32 accumulation."""

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        return getattr(self._real, name)

    def conv2d(self, *a, **k):
        return _QuantBoth.apply(self._real.conv2d(*a, **k))

    def conv_transpose2d(self, *a, **k):
        return _QuantBoth.apply(self._real.conv_transpose2d(*a, **k))

    def relu(self, *a, **k):
        return _QuantBoth.apply(self._real.relu(*a, **k))

    def batch_norm(self, *a, **k):
        return _QuantBoth.apply(self._real.batch_norm(*a, **k))


def bf16_emulated_oracle(forward, sd, inputs, probe, round_inputs=True):
    """-> (pred, grads) of the fp64 oracle with bf16 storage roundings (inputs rounded like engine.from_nchw does)."""
    q = lambda t: t.to(torch.bfloat16).to(t.dtype)
    real = R.F
    try:
        R.F = _Bf16F(real)
        return oracle_gradients(forward, sd, [q(t) for t in inputs] if round_inputs else inputs, probe)
    finally:
        R.F = real
