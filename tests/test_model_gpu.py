"""GPU parity: the product Model (HIP path) against the reference-made fixtures and the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import jspsr_ref as R
from tests import fixtures as Fx

IMG, MSK = Fx.IMG, Fx.MSK
CASES = [
    ("g3_img_nf32_64_train.npz", IMG),
    ("g3_img_nf32_64_eval.npz", IMG),
    ("g3_img_nf8_b2_48x80_train.npz", IMG),
    ("g4_msk_nf8_b2_64_train.npz", MSK),
    ("g4_msk_nf8_b2_64_eval.npz", MSK),
    ("g4_msk_nf32_b1_64_train.npz", MSK),     # the benched architecture: image+mask, num_feature 32
]


def _build(z, ic):
    from jspsr_amd.JSPSR import Model
    sd, inputs, gt = Fx.regen_jspsr(z, ic)        # fails (never skips) if the fixture does not regenerate
    m = Model(dict(ic, COP30=1), num_feature=int(z["nf"]))
    m.load_state_dict(Fx.as_f32(sd))
    return m.cuda(), [t.float().cuda() for t in inputs], gt.float().cuda(), (sd, inputs)


_rel = Fx.rel


@pytest.mark.parametrize("name,ic", CASES)
def test_model_matches_reference_fixture(golden_dir, name, ic):
    """north_star tolerance: 1e-4 relative (fp32) against the reference's CPU result.  Gradients: every parameter,
    element-wise, against the fp64 oracle (itself held to the fixture at 1e-8 by test_oracle_golden.py and re-checked
    here), within a tolerance DERIVED IN THIS TEST from the oracle's own sensitivity to fp32-sized disturbances."""
    z = Fx.load(golden_dir, name)
    m, inputs, gt, (sd64, in64) = _build(z, ic)
    training = bool(z["training"])
    m.train(training)
    pred = m(*inputs)
    assert pred.shape == gt.shape and pred.is_cuda
    ref = torch.from_numpy(z["pred"])
    assert (pred.detach().cpu().double() - ref).abs().max().item() < 1e-4 * ref.abs().max().item()
    if not training:
        # inference path proper: under no_grad every conv -> BatchNorm(eval) (-> + shortcut) (-> ReLU) is one launch
        # with the normalisation folded into the conv epilogue (ops.conv_bn_infer) -- same bound in fp32
        with torch.no_grad():
            pred_i = m(*inputs)
        assert (pred_i.cpu().double() - ref).abs().max().item() < 1e-4 * ref.abs().max().item()
        m.compute_dtype = torch.bfloat16
        with torch.no_grad():
            pred_b = m(*inputs)
        assert (pred_b.cpu().double() - ref).abs().max().item() < 3e-2 * ref.abs().max().item()
        return
    loss = (pred - gt).abs().mean() + ((pred - gt) ** 2).mean()
    assert abs(loss.item() - float(z["loss"])) < 1e-4 * abs(float(z["loss"]))
    probe = R.probe_gradient(pred.shape, int(z["seed"]) + 2)                                             # fixed linear probe
    (pred * probe.float().cuda()).mean().backward()
    grads = {k: p.grad.detach().double().cpu() for k, p in m.named_parameters()}
    # fp64 oracle gradients of every parameter; chain of trust to the reference: its gradient norms and stored tensors
    fwd = lambda sd_, inp: R.jspsr_forward(sd_, inp, True)
    _, g_ref = Fx.oracle_gradients(fwd, sd64, in64, probe)
    for k, n in zip(z["grad_names"], z["grad_norms"]):
        assert abs(g_ref[str(k)].norm().item() - n) <= 1e-8 * max(n, 1e-30) + 1e-13, k
    # Measured noise floor (tests/fixtures.py::gradient_noise_floor): how far the ORACLE's gradient of each parameter
    # moves under (a) an fp32 rounding of inputs and parameters, (b) evaluation in fp32 and (c) fp32-accumulation-sized
    # noise on every convolution output, scaled to the forward deviation this very run shows.  Which parameters may see a
    # DISCRETE event (a ReLU mask, a max-pool arg-max or a sampler cell flipping) is not assumed but counted
    # (Fx.kink_census: elements within 10 x the local fp32 deviation of a kink, and the parameters upstream of each such
    # kink): a parameter with no at-risk kink downstream must meet 2 x its own floor; one upstream of an at-risk kink
    # 2 x max(own floor, the network's median floor) -- and the log names the kinks.
    dev = (pred.detach().cpu().double() - ref).abs().max().item()
    floor = Fx.gradient_noise_floor(fwd, sd64, in64, probe, g_ref, forward_dev=dev, pred_ref=ref)
    census = Fx.kink_census(fwd, sd64, in64, forward_dev=dev, pred_ref=ref)
    tols, risky = Fx.gradient_tolerances(floor, census)
    print(f"{name}: " + Fx.describe_census(census))
    print(f"{name}: {len(g_ref) - len(risky)} parameters have no at-risk kink downstream (held to 2 x their own floor), {len(risky)} have")
    worst, by_floor = [], []
    for k, ref_g in g_ref.items():
        err = _rel(grads[k], ref_g)
        worst.append((err / tols[k], k, err, tols[k]))
        by_floor.append((err / max(floor[k][1], 1e-30), k, err))
    by_floor.sort(reverse=True)
    for ratio, k, err in by_floor[:3]:      # the parameters furthest above their OWN floor, and the at-risk kinks that explain it
        print(f"  {k}: error {err:.2e} = {ratio:.1f} x own floor {floor[k][1]:.1e}; at-risk kinks downstream: {len(risky.get(k, []))} {risky.get(k, [])[:12]}")
        assert err <= 2 * floor[k][1] + 1e-5 or k in risky, (k, ratio)        # beyond its own floor only if a listed kink can explain it
    for k in ("layer3d.dconv.bn.bias",):      # the round-2 outlier of this architecture (146 x its floor on one run)
        print(f"  {k}: at-risk kinks downstream: {len(risky.get(k, []))} {risky.get(k, [])[:12]}")
    worst.sort(reverse=True)
    print(f"{name}: worst gradient error / derived tolerance: " + "; ".join(f"{k} {e:.2e}/{t:.2e}" for _, k, e, t in worst[:4]))
    assert worst[0][0] < 1.0, worst[:5]
    for k in z.files:
        if k.startswith("buf:"):
            assert _rel(m.state_dict()[k[4:]], z[k]) < 1e-4, k


def test_benched_configuration_bf16_training_step(golden_dir):
    """The benchmarked path -- image+mask, num_feature 32, bf16 storage / fp32 accumulate, TRAINING mode -- against the
    reference-made fixture.  bf16 cannot meet 1e-4: every stored activation and every stored gradient is rounded to
    8 significant bits, and every ReLU mask within that rounding of zero flips (0.2-0.4 % of them per layer, which
    alone moves a gradient by sqrt(0.003) ~ 5 % per layer it crosses).  The yardstick is therefore MEASURED: the fp64
    oracle with exactly those storage roundings inserted (tests/fixtures.py::bf16_emulated_oracle) -- what any
    bf16-storage implementation of this network does to the numbers.  Stated tolerances, HIP bf16 vs the fp64
    reference fixture:
      * prediction: relative L2 within 1.5 x the emulated oracle's own deviation (and < 1.5e-2 absolute); the tail of
        |diff| at its 99 % / 99.9 % points within 1.5 x / 2 x the emulated oracle's; the single worst pixel of 4096 -- one
        draw from a heavy tail, see below -- only by backstops (3 x the emulated oracle's, 1e-1 x max |ref|);
      * loss L1+L2 within 1.5 x the emulated oracle's deviation + 1e-3 relative;
      * parameter gradients: median and 90th percentile of the per-tensor relative L2 errors within 1.5 x the emulated
        oracle's (+ 0.02), every tensor within 3 x its own yardstick + 0.15, cosine to the fp64 gradient no worse than the
        emulated oracle's (median - 0.05); that bf16 TRAINS like fp32 is tests/test_model_scale_gpu.py::test_bf16_trains_like_fp32;
      * evaluation scores ("compare metrics, not tensors") through jspsr_amd.metrics on de-scaled
        elevations, bf16 vs fp32 prediction of the same module: |dRMSE| < 0.5 % of RMSE + 0.05 m, |dPSNR| < 0.05 dB.
    """
    from jspsr_amd import metrics as M
    z = Fx.load(golden_dir, "g4_msk_nf32_b1_64_train.npz")
    m, inputs, gt, (sd64, in64) = _build(z, MSK)
    m.train()
    state = {k: v.clone() for k, v in m.state_dict().items()}
    ref = torch.from_numpy(z["pred"])
    probe = R.probe_gradient(ref.shape, int(z["seed"]) + 2)
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        m.load_state_dict(state)             # same running statistics going in
        m.compute_dtype = dt
        m.zero_grad(set_to_none=True)
        pred = m(*inputs)
        (pred * probe.float().cuda()).mean().backward()
        out[dt] = (pred.detach(), {k: p.grad.detach().double().cpu() for k, p in m.named_parameters()})
    fwd = lambda sd_, inp: R.jspsr_forward(sd_, inp, True)
    _, g_ref = Fx.oracle_gradients(fwd, sd64, in64, probe)
    pe, ge = Fx.bf16_emulated_oracle(fwd, sd64, in64, probe)                 # the yardstick
    pb, gt64 = out[torch.bfloat16][0].cpu().double(), gt.cpu().double()
    loss = lambda p: ((p - gt64).abs().mean() + ((p - gt64) ** 2).mean()).item()
    d_hip = ((pb - ref).abs().max().item(), _rel(pb, ref), abs(loss(pb) - float(z["loss"])) / float(z["loss"]))
    d_emu = ((pe - ref).abs().max().item(), _rel(pe, ref), abs(loss(pe) - float(z["loss"])) / float(z["loss"]))
    print(f"bf16 prediction (max|diff|, rel L2, rel loss diff): HIP {d_hip}  emulated oracle {d_emu}; ref max {ref.abs().max().item():.3f}")
    q = lambda p, f: torch.quantile((p - ref).abs().flatten(), f).item()
    q_hip, q_emu = [q(pb, f) for f in (0.99, 0.999)], [q(pe, f) for f in (0.99, 0.999)]
    print(f"bf16 prediction |diff| quantiles (99 %, 99.9 %): HIP {q_hip}  emulated oracle {q_emu}")
    # The single worst pixel of 4096 is one draw from a heavy tail: the 32x32x16 and the 16x16x32 forms of the same conv
    # kernels -- which differ only in the fp32 summation order inside an MFMA -- give 0.0722 and 0.0769 at identical
    # relative L2 (7.04e-3 / 7.09e-3; profiles/r03_conv_mfma16.txt), i.e. the draw alone moves it by 6 %.  The tail is
    # therefore held at its 99 % and 99.9 % points (41 and 4 pixels) against the emulated oracle's own, and the worst
    # pixel only by the backstops.
    assert q_hip[0] < 1.5 * q_emu[0] and q_hip[1] < 2.0 * q_emu[1]
    assert d_hip[0] < 3.0 * d_emu[0] and d_hip[0] < 1e-1 * ref.abs().max().item()
    assert d_hip[1] < 1.5 * d_emu[1] and d_hip[1] < 1.5e-2
    assert d_hip[2] < 1.5 * d_emu[2] + 1e-3
    # per-tensor gradient errors: printed, not asserted -- on a random-init network under 8-bit storage the emulated
    # oracle's own median error is 0.5 (every layer flips 0.2-0.4 % of its ReLU masks), a yardstick that holds nothing.
    # That the bf16 step TRAINS like the fp32 one is asserted where it can be measured:
    # tests/test_model_scale_gpu.py::test_bf16_trains_like_fp32 (100 steps, loss windows and held-out scores).
    e_hip = np.array([_rel(out[torch.bfloat16][1][k], g_ref[k]) for k in g_ref])
    e_emu = np.array([_rel(ge[k], g_ref[k]) for k in g_ref])
    print(f"bf16 gradient error over {e_hip.size} tensors: HIP max {e_hip.max():.3f} median {np.median(e_hip):.3f}; "
          f"emulated oracle max {e_emu.max():.3f} median {np.median(e_emu):.3f}")
    assert np.isfinite(e_hip).all()
    # What IS held (a wrong gradient in one small tensor moves no 100-step curve): the DISTRIBUTION of the
    # per-tensor errors against the emulated oracle's own -- median and 90th percentile within 1.5 x (+ 0.02) -- and every
    # single tensor within 3 x the emulated oracle's error for that tensor + 0.15 (a sign, a factor 2, a missing term or
    # a wrong slice is an error of 1 on a tensor whose yardstick reads 0.1-0.6).
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm()).clamp_min(1e-300))
    c_hip = np.array([cos(out[torch.bfloat16][1][k], g_ref[k]) for k in g_ref])
    c_emu = np.array([cos(ge[k], g_ref[k]) for k in g_ref])
    names = list(g_ref)
    worst = np.argsort(e_hip - 3.0 * e_emu)[::-1][:3]
    print(f"bf16 gradient error quantiles (50 %, 90 %): HIP {np.median(e_hip):.3f} {np.quantile(e_hip, 0.9):.3f}; emulated oracle "
          f"{np.median(e_emu):.3f} {np.quantile(e_emu, 0.9):.3f}; cosine to the fp64 gradient, min / median: HIP {c_hip.min():.3f} {np.median(c_hip):.3f}, "
          f"emulated {c_emu.min():.3f} {np.median(c_emu):.3f}; furthest above 3 x their yardstick: "
          + ", ".join(f"{names[i]} {e_hip[i]:.3f} vs {e_emu[i]:.3f}" for i in worst))
    assert np.median(e_hip) < 1.5 * np.median(e_emu) + 0.02 and np.quantile(e_hip, 0.9) < 1.5 * np.quantile(e_emu, 0.9) + 0.02
    assert (e_hip < 3.0 * e_emu + 0.15).all(), [(names[i], e_hip[i], e_emu[i]) for i in worst]
    assert np.median(c_hip) > np.median(c_emu) - 0.05 and c_hip.min() > min(c_emu.min(), 0.5) - 0.25
    # scores on de-scaled elevations (configs/jspsr_r8_img_msk.yml: min -80, max 929, log scaling)
    sc = {}
    for dt in out:
        meter = M.Meter(-80.0, 929.0, border=0.05, elev_log=True)
        meter.update(out[dt][0], gt)
        sc[dt] = meter.scores()
    s32, s16 = sc[torch.float32], sc[torch.bfloat16]
    print("scores fp32", s32, "bf16", s16)
    assert abs(s16["RMSE"] - s32["RMSE"]) < 0.005 * s32["RMSE"] + 0.05
    assert abs(s16["PSNR"] - s32["PSNR"]) < 0.05


def test_generator_postprocessor_public_api():
    """Generator.forward returns the 18-channel torchvision layout; PostProcessor accepts it."""
    from jspsr_amd.spn import Generator, PostProcessor
    torch.manual_seed(0)
    gen, pp = Generator(16, 3, bc=8).cuda().eval(), PostProcessor().cuda()
    dem = torch.rand(2, 1, 32, 48, device="cuda")
    ctx = torch.randn(2, 16, 32, 48, device="cuda")
    weight, offset = gen(dem, ctx)
    assert weight.shape == (2, 9, 32, 48) and offset.shape == (2, 18, 32, 48)
    assert offset[:, 8:10].abs().max().item() == 0 and weight.min().item() > 0 and weight.max().item() < 1
    out = pp(dem, weight, offset)
    ref = R.propagate(dem.cpu().double(), weight.detach().cpu().double(), offset.detach().cpu().double(),
                      pp.w.detach().cpu().double(), pp.b.detach().cpu().double())
    assert (out.detach().cpu().double() - ref).abs().max().item() < 5e-6


def test_b2_dummy_batch_like_torchinfo():
    """main.py:92 feeds B=2 dummy batches through the model at start-up (utils/utils.py:83-100)."""
    from jspsr_amd.JSPSR import Model
    m = Model({"COP30": 1, "image": 3, "mask": 15, "lr_dem": 1}, num_feature=8).cuda().eval()
    with torch.no_grad():
        y = m(torch.rand(2, 1, 128, 128, device="cuda"), torch.rand(2, 3, 128, 128, device="cuda"),
              torch.rand(2, 15, 128, 128, device="cuda"))
    assert y.shape == (2, 1, 128, 128) and torch.isfinite(y).all()


def test_weight_gradients_straight_into_the_flat_buffer():
    """With a GradReducer the conv weight-gradient kernels add directly into the reducer's flat buffer and the
    autograd node reports no gradient for the weight (ops._wgrad_into).  Same numbers as autograd's own
    accumulation, two steps in a row (the buffer is re-zeroed in between), every parameter."""
    from jspsr_amd.JSPSR import Model
    from jspsr_amd.ddp import GradReducer
    torch.manual_seed(0)
    ic = dict(MSK, COP30=1)
    a, b = Model(ic, num_feature=8).cuda(), Model(ic, num_feature=8).cuda()
    b.load_state_dict(a.state_dict())
    inputs, gt = R.synthetic_batch(2, 32, 64, True, seed=5, dtype=torch.float32)
    inputs = [t.cuda() for t in inputs]
    probe = R.probe_gradient((2, 1, 32, 64), 7, torch.float32).cuda()
    red = GradReducer(b.parameters())
    direct = [p for p in b.parameters() if getattr(p, "_jspsr_direct_grad", False)]
    assert len(direct) > 150   # conv weights and BatchNorm scale/shift
    for _ in range(2):
        a.zero_grad(set_to_none=True)
        (a(*inputs) * probe).mean().backward()
        red.zero_grad()
        (b(*inputs) * probe).mean().backward()
        red.finish()
        for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
            assert pb.grad.data_ptr() >= red.flat.data_ptr() and torch.equal(pa.grad, pb.grad), n


def test_bucket_bookkeeping_with_direct_gradients(monkeypatch):
    """Data-parallel path: every bucket's all-reduce must be issued exactly once per step, after the last
    gradient of the bucket has been produced -- whether that gradient arrived through autograd's accumulation hook
    (biases, propagation parameters) or was written straight into the flat buffer by a kernel (conv weights,
    BatchNorm scale/shift).  The collective is stubbed; a second rank is not needed to check the counting."""
    import torch.distributed as dist
    from jspsr_amd.JSPSR import Model
    from jspsr_amd.ddp import GradReducer
    torch.manual_seed(0)
    m = Model(dict(MSK, COP30=1), num_feature=8).cuda()
    calls = []

    class _Work:
        def wait(self):
            return True

    def fake_all_reduce(t, op=None, group=None, async_op=False):
        calls.append((t.data_ptr(), t.numel(), [p for p in red._pending]))
        return _Work()

    monkeypatch.setattr(dist, "all_reduce", fake_all_reduce)
    red = GradReducer(m.parameters(), bucket_bytes=256 << 10, world=2)   # several buckets
    assert len(red.buckets) > 3
    inputs, _ = R.synthetic_batch(2, 32, 64, True, seed=5, dtype=torch.float32)
    inputs = [t.cuda() for t in inputs]
    for _ in range(2):
        calls.clear()
        red.zero_grad()
        m(*inputs).mean().backward()
        assert len(calls) == len(red.buckets), "every bucket reduced during backward, none left for finish()"
        assert all(v == 0 for v in red._pending)
        es = red.flat.element_size()
        got = sorted(((ptr - red.flat.data_ptr()) // es, n) for ptr, n, _ in calls)
        assert got == sorted((s, e - s) for s, e, _ in red.buckets)
        red.finish()
        assert len(calls) == len(red.buckets)


def test_block_level_autograd_node_equals_op_by_op_graph():
    """ResUnit.fused (one autograd node per BasicBlock, shortcut and parked skip gradients added inside the conv1
    data-gradient kernel) against the op-by-op graph where autograd performs those additions: the forward is the
    same kernel sequence (bit-identical), so every gradient may differ only by the order of a few additions."""
    from jspsr_amd.JSPSR import Model
    from jspsr_amd.blocks import ResUnit
    torch.manual_seed(0)
    ic = dict(MSK, COP30=1)
    m = Model(ic, num_feature=8).cuda()
    inputs, _ = R.synthetic_batch(2, 32, 64, True, seed=9, dtype=torch.float32)
    inputs = [t.cuda() for t in inputs]
    probe = R.probe_gradient((2, 1, 32, 64), 3, torch.float32).cuda()
    res = []
    for fused in (True, False):
        ResUnit.fused = fused
        try:
            m.zero_grad(set_to_none=True)
            out = m(*inputs)
            (out * probe).mean().backward()
            res.append((out.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters()}))
        finally:
            ResUnit.fused = True
    assert torch.equal(res[0][0], res[1][0])
    worst = max(_rel(res[0][1][n], res[1][1][n]) for n in res[0][1])
    assert worst < 2e-5, worst


@pytest.mark.parametrize("nf,B,H,W,dtype", [(8, 2, 64, 64, torch.float32), (32, 2, 256, 256, torch.bfloat16)])
def test_branch_streams_change_nothing(nf, B, H, W, dtype):
    """The guidance branches run on side streams beside the dem branch (forward and, through autograd's stream
    replay, backward), and every weight-gradient kernel runs on an auxiliary stream beside the data-gradient chain
    (ops._wgrad_async).  Same kernels on the same data in the same per-tensor order: outputs, every parameter
    gradient and the BatchNorm buffers must be bit-identical to the single-stream run, on every repetition."""
    from jspsr_amd import ops
    from jspsr_amd.JSPSR import Model
    from jspsr_amd.ddp import GradReducer
    torch.manual_seed(0)
    ic = dict(MSK, COP30=1)
    ref_m = Model(ic, num_feature=nf).cuda()
    ref_m.compute_dtype = dtype
    state = {k: v.clone() for k, v in ref_m.state_dict().items()}
    inputs, _ = R.synthetic_batch(B, H, W, True, seed=11, dtype=torch.float32)
    inputs = [t.cuda() for t in inputs]
    probe = R.probe_gradient((B, 1, H, W), 13, torch.float32).cuda()

    def run(streams, direct):
        m = Model(ic, num_feature=nf).cuda()
        m.load_state_dict(state)
        m.compute_dtype = dtype
        m.branch_streams = streams
        keep, ops.wgrad_async = ops.wgrad_async, streams
        try:
            red = GradReducer(m.parameters()) if direct else None
            for _ in range(2):                       # second step: allocator blocks are being recycled
                if red is not None:
                    red.zero_grad()
                else:
                    m.zero_grad(set_to_none=True)
                out = m(*inputs)
                (out * probe).mean().backward()
                if red is not None:
                    red.finish()                     # orders this stream after the gradient streams
            torch.cuda.synchronize()
        finally:
            ops.wgrad_async = keep
        return out.detach().clone(), [p.grad.clone() for p in m.parameters()], [b.clone() for b in m.buffers()]

    base = run(False, False)
    assert len(Model(ic, num_feature=nf).cuda().side_streams()) == 2
    for rep in range(3):
        for direct in (False, True):
            got = run(True, direct)
            assert torch.equal(got[0], base[0]), (rep, direct)
            for i, (a, b) in enumerate(zip(got[1], base[1])):
                assert torch.equal(a, b), (rep, direct, i)
            for a, b in zip(got[2], base[2]):
                assert torch.equal(a, b)
