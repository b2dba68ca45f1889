"""GPU parity: fused loss (forward + gradient) against the oracle's formula, fused AdamW against torch."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import jspsr_ref as R


@pytest.mark.parametrize("shape", [(2, 1, 37, 53), (8, 1, 128, 128), (1, 1, 5, 3)])
def test_fused_loss_matches_oracle(shape):
    from jspsr_amd.losses import MultiLoss
    g = torch.Generator().manual_seed(sum(shape))
    pred = torch.rand(shape, generator=g)
    gt = (pred + 0.05 * torch.randn(shape, generator=g))
    p64 = pred.double().requires_grad_()
    ref = R.multi_loss(p64, gt.double(), 1.0, 1.0, 0.1)
    (3.0 * ref["Total"]).backward()
    pc = pred.cuda().requires_grad_()
    out = MultiLoss(1.0, 1.0, 0.1)(pc, gt.cuda())
    for k in ("L1", "L2", "Grad", "Total"):
        assert abs(out[k].item() - ref[k].item()) < 2e-6 * max(1.0, abs(ref[k].item())), k
    (3.0 * out["Total"]).backward()
    err = (pc.grad.cpu().double() - p64.grad).abs().max().item()
    assert err < 1e-6 * p64.grad.abs().max().item() + 1e-9


def test_flat_adamw_matches_torch():
    from jspsr_amd.ddp import GradReducer
    from jspsr_amd.optim import FlatAdamW
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.Conv2d(8, 5, 1)).cuda()
    ref = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.Conv2d(8, 5, 1)).cuda()
    ref.load_state_dict(net.state_dict())
    red = GradReducer(net.parameters())
    opt = FlatAdamW(red, lr=1e-3, weight_decay=1e-2)
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=1e-2)
    x = torch.randn(4, 3, 9, 9, device="cuda")
    for _ in range(3):
        opt.zero_grad()
        net(x).square().mean().backward()
        red.finish()
        opt.step()
        ropt.zero_grad()
        ref(x).square().mean().backward()
        ropt.step()
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    assert set(net.state_dict()) == set(ref.state_dict())


def test_flat_adamw_second_lr_group_and_schedule():
    """diff_lr grouping of utils/common_config.py:247-258 (named parameters containing a key run at their own
    learning rate) + the WarmupStepLR schedule, against torch.optim.AdamW with two param groups."""
    from jspsr_amd.ddp import GradReducer
    from jspsr_amd.optim import FlatAdamW, WarmupStepLR
    torch.manual_seed(1)
    mk = lambda: torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.Conv2d(8, 8, 1), torch.nn.Conv2d(8, 5, 1)).cuda()
    net, ref = mk(), mk()
    ref.load_state_dict(net.state_dict())
    red = GradReducer(net.parameters())
    opt = FlatAdamW(red, lr=1e-3, weight_decay=1e-6, lr_overrides={p: 3e-4 for p in net[1].parameters()})
    assert len(opt.param_groups) == 2 and sum(len(g["ranges"]) for g in opt.param_groups) == 3
    ropt = torch.optim.AdamW([{"params": list(ref[0].parameters()) + list(ref[2].parameters())},
                              {"params": list(ref[1].parameters()), "lr": 3e-4}], lr=1e-3, weight_decay=1e-6)
    sch = WarmupStepLR(opt, warmup_epoch=2, step_size=2, gamma=0.5)
    rsch = WarmupStepLR(ropt, warmup_epoch=2, step_size=2, gamma=0.5)
    x = torch.randn(4, 3, 9, 9, device="cuda")
    for _ in range(6):
        opt.zero_grad()
        net(x).square().mean().backward()
        red.finish()
        opt.step()
        sch.step()
        ropt.zero_grad()
        ref(x).square().mean().backward()
        ropt.step()
        rsch.step()
        assert sch.get_last_lr() == rsch.get_last_lr()
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)


def test_flat_adamw_checkpoint_round_trip_is_bit_exact(tmp_path):
    """The reference saves optimizer.state_dict() with every improved model (main.py:246-252) and restores it on resume
    (utils/utils.py:394): save after 3 steps -> torch.save / torch.load -> fresh model + optimizer -> 3 more steps
    must equal 6 uninterrupted steps bit for bit, with the parameter / gradient aliases into the flat buffers intact.
    The toy model is element-wise on purpose: its gradients are bit-reproducible, so any difference is the optimizer's."""
    from jspsr_amd.ddp import GradReducer
    from jspsr_amd.optim import FlatAdamW, WarmupStepLR
    torch.manual_seed(2)

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Parameter(torch.randn(3, 1000))
            self.b = torch.nn.Parameter(torch.randn(1000))
            self.c = torch.nn.Parameter(torch.randn(7))

        def forward(self, x):
            return (torch.tanh(self.a * x) * self.b).sum(0) + self.c.repeat(143)[:1000]

    x = torch.randn(3, 1000, device="cuda")
    mk = lambda: Toy().cuda()

    def make(net):
        red = GradReducer(net.parameters())
        opt = FlatAdamW(red, lr=1e-3, weight_decay=1e-6, lr_overrides={net.c: 3e-4})
        return red, opt, WarmupStepLR(opt, warmup_epoch=2, step_size=2, gamma=0.5)

    def steps(net, red, opt, sch, n):
        for _ in range(n):
            opt.zero_grad()
            (net(x) ** 2).mean().backward()
            red.finish()
            opt.step()
            sch.step()

    a = mk()
    start = {k: v.clone() for k, v in a.state_dict().items()}
    ra, oa, sa = make(a)
    steps(a, ra, oa, sa, 6)
    b = mk()
    b.load_state_dict(start)
    rb, ob, sb = make(b)
    steps(b, rb, ob, sb, 3)
    path = tmp_path / "ckpt.pt"
    torch.save({"optimizer": ob.state_dict(), "state_dict": b.state_dict(), "scheduler": sb.state_dict()}, path)
    ck = torch.load(path)
    c = mk()
    rc, oc, sc = make(c)
    c.load_state_dict(ck["state_dict"])
    oc.load_state_dict(ck["optimizer"])
    sc.load_state_dict(ck["scheduler"])
    for p in c.parameters():                      # still views of the optimizer's flat buffers
        assert oc.flat_p.data_ptr() <= p.data_ptr() < oc.flat_p.data_ptr() + oc.flat_p.numel() * 4
        assert rc.flat.data_ptr() <= p.grad.data_ptr() < rc.flat.data_ptr() + rc.flat.numel() * 4
    steps(c, rc, oc, sc, 3)
    assert oc.steps == oa.steps == 6 and sc.get_last_lr() == sa.get_last_lr()
    for pa, pc in zip(a.parameters(), c.parameters()):
        assert torch.equal(pa, pc)
    assert torch.equal(oa.exp_avg, oc.exp_avg) and torch.equal(oa.exp_avg_sq, oc.exp_avg_sq)


def test_flat_adamw_exchanges_checkpoints_with_torch_adamw():
    """The reference stores torch.optim.AdamW.state_dict() in its checkpoints (main.py:246-252) and resumes from it
    (utils/utils.py:394).  Both directions: a torch-written state resumes here, a state written here
    (state_dict(layout="torch")) resumes in torch; three more steps on either side must agree with the uninterrupted
    run.  Second learning-rate group (diff_lr) included; a foreign dict raises a clear error."""
    from jspsr_amd.ddp import GradReducer
    from jspsr_amd.optim import FlatAdamW
    torch.manual_seed(3)
    mk = lambda: torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.Conv2d(8, 8, 1), torch.nn.Conv2d(8, 5, 1)).cuda()
    x = torch.randn(4, 3, 9, 9, device="cuda")

    def torch_opt(net):
        return torch.optim.AdamW([{"params": list(net[0].parameters()) + list(net[2].parameters())},
                                  {"params": list(net[1].parameters()), "lr": 3e-4}], lr=1e-3, weight_decay=1e-6)

    def flat_opt(net):
        red = GradReducer(net.parameters())
        return red, FlatAdamW(red, lr=1e-3, weight_decay=1e-6, lr_overrides={p: 3e-4 for p in net[1].parameters()})

    def tsteps(net, opt, n):
        for _ in range(n):
            opt.zero_grad()
            net(x).square().mean().backward()
            opt.step()

    def fsteps(net, red, opt, n):
        for _ in range(n):
            opt.zero_grad()
            net(x).square().mean().backward()
            red.finish()
            opt.step()

    ref = mk()
    start = {k: v.clone() for k, v in ref.state_dict().items()}
    ropt = torch_opt(ref)
    tsteps(ref, ropt, 3)
    import copy
    mid_model, mid_opt = {k: v.clone() for k, v in ref.state_dict().items()}, copy.deepcopy(ropt.state_dict())    # (torch hands out its live tensors)
    tsteps(ref, ropt, 3)                                     # the uninterrupted run: 6 steps
    # torch checkpoint -> FlatAdamW
    a = mk()
    a.load_state_dict(mid_model)
    ra, oa = flat_opt(a)
    oa.load_state_dict(mid_opt)
    assert oa.steps == 3
    fsteps(a, ra, oa, 3)
    for pa, pr in zip(a.parameters(), ref.parameters()):
        assert torch.allclose(pa, pr, rtol=1e-5, atol=1e-7)
    # FlatAdamW checkpoint (torch layout) -> torch
    b = mk()
    b.load_state_dict(start)
    rb, ob = flat_opt(b)
    fsteps(b, rb, ob, 3)
    c = mk()
    c.load_state_dict(b.state_dict())
    copt = torch_opt(c)
    copt.load_state_dict(ob.state_dict(layout="torch"))
    tsteps(c, copt, 3)
    for pc, pr in zip(c.parameters(), ref.parameters()):
        assert torch.allclose(pc, pr, rtol=1e-5, atol=1e-7)
    with pytest.raises(ValueError, match="unknown optimizer checkpoint format|not an optimizer state dict"):
        oa.load_state_dict({"state": {}, "param_groups": [{"lr": 1e-3}]})
    with pytest.raises(ValueError, match="groups its parameters differently"):
        oa.load_state_dict(torch.optim.AdamW(mk().parameters()).state_dict())


def test_graphed_step_is_bit_identical_to_the_eager_step():
    """jspsr_amd.graph.GraphedStep: the training step captured in a hipGraph (forward on three streams, fused loss, backward
    with the weight gradients on their auxiliary streams, FlatAdamW with its scalars in device memory) replays to the
    SAME BITS as the eager step -- parameters, optimizer moments, BatchNorm buffers and the loss, after 3 eager + 4 replayed
    steps against 7 eager ones, with a learning-rate change in between (the schedule reaches the replayed launches through
    the device-side scalars) and a new batch copied into the static inputs."""
    from jspsr_amd.JSPSR import Model
    from jspsr_amd.ddp import GradReducer
    from jspsr_amd.graph import GraphedStep
    from jspsr_amd.losses import MultiLoss
    from jspsr_amd.optim import FlatAdamW
    from oracle import jspsr_ref as R
    ic = {"lr_dem": 1, "image": 3, "mask": 15}
    sd = R.make_state_dict(R.jspsr_param_shapes(ic, 8), seed=21)
    batches = []
    for s in range(2):
        inp, gt = R.synthetic_batch(2, 64, 64, True, seed=22 + s)
        batches.append(([t.cuda() for t in inp], gt.cuda()))

    def build(dtype):
        m = Model(dict(ic, COP30=1), num_feature=8)
        m.load_state_dict(sd)
        m = m.cuda().train()
        m.compute_dtype = dtype
        red = GradReducer(m.parameters())
        red.watch_streams(m.side_streams("cuda"))
        opt = FlatAdamW(red, lr=1e-3, weight_decay=1e-6)
        return m, red, opt, MultiLoss(1.0, 1.0, 0.1)

    for dtype in (torch.float32, torch.bfloat16):
        # eager: 7 steps, batch 0 for steps 1-5, batch 1 from step 6; lr halves from step 5 on
        m, red, opt, crit = build(dtype)
        losses_e = []
        for i in range(7):
            if i == 4:
                opt.lr = 5e-4
            inp, gt = batches[0 if i < 5 else 1]
            red.zero_grad()
            loss = crit(m(*inp), gt)["Total"]
            loss.backward()
            red.finish()
            opt.step()
            losses_e.append(loss.item())
        ref = {k: v.clone() for k, v in m.state_dict().items()}
        ref_m, ref_v = opt.exp_avg.clone(), opt.exp_avg_sq.clone()
        # graphed: 3 eager warm-up steps inside the constructor, then 4 replays
        m, red, opt, crit = build(dtype)
        step = GraphedStep(m, red, opt, crit, *batches[0], warmup=3)
        losses_g = []
        for i in range(3, 7):
            if i == 4:
                opt.lr = 5e-4
            loss = step(*batches[1]) if i == 5 else step()
            losses_g.append(loss.item())
        assert opt.steps == 7 and step.replays == 4
        assert losses_g == losses_e[3:], (dtype, losses_g, losses_e)
        for k, v in m.state_dict().items():
            assert torch.equal(v, ref[k]), (dtype, k)
        assert torch.equal(opt.exp_avg, ref_m) and torch.equal(opt.exp_avg_sq, ref_v)
        # and the module keeps working eagerly afterwards (packed-weight caches were invalidated)
        m.eval()
        with torch.no_grad():
            assert torch.isfinite(m(*batches[0][0])).all()
