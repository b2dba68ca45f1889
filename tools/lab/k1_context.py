"""Lab: does the planar K1 backward's launch time depend on what else the process has set up?  tools/k1_lab.py's measurement
(20 back-to-back launches between events, operand sets rotating over 700 MB) repeated (a) in a bare process, (b) after
creating six more streams, (c) after building the benchmark model, (d) after one training step, (e) after 10 more."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jspsr_amd import ops  # noqa: E402

B, H, W = 8, 512, 512
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
dem = torch.rand(B, 1, H, W, device=dev, generator=g)
nset = max(2, int(700e6 // (B * H * W * 4 * 26)) + 1)
sets = [(torch.sigmoid(torch.randn(B, 9, H, W, device=dev, generator=g)), 1.5 * torch.randn(B, 16, H, W, device=dev, generator=g)) for _ in range(nset)]
gsets = [(torch.empty(B, 9, H, W, device=dev), torch.empty(B, 16, H, W, device=dev)) for _ in range(nset)]
w, b = torch.ones(1, 1, 3, 3, device=dev), torch.zeros(1, device=dev)
gout = torch.randn(B, 1, H, W, device=dev, generator=g)
out = torch.empty_like(dem)
ws = ops.prop_backward_workspace(B, H, W, dev)
fwd = lambda i: ops.prop_forward_raw(dem, sets[i % nset][0], sets[i % nset][1], w, b, 1.0, out)
bwd = lambda i: ops.prop_backward_raw(gout, dem, sets[i % nset][0], sets[i % nset][1], w, gsets[i % nset][0], gsets[i % nset][1], None, None, ws)


def measure(tag):
    res = []
    for fn in (fwd, bwd):
        v = []
        for rep in range(3):
            for i in range(3):
                fn(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(20):
                fn(i)
            e1.record()
            e1.synchronize()
            v.append(e0.elapsed_time(e1) / 20 * 1e3)
        res.append(" ".join(f"{x:6.1f}" for x in v))
    print(f"{tag:34s} fwd us {res[0]} | bwd us {res[1]}", flush=True)


measure("(a) bare process")
streams = [torch.cuda.Stream() for _ in range(6)]
for s_ in streams:
    with torch.cuda.stream(s_):
        torch.zeros(1, device=dev)
torch.cuda.synchronize()
measure("(b) + six streams")
sys.argv = ["bench.py"]
import bench  # noqa: E402
from jspsr_amd.JSPSR import Model  # noqa: E402
from jspsr_amd.ddp import GradReducer, broadcast_module  # noqa: E402
from jspsr_amd.losses import MultiLoss  # noqa: E402
from jspsr_amd.optim import FlatAdamW  # noqa: E402
model = Model(in_channels=bench.IN_CHANNELS, out_channels=1, num_feature=32, layers=(2, 2, 2, 2), spn=True).to(dev).train()
model.compute_dtype = torch.bfloat16
broadcast_module(model)
reducer = GradReducer(model.parameters())
reducer.watch_streams(model.side_streams(dev))
opt = FlatAdamW(reducer, lr=1e-3, weight_decay=1e-6)
crit = MultiLoss(1.0, 1.0, 0.1)
inputs, gt = bench.synthetic_batch(8, 512, 512, dev, seed=1000)
torch.cuda.synchronize()
measure("(c) + model built")


def step():
    reducer.zero_grad()
    loss = crit(model(*inputs), gt)["Total"]
    loss.backward()
    reducer.finish()
    opt.step()


step()
torch.cuda.synchronize()
measure("(d) + one training step")
for _ in range(10):
    step()
torch.cuda.synchronize()
measure("(e) + ten more steps")
time.sleep(3)
measure("(f) after 3 s idle")
torch.cuda.empty_cache()
measure("(g) after empty_cache()")


def measure_w(tag, warm):
    res = []
    for fn in (fwd, bwd):
        v = []
        for rep in range(3):
            time.sleep(1.0)
            for i in range(warm):
                fn(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(20):
                fn(i)
            e1.record()
            e1.synchronize()
            v.append(e0.elapsed_time(e1) / 20 * 1e3)
        res.append(" ".join(f"{x:6.1f}" for x in v))
    print(f"{tag:34s} fwd us {res[0]} | bwd us {res[1]}", flush=True)


for warm in (3, 30, 100, 300, 1000, 3000):
    measure_w(f"(h) 1 s idle, then {warm} warm-up launches", warm)
