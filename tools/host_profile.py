"""Lab: where the HOST spends its ~40 ms enqueuing one benchmark step (cProfile over 3 steps, top functions by own time)."""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from jspsr_amd.JSPSR import Model  # noqa: E402
from jspsr_amd.ddp import GradReducer  # noqa: E402
from jspsr_amd.losses import MultiLoss  # noqa: E402
from jspsr_amd.optim import FlatAdamW  # noqa: E402

dev = torch.device("cuda", 0)
np.random.seed(0)
model = Model(in_channels=bench.IN_CHANNELS, out_channels=1, num_feature=32).to(dev).train()
model.compute_dtype = torch.bfloat16
red = GradReducer(model.parameters())
red.watch_streams(model.side_streams(dev))
opt = FlatAdamW(red, lr=1e-3, weight_decay=1e-6)
crit = MultiLoss(1.0, 1.0, 0.1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
inputs, gt = bench.synthetic_batch(B, 512, 512, dev, seed=1000)


def step():
    red.zero_grad()
    crit(model(*inputs), gt)["Total"].backward()
    red.finish()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
