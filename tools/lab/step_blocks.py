"""Lab: does the eager benchmark step settle?  The headline loop of bench.py (same model, data, step) timed in consecutive blocks of
10.  bench.py's own line on this pool read 62-64 ms while the same eager loop in its graph-leg child read 59-60: this loop reads
60.0 in every block; bench.py 62.4 over the 20 steps behind 5 warm-up steps, 61.0 over 60, 59.5 behind 25 warm-up steps -- one
full pass of the interpreter's cyclic collector (tens of ms with the host not enqueueing, the GPU's queue running dry) falls
into bench.py's timed steps and into this script's warm-up.  bench.py now collects and freezes at the step boundary before the
timed region (62.9 / 62.2 -> 60.3 / 60.8 ms, interleaved on one box).  Arguments: [2-tile steps first] [keep: hold the last loss]."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from jspsr_amd import _lib  # noqa: E402
from jspsr_amd.JSPSR import Model  # noqa: E402
from jspsr_amd.ddp import GradReducer  # noqa: E402
from jspsr_amd.losses import MultiLoss  # noqa: E402
from jspsr_amd.optim import FlatAdamW  # noqa: E402

_lib.load()
device = torch.device("cuda", 0)
torch.manual_seed(0)
model = Model(in_channels=bench.IN_CHANNELS, out_channels=1, num_feature=32, layers=(2, 2, 2, 2), spn=True).to(device).train()
model.compute_dtype = torch.bfloat16
reducer = GradReducer(model.parameters())
reducer.watch_streams(model.side_streams(device))
opt = FlatAdamW(reducer, lr=1e-3, weight_decay=1e-6)
criterion = MultiLoss(1.0, 1.0, 0.1)
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 0          # 2-tile steps first, as the graph child does
keep = len(sys.argv) > 2 and sys.argv[2] == "keep"
if pre:
    i2, g2 = bench.synthetic_batch(2, bench.TILE, bench.TILE, device, seed=3002)
    for _ in range(pre):
        reducer.zero_grad(); criterion(model(*i2), g2)["Total"].backward(); reducer.finish(); opt.step()
    del i2, g2
    criterion.reset()
    torch.cuda.empty_cache()
inputs, gt = bench.synthetic_batch(8, bench.TILE, bench.TILE, device, seed=1000)


def step():
    reducer.zero_grad()
    loss = criterion(model(*inputs), gt)["Total"]
    loss.backward()
    reducer.finish()
    opt.step()
    return loss


for _ in range(5):
    step()
out = []
for blk in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        if keep:
            loss = step()          # bench.py's form: the previous step's loss (and what its graph still references) lives through the next step
        else:
            step()
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 10 * 1e3)
print("pre", pre, "keep", keep, "ms/step per block of 10:", " ".join(f"{x:.1f}" for x in out), "reserved GB", round(torch.cuda.memory_reserved() / 2**30, 2), flush=True)
