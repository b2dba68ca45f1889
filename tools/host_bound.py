"""Is the training step host-bound?  Times the host-side enqueue of one step (no synchronisation inside) against the
synchronised step time, for the benchmark workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from jspsr_amd.JSPSR import Model
from jspsr_amd.ddp import GradReducer
from jspsr_amd.losses import MultiLoss
from jspsr_amd.optim import FlatAdamW

dev = torch.device("cuda", 0)
np.random.seed(0); torch.manual_seed(0)
model = Model(in_channels=bench.IN_CHANNELS, out_channels=1, num_feature=32).to(dev).train()
model.compute_dtype = torch.bfloat16
red = GradReducer(model.parameters()); red.watch_streams(model.side_streams(dev))
opt = FlatAdamW(red, lr=1e-3, weight_decay=1e-6)
crit = MultiLoss(1.0, 1.0, 0.1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
inputs, gt = bench.synthetic_batch(B, 512, 512, dev, seed=1000)

def step():
    red.zero_grad(); loss = crit(model(*inputs), gt)["Total"]; loss.backward(); red.finish(); opt.step()

for _ in range(3): step()
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B}: host enqueue {1e3 * (t1 - t0):.1f} ms, until GPU done {1e3 * (t2 - t0):.1f} ms", flush=True)
from jspsr_amd import ops
ops.RUN_AHEAD = 10 ** 9
for trial in range(2):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B} no run-ahead throttle: host enqueue {1e3 * (t1 - t0):.1f} ms, until GPU done {1e3 * (t2 - t0):.1f} ms", flush=True)
