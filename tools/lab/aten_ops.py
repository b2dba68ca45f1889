"""Lab: which aten operators (torch's own small kernels) still run inside the benchmark step, by call count."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from jspsr_amd.JSPSR import Model
from jspsr_amd.ddp import GradReducer
from jspsr_amd.losses import MultiLoss
from jspsr_amd.optim import FlatAdamW
dev = torch.device("cuda", 0)
model = Model(in_channels=bench.IN_CHANNELS, num_feature=32).to(dev).train()
model.compute_dtype = torch.bfloat16
red = GradReducer(model.parameters()); red.watch_streams(model.side_streams(dev))
opt = FlatAdamW(red, lr=1e-3, weight_decay=1e-6); crit = MultiLoss(1.0, 1.0, 0.1)
inputs, gt = bench.synthetic_batch(bench.TILES_PER_GPU, bench.TILE, bench.TILE, dev, seed=1000)
def step():
    red.zero_grad(); crit(model(*inputs), gt)["Total"].backward(); red.finish(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages(group_by_stack_n=4) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda r: -r[1])
for e in sorted(prof.key_averages(group_by_stack_n=4), key=lambda e: -e.count)[:40]:
    if e.key.startswith("aten::") and e.device_time_total > 0:
        print(f"{e.key:28s} n={e.count:4d} dev={e.device_time_total:9.1f}us  {' <- '.join(s.split('/')[-1] for s in e.stack[:3])}")
