"""Lab: shapes of the aten::add / add_ kernels autograd still runs in the benchmark step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from jspsr_amd.JSPSR import Model
from jspsr_amd.ddp import GradReducer
from jspsr_amd.losses import MultiLoss
from jspsr_amd.optim import FlatAdamW
from torch.profiler import profile, ProfilerActivity
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::add", "aten::add_", "aten::copy_", "aten::fill_", "aten::zeros", "aten::mul", "aten::sum")]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:30]:
    print(f"{e.key:12s} n={e.count:3d} dev={e.device_time_total:8.1f}us  {str(e.input_shapes)[:110]}")
