"""Lab: which aten operators (torch's own small kernels) still run inside the benchmark step, and from where."""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from jspsr_amd.JSPSR import Model
from jspsr_amd.ddp import GradReducer
from jspsr_amd.losses import MultiLoss
from jspsr_amd.optim import FlatAdamW
from torch.utils._python_dispatch import TorchDispatchMode
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
counts = collections.Counter()
SKIP = ("aten.view", "aten.detach", "aten._unsafe_view", "aten.alias", "aten.t.", "aten.slice", "aten.select", "aten.permute",
        "aten.reshape", "aten.expand", "aten.as_strided", "aten.unsqueeze", "aten.squeeze", "aten.transpose", "aten.empty", "aten.record_stream",
        "aten.is_", "aten.sym_", "aten.split", "aten.unbind", "aten.narrow", "aten._local_scalar", "aten.lift_fresh", "aten.flatten")
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            where = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "jspsr_amd" in fr.filename or fr.filename.endswith("bench.py"):
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            counts[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Log():
    step()
torch.cuda.synchronize()
for (name, where), n in counts.most_common(45):
    print(f"{n:5d}  {name:34s} {where}")
