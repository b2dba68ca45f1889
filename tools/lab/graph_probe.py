"""Lab: which (width, raster) makes hipGraph capture of the training step fall over?  One child process per case."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from jspsr_amd.JSPSR import Model
from jspsr_amd.ddp import GradReducer
from jspsr_amd.graph import GraphedStep
from jspsr_amd.losses import MultiLoss
from jspsr_amd.optim import FlatAdamW
nf, B, H, dt, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
m = Model({"COP30": 1, "image": 3, "mask": 15, "lr_dem": 1}, num_feature=nf).cuda().train()
m.compute_dtype = torch.bfloat16 if dt == "bf16" else torch.float32
red = GradReducer(m.parameters()); red.watch_streams(m.side_streams("cuda"))
opt = FlatAdamW(red, lr=1e-3, weight_decay=1e-6); crit = MultiLoss(1.0, 1.0, 0.1)
g = torch.Generator(device="cuda").manual_seed(0)
inp = [torch.rand(B, c, H, H, device="cuda", generator=g) for c in (1, 3, 15)]
gt = torch.rand(B, 1, H, H, device="cuda", generator=g)
if mode != "default":
    import jspsr_amd.graph as G
    import torch.cuda
    orig = torch.cuda.graph
    class graph_(orig):
        def __init__(self, g_, **k):
            super().__init__(g_, capture_error_mode=mode, **k)
    torch.cuda.graph = graph_
step = GraphedStep(m, red, opt, crit, inp, gt)
for _ in range(3):
    l = step()
torch.cuda.synchronize()
print("ok loss", l.item(), "reserved GiB", round(torch.cuda.memory_reserved() / 2**30, 2))
''' % ROOT
cases = [a.split(",") for a in sys.argv[1:]] or [["8", "2", "64", "bf16", "default"]]
for c in cases:
    env = dict(os.environ)
    for kv in c[5:]:
        k, v = kv.split("=")
        env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD] + c[:5], capture_output=True, text=True, env=env, timeout=280)
    tail = [l for l in (r.stdout + r.stderr).splitlines() if l.strip() and "amdgpu.ids" not in l and "Warning" not in l and "run_backward" not in l]
    print(c, "rc", r.returncode, "|", (tail[-1] if tail else "")[:200], flush=True)
