"""Lab: what the GPU box gives the CPU oracle (visible CPUs, affinity, cgroup quota) and how the fp64 oracle step scales
with torch's thread count -- the fixture tests spend their time there (tests/fixtures.py::gradient_noise_floor)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import jspsr_ref as R  # noqa: E402

print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), flush=True)
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip(), flush=True)
MSK = {"lr_dem": 1, "image": 3, "mask": 15}
sd = R.make_state_dict(R.jspsr_param_shapes(MSK, 32), 3, torch.float64)
params = {k: v.requires_grad_() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
inp, gt = R.synthetic_batch(1, 64, 64, True, seed=4, dtype=torch.float64)
for n in [int(a) for a in sys.argv[1:]] or (128, 32, 16, 8):
    torch.set_num_threads(n)
    ts = []
    for _ in range(2):
        t = time.time()
        for v in params.values():
            v.grad = None
        R.jspsr_forward(sd, inp, True).mean().backward()
        ts.append(time.time() - t)
    print(f"threads {n:4d}: fp64 oracle fwd+bwd nf32 1x64x64: {min(ts):.2f} s", flush=True)
