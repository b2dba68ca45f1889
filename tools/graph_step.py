"""Lab: eager vs hipGraph-replayed training step (jspsr_amd.graph.GraphedStep) -- ms/step and host ms/step at B tiles of 512^2."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from jspsr_amd.JSPSR import Model  # noqa: E402
from jspsr_amd.ddp import GradReducer  # noqa: E402
from jspsr_amd.graph import GraphedStep  # noqa: E402
from jspsr_amd.losses import MultiLoss  # noqa: E402
from jspsr_amd.optim import FlatAdamW  # noqa: E402

dev = torch.device("cuda", 0)
for B in [int(a) for a in sys.argv[1:]] or (2, 8):
    np.random.seed(0)
    model = Model(in_channels=bench.IN_CHANNELS, out_channels=1, num_feature=32).to(dev).train()
    model.compute_dtype = torch.bfloat16
    red = GradReducer(model.parameters())
    red.watch_streams(model.side_streams(dev))
    opt = FlatAdamW(red, lr=1e-3, weight_decay=1e-6)
    crit = MultiLoss(1.0, 1.0, 0.1)
    inputs, gt = bench.synthetic_batch(B, 512, 512, dev, seed=1000)

    def eager():
        red.zero_grad()
        loss = crit(model(*inputs), gt)["Total"]
        loss.backward()
        red.finish()
        opt.step()
        return loss

    def timed(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        host = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, host * 1e3

    e_ms, e_host = timed(eager)
    print(f"{B} tiles: eager {e_ms:.2f} ms/step (host enqueue {e_host:.2f}); capturing ...", flush=True)
    t0 = time.perf_counter()
    step = GraphedStep(model, red, opt, crit, inputs, gt)
    t_cap = time.perf_counter() - t0
    print(f"  captured in {t_cap:.1f} s", flush=True)
    g_ms, g_host = timed(step)
    print(f"{B} tiles: eager {e_ms:.2f} ms/step (host enqueue {e_host:.2f}); graph {g_ms:.2f} ms/step (host {g_host:.2f}); "
          f"capture + 3 warm-up steps {t_cap:.1f} s; loss {step().item():.5f}; reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB", flush=True)
    del step, model, red, opt
    torch.cuda.empty_cache()
