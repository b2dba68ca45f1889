"""Soak: N training steps of the benchmark configuration; prints loss, step time and allocator high-water marks every
25 steps (the multi-stream step defers block reuse through record_stream: memory must plateau)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from jspsr_amd.JSPSR import Model  # noqa: E402
from jspsr_amd.ddp import GradReducer  # noqa: E402
from jspsr_amd.losses import MultiLoss  # noqa: E402
from jspsr_amd.optim import FlatAdamW  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = Model(in_channels=bench.IN_CHANNELS, num_feature=32).to(dev).train()
    model.compute_dtype = torch.bfloat16
    red = GradReducer(model.parameters())
    red.watch_streams(model.side_streams(dev))
    opt = FlatAdamW(red, lr=1e-3, weight_decay=1e-6)
    crit = MultiLoss(1.0, 1.0, 0.1)
    inputs, gt = bench.synthetic_batch(bench.TILES_PER_GPU, bench.TILE, bench.TILE, dev, seed=1000)
    t0 = time.perf_counter()
    for i in range(1, n + 1):
        red.zero_grad()
        loss = crit(model(*inputs), gt)["Total"]
        loss.backward()
        red.finish()
        opt.step()
        if i % every == 0:
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / every
            t0 = time.perf_counter()
            print(f"step {i}: loss {loss.item():.6f}  {dt*1e3:.1f} ms/step  allocated {torch.cuda.max_memory_allocated()/2**30:.1f} GiB  "
                  f"reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB", flush=True)
            assert torch.isfinite(loss).item()


if __name__ == "__main__":
    main()
