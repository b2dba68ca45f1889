"""BASELINE config 5 on one GPU: eval-mode forward of the image+mask JSPSR on ONE rank's strip of a 4096 x 4096
scene split over 8 ranks (512 interior rows + 128-row halos = what a rank computes), Mpixel/s of INTERIOR pixels.
Usage: python tools/bench_scene.py [scene=4096] [world=8] [rank=3] [bf16|f32]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jspsr_amd.JSPSR import Model  # noqa: E402
from jspsr_amd import tiling  # noqa: E402


def main():
    a = sys.argv[1:]
    S = int(a[0]) if len(a) > 0 else 4096
    world = int(a[1]) if len(a) > 1 else 8
    rank = int(a[2]) if len(a) > 2 else 3
    dt = torch.float32 if len(a) > 3 and a[3] == "f32" else torch.bfloat16
    torch.manual_seed(0)
    model = Model({"COP30": 1, "image": 3, "mask": 15, "lr_dem": 1}).cuda().eval()
    model.compute_dtype = dt
    s = tiling.plan_strips(S, world, 128)[rank]
    g = torch.Generator().manual_seed(1)
    rows = s.ty1 - s.ty0
    tiles = [torch.rand(1, c, rows, S, generator=g).cuda() for c in (1, 3, 15)]
    run = lambda: tiling._run(model, tiles, [s], tiling._combine_batch)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        out = run()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / n
    interior = (s.y1 - s.y0) * S
    print(f"scene {S}x{S} / {world} ranks, rank {rank}: window rows {rows} (interior {s.y1 - s.y0}), {str(dt)[6:]} "
          f"eval forward {t*1e3:.1f} ms -> {interior/t/1e6:.1f} Mpixel/s interior per GPU "
          f"({rows*S/t/1e6:.1f} computed), x{world} = {world*interior/t/1e6:.0f} Mpixel/s scene rate if ranks overlap fully")


if __name__ == "__main__":
    main()
