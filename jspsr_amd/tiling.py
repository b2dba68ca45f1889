"""Whole-scene inference sharded over GPUs (BASELINE config 5).

The raster is cut into horizontal strips, one per rank.  Each rank runs the unmodified model on its
strip plus a halo (default 128 px >= receptive radius ~97 px + learned offset reach) and keeps the
strip's interior.  Two things couple the strips:

* **halo rows** of the inputs -- exchanged once per scene with the neighbouring ranks
  (`exchange_halo`, point-to-point: xGMI links are point-to-point, so neighbour send/recv is the
  natural pattern, not a collective);
* **ChannelAttention statistics** -- the global average / max pool of `resnet_cbam.py:39-40,50-51`
  spans the whole scene: each rank pools its *interior* rows only, then one tiny all-reduce (SUM of
  sums and counts, MAX of maxima) per gated layer makes every rank use the scene-wide statistics
  (4 sync points of a few kB: latency-bound, ~10-20 us each over xGMI).

BatchNorm runs in eval mode (per-channel affine): no coupling.  Zero padding applies only at true
scene borders: windows are clamped inside the scene, so interior tile edges compute on real halo
data and are cropped.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import engine as E
from . import kernels as K


@dataclass
class Strip:
    y0: int      # interior rows [y0, y1) of the scene owned by this rank
    y1: int
    ty0: int     # window rows [ty0, ty1) actually fed to the model (interior + halo, clamped)
    ty1: int


def plan_strips(H: int, world: int, halo: int = 128) -> List[Strip]:
    """Equal strips (multiples of 8 rows) with equal-height windows clamped inside the scene."""
    if H % 8 or halo % 8:
        raise ValueError("scene height and halo must be multiples of 8 (three stride-2 stages)")
    rows = -(-(H // 8) // world) * 8
    win = min(H, rows + 2 * halo)
    out = []
    for r in range(world):
        y0, y1 = min(H, r * rows), min(H, (r + 1) * rows)
        ty0 = max(0, min(y0 - halo, H - win))
        out.append(Strip(y0, y1, ty0, ty0 + win))
    return out


class _GateSync:
    """Replaces the per-tensor pooling of the channel gate by scene-wide statistics."""

    def __init__(self, windows: Sequence[Strip], combine: Callable):
        self.windows = windows      # one per batch entry of the tensors flowing through the model
        self.combine = combine      # (sum[B,C], count[B], max[B,C]) -> (avg[B,C], max[B,C])

    def pool(self, x: torch.Tensor):
        B, h, w, C = x.shape
        sums, cnts, maxs = [], [], []
        for b, s in enumerate(self.windows):
            f = (s.ty1 - s.ty0) // h                      # down-sampling factor of this level
            r0, r1 = (s.y0 - s.ty0) // f, (s.y1 - s.ty0) // f
            if r1 > r0:
                avg, mx, _ = K.gate_pool(x[b:b + 1, r0:r1].contiguous())
                n = (r1 - r0) * w
                sums.append(avg * n)
                maxs.append(mx)
            else:                                          # a rank whose strip is empty
                sums.append(torch.zeros(1, C, device=x.device))
                maxs.append(torch.full((1, C), float("-inf"), device=x.device))
                n = 0
            cnts.append(n)
        cnt = torch.tensor(cnts, dtype=torch.float32, device=x.device)
        return self.combine(torch.cat(sums), cnt, torch.cat(maxs))


def _combine_ranks(group=None):
    def f(s, cnt, mx):
        buf = torch.cat((s.flatten(), cnt))
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
        n = s.numel()
        return buf[:n].view_as(s) / buf[n:].sum(), mx
    return f


def _combine_batch(s, cnt, mx):
    """Single-process emulation: the batch entries ARE the strips of one scene."""
    B = s.shape[0]
    return (s.sum(0, keepdim=True) / cnt.sum()).expand(B, -1).contiguous(), mx.amax(0, keepdim=True).expand(B, -1).contiguous()


def _run(model, tiles, windows, combine):
    if model.training:
        raise RuntimeError("sharded inference needs model.eval(): BatchNorm batch statistics would couple the strips")
    prev = E._gate_sync
    E._gate_sync = _GateSync(windows, combine)
    try:
        with torch.no_grad():
            return model(*tiles)
    finally:
        E._gate_sync = prev


def sharded_forward(model, inputs: Sequence[torch.Tensor], rank: int, world: int, halo: int = 128, group=None):
    """This rank's interior rows of model(*inputs) for a scene every rank can address
    (`inputs`: full-scene (B=1,C,H,W) tensors, e.g. memory-mapped; only the window is touched)."""
    H = inputs[0].shape[2]
    s = plan_strips(H, world, halo)[rank]
    tiles = [t[:, :, s.ty0:s.ty1].contiguous().cuda() for t in inputs]
    out = _run(model, tiles, [s], _combine_ranks(group))
    return out[:, :, s.y0 - s.ty0:s.y1 - s.ty0]


def emulate_sharded_forward(model, inputs: Sequence[torch.Tensor], world: int, halo: int = 128):
    """Same algorithm in ONE process: the strips are stacked along the batch axis and the
    cross-rank reductions become reductions over that axis.  Used to prove exactness against the
    monolithic forward on a single GPU (tests/test_tiling_gpu.py)."""
    assert inputs[0].shape[0] == 1
    H = inputs[0].shape[2]
    strips = plan_strips(H, world, halo)
    tiles = [torch.cat([t[:, :, s.ty0:s.ty1] for s in strips]).contiguous() for t in inputs]
    out = _run(model, tiles, strips, _combine_batch)
    return torch.cat([out[i:i + 1, :, s.y0 - s.ty0:s.y1 - s.ty0] for i, s in enumerate(strips)], 2)


def exchange_halo(strip: torch.Tensor, halo: int, group=None) -> torch.Tensor:
    """Each rank holds only its own rows (B,C,rows,W); returns them with up to `halo` rows from the
    previous / next rank attached (fewer at the scene borders).  Neighbour send/recv, one exchange per
    scene.  Requires rows >= halo (one neighbour per side)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if strip.shape[2] < halo:
        raise ValueError("strip shorter than the halo: use fewer ranks or a smaller halo")
    ops, top, bot = [], None, None
    if rank > 0:
        top = torch.empty_like(strip[:, :, :halo])
        ops += [dist.P2POp(dist.isend, strip[:, :, :halo].contiguous(), rank - 1, group),
                dist.P2POp(dist.irecv, top, rank - 1, group)]
    if rank < world - 1:
        bot = torch.empty_like(strip[:, :, :halo])
        ops += [dist.P2POp(dist.isend, strip[:, :, -halo:].contiguous(), rank + 1, group),
                dist.P2POp(dist.irecv, bot, rank + 1, group)]
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    return torch.cat([t for t in (top, strip, bot) if t is not None], 2)
