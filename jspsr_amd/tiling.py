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
    H: int = 0   # scene height (0: unknown)

    def margins(self):
        """Rows of real neighbour data above / below the interior; an edge on the scene border needs none (the window's
        zero padding there IS the scene's)."""
        big = 1 << 30
        return (self.y0 - self.ty0 if self.ty0 > 0 else big), (self.ty1 - self.y1 if (self.H == 0 or self.ty1 < self.H) else big)


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
        out.append(Strip(y0, y1, ty0, ty0 + win, H))
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


RECEPTIVE_RADIUS = 97   # JSPSR: stem 2 + encoder 4+8+16+32 + decoder 16+8+4 + conv0 1 + generator 5 + 3x3 sampler 1;
                        # the certified figure is a per-model attribute (`Model.receptive_radius`), this is JSPSR's


class HaloTooSmall(RuntimeError):
    pass


def _combine_ranks(group=None):
    def f(s, cnt, mx):
        if s.shape[0] != 1:
            raise ValueError("cross-rank gate statistics are defined for one scene per call (batch size 1)")
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


def _run(model, tiles, windows, combine, check_reach=True):
    """-> (output, reach); reach[b] = max |learned offset| in pixels over the interior rows of window b.
    check_reach: raise HaloTooSmall unless reach + RECEPTIVE_RADIUS fits into the rows of real neighbour data the window
    holds beyond every interior edge -- the condition under which a strip equals the monolithic forward (SURVEY.md 8e:
    the offsets are unbounded reals, so exactness is checked per scene, not assumed)."""
    if model.training:
        raise RuntimeError("sharded inference needs model.eval(): BatchNorm batch statistics would couple the strips")
    radius = getattr(model, "receptive_radius", None)
    if check_reach and radius is None:
        raise RuntimeError(f"{type(model).__name__} has no certified receptive radius (Model.receptive_radius): the strip result "
                           "cannot be certified equal to the monolithic forward; pass check_reach=False to run it uncertified")
    prev, prev_probe = E._gate_sync, E._offset_probe
    E._gate_sync = _GateSync(windows, combine)
    E._offset_probe = probe = []
    try:
        with torch.no_grad():
            out = model(*tiles)
    finally:
        E._gate_sync, E._offset_probe = prev, prev_probe
    if check_reach and not probe:
        raise RuntimeError(f"{type(model).__name__} reported no learned offsets during the sharded forward: the halo check "
                           "would pass vacuously (every propagation step must append to engine._offset_probe)")
    chained = bool(getattr(model, "offsets_chain", False))   # steps applied to their own output: the reaches add up
    reach = []
    for b, s in enumerate(windows):
        r = 0.0
        for off in probe:                                   # NHWC (B,h,w,16) learned offsets of a propagation step
            rows = off[b, s.y0 - s.ty0:s.y1 - s.ty0]
            if rows.numel():
                m = rows.abs().max().item()
                r = r + m if chained else max(r, m)
        reach.append(r)
        margin = min(s.margins())
        if check_reach and r + radius > margin:
            raise HaloTooSmall(f"strip rows [{s.y0},{s.y1}): learned offsets reach {r:.1f} px; with the receptive radius of "
                               f"{radius} px that needs {r + radius:.0f} halo rows, the window holds {margin}")
    return out, reach


def sharded_forward(model, inputs: Sequence[torch.Tensor], rank: int, world: int, halo: int = 128, group=None,
                    check_reach=True, return_reach=False):
    """This rank's interior rows of model(*inputs) for a scene every rank can address
    (`inputs`: full-scene (B=1,C,H,W) tensors, e.g. memory-mapped; only the window is touched)."""
    if inputs[0].shape[0] != 1:
        raise ValueError("sharded_forward: one scene per call (batch size 1)")
    H = inputs[0].shape[2]
    s = plan_strips(H, world, halo)[rank]
    tiles = [t[:, :, s.ty0:s.ty1].contiguous().cuda() for t in inputs]
    out, reach = _run(model, tiles, [s], _combine_ranks(group), check_reach)
    out = out[:, :, s.y0 - s.ty0:s.y1 - s.ty0]
    return (out, reach[0]) if return_reach else out


def sharded_forward_owned(model, own: Sequence[torch.Tensor], scene_h: int, halo: int = 128, group=None,
                          check_reach=True, return_reach=False):
    """Config 5 as SURVEY.md 8e states it: every rank HOLDS ONLY ITS STRIP.  `own`: this rank's rows
    [y0, y1) of each input, (1,C,y1-y0,W) on the GPU.  The window plan is `plan_strips` (the same one the
    single-process emulation and `sharded_forward` use); the window's missing rows come from the neighbouring ranks
    by point-to-point send/recv (`exchange_window`), once per scene; the channel-gate statistics are all-reduced
    (SUM, MAX) at the four gated layers.  Returns this rank's interior rows of the scene's prediction."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    strips = plan_strips(scene_h, world, halo)
    s = strips[rank]
    if any(t.shape[0] != 1 or t.shape[2] != s.y1 - s.y0 for t in own):
        raise ValueError(f"sharded_forward_owned: rank {rank} owns rows [{s.y0},{s.y1}) -- got {[tuple(t.shape) for t in own]}")
    tiles = [exchange_window(t.cuda(), strips, group) for t in own]
    out, reach = _run(model, tiles, [s], _combine_ranks(group), check_reach)
    out = out[:, :, s.y0 - s.ty0:s.y1 - s.ty0]
    return (out, reach[0]) if return_reach else out


def emulate_sharded_forward(model, inputs: Sequence[torch.Tensor], world: int, halo: int = 128, check_reach=True,
                            return_reach=False):
    """Same algorithm in ONE process: the strips are stacked along the batch axis and the
    cross-rank reductions become reductions over that axis.  Used to prove exactness against the
    monolithic forward on a single GPU (tests/test_tiling_gpu.py)."""
    assert inputs[0].shape[0] == 1
    H = inputs[0].shape[2]
    strips = plan_strips(H, world, halo)
    tiles = [torch.cat([t[:, :, s.ty0:s.ty1] for s in strips]).contiguous() for t in inputs]
    out, reach = _run(model, tiles, strips, _combine_batch, check_reach)
    out = torch.cat([out[i:i + 1, :, s.y0 - s.ty0:s.y1 - s.ty0] for i, s in enumerate(strips)], 2)
    return (out, reach) if return_reach else out


def exchange_window(strip: torch.Tensor, strips: Sequence[Strip], group=None) -> torch.Tensor:
    """Each rank holds only its own rows (B,C,y1-y0,W); returns its window [ty0, ty1) of `plan_strips`: the rows it
    lacks arrive from the previous / next rank.  Windows clamped at a scene border reach further into the one
    neighbour they have (up to 2 x halo rows), so the sizes differ per edge; every rank derives all of them from the
    shared plan.  Neighbour send/recv (xGMI links are point-to-point), one exchange per scene."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    s = strips[rank]
    need_top, need_bot = s.y0 - s.ty0, s.ty1 - s.y1

    def rows_needed_by(r):     # (rows rank r needs from r-1, rows it needs from r+1)
        t = strips[r]
        return t.y0 - t.ty0, t.ty1 - t.y1

    ops, top, bot = [], None, None
    B, C, rows, W = strip.shape
    if rank > 0:
        give = rows_needed_by(rank - 1)[1]                 # what the previous rank needs from the top of mine
        if give > rows or need_top > strips[rank - 1].y1 - strips[rank - 1].y0:
            raise ValueError("exchange_window: a window spans more than one neighbour; use fewer ranks or a smaller halo")
        if give:
            ops.append(dist.P2POp(dist.isend, strip[:, :, :give].contiguous(), rank - 1, group))
        if need_top:
            top = strip.new_empty((B, C, need_top, W))
            ops.append(dist.P2POp(dist.irecv, top, rank - 1, group))
    if rank < world - 1:
        give = rows_needed_by(rank + 1)[0]
        if give > rows or need_bot > strips[rank + 1].y1 - strips[rank + 1].y0:
            raise ValueError("exchange_window: a window spans more than one neighbour; use fewer ranks or a smaller halo")
        if give:
            ops.append(dist.P2POp(dist.isend, strip[:, :, rows - give:].contiguous(), rank + 1, group))
        if need_bot:
            bot = strip.new_empty((B, C, need_bot, W))
            ops.append(dist.P2POp(dist.irecv, bot, rank + 1, group))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    return torch.cat([t for t in (top, strip, bot) if t is not None], 2)


def exchange_halo(strip: torch.Tensor, halo: int, group=None) -> torch.Tensor:
    """Symmetric special case: up to `halo` rows from each neighbour (fewer at the scene borders)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    rows = strip.shape[2]
    if rows < halo:
        raise ValueError("strip shorter than the halo: use fewer ranks or a smaller halo")
    H = rows * world
    strips = [Strip(r * rows, (r + 1) * rows, max(0, r * rows - halo), min(H, (r + 1) * rows + halo), H) for r in range(world)]
    return exchange_window(strip, strips, group)
