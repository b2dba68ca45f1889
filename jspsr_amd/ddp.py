"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI (backend "nccl" on ROCm).

The reference is single-GPU; this is the new exchange step of BASELINE config 4.
Design for xGMI (point-to-point links, ring collectives per-link bound): few, large buckets
(default 64 MiB) carved from ONE flat fp32 buffer that the parameters' .grad tensors alias, so
autograd accumulates straight into the communication buffer (no copy in, no copy out); buckets
are filled in reverse parameter order -- the order backward produces gradients -- and each
bucket's all-reduce is issued asynchronously from an autograd hook the moment its last
gradient lands, overlapping the remaining backward.  BatchNorm statistics stay per replica
(plain data-parallel semantics, matching the reference's numerics at the per-GPU batch); the running statistics are
averaged over the replicas on demand (`sync_buffers`, `GradReducer.sync_buffers`: before a checkpoint / evaluation).
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, params, bucket_bytes: int = 64 << 20, process_group=None, world=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = process_group
        # `world` overrides the process group's size (tests drive the bucket bookkeeping without a second rank)
        self.world = world if world is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)
        total = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        if self.world > 1 and p0.is_cuda and dist.is_initialized():
            # RCCL's all-reduce kernels hold compute units beside the backward pass; the persistent register-resident conv
            # kernels need whole CUs: let their late workgroups find an empty tile queue instead of a full static share
            from . import _lib
            _lib.load().jspsr_conv_dynamic_queue(1)
        self.flat = torch.zeros(total, dtype=p0.dtype, device=p0.device)
        # reverse order = gradient production order; contiguous bucket ranges in the flat buffer
        self.buckets = []  # (start, end, n_params)
        self._bucket_of = {}
        start = off = 0
        count = 0
        per = max(1, bucket_bytes // self.flat.element_size())
        offsets = {}
        for p in reversed(self.params):
            n = p.numel()
            offsets[id(p)] = off
            p.grad = self.flat[off:off + n].view_as(p)
            self._bucket_of[p] = len(self.buckets)
            off += n
            count += 1
            if off - start >= per:
                self.buckets.append((start, off, count))
                start, count = off, 0
        if count:
            self.buckets.append((start, off, count))
        self._offsets = [offsets[id(p)] for p in self.params]
        self._pending = [b[2] for b in self.buckets]    # a backward before the first zero_grad() counts correctly too
        self._reduced = [0] * len(self.buckets)
        self._seen = set()
        self._direct = set()
        self._works = []
        self._streams, self._home = [], None
        if self.world > 1:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)
        # conv weights and BatchNorm scale/shift: the gradient kernels add straight into the flat buffer
        # (ops._wgrad_into, ops._bn_sink) and report readiness here instead of through autograd's accumulation
        # hook; operators that do not know the protocol (biases, propagation parameters) keep returning their
        # gradients to autograd
        for p in self.params:
            p._jspsr_direct_grad = True
            p._jspsr_grad_ready = self._direct_ready

    def sync_buffers(self, module: torch.nn.Module):
        """Average the module's BatchNorm running statistics over this reducer's process group (module-level
        `sync_buffers`); call before writing a checkpoint or evaluating, so that what rank 0 saves is the job's, not its own."""
        return sync_buffers(module, self.group)

    def attach(self, module: torch.nn.Module):
        """Let `module.zero_grad(...)` -- what the reference's loop calls every iteration with set_to_none=True
        (train/train_utils.py:210) -- zero this reducer's flat buffer and keep the gradient aliases, instead of
        dropping them (jspsr_amd.JSPSR.Model / LRRU.Model / EDSR honour `_grad_reducer`)."""
        module.__dict__["_grad_reducer"] = self
        return self

    def _direct_ready(self, p):
        """Readiness report of a kernel that added its gradient straight into the flat buffer.  It is issued when the
        FIRST contribution is enqueued; a layer applied twice in one forward would have its bucket reduced before the
        second contribution lands, so a second direct report for the same parameter in one step is an error."""
        if id(p) in self._direct:
            if self.world > 1:
                raise RuntimeError("GradReducer: a layer with in-place weight gradients was applied more than once in one "
                                   "forward; shared layers are not supported by the direct-gradient path under data "
                                   "parallelism")
            return
        self._direct.add(id(p))
        if self.world > 1:
            self._hook(p)

    def watch_streams(self, streams):
        """Side streams whose backward kernels write gradients into the flat buffer (Model.side_streams()): a
        bucket's all-reduce is ordered after all of them, not only after the stream its last gradient came from."""
        self._streams = list(streams)

    def _join_streams(self):
        if self.flat.is_cuda:
            from . import ops
            cur = torch.cuda.current_stream()
            for st in self._streams + ops.aux_streams() + ([self._home] if self._home is not None else []):
                if st != cur:
                    cur.wait_stream(st)

    def zero_grad(self):
        """Gradients alias the flat buffer: zero it in one kernel, keep the aliases."""
        self._home = torch.cuda.current_stream() if self.flat.is_cuda else None
        self.flat.zero_()
        base, es = self.flat.data_ptr(), self.flat.element_size()
        for p, off in zip(self.params, self._offsets):   # aliases dropped or replaced (zero_grad(set_to_none=True)): restore
            g = p.grad
            if g is None or g.data_ptr() != base + off * es:
                p.grad = self.flat[off:off + p.numel()].view_as(p)
        self._pending = [b[2] for b in self.buckets]
        self._reduced = [0] * len(self.buckets)
        self._seen = set()
        self._direct = set()
        self._works = []

    def _check_aliases(self):
        base, es = self.flat.data_ptr(), self.flat.element_size()
        for p, off in zip(self.params, self._offsets):
            g = p.grad
            if g is None or g.data_ptr() != base + off * es or g.shape != p.shape:
                raise RuntimeError(
                    "GradReducer: a parameter's .grad no longer aliases the flat gradient buffer (was "
                    "model.zero_grad(set_to_none=True) or optimizer.zero_grad() of another optimizer called after "
                    "reducer.zero_grad()?).  Use reducer.zero_grad() / FlatAdamW.zero_grad(), or reducer.attach(model) "
                    "so that model.zero_grad() does the right thing.")

    def _hook(self, p):
        # A parameter reports once per step.  Kernels that write a gradient straight into the flat buffer report by
        # hand (ops._wgrad_into / _bn_ready); depending on the PyTorch version autograd's post-accumulate hook fires
        # for such a parameter as well (its node runs with an undefined gradient) -- the second report is dropped.
        if id(p) in self._seen:
            return
        self._seen.add(id(p))
        i = self._bucket_of[p]
        self._pending[i] -= 1
        if self._pending[i] == 0:
            s, e, _ = self.buckets[i]
            self._join_streams()
            self._reduced[i] += 1
            self._works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Wait for the in-flight bucket reductions and average.  Also the point where the calling stream is ordered
        after the streams gradient kernels were launched on (branch streams, weight-gradient streams): call it after
        backward() and before the optimizer step, single-GPU runs included."""
        self._join_streams()
        self._check_aliases()
        if self.world == 1:
            return
        for i, left in enumerate(self._pending):  # parameters that received no gradient this step
            if left > 0:
                s, e, _ = self.buckets[i]
                self._reduced[i] += 1
                self._works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                self._pending[i] = 0
        if any(left < 0 for left in self._pending) or any(n != 1 for n in self._reduced):
            raise RuntimeError(f"GradReducer: bucket bookkeeping broken (pending {self._pending}, reductions per bucket "
                               f"{self._reduced}): every bucket must be reduced exactly once per step -- call "
                               "reducer.zero_grad() before every backward pass")
        for w in self._works:
            w.wait()
        self._works = []
        self.flat.div_(self.world)
        self._reduced = [0] * len(self.buckets)      # a second finish() without a new step is an error too


def sync_buffers(module: torch.nn.Module, group=None):
    """Make every replica's BatchNorm running statistics the MEAN over the replicas ("running stats kept in
    sync").  Training keeps per-replica batch statistics (plain data-parallel semantics, the reference's numerics at the
    per-GPU batch), so after the initial broadcast the running_mean / running_var of the replicas drift apart and a
    checkpoint written by rank 0 would carry rank 0's only.  Call before a checkpoint is written and before evaluation:
    ONE all-reduce over a flat copy of all floating-point buffers (a few hundred KB), integer buffers
    (num_batches_tracked: identical on every replica by construction) are checked, not averaged."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    fl = [b for b in module.buffers() if b.is_floating_point()]
    ints = [b for b in module.buffers() if not b.is_floating_point()]
    if fl:
        flat = torch.cat([b.detach().reshape(-1).float() for b in fl])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        off = 0
        for b in fl:
            n = b.numel()
            b.data.copy_(flat[off:off + n].view_as(b))
            off += n
    if ints:
        mine = torch.stack([b.detach().reshape(-1)[0].to(torch.int64) for b in ints])
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        if not torch.equal(lo, hi):
            raise RuntimeError("sync_buffers: the replicas disagree on an integer buffer (num_batches_tracked): they did not "
                               "run the same number of training steps")
    return len(fl)


def broadcast_module(module: torch.nn.Module, src: int = 0, group=None):
    """Start every replica from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src, group=group)
    from . import ops
    ops.invalidate_packed_weights()          # written through .data: torch's version counter did not move
