"""One training step captured in a hipGraph and replayed (round 4).

The reference's loop body (train/train_utils.py:205-219: zero_grad, forward, MultiLoss, backward, optimizer step) costs the
host ~40 ms to ENQUEUE through Python, autograd and ctypes -- ~560 C-ABI calls, ~1 400 launches on seven streams -- while
the GPU needs 62 ms for it at 8 tiles of 512 x 512 and 21 ms at 2 tiles.  For a static-shape step all of that host work is
the same every iteration, so it is done ONCE: the step is captured (torch.cuda.CUDAGraph = hipStreamBeginCapture over the
calling stream; the side streams of the guidance branches and of the weight gradients fork from and join back into it by
events, so their concurrency is part of the graph) and every later step is one hipGraphLaunch.

What had to change for the capture to be legal and correct:
  * FlatAdamW's scalars (learning rate, bias corrections) live in device memory (jspsr_adamw_step_dev) and are refreshed
    by an ordinary copy before each replay -- a captured launch carries its kernel arguments verbatim;
  * the run-ahead throttle of the weight-gradient streams (a host-side event wait) is skipped under capture: a graph has a
    fixed memory plan (its own allocator pool) and no host to throttle;
  * the BatchNorm step counters, the packed weight copies and the loss bookkeeping are device work and simply part of the
    graph; Python-side state that a replay does not run (the optimizer's step count, the packed-weights epoch) is advanced
    by `GraphedStep.__call__`.
Single process only: a captured step contains no collective (data-parallel runs keep the eager step).
"""
from __future__ import annotations

import torch

from . import ops


class GraphedStep:
    """step = GraphedStep(model, reducer, optimizer, criterion, inputs, target); loss = step() replays the captured step on
    the tensors it was built with; step(inputs, target) first copies a new batch (same shapes) into them.  `loss` is the
    graph's own output tensor (overwritten by every replay)."""

    def __init__(self, model, reducer, optimizer, criterion, inputs, target, warmup: int = 3):
        if reducer.world != 1:
            raise RuntimeError("GraphedStep: single-process steps only (a captured step holds no collective)")
        if not model.training:
            raise RuntimeError("GraphedStep captures a TRAINING step: call model.train() first")
        self.model, self.reducer, self.opt, self.criterion = model, reducer, optimizer, criterion
        self.inputs = [t.detach().clone() for t in inputs]
        self.target = target.detach().clone()
        optimizer.enable_device_hyper()
        # No autograd graph of an earlier EAGER step may be alive when the capture begins: its AccumulateGrad nodes belong to
        # the stream that step ran on (usually the legacy default stream), the engine would synchronise the capturing stream
        # with it, and hipStreamEndCapture falls over (seen as a segmentation fault in capture_end).  The criterion's result
        # dict is the usual keeper of the last step's graph.
        if hasattr(criterion, "reset"):
            criterion.reset()
        import gc
        gc.collect()
        # warm-up on the side stream the capture will use (torch's capture protocol): lazy state -- side streams, workspaces,
        # packed-weight caches, the library's static initialisers, the AccumulateGrad nodes -- exists, on THAT stream, before
        # the capture begins.  These are REAL steps.
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._body()
            if hasattr(criterion, "reset"):
                criterion.reset()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.warmup_steps = warmup
        steps_before = optimizer.steps
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self.loss = self._body()
        optimizer.steps = steps_before          # the capture enqueued nothing: its optimizer step has not happened
        self.replays = 0

    def _body(self):
        self.reducer.zero_grad()
        loss = self.criterion(self.model(*self.inputs), self.target)["Total"]
        loss.backward()
        self.reducer.finish()
        self.opt.step()
        return loss

    def __call__(self, inputs=None, target=None):
        if inputs is not None:
            for dst, src in zip(self.inputs, inputs):
                dst.copy_(src, non_blocking=True)
        if target is not None:
            self.target.copy_(target, non_blocking=True)
        self.opt.steps += 1
        self.opt.upload_hyper()                  # this step's learning rate and bias corrections
        self.graph.replay()
        ops.invalidate_packed_weights()          # the replay rewrote the parameters (and re-made the packed copies it uses)
        self.replays += 1
        return self.loss
