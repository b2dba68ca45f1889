"""AdamW over ONE flat parameter buffer (reference: torch.optim.AdamW built by
utils/common_config.py:241-291 from configs/*.yml:71-76 -- lr 1e-3, weight_decay 1e-6).

`GradReducer` already aliases every `.grad` into one flat fp32 buffer (the RCCL buckets); this
optimizer re-points every `.data` into a second flat buffer laid out identically, so a whole step is a
single HBM-bound kernel launch (jspsr_adamw_step: 16 B read + 12 B written per parameter) instead of
per-tensor launches.  `state_dict()` / `load_state_dict()` of the model keep working: parameters are
ordinary views.
"""
from __future__ import annotations

import torch

from . import _lib
from .ddp import GradReducer


class FlatAdamW:
    def __init__(self, reducer: GradReducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.reducer = reducer
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.flat_p = torch.empty_like(reducer.flat)
        off = 0
        for p in reversed(reducer.params):          # same order as the gradient buffer
            n = p.numel()
            self.flat_p[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + n].view_as(p)
            off += n
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.steps = 0

    def step(self):
        if not self.flat_p.is_cuda:
            raise RuntimeError("FlatAdamW runs on the GPU only")
        self.steps += 1
        lib = _lib.load()
        _lib.check(lib.jspsr_adamw_step(self.flat_p.data_ptr(), self.reducer.flat.data_ptr(), self.exp_avg.data_ptr(),
                                        self.exp_avg_sq.data_ptr(), self.flat_p.numel(), self.lr, self.betas[0],
                                        self.betas[1], self.eps, self.weight_decay, self.steps,
                                        torch.cuda.current_stream().cuda_stream), "jspsr_adamw_step")

    def zero_grad(self):
        self.reducer.zero_grad()
