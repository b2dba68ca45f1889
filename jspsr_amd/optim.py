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
from . import ops
from .ddp import GradReducer


class FlatAdamW:
    """lr_overrides: {parameter: lr} -- the reference's optional second parameter group (`diff_lr`: the
    `postprocessor` parameters at lr 3e-4, utils/common_config.py:247-258).  Overridden parameters that are
    neighbours in the flat buffer form one range; a step is one launch per range (2 for the reference's grouping).
    `param_groups` mirrors torch's list of dicts ("lr", "initial_lr") so schedulers can drive it."""

    def __init__(self, reducer: GradReducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 lr_overrides=None):
        self.reducer = reducer
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.flat_p = torch.empty_like(reducer.flat)
        over = {id(p): float(v) for p, v in (lr_overrides or {}).items()}
        self.param_groups = [{"lr": float(lr), "initial_lr": float(lr), "ranges": []}]
        by_lr = {}
        off = 0
        self._slices = {}                            # id(parameter) -> (offset, numel) in the flat buffers
        self._over = over
        self._group_of = {}                          # id(parameter) -> its entry of param_groups
        for p in reversed(reducer.params):          # same order as the gradient buffer
            n = p.numel()
            self._slices[id(p)] = (off, n)
            self.flat_p[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + n].view_as(p)
            if id(p) in over:
                g = by_lr.get(over[id(p)])
                if g is None:
                    g = by_lr[over[id(p)]] = {"lr": over[id(p)], "initial_lr": over[id(p)], "ranges": []}
                    self.param_groups.append(g)
            else:
                g = self.param_groups[0]
            # membership is recorded HERE, by identity -- not re-derived later from the groups' mutable "lr" / "initial_lr"
            # (a scheduler step or a loaded checkpoint rewrites those: ADVICE r3)
            self._group_of[id(p)] = g
            if g["ranges"] and g["ranges"][-1][1] == off:
                g["ranges"][-1][1] = off + n
            else:
                g["ranges"].append([off, off + n])
            off += n
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.steps = 0
        ops.invalidate_packed_weights()      # the parameters now live elsewhere

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, v):
        self.param_groups[0]["lr"] = float(v)

    # -- scalars in device memory (graph capture: jspsr_amd/graph.py) ----------------------------------------------------
    def enable_device_hyper(self):
        """From now on the kernels read lr / betas / eps / weight decay / bias corrections from a small device tensor per
        parameter group (jspsr_adamw_step_dev) that `upload_hyper()` refreshes -- the launches can then be captured in a
        hipGraph and replayed while the learning rate and the step count move on.  Same arithmetic, same bits."""
        if getattr(self, "_hyper_dev", None) is None:
            n = len(self.param_groups)
            self._hyper_host = torch.zeros((n, 8), dtype=torch.float32).pin_memory()
            self._hyper_dev = torch.zeros((n, 8), dtype=torch.float32, device=self.flat_p.device)
        return self

    def upload_hyper(self, step=None):
        """Write the seven scalars of every group for step count `step` (default: the current one) into the device tensor;
        an ordinary stream-ordered copy from pinned memory -- call it OUTSIDE a captured region."""
        import numpy as np
        step = self.steps if step is None else step
        b1, b2 = float(np.float32(self.betas[0])), float(np.float32(self.betas[1]))      # the C side raises the float values
        bc1, bc2s = 1.0 - b1 ** step, (1.0 - b2 ** step) ** 0.5
        for i, g in enumerate(self.param_groups):
            self._hyper_host[i, :7] = torch.tensor([g["lr"], self.betas[0], self.betas[1], self.eps, self.weight_decay, bc1, bc2s],
                                                   dtype=torch.float64).float()
        self._hyper_dev.copy_(self._hyper_host, non_blocking=True)

    def step(self):
        if not self.flat_p.is_cuda:
            raise RuntimeError("FlatAdamW runs on the GPU only")
        self.steps += 1
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        es = self.flat_p.element_size()
        dev = getattr(self, "_hyper_dev", None)
        if dev is not None and not torch.cuda.is_current_stream_capturing():
            self.upload_hyper()                 # (under capture the owner of the graph uploads before every replay)
        for i, g in enumerate(self.param_groups):
            for lo, hi in g["ranges"]:
                if dev is not None:
                    _lib.check(lib.jspsr_adamw_step_dev(self.flat_p.data_ptr() + lo * es, self.reducer.flat.data_ptr() + lo * es,
                                                        self.exp_avg.data_ptr() + lo * es, self.exp_avg_sq.data_ptr() + lo * es,
                                                        hi - lo, dev[i].data_ptr(), stream), "jspsr_adamw_step_dev")
                else:
                    _lib.check(lib.jspsr_adamw_step(self.flat_p.data_ptr() + lo * es, self.reducer.flat.data_ptr() + lo * es,
                                                    self.exp_avg.data_ptr() + lo * es, self.exp_avg_sq.data_ptr() + lo * es,
                                                    hi - lo, g["lr"], self.betas[0], self.betas[1], self.eps,
                                                    self.weight_decay, self.steps, stream), "jspsr_adamw_step")
        ops.invalidate_packed_weights()      # the kernel wrote the parameters through raw pointers ...
        if ops.repack_after_step:
            ops.repack_all()                 # ... and every cached packed copy is re-made here, in one launch

    def zero_grad(self, set_to_none=False):
        self.reducer.zero_grad()

    # -- checkpointing: the reference saves optimizer.state_dict() with every best model (main.py:246-252) and
    #    restores it on resume (utils/utils.py:394) -------------------------------------------------------------------
    def _torch_order(self):
        """The parameters in the order torch.optim.AdamW numbers them when built as the reference builds it
        (utils/common_config.py:241-258): model.parameters() order, the `diff_lr` parameters moved to a second group."""
        return [[p for p in self.reducer.params if self._group_of[id(p)] is g] for g in self.param_groups]

    def state_dict(self, layout="flat"):
        """layout="flat" (default): flat moments + step count + per-group learning rates; tensors are clones (safe to
        torch.save).  layout="torch": the dict torch.optim.AdamW.state_dict() would hold for the same parameters
        (per-parameter `exp_avg` / `exp_avg_sq` / `step`, `param_groups` with `params` indices) -- what the reference
        writes into its checkpoints (main.py:246-252) and can resume from (utils/utils.py:394)."""
        if layout == "torch":
            state, groups, idx = {}, [], 0
            for g, plist in zip(self.param_groups, self._torch_order()):
                ids = []
                for p in plist:
                    off, n = self._slices[id(p)]
                    state[idx] = {"step": torch.tensor(float(self.steps)),
                                  "exp_avg": self.exp_avg[off:off + n].detach().clone().view_as(p),
                                  "exp_avg_sq": self.exp_avg_sq[off:off + n].detach().clone().view_as(p)}
                    ids.append(idx)
                    idx += 1
                groups.append({"lr": g["lr"], "initial_lr": g["initial_lr"], "betas": tuple(self.betas), "eps": self.eps,
                               "weight_decay": self.weight_decay, "amsgrad": False, "maximize": False, "foreach": None,
                               "capturable": False, "differentiable": False, "fused": None, "params": ids})
            return {"state": state if self.steps else {}, "param_groups": groups}
        if layout != "flat":
            raise ValueError("FlatAdamW.state_dict: layout must be 'flat' or 'torch'")
        return {
            "state": {"exp_avg": self.exp_avg.detach().clone(), "exp_avg_sq": self.exp_avg_sq.detach().clone(),
                      "step": int(self.steps)},
            "param_groups": [{"lr": g["lr"], "initial_lr": g["initial_lr"], "ranges": [list(r) for r in g["ranges"]],
                              "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay}
                             for g in self.param_groups],
            "numel": int(self.flat_p.numel()),
        }

    def load_state_dict(self, sd):
        """In place: the parameters keep aliasing `flat_p`, the gradients keep aliasing the reducer's buffer.  Accepts
        this class's flat layout and torch.optim.AdamW's (a checkpoint the reference wrote: per-parameter state scattered
        into the flat buffers in parameter order)."""
        if not isinstance(sd, dict) or "param_groups" not in sd or "state" not in sd:
            raise ValueError("FlatAdamW.load_state_dict: not an optimizer state dict (no 'state' / 'param_groups')")
        if "numel" not in sd:
            if not all("params" in g for g in sd["param_groups"]):
                raise ValueError("FlatAdamW.load_state_dict: unknown optimizer checkpoint format (neither FlatAdamW's flat "
                                 "layout nor torch.optim.AdamW's)")
            return self._load_torch(sd)
        if int(sd["numel"]) != self.flat_p.numel() or len(sd["param_groups"]) != len(self.param_groups):
            raise ValueError("FlatAdamW.load_state_dict: checkpoint was written for a different parameter layout")
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            if [list(r) for r in sg["ranges"]] != [list(r) for r in g["ranges"]]:
                raise ValueError("FlatAdamW.load_state_dict: parameter-group ranges differ")
            g["lr"], g["initial_lr"] = float(sg["lr"]), float(sg["initial_lr"])
        first = sd["param_groups"][0]
        self.betas, self.eps, self.weight_decay = tuple(first["betas"]), float(first["eps"]), float(first["weight_decay"])
        self.exp_avg.copy_(sd["state"]["exp_avg"])
        self.exp_avg_sq.copy_(sd["state"]["exp_avg_sq"])
        self.steps = int(sd["state"]["step"])


    def _load_torch(self, sd):
        order = self._torch_order()
        if len(sd["param_groups"]) != len(order) or any(len(g["params"]) != len(pl) for g, pl in zip(sd["param_groups"], order)):
            raise ValueError("FlatAdamW.load_state_dict: the torch AdamW checkpoint groups its parameters differently "
                             f"({[len(g['params']) for g in sd['param_groups']]} vs {[len(pl) for pl in order]}): build the "
                             "optimizer with the same lr_overrides (diff_lr) as the run that wrote it")
        # validate everything first, write afterwards: a checkpoint that is refused leaves the optimizer as it was
        steps, writes = set(), []
        for sg, plist in zip(sd["param_groups"], order):
            for pid, p in zip(sg["params"], plist):
                st = sd["state"].get(pid)
                if st is None:
                    continue                          # torch keeps no state for a parameter that never had a gradient
                if tuple(st["exp_avg"].shape) != tuple(p.shape) or tuple(st["exp_avg_sq"].shape) != tuple(p.shape):
                    raise ValueError(f"FlatAdamW.load_state_dict: parameter {pid} has shape {tuple(st['exp_avg'].shape)} in the "
                                     f"checkpoint, {tuple(p.shape)} here")
                writes.append((self._slices[id(p)], st))
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"FlatAdamW.load_state_dict: parameters at different step counts {sorted(steps)} (one fused step "
                             "count is kept)")
        first = sd["param_groups"][0]
        if first.get("amsgrad") or first.get("maximize"):
            raise ValueError("FlatAdamW.load_state_dict: amsgrad / maximize checkpoints are not supported")
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for (off, n), st in writes:
            self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            g["lr"], g["initial_lr"] = float(sg["lr"]), float(sg.get("initial_lr", sg["lr"]))
        self.betas, self.eps, self.weight_decay = tuple(first["betas"]), float(first["eps"]), float(first["weight_decay"])
        self.steps = steps.pop() if steps else 0


class WarmupStepLR:
    """The reference's "warmupsteplr" schedule (utils/common_config.py:339-358): SequentialLR of
    LambdaLR(1 / 10**(warmup_epoch - e)) for the first `warmup_epoch` epochs, then StepLR(step_size, gamma)
    counted from the hand-over.  Closed form per epoch e (one `step()` per epoch, as train loops call it):

        e <  warmup_epoch :  lr = initial_lr * 10**-(warmup_epoch - e)
        e >= warmup_epoch :  lr = initial_lr * gamma**((e - warmup_epoch) // step_size)

    Works on anything with torch-style `param_groups` (FlatAdamW, torch optimizers)."""

    def __init__(self, optimizer, warmup_epoch=0, step_size=100, gamma=0.5):
        self.optimizer = optimizer
        self.warmup_epoch, self.step_size, self.gamma = int(warmup_epoch), int(step_size), float(gamma)
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self.last_epoch = 0
        self._apply()

    def factor(self, epoch: int) -> float:
        if epoch < self.warmup_epoch:
            return 1.0 / (10.0 ** float(self.warmup_epoch - epoch))
        if self.warmup_epoch == 0:  # SequentialLR never "reaches" a milestone at 0: StepLR then runs one epoch late
            return self.gamma ** (max(epoch - 1, 0) // self.step_size)
        return self.gamma ** ((epoch - self.warmup_epoch) // self.step_size)

    def _apply(self):
        f = self.factor(self.last_epoch)
        for g in self.optimizer.param_groups:
            g["lr"] = g["initial_lr"] * f

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        return {"last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.last_epoch = int(sd["last_epoch"])
        self._apply()
