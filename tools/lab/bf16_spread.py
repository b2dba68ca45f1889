"""Lab: run-to-run spread of the 100-step convergence test's bf16 arm (tests/test_model_scale_gpu.py::test_bf16_trains_like_fp32):
the same run from initial parameters perturbed by one fp32 rounding, several draws; prints the last loss windows and the
held-out scores.  JSPSR_PROP_HEAD_DMA=0|1 python tools/lab/bf16_spread.py [draws=4] [dtype=bf16|f32]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import jspsr_ref as R  # noqa: E402
from tests import fixtures as Fx  # noqa: E402
from tests.test_model_scale_gpu import _hip_trainer  # noqa: E402
from jspsr_amd import metrics as M  # noqa: E402

draws = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dt = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.bfloat16
decay_from = int(os.environ.get("SPREAD_DECAY_FROM", "0"))     # multiply the learning rate by 0.1 from this step on
B, H, W = 2, 128, 128
sd64 = R.make_state_dict(R.jspsr_param_shapes(Fx.MSK, 8), 991, torch.float64)
batches = []
for s in range(5):
    i64, g64 = R.synthetic_batch(B, H, W, True, seed=1000 + s, dtype=torch.float32)
    batches.append(([t.cuda() for t in i64], g64.cuda()))
held = batches.pop()
for d in range(draws):
    rs = np.random.RandomState(100 + d)
    sd = sd64 if d == 0 else {k: (v * (1 + 2.0 ** -23 * torch.from_numpy(rs.uniform(-1, 1, tuple(v.shape)))) if v.is_floating_point() and v.dim() > 0 and "running" not in k else v.clone())
                              for k, v in sd64.items()}
    m, step = _hip_trainer(sd, dt)
    losses, evals = [], []
    for i in range(100):
        if decay_from and i == decay_from:
            step.opt.lr = step.opt.lr * 0.1
        losses.append(step(*batches[i % 4]).item())
        if i + 1 in (70, 80, 90, 100):
            m.eval()
            with torch.no_grad():
                pred = m(*held[0])
            meter = M.Meter(-80.0, 929.0, border=0.05, elev_log=True)
            meter.update(pred, held[1])
            evals.append(meter.scores())
            m.train()
    w = np.array(losses).reshape(10, 10).mean(1)
    print(f"draw {d}: windows " + " ".join(f"{v:.5f}" for v in w) + f"; held-out RMSE {np.mean([e['RMSE'] for e in evals]):.3f} "
          f"({' '.join('%.2f' % e['RMSE'] for e in evals)}) PSNR {np.mean([e['PSNR'] for e in evals]):.2f}", flush=True)
