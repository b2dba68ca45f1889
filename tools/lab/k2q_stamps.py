"""Lab: where a K2q wave's cycles go (the -DK2Q_STAMPS build: tools/lab/build_k2q_variants.sh stamps=-DK2Q_STAMPS, loaded through
JSPSR_LAB_LIB).  One launch of each form at 8 x 512^2 after a warm-up; waves 0 and 2 of workgroup 0 print their per-tile cycles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jspsr_amd import kernels as K  # noqa: E402

B, H, W = 8, 512, 512
x = torch.randn(B, H, W, 128, device="cuda").to(torch.bfloat16)
w = torch.randn(128, 128, 3, 3, device="cuda") / (128 * 9) ** 0.5
wp, wpt = K.pack_weight(w, 0, 128, torch.bfloat16), K.pack_weight(w, 1, 128, torch.bfloat16)
add = torch.randn(B, H, W, 128, device="cuda").to(torch.bfloat16)
for name, fn in (("fwd", lambda: K.conv2d_forward(x, wp, None, 1, 1)), ("fwd+stats", lambda: K.conv2d_forward(x, wp, None, 1, 1, stats=True)),
                 ("dgrad+addend", lambda: K.conv2d_dgrad(x, wpt, (H, W), 1, 1, addend=add))):
    print("==", name, flush=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
