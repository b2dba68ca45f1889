"""Lab: what the tile queue of the persistent conv kernels (K2r, K2q) is for.  One workgroup of these kernels needs a WHOLE
compute unit; in a data-parallel step RCCL's all-reduce kernels hold some CUs for milliseconds beside the backward pass.  No
second GPU here, so the holders are emulated: S single-wave spin kernels (torch.cuda._sleep) on S streams, each keeping one CU
from hosting a 512-register wave for ~4 ms.  The same conv launch is then timed with the static stride walk
(jspsr_conv_dynamic_queue(0)) and with the global ticket (1): with the static walk the workgroups that cannot be placed start
when the first ones finish and still do their full share; with the ticket they find it empty.
  python tools/lab/persistent_under_contention.py [S = 3]
Result on one box (round 4): NO difference -- K2q alone static 604 / ticket 535 us, beside 3 holders 535 / 521 us; K2r 140 / 147 and
140 / 145 us.  The emulation does not reproduce the situation (the spinning waves evidently do not keep a workgroup off their CU,
or were not resident when the conv started), so the switch GradReducer sets for world sizes > 1 stays a precaution that costs
nothing measurable in the step (tools/ab_step.sh: 63.6 / 64.3 / 64.4 against 63.1 / 64.3 / 64.4 ms), not a measured gain."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jspsr_amd import _lib  # noqa: E402
from jspsr_amd import kernels as K  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 3          # (the runtime maps streams onto 4 hardware queues: more than 3 holders beside the main stream would queue behind one another)
    lib = _lib.load()
    streams = [torch.cuda.Stream() for _ in range(S)]
    main_s = torch.cuda.Stream()
    for C, shape, name in ((128, (8, 512, 512), "K2q 8x512^2 128->128 dgrad"), (64, (8, 512, 512), "K2r 8x512^2 64->64 dgrad")):
        x = torch.randn(*shape, C, device="cuda").to(torch.bfloat16)
        w = torch.randn(C, C, 3, 3, device="cuda") / (C * 9) ** 0.5
        wpt = K.pack_weight(w, 1, C, torch.bfloat16)
        out = torch.empty_like(x)
        res = {}
        for holders in (0, S):
            for mode in (0, 1):
                lib.jspsr_conv_dynamic_queue(mode)
                ts = []
                for rep in range(7):
                    torch.cuda.synchronize()
                    for st in streams[:holders]:
                        with torch.cuda.stream(st):
                            torch.cuda._sleep(10_000_000)         # ~4-5 ms of one spinning wave
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    with torch.cuda.stream(main_s):
                        torch.cuda._sleep(200_000)                # let the holders get their CUs first
                        e0.record()
                        K.conv2d_dgrad(x, wpt, shape[1:3], 1, 1, out=out)
                        e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                lib.jspsr_conv_dynamic_queue(-1)
                ts.sort()
                res[(holders, mode)] = ts[len(ts) // 2]
        print(f"{name}: alone static {res[(0, 0)]:.0f} us, ticket {res[(0, 1)]:.0f} us | beside {S} CU holders static {res[(S, 0)]:.0f} us, ticket {res[(S, 1)]:.0f} us", flush=True)


if __name__ == "__main__":
    main()
