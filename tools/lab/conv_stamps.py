"""Lab: per-phase cycle sums of conv_patch_kernel's waves (conv.hip, -DCONVLAB_STAMPS build).
JSPSR_LAB_LIB=jspsr_amd/lib_lab/libjspsr_conv_stamps.so python tools/lab/conv_stamps.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jspsr_amd import _lib, kernels as K  # noqa: E402

lib = _lib.load()
rd = lib.jspsr_lab_conv_stamps
rd.restype = ctypes.c_int
rd.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
SHAPES = [(8, 512, 512, 128, 128, 3), (8, 256, 256, 128, 128, 3), (8, 128, 128, 256, 256, 3), (8, 64, 64, 512, 512, 3),
          (8, 512, 512, 128, 128, 5), (8, 512, 512, 256, 64, 3)]
names = ["load issue", "compute", "patch handover", "weight store + stage barrier"]
FINE = os.environ.get("CONVLAB_FINE", "0") == "1"      # a -DCONVLAB_STAMPS=2 build
for B, H, W, Ci, Co, k in SHAPES:
    x = torch.randn(B, H, W, Ci, device="cuda").to(torch.bfloat16)
    w = torch.randn(Co, Ci, k, k, device="cuda") / (Ci * k * k) ** 0.5
    wp = K.pack_weight(w, 0, Ci, torch.bfloat16)
    for _ in range(20):
        K.conv2d_forward(x, wp, None, 1, k // 2, relu=False, stats=False)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    rd(buf, 1)
    n = 1        # the stamp rows hold one launch (rows are per wave, overwritten by the next launch)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        K.conv2d_forward(x, wp, None, 1, k // 2, relu=False, stats=False)
    e1.record()
    torch.cuda.synchronize()
    rd(buf, 1)
    v = [float(b) for b in buf]
    stages = v[7]                      # sum over waves of the stage count
    if stages == 0:
        print(f"B{B} {H}x{W} {Ci}->{Co} k{k}: not the patch kernel")
        continue
    ms = e0.elapsed_time(e1) / n
    waves = min(4 * B * ((H + 7) // 8) * ((W + 15) // 16) * ((Co + 127) // 128), 1 << 17)
    life = (v[4] + v[5] + v[6]) / waves
    print(f"B{B} {H}x{W} {Ci}->{Co} k{k}: {ms:.3f} ms  {2.0*B*H*W*Co*Ci*k*k/ms/1e9:.0f} TF/s with stamps; {stages/waves:.0f} stages per wave; shader cycles per wave (s_memtime):")
    print(f"   entry -> first stage ready {v[4]/waves:9.0f}  ({100*v[4]/waves/life:4.1f} %)")
    print(f"   main loop                  {v[5]/waves:9.0f}  ({100*v[5]/waves/life:4.1f} %)  = {v[5]/stages:7.0f} per stage (16 MFMAs = 512 cycles of one wave's matrix pipe, 2 waves per SIMD)")
    print(f"   epilogue                   {v[6]/waves:9.0f}  ({100*v[6]/waves/life:4.1f} %)")
    if os.environ.get("CONVLAB_PIECES", "0") == "1":      # a -DCONVLAB_STAMPS=3 build
        print(f"      prologue: tile coordinates {v[11]/waves:7.0f}   rest of the plan {v[8]/waves:7.0f}   loads issued {v[9]/waves:7.0f}   waited for + stored {v[10]/waves:7.0f}   barrier {(v[4]-v[11]-v[8]-v[9]-v[10])/waves:7.0f}")
        print(f"      epilogue: piece plan + accumulators -> LDS {v[0]/waves:7.0f}   barrier {v[1]/waves:7.0f}   pieces read back {v[2]/waves:7.0f}   stores issued {(v[6]-v[0]-v[1]-v[2])/waves:7.0f}")
    if FINE:
        for i, nm in enumerate(names):
            print(f"      per stage: {nm:30s} {v[i]/stages:8.1f}")
