"""CPU: plain-C propagation oracle (oracle/prop_ref.c) against the reference-made fixture."""
import os

import numpy as np

from oracle import prop_ref


def test_c_oracle_matches_reference_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "g1_postprocessor.npz"))
    out = prop_ref.forward(z["dem"], z["weight"], z["offset"], z["w"], z["b"][0], 1.0)
    assert np.abs(out - z["out"]).max() < 1e-12
    gw, go, gwk, gb = prop_ref.backward(z["grad_out"], z["dem"], z["weight"], z["offset"], z["w"])
    assert np.abs(gw - z["grad_weight"]).max() < 1e-12
    smooth = z["offset"] != np.round(z["offset"])  # kink of the bilinear sampler: see test_oracle_golden
    assert np.abs((go - z["grad_offset"]) * smooth).max() < 1e-11
    assert np.abs(gwk - z["grad_w"]).max() < 1e-10
    assert np.abs(gb - z["grad_b"]).max() < 1e-10


def test_c_oracle_fp32_close_to_fp64(golden_dir):
    z = np.load(os.path.join(golden_dir, "g1_postprocessor.npz"))
    f = lambda k: z[k].astype(np.float32)
    out = prop_ref.forward(f("dem"), f("weight"), f("offset"), f("w"), z["b"][0], 1.0)
    far = np.abs(z["offset"]).reshape(2, 9, 2, 20, 24).max((1, 2)) > 20  # fp32 coordinate rounding at |p|~100
    err = np.abs(out - z["out"])[:, 0][~far]
    assert err.max() < 2e-5
