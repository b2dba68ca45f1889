"""CPU: host logic of the product package -- parameter tree, input contract, C-ABI exports."""
import ctypes
import os
import re

import pytest
import torch

from oracle import jspsr_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ic", [{"lr_dem": 1, "image": 3}, {"lr_dem": 1, "image": 3, "mask": 15},
                                {"lr_dem": 1, "mask": 15}])
def test_state_dict_keys_match_reference_table(ic):
    from jspsr_amd.JSPSR import Model
    m = Model(dict(ic, COP30=1), num_feature=8)
    got = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert got == list(R.jspsr_param_shapes(ic, 8).items())
    assert m.name == "JSPSR" and m.in_channels.get("lr_dem") == 1
    assert any("postprocessor" in k for k, _ in m.named_parameters())  # diff-LR group hook


def test_reference_param_counts():
    from jspsr_amd.JSPSR import Model
    m = Model({"lr_dem": 1, "image": 3, "COP30": 1})
    assert sum(p.numel() for p in m.parameters()) == 29_162_435 and len(m.state_dict()) == 326
    assert torch.equal(m.postprocessor.w, torch.ones(1, 1, 3, 3)) and m.postprocessor.b.item() == 0


def test_input_contract_errors():
    from jspsr_amd.JSPSR import Model
    m = Model({"lr_dem": 1, "image": 3}, num_feature=8)
    x = torch.zeros(1, 1, 16, 16)
    with pytest.raises(NotImplementedError):
        m(x)
    with pytest.raises(NotImplementedError):
        m(x, x, x, x)
    with pytest.raises(AssertionError):
        Model({"lr_dem": 1})
    with pytest.raises(RuntimeError, match="GPU only"):
        m(x, torch.zeros(1, 3, 16, 16))  # CPU tensors: no fallback


def test_cabi_exports_every_declared_symbol():
    from jspsr_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "jspsr_hip.h")).read()
    declared = set(re.findall(r"\b(jspsr_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.SO_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().jspsr_abi_version() == _lib.ABI_VERSION


def test_cabi_argument_validation_without_gpu():
    """Argument checks return error codes before any launch (safe on a box without a GPU)."""
    from jspsr_amd import _lib
    lib = _lib.load()
    assert lib.jspsr_prop_forward_f32(None, None, None, 18, None, None, 1.0, None, 1, 8, 8, None) == -1
    assert b"null" in lib.jspsr_last_error()
    assert lib.jspsr_prop_backward_workspace_bytes(8, 512, 512) == 16 + 4096 * 10 * 4  # row-count header + 64x8 tiles
    assert lib.jspsr_prop_backward_workspace_bytes(0, 512, 512) == 0


def test_warmup_step_lr_matches_the_reference_composition():
    """utils/common_config.py:339-358 builds SequentialLR([LambdaLR(warmup), StepLR], [warmup_epoch]) on a torch
    optimizer; the closed form must give the same learning rate every epoch, for both parameter groups."""
    import warnings
    import torch
    from jspsr_amd.optim import WarmupStepLR

    for warm, step_size, gamma, epochs in ((3, 100, 0.5, 320), (5, 7, 0.3, 40), (0, 4, 0.5, 14)):
        w = [torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))]
        ropt = torch.optim.AdamW([{"params": [w[0]]}, {"params": [w[1]], "lr": 3e-4}], lr=1e-3)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            train = torch.optim.lr_scheduler.StepLR(ropt, step_size=step_size, gamma=gamma)
            warmup = torch.optim.lr_scheduler.LambdaLR(ropt, lr_lambda=lambda e, warm=warm: 1 / (10 ** float(warm - e)))
            ref = torch.optim.lr_scheduler.SequentialLR(ropt, [warmup, train], [warm])

            class Opt:
                param_groups = [{"lr": 1e-3}, {"lr": 3e-4}]
            mine = WarmupStepLR(Opt(), warm, step_size, gamma)
            for e in range(epochs):
                a, b = ref.get_last_lr(), mine.get_last_lr()
                assert all(abs(x - y) <= 1e-12 * max(abs(x), 1e-30) + 1e-18 for x, y in zip(a, b)), (warm, e, a, b)
                ropt.step()
                ref.step()
                mine.step()


def test_fastdiv_is_exact(tmp_path):
    """csrc/common.h FastDiv (the tile kernels' division by launch constants) against `/` on the host: every divisor up
    to 4096 and a spread of large ones, dividends at the quotient boundaries and across the 32-bit range."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    src = tmp_path / "fd.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cstdint>
#define __host__
#define __device__
#define __forceinline__ inline
struct FastDiv { unsigned m, s1, s2; };
#define JSPSR_FASTDIV_ONLY
''' + _extract_fastdiv(os.path.join(here, "..", "jspsr_amd", "csrc", "common.h")) + r'''
int main() {
  unsigned long long bad = 0, n_checked = 0;
  auto check = [&](unsigned d) {
    const FastDiv f = make_fastdiv(d);
    const unsigned ns[] = {0u, 1u, d - 1, d, d + 1, 2 * d - 1, 2 * d, 12345u * d, 12345u * d - 1, 0x7fffffffu, 0x80000000u, 0xfffffffeu, 0xffffffffu,
                           0xffffffffu / d * d, 0xffffffffu / d * d - 1};
    for (unsigned n : ns) { ++n_checked; if (fastdiv(n, f) != n / d) ++bad; }
    for (unsigned k = 0; k < 64; ++k) { const unsigned n = k * 0x04000001u + k * k * 977u; ++n_checked; if (fastdiv(n, f) != n / d) ++bad; }
  };
  for (unsigned d = 1; d <= 4096; ++d) check(d);
  for (unsigned d = 4097; d < 0xfff00000u; d += 104729u) check(d);
  check(0x7fffffffu); check(0x80000000u); check(0x80000001u); check(0xffffffffu);
  std::printf("%llu %llu\n", n_checked, bad);
  return bad != 0;
}
''')
    exe = tmp_path / "fd"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert int(out[0]) > 300000 and int(out[1]) == 0


def _extract_fastdiv(path):
    """the two functions of common.h that implement FastDiv, as text (compiled by g++ beside a reference division)"""
    s = open(path).read()
    a = s.index("inline FastDiv make_fastdiv")
    b = s.index("}  // namespace jspsr", a)
    body = s[a:b]
    return body.replace("__umulhi(f.m, n)", "0")      # (the device branch is preprocessed away; keep g++ happy anyway)
