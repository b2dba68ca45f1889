"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's NLSPN (models/components/nlspn.py), the
N-iteration fixed-affinity user of the propagation sampler.  Only tests/ may import this module.

  * offset_affinity()   nlspn.py:77-175  (zero reference offset :84-90; TC / TGASS tanh scaling :94-99; confidence
                        modulation by eight 1x1 deform_conv2d samples :104-154; abs-sum normalisation :158-173)
  * propagate()         nlspn.py:177-187,219-233  (prop_time steps, optional preserve_input blend :221-224)
The sampler is formulation A of oracle/jspsr_ref.py (explicit 4-corner gather).  Pinned by
tests/golden/g8_nlspn_*.npz, made by the reference's own NLSPN class with the grid_sample stand-in (formulation B).
"""
import torch

from oracle import jspsr_ref as R


def sample_1x1(x, off2, legacy_shift=(0.0, 0.0)):
    """deform_conv2d(x, off2, weight=ones(1,1,1,1), padding 0): x (B,1,H,W) sampled at (y + dy, x + dx)."""
    B, _, H, W = x.shape
    off = torch.zeros(B, 18, H, W, dtype=x.dtype)
    # tap 4 of the 3x3 sampler sits at (y, x): put the pair there
    off[:, 8] = off2[:, 0] + legacy_shift[0]
    off[:, 9] = off2[:, 1] + legacy_shift[1]
    return R.sample_taps(x, off)[:, 4:5]


def offset_affinity(raw, affinity, scale_const, confidence=None, conf_prop=False, legacy=False):
    """raw (B,24,H,W) = conv_offset_aff(guidance) -> (offset (B,18,H,W), aff (B,9,H,W))."""
    B, _, H, W = raw.shape
    off16, aff = raw[:, :16], raw[:, 16:]
    offset = torch.cat((off16[:, :8], torch.zeros(B, 2, H, W, dtype=raw.dtype), off16[:, 8:]), 1)
    if affinity == "TC":
        aff = torch.tanh(aff / 100) / scale_const
    elif affinity == "TGASS":
        aff = torch.tanh(aff / 100) / (scale_const + 1e-8)
    if conf_prop:
        confs = []
        for idx in range(9):
            ww, hh = idx % 3, idx // 3
            if ww == 1 and hh == 1:
                continue
            o = offset[:, 2 * idx:2 * idx + 2].detach()
            confs.append(sample_1x1(confidence, o, (hh - 1.0, ww - 1.0) if legacy else (0.0, 0.0)))
        aff = aff * torch.cat(confs, 1)
    s = aff.abs().sum(1, keepdim=True) + 1e-4
    if affinity in ("ASS", "TGASS"):
        s = torch.where(s < 1.0, torch.ones_like(s), s)
    if affinity in ("AS", "ASS", "TGASS"):
        aff = aff / s
    ref = 1.0 - aff.sum(1, keepdim=True)
    return offset, torch.cat((aff[:, :4], ref, aff[:, 4:]), 1)


def propagate(feat, offset, aff, prop_time, feat_fix=None):
    """-> list of prop_time rasters; each step = sum_k aff_k S_k(feat; offset) (w = 1, b = 0)."""
    mask = None
    if feat_fix is not None:
        mask = ((feat_fix > 0.0).sum(1, keepdim=True) > 0.0).to(feat.dtype)
    out, cur = [], feat
    for _ in range(prop_time):
        if mask is not None:
            cur = (1.0 - mask) * cur + mask * feat_fix
        cur = (aff * R.sample_taps(cur, offset)).sum(1, keepdim=True)
        out.append(cur)
    return out
