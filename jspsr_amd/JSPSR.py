"""``Model`` -- drop-in for the reference's ``models.JSPSR.Model`` (models/JSPSR.py:9-550).

Same constructor, ``forward(*in_tensor)`` contract ([dem, img] or [dem, img, aux], contiguous
fp32 NCHW on the device), attributes (``name``, ``in_channels``) and ``state_dict`` key set, so
it can be handed to the reference's ``main.py`` train/eval loops and checkpoints unchanged
(INTEGRATION.md).  All arithmetic runs in HIP kernels on gfx950 via jspsr_amd.engine.
"""
from __future__ import annotations

import contextlib
import math
import os

import torch
import torch.nn as nn

from . import engine as E
from . import ops
from .blocks import ConvUnit, HotPathModule, ResUnit, UpUnit
from .spn import Generator, PostProcessor

_AUX_KEYS = ("mask", "canopy", "coord")


class Model(HotPathModule):
    # pixels of input a prediction depends on beyond its own position, learned offsets aside (tiling.py certifies strips
    # with it): stem 2 + encoder 4+8+16+32 + decoder 16+8+4 + conv0 1 + generator 5 + 3x3 sampler 1
    receptive_radius = 97

    def __init__(self, in_channels: dict, out_channels: int = 1, num_feature: int = 32,
                 layers: tuple = (2, 2, 2, 2), res_scale: tuple = (1, 1, 1, 1), spn: bool = True,
                 spn_scale: int = 1):
        super().__init__()
        self.name = "JSPSR"
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.spn = spn
        self.spn_scale = spn_scale
        self.compute_dtype = torch.float32  # or torch.bfloat16: bf16 storage, fp32 accumulate/statistics
        assert len(in_channels) > 1, "At least 2 input data are required"
        if not spn:
            raise NotImplementedError("spn=False (plain conv head) is outside the hot path")
        nf = num_feature
        self.flag_dem_img = "image" in in_channels
        self.flag_dem_msk = "mask" in in_channels
        self.flag_dem_canopy = "canopy" in in_channels
        self.flag_dem_coord = "coord" in in_channels
        # the reference wires exactly one auxiliary branch: mask, else canopy, else coord (:74-87)
        self._aux = next((k for k in _AUX_KEYS if k in in_channels), None)
        self._branches = ["dem"] + (["img"] if self.flag_dem_img else []) + (["aux"] if self._aux else [])
        nb = len(self._branches)
        if nb < 2:
            raise AssertionError("At least one of image or mask is required")

        self.conv_dem = ConvUnit(in_channels["lr_dem"], nf, 5, bn=False)
        self.conv_img = ConvUnit(in_channels["image"], nf, 5, bn=True) if self.flag_dem_img else None
        self.conv_aux = ConvUnit(in_channels[self._aux], nf, 5, bn=False) if self._aux else None

        cin = nf
        for s in range(4):
            planes = nf * 2 * 2**s
            stride = 1 if s == 0 else 2
            for br in ("dem", "img", "aux"):
                seq = None
                if br in self._branches:
                    first_in = cin * (nb if (br == "dem" and s > 0) else 1)
                    units = [ResUnit(first_in, planes, stride, project=True, scale=res_scale[s])]
                    units += [ResUnit(planes, planes, scale=res_scale[s]) for _ in range(1, layers[s])]
                    seq = nn.Sequential(*units)
                setattr(self, f"layer{s + 1}_{br}", seq)
            setattr(self, f"guide{s + 1}", nn.Module())  # parameter-free concat (Guide, cat_only)
            cin = planes
        self.layer3d = UpUnit(nf * 16 * nb, nf * 8)
        self.layer2d = UpUnit(nf * 8 + nf * 8 * nb, nf * 4)
        self.layer1d = UpUnit(nf * 4 + nf * 4 * nb, nf * 2)
        self.conv0 = ConvUnit(nf * 2 + nf * 2 * nb, nf * 2, 3, bn=True, relu=True, gate=True)
        self.generator = Generator(in_channels=nf * 2, kernel_size=3, bc=nf)
        self.postprocessor = PostProcessor(kernel_size=3, residual=True, scale=self.spn_scale)
        self._initialize_weights()

    # -- init: truncated normal +-2 sigma, sigma = sqrt(2.6 / (k*k*C_in)) (JSPSR.py:494-517) -------
    def _initialize_weights(self):
        try:
            from scipy.stats import truncnorm  # same sampler / RNG stream as the reference
        except ImportError:  # pragma: no cover
            truncnorm = None
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                n = m.kernel_size[0] * m.kernel_size[1] * m.in_channels
                std = math.sqrt(1.3 * 2.0 / n)
                if truncnorm is not None:
                    vals = truncnorm(-2.0, 2.0, loc=0.0, scale=std).rvs(m.weight.nelement())
                    m.weight.data = torch.from_numpy(vals).type_as(m.weight.data).view_as(m.weight.data)
                else:
                    nn.init.trunc_normal_(m.weight, 0.0, std, -2 * std, 2 * std)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    @staticmethod
    def parse_input(flag_dem_img, flag_dem_msk, flag_dem_canopy, flag_dem_coord, *in_tensor):
        """Same contract and errors as models/JSPSR.py:519-550."""
        assert flag_dem_msk or flag_dem_img or flag_dem_canopy, "At least one of image or mask is required"
        dem = img = msk = canopy = coord = None
        if len(in_tensor) == 3 and flag_dem_img:
            if flag_dem_msk:
                dem, img, msk = in_tensor
            elif flag_dem_canopy:
                dem, img, canopy = in_tensor
            elif flag_dem_coord:
                dem, img, coord = in_tensor
            else:
                raise NotImplementedError
        elif len(in_tensor) == 2:
            if flag_dem_img:
                dem, img = in_tensor
            elif flag_dem_msk:
                dem, msk = in_tensor
            elif flag_dem_canopy:
                dem, canopy = in_tensor
            elif flag_dem_coord:
                dem, coord = in_tensor
            else:
                raise NotImplementedError
        else:
            raise NotImplementedError
        return dem, img, msk, canopy, coord

    def forward(self, *in_tensor):
        dem, img, msk, canopy, coord = self.parse_input(
            self.flag_dem_img, self.flag_dem_msk, self.flag_dem_canopy, self.flag_dem_coord, *in_tensor)
        aux = msk if msk is not None else (canopy if canopy is not None else coord)
        with E.compute_dtype(self.compute_dtype), E.count_batches():
            return self._forward(dem, img, aux)

    branch_streams = True   # guidance branches on side streams (class-wide switch; JSPSR_BRANCH_STREAMS=0 disables)

    def _side_streams(self, device):
        """{branch: stream} for the img / aux branches on `device`, or None when disabled."""
        if not self.branch_streams or os.environ.get("JSPSR_BRANCH_STREAMS", "1") == "0" or device.type != "cuda":
            return None
        cache = self.__dict__.setdefault("_branch_stream_cache", {})
        key = (device.index if device.index is not None else torch.cuda.current_device())
        if key not in cache:
            cache[key] = {br: torch.cuda.Stream(device=device) for br in ("img", "aux")}
        return {br: st for br, st in cache[key].items() if br in self._branches}

    def side_streams(self, device=None):
        """Streams the backward pass may write parameter gradients from (for GradReducer.watch_streams)."""
        device = device if device is not None else next(self.parameters()).device
        side = self._side_streams(torch.device(device))
        return list(side.values()) if side else []

    def _forward(self, dem, img, aux):
        dem_a = E.from_nchw(dem)
        feats = {"dem": self.conv_dem(dem_a)}
        if img is not None:
            feats["img"] = self.conv_img(E.from_nchw(img))
        if aux is not None:
            feats["aux"] = self.conv_aux(E.from_nchw(aux))
        order = [b for b in self._branches if b in feats]
        if len(order) < 2:
            raise NotImplementedError
        # The reference concatenates per stage (Guide cat_only, :230-352) and again in the decoder (:354-368).
        # Here each stage owns one wide NHWC buffer [up | dem | img | aux]; the last unit of every branch and the
        # decoder's UpUnit write their channel slice of it directly, so neither concat moves any data.
        B, H, W = dem_a.shape[:3]
        nb, nf2 = len(order), self.conv0.conv[0].out_channels
        bufs, fused, joined = [], [], []
        defer_skip = all(getattr(self, f"layer{s}_dem")[0].fused for s in (2, 3, 4))
        # deposits (below) rely on the dem branch being the LAST of a stage to run in the backward pass: it is first here
        deposit_ok = os.environ.get("JSPSR_GRAD_DEPOSIT", "1") != "0" and torch.is_grad_enabled() and order[0] == "dem"
        for s in range(1, 5):
            planes = nf2 * 2 ** (s - 1)
            if s > 1:
                H, W = (H + 1) // 2, (W + 1) // 2  # 3x3 stride-2 pad-1 (k1 projection: same size)
            lead = planes if s < 4 else 0       # room for the decoder's up-sampled features
            buf = E.SliceBuffer(B, H, W, lead + nb * planes, dem_a.dtype, dem_a.device)
            nxt = {}
            # The branches of a stage are independent until the concat: the guidance branches run on side streams
            # beside the dem branch (their deep-stage grids are too small to fill the chip alone).  autograd replays
            # each node's backward on its forward stream, so the backward pass gets the same concurrency.
            side = self._side_streams(dem_a.device) if len(order) > 1 else None
            if side is not None:
                main = torch.cuda.current_stream()
                forked = main.record_event()              # everything the branches read, and `buf`, exists by now
            for i, br in enumerate(order):
                from_fused = br == "dem" and bool(fused)
                src = fused[-1] if from_fused else feats[br]
                units = list(getattr(self, f"layer{s}_{br}"))
                st = side.get(br) if side is not None else None
                if st is not None:
                    st.wait_event(forked)
                    # tensors of the main stream's allocator pool that this stream touches (now, and again in
                    # the backward pass through saved references): tell the allocator, or it may hand their memory
                    # out again while work queued on `st` still uses it
                    buf.buf.record_stream(st)
                    if s == 1:
                        src.record_stream(st)
                with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                    for k, u in enumerate(units):
                        kw = {}
                        if k == 0 and from_fused and defer_skip:
                            kw["grad_extra"] = bufs[-1][0]      # the decoder's gradient of fused[s-1] is parked there
                        elif k == 0 and s > 1 and defer_skip and br != "dem" and getattr(u, "fused", False) and deposit_ok:
                            # this branch's previous-stage output feeds the stage join AND this unit: the unit adds its
                            # gradient of it into the parked gradient of the join (its slice), in place, and the dem
                            # branch's first block (which runs later in the backward pass) hands the sum on -- instead of
                            # autograd adding two full tensors per branch and stage
                            holder, w_prev = bufs[-1][0], planes // 2
                            kw["grad_extra"] = (holder, i * w_prev, w_prev)
                            holder.expected_deposits += 1
                        if k == len(units) - 1:
                            kw["dest"] = (buf, lead + i * planes)
                        src = u(src, **kw)
                nxt[br] = src
            if side is not None:
                for st in side.values():
                    main.wait_stream(st)                  # the concat (and everything after it) sees all branches
            feats = nxt
            bufs.append((buf, lead))
            joined.append([feats[b] for b in order])
            fused.append(buf.join(joined[-1], lead))
        x = fused[3]
        for up, s in ((self.layer3d, 2), (self.layer2d, 1), (self.layer1d, 0)):
            buf, lead = bufs[s]
            # cat((up, skip)), :354-368.  fused[s] has two consumers (this decoder stage and the next encoder stage's
            # dem branch): the decoder's share of its gradient is parked in the buffer and added inside that branch's
            # first block (grad_extra above) instead of by autograd
            x = buf.join([up(x, dest=(buf, 0))] + joined[s], 0, defer=nb if defer_skip else 0)
        c0 = self.conv0(x)
        dem = dem.detach()  # :372
        # the two 1x1 heads write the planes the propagation kernel reads (sigmoid, zero centre offset, mean subtraction,
        # gather, residual: one kernel -- the one of the public PostProcessor boundary; engine.heads_propagate)
        feature = self.generator.features(dem_a.detach(), c0)
        return self.postprocessor.from_feature(dem.float(), feature, self.generator)
