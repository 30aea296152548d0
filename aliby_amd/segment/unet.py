"""
The segmentation network: a residual U-Net with a global style vector, in plain PyTorch-ROCm.

The module DEFINES the network (parameter layout, state_dict keys) and serves as the fp32 reference the GPU tests
compare against; inference runs through `fused_unet.FusedUNet`, which executes the same arithmetic with the
hand-written kernels of aliby_amd/csrc/nn_*.hip (north_star allowed PyTorch for this forward; nothing of it is left there).
Architecture as published for Cellpose's U-Net family (nbase = [2, 32, 64, 128, 256], 3x3 kernels,
pre-activation BatchNorm-ReLU-Conv units, two residual pairs per scale with a 1x1 projection, max-pool
down, nearest up, additive skips, style = L2-normalised global average of the deepest map injected
through a linear layer before every up-convolution, 3 outputs = dY, dX, cell probability).

Pretrained weights are fetched from the network by the reference at model construction
(src/aliby/segment/dispatch.py:171-175) and are NOT obtainable offline: without a
`pretrained_model` file the module is randomly initialised with a fixed seed — good for throughput and
MFMA-utilisation measurements, meaningless for masks.  `load_cellpose_state_dict` maps the public CPnet checkpoint
key layout (`downsample.down.res_down_k.…`, `upsample.up.res_up_k.…`, `make_style`, `output.…`, `diam_*`) onto this
module's names; `build_network(pretrained_model=path)` uses it with a weights-only loader.
"""

from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def _unit(cin, cout, k, relu=True):
    layers = [nn.BatchNorm2d(cin, eps=1e-5, momentum=0.05)]
    if relu:
        layers.append(nn.ReLU(inplace=True))
    layers.append(nn.Conv2d(cin, cout, k, padding=k // 2))
    return nn.Sequential(*layers)


class DownBlock(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.proj = _unit(cin, cout, 1, relu=False)
        self.conv = nn.ModuleList([_unit(cin if t == 0 else cout, cout, k) for t in range(4)])

    def forward(self, x):
        x = self.proj(x) + self.conv[1](self.conv[0](x))
        return x + self.conv[3](self.conv[2](x))


class StyledUnit(nn.Module):
    def __init__(self, cin, cout, cstyle, k):
        super().__init__()
        self.conv = _unit(cin, cout, k)
        self.full = nn.Linear(cstyle, cout)

    def forward(self, style, x, y=None):
        if y is not None:
            x = x + y
        return self.conv(x + self.full(style)[:, :, None, None])


class UpBlock(nn.Module):
    def __init__(self, cin, cout, cstyle, k):
        super().__init__()
        self.conv0 = _unit(cin, cout, k)
        self.conv1 = StyledUnit(cout, cout, cstyle, k)
        self.conv2 = StyledUnit(cout, cout, cstyle, k)
        self.conv3 = StyledUnit(cout, cout, cstyle, k)
        self.proj = _unit(cin, cout, 1, relu=False)

    def forward(self, x, skip, style):
        x = self.proj(x) + self.conv1(style, self.conv0(x), y=skip)
        return x + self.conv3(style, self.conv2(style, x))


class ResidualUNet(nn.Module):
    def __init__(self, nbase=(2, 32, 64, 128, 256), nout=3, k=3):
        super().__init__()
        nbase = list(nbase)
        self.nbase = nbase
        self.down = nn.ModuleList([DownBlock(nbase[i], nbase[i + 1], k) for i in range(len(nbase) - 1)])
        up = nbase[1:] + [nbase[-1]]
        self.up = nn.ModuleList([UpBlock(up[i], up[i - 1], up[-1], k) for i in range(1, len(up))])
        self.output = _unit(up[0], nout, 1)

    def forward(self, x):
        feats = []
        for i, blk in enumerate(self.down):
            x = blk(x if i == 0 else F.max_pool2d(feats[-1], 2, 2))
            feats.append(x)
        style = F.adaptive_avg_pool2d(feats[-1], 1).flatten(1)
        style = style / torch.sum(style**2, dim=1, keepdim=True) ** 0.5
        x = self.up[-1](feats[-1], feats[-1], style)
        for i in range(len(self.up) - 2, -1, -1):
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = self.up[i](x, feats[i], style)
        return self.output(x), style

    def flops_per_pixel(self) -> float:
        """2*MACs of the convolutions per input pixel (for MFMA-utilisation reporting)."""
        total = 0.0
        scale = 1.0
        for i, blk in enumerate(self.down):
            if i > 0:
                scale /= 4.0
            for m in blk.modules():
                if isinstance(m, nn.Conv2d):
                    total += 2.0 * m.in_channels * m.out_channels * m.kernel_size[0] * m.kernel_size[1] * scale
        scale_up = [1.0 / 4.0**i for i in range(len(self.up))]
        for i, blk in enumerate(self.up):
            for m in blk.modules():
                if isinstance(m, nn.Conv2d):
                    total += 2.0 * m.in_channels * m.out_channels * m.kernel_size[0] * m.kernel_size[1] * scale_up[i]
        for m in self.output.modules():
            if isinstance(m, nn.Conv2d):
                total += 2.0 * m.in_channels * m.out_channels
        return total


# Public CPnet checkpoint layout (cellpose 2.x / 3.x `resnet_torch.CPnet`, nbase = [nchan, 32, 64, 128, 256], residual_on,
# style_on, no concatenation; [UPSTREAM-RECALL], the package is not in this image) -> names of ResidualUNet:
#   downsample.down.res_down_K.conv.conv_J.{0,2}.*        -> down.K.conv.J.{0,2}.*        (0 = BatchNorm, 2 = Conv)
#   downsample.down.res_down_K.proj.{0,1}.*               -> down.K.proj.{0,1}.*          (0 = BatchNorm, 1 = 1x1 Conv)
#   upsample.up.res_up_K.conv.conv_0.{0,2}.*              -> up.K.conv0.{0,2}.*
#   upsample.up.res_up_K.conv.conv_J.conv.{0,2}.*  J=1..3 -> up.K.convJ.conv.{0,2}.*
#   upsample.up.res_up_K.conv.conv_J.full.{weight,bias}   -> up.K.convJ.full.{weight,bias}
#   upsample.up.res_up_K.proj.{0,1}.*                     -> up.K.proj.{0,1}.*
#   output.{0,2}.*                                        -> output.{0,2}.*
#   diam_mean, diam_labels                                -> returned separately (they rescale images, not weights)
_CPNET_RULES = (
    (r"^downsample\.down\.res_down_(\d+)\.conv\.conv_(\d+)\.(\d+)\.(\w+)$", r"down.\1.conv.\2.\3.\4"),
    (r"^downsample\.down\.res_down_(\d+)\.proj\.(\d+)\.(\w+)$", r"down.\1.proj.\2.\3"),
    (r"^upsample\.up\.res_up_(\d+)\.conv\.conv_0\.(\d+)\.(\w+)$", r"up.\1.conv0.\2.\3"),
    (r"^upsample\.up\.res_up_(\d+)\.conv\.conv_([123])\.conv\.(\d+)\.(\w+)$", r"up.\1.conv\2.conv.\3.\4"),
    (r"^upsample\.up\.res_up_(\d+)\.conv\.conv_([123])\.full\.(\w+)$", r"up.\1.conv\2.full.\3"),
    (r"^upsample\.up\.res_up_(\d+)\.proj\.(\d+)\.(\w+)$", r"up.\1.proj.\2.\3"),
    (r"^output\.(\d+)\.(\w+)$", r"output.\1.\2"),
)
_CPNET_EXTRA = ("diam_mean", "diam_labels")


def cpnet_key_to_local(key: str) -> str | None:
    """CPnet checkpoint key -> ResidualUNet key (None for the diameter buffers); KeyError for anything else."""
    import re

    key = key.removeprefix("module.")  # checkpoints saved from nn.DataParallel
    if key in _CPNET_EXTRA:
        return None
    for pattern, repl in _CPNET_RULES:
        if re.match(pattern, key):
            return re.sub(pattern, repl, key)
    raise KeyError(f"unrecognised CPnet checkpoint key: {key}")


def load_cellpose_state_dict(net: "ResidualUNet", state: dict, strict: bool = True) -> dict:
    """Load a state dict in the public CPnet key layout (or in this module's own layout) into `net`.

    Returns {"diam_mean": float | None, "diam_labels": float | None}.  Shapes are checked by `load_state_dict`; a
    checkpoint of another architecture (different nbase, concatenation=True, the 4.x transformer) fails there with the
    offending keys named.  Call site replaced: CellposeModel(pretrained_model=...) at segment/dispatch.py:171-175."""
    own = set(net.state_dict())
    if all(k in own for k in state):
        net.load_state_dict(state, strict=strict)
        return {"diam_mean": None, "diam_labels": None}
    mapped, extra = {}, {"diam_mean": None, "diam_labels": None}
    for key, value in state.items():
        local = cpnet_key_to_local(key)
        if local is None:
            extra[key.removeprefix("module.")] = float(torch.as_tensor(value).reshape(-1)[0])
        else:
            mapped[local] = value
    net.load_state_dict(mapped, strict=strict)
    return extra


def build_network(seed: int = 0, pretrained_model: str | None = None, device="cuda") -> ResidualUNet:
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    net = ResidualUNet()
    # non-trivial BatchNorm statistics so that a random-weight forward keeps a sane dynamic range
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.running_var.fill_(1.0)
            m.running_mean.zero_()
    torch.random.set_rng_state(gen_state)
    if pretrained_model is not None:
        state = torch.load(pretrained_model, map_location="cpu", weights_only=True)  # executes nothing from the file
        net.diam = load_cellpose_state_dict(net, state)
    return net.eval().to(device)
