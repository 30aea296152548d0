"""
Fused inference executor for the residual U-Net (aliby_amd/segment/unet.py).

Same arithmetic as `ResidualUNet.forward` in eval mode, reorganised for MI355X:
  * every 3x3 convolution unit of all four levels runs on the hand-written MFMA convolution unit `k_conv3x3`
    (aliby_amd/csrc/nn_conv.hip): BatchNorm + ReLU + style shift in the prologue, bias + residual / skip add (+ the
    block's 1x1 projection at levels 0-1, + the 2x2 max pool) in the epilogue, so a conv unit is ONE pass over HBM;
    layers wider than one launch holds (128 / 256 channels) are split along K and N (`mfma_levels` selects the
    levels; the others fall back to MIOpen convolutions with the fused pointwise kernel `k_fused_act` between them);
  * the output head (BatchNorm + ReLU + 1x1 conv + NHWC->NCHW) is one kernel (`k_out_head`);
  * the 2-channel first layer (`k_first_conv`: im2col gathered from LDS into two MFMA k-steps) and the 1x1 projections of
    the deep / up blocks (`k_conv1x1`; BatchNorm folded into their weights; the up path's run at the low resolution and
    are read through the upsample: a 1x1 conv commutes with nearest upsampling) are hand-written too
    (aliby_amd/csrc/nn_conv1x1.hip): no MIOpen / rocBLAS convolution is called;
  * the style vector (spatial mean, L2 normalisation) and the per-sample shifts of all styled units derived from it are one
    small kernel (`k_style`): torch computes nothing in the forward pass, it only owns the buffers.
"""

from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from aliby_amd import _lib
from aliby_amd.extraction.engine import _ptr, _stream_ptr

CL = torch.channels_last


def _bn_affine(bn):
    s = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
    t = bn.bias.float() - bn.running_mean.float() * s
    return s.contiguous(), t.contiguous()


class _Unit:
    """BN -> ReLU -> Conv with the BN/ReLU done by the fused kernel (scale, shift kept in fp32)."""

    def __init__(self, seq, dtype, pad_in=None):
        bn, conv = seq[0], seq[-1]
        self.scale, self.shift = _bn_affine(bn)
        w = conv.weight.detach().float()
        if pad_in is not None and w.shape[1] < pad_in:
            w = F.pad(w, (0, 0, 0, 0, 0, pad_in - w.shape[1]))
            self.scale = F.pad(self.scale, (0, pad_in - self.scale.shape[0]), value=1.0)
            self.shift = F.pad(self.shift, (0, pad_in - self.shift.shape[0]))
        self.w = w.to(dtype).contiguous(memory_format=CL)
        self.w32 = w.contiguous()  # fp32 OIHW, packed for the MFMA unit on demand
        self.wpk = None
        self.bias = conv.bias.detach().float().contiguous()  # folded into the next fused pointwise pass
        self.pad = conv.padding


class _Proj:
    """BN -> 1x1 Conv folded into a single 1x1 conv."""

    def __init__(self, seq, dtype, pad_in=None):
        bn, conv = seq[0], seq[-1]
        s, t = _bn_affine(bn)
        w = conv.weight.detach().float()  # [O, I, 1, 1]
        b = conv.bias.detach().float() + (w[:, :, 0, 0] @ t)
        w = w * s[None, :, None, None]
        if pad_in is not None and w.shape[1] < pad_in:
            w = F.pad(w, (0, 0, 0, 0, 0, pad_in - w.shape[1]))
        self.w = w.to(dtype).contiguous(memory_format=CL)
        self.w32 = w[:, :, 0, 0].contiguous()  # [O, I] fp32, BatchNorm folded: packed for the fused projection on demand
        self.wpk = None   # packed for the projection fused into conv1 (levels 0-1)
        self.wpk1 = None  # packed for the standalone 1x1 kernel
        self.bias = b.contiguous()


# (CIN, COUT, input read through the 2x upsample) instantiations of k_conv3x3
_MFMA_SHAPES = {(32, 32, False), (32, 64, False), (64, 64, False), (64, 32, True), (64, 64, True), (64, 128, False), (64, 128, True)}
_POOL_SHAPES = {(32, 32, False), (32, 64, False), (64, 64, False), (64, 128, False)}


class FusedUNet:
    @property
    def h(self):
        """Context handle of the calling thread (contexts are per thread: aliby_amd/_lib.py default_context)."""
        return self.eng.ctx.handle

    def __init__(self, net, eng, dtype=torch.bfloat16, mfma_levels=(0, 1, 2, 3)):
        assert dtype == torch.bfloat16, "the fused kernels are bf16"
        self.eng, self.dtype = eng, dtype
        self.mfma_levels = tuple(mfma_levels)
        self.fused_head = os.environ.get("ALIBY_NET_FUSED_HEAD", "1") != "0"
        self.fused_pair = os.environ.get("ALIBY_NET_FUSED_PAIR", "1") != "0"  # level 0: conv2 + conv3 of a block in one launch
        # the last block (conv2 + conv3 + output head) as one launch: built and bit-identical, but no faster than its two launches
        # (A/B/A/B 588.6 / 586.5 / 579.9 / 585.2 tiles/s), so off by default
        self.pair_head = os.environ.get("ALIBY_NET_PAIR_HEAD", "0") != "0"
        self.fused_first = os.environ.get("ALIBY_NET_FUSED_FIRST", "1") != "0"  # the first layer + conv1 + projection in one launch
        # deep levels (128 / 256 channels): one K-loop launch per convolution (csrc/nn_conv_deep.hip); 0 = the K/N-slice launches
        self.deep_kernel = os.environ.get("ALIBY_CONV_DEEP", "1") != "0"
        self.conv_stats = {}  # timing group -> [algorithmic bytes, flops] of the MFMA conv launches bracketed with events
        # every call made through self.lib is an asynchronous kernel launch: they keep the interpreter lock (see _lib.load_fast)
        self.lib = _lib.load_fast() if os.environ.get("ALIBY_FAST_LAUNCH", "1") != "0" else eng.lib
        net = net.float().eval()
        self.down = []
        for i, blk in enumerate(net.down):
            pad = 8 if i == 0 else None
            d = dict(proj=_Proj(blk.proj, dtype, pad), u=[_Unit(blk.conv[0], dtype, pad)] + [_Unit(blk.conv[k], dtype) for k in (1, 2, 3)])
            d["pb1"] = (d["proj"].bias + d["u"][1].bias).contiguous()
            d["shift1_b0"] = (d["u"][1].scale * d["u"][0].bias + d["u"][1].shift).contiguous()
            self.down.append(d)
        self.up = []
        for blk in net.up:
            d = dict(proj=_Proj(blk.proj, dtype), u=[_Unit(blk.conv0, dtype)])
            for su in (blk.conv1, blk.conv2, blk.conv3):
                u = _Unit(su.conv, dtype)
                u.full_w = su.full.weight.detach().float().t().contiguous()  # [style, C]
                u.full_b = su.full.bias.detach().float()
                d["u"].append(u)
            d["pb1"] = (d["proj"].bias + d["u"][1].bias).contiguous()
            self.up.append(d)
        # all style projections of the up path as ONE GEMM: shift[n, :] = style[n] @ (full_w * scale) + (full_b * scale + shift)
        ws, bs, off = [], [], 0
        for d in self.up:
            for k in d["u"][1:]:
                ws.append(k.full_w * k.scale[None, :])
                bs.append(k.full_b * k.scale + k.shift)
                k.style_slice = (off, off + k.scale.numel())
                off += k.scale.numel()
        self.style_w = torch.cat(ws, dim=1).contiguous()  # [style, sum C]
        self.style_b = torch.cat(bs).contiguous()
        # first layer's weights [32, cin, 3, 3], bf16-rounded like the convolution it replaces
        self.first_w = net.down[0].conv[0][-1].weight.detach().float().to(dtype).float().contiguous()
        self.out = _Unit(net.output, dtype)
        self.out_w = self.out.w32[:, :, 0, 0].to(dtype).float().contiguous()  # [O, 32], bf16-rounded like the conv it replaces
        self.cin = net.nbase[0]
        self.bytes_moved = 0  # bytes read+written by the fused pointwise launches while profiling is on

    # -------------------------------------------------------------------------------- kernels
    def _new(self, n, c, h, w):
        return torch.empty((n, c, h, w), dtype=self.dtype, device="cuda", memory_format=CL)

    def _fused(self, A, B=None, want_sum=False, act=None, shift=None, upA=False, upB=False, relu=True, bias=None):
        """A, B: [N,C,h,w] channels_last bf16.  Returns (SUM or None, ACT or None) at the output resolution."""
        n, c = A.shape[0], A.shape[1]
        H = A.shape[2] * (2 if upA else 1)
        W = A.shape[3] * (2 if upA else 1)
        S = self._new(n, c, H, W) if want_sum else None
        T = self._new(n, c, H, W) if act is not None else None
        per_sample = 0
        sh = None
        if act is not None:
            sh = shift if shift is not None else act.shift
            per_sample = self._sps(sh)
        timer = self.eng.timed("fused_pointwise")
        if timer.active:
            self.bytes_moved += 2 * (A.numel() + (B.numel() if B is not None else 0) + (n * c * H * W) * ((S is not None) + (T is not None)))
        with timer:
          _lib.check(self.lib.aliby_nn_fused_act_bf16(
            self.h, _ptr(A), _ptr(B) if B is not None else 0, _ptr(S) if S is not None else 0, _ptr(T) if T is not None else 0,
            _ptr(bias) if bias is not None else 0,
            _ptr(act.scale) if act is not None else 0, _ptr(sh) if sh is not None else 0, n, H, W, c, 1 if upA else 0,
            1 if upB else 0, 1 if relu else 0, per_sample, _stream_ptr()))
        return S, T

    @staticmethod
    def _conv(x, unit, pad=1):
        return F.conv2d(x, unit.w, None, padding=pad)  # bias is applied by the next fused pass

    def _proj(self, x, proj):
        """1x1 projection (BatchNorm folded, bias added later by the consumer): hand-written MFMA GEMM for the widths
        the kernel is built for, MIOpen otherwise."""
        n, cin, H, W = x.shape
        cout = proj.w32.shape[0]
        if cin not in (32, 64, 128, 256) or not (cout in (32, 64) or cout % 128 == 0) or proj.w32.shape[1] != cin:
            return self._conv(x, proj, pad=0)
        if proj.wpk1 is None:
            proj.wpk1 = torch.empty(cout * cin, dtype=torch.bfloat16, device="cuda")
            _lib.check(self.lib.aliby_nn_pack_conv1x1_bf16(self.h, _ptr(proj.w32), cout, cin, cin, _ptr(proj.wpk1), _stream_ptr()))
        out = self._new(n, cout, H, W)
        with self.eng.timed("conv1x1_mfma"):
            _lib.check(self.lib.aliby_nn_conv1x1_bf16(self.h, _ptr(x), _ptr(proj.wpk1), 0, _ptr(out), n, H, W, cin, cout, _stream_ptr()))
        return out

    def _unit(self, x, unit, shift=None, bias=None, res=None, res_up=False, in_up=False, pool=False):
        """conv3x3(relu(scale*x + shift)) + bias + res on the MFMA convolution unit.  One launch when the shape is
        one of the kernel's instantiations; otherwise the convolution is split along K (64 input channels per
        launch, each launch adding to the previous one through RES, in place) and along N (128 output channels
        per launch, written into channel slices of the output)."""
        n, cin, cout = x.shape[0], x.shape[1], unit.w32.shape[0]
        H, W = (x.shape[2] * 2, x.shape[3] * 2) if in_up else (x.shape[2], x.shape[3])
        out = self._new(n, cout, H, W)
        pooled = self._new(n, cout, H // 2, W // 2) if pool else None
        if self.deep_kernel and cout % 128 == 0 and cin in (64, 128, 256) and W <= 56 and (H + 1) * (W + 2) >= 226 + 2 * (W + 2):
            self._launch_deep(x, unit, out, shift, bias, res, res_up, in_up)
            if pool:
                with self.eng.timed("maxpool"):
                    _lib.check(self.lib.aliby_nn_maxpool2_bf16(self.h, _ptr(out), _ptr(pooled), n, H, W, cout, _stream_ptr()))
            return (out, pooled) if pool else out
        if (cin, cout, bool(in_up)) in (_POOL_SHAPES if pool else _MFMA_SHAPES):
            self._launch_unit(x, unit, (0, cin), (0, cout), out, shift, bias, res, res_up, in_up, pooled)
            return (out, pooled) if pool else out
        ns = 128 if cout > 128 else cout
        ks = 64 if cin > 64 else cin
        assert cin % ks == 0 and cout % ns == 0 and (ks, ns, bool(in_up)) in _MFMA_SHAPES, (cin, cout, in_up)
        for n0 in range(0, cout, ns):
            cur, cur_up = res, res_up
            for k0 in range(0, cin, ks):
                last = k0 + ks == cin
                self._launch_unit(x, unit, (k0, k0 + ks), (n0, n0 + ns), out, shift, bias if last else None, cur, cur_up, in_up,
                                  pooled if last else None)
                cur, cur_up = out, False
        return (out, pooled) if pool else out

    def _pack(self, unit):
        cout, cin = unit.w32.shape[0], unit.w32.shape[1]
        if unit.wpk is None:
            unit.wpk = {}
        key = (0, cin, 0, cout)
        if key not in unit.wpk:
            pk = torch.empty(cout * cin * 9, dtype=torch.bfloat16, device="cuda")
            _lib.check(self.lib.aliby_nn_pack_conv3x3_bf16(self.h, _ptr(unit.w32), cout, cin, cin, _ptr(pk), _stream_ptr()))
            unit.wpk[key] = pk
        return unit.wpk[key]

    def _pair(self, x, ua, ub, shift_a, shift_b, bias_a, bias_b, res, pool=False, head_out=None):
        """Two consecutive 32-channel units in one launch (aliby_nn_conv3x3_pair_bf16): the tensor between them stays in
        LDS.  With `head_out` the network's output head is taken from the second unit's accumulators and its own output is
        not written; with `pool` the next level's input is written beside the output."""
        n, cin, H, W = x.shape
        sa = ua.shift if shift_a is None else shift_a
        sb = ub.shift if shift_b is None else shift_b
        out = None if head_out is not None else self._new(n, 32, H, W)
        pooled = self._new(n, 32, H // 2, W // 2) if pool else None
        group = "conv3x3_mfma_pair"
        timer = self.eng.timed(group)
        if timer.active:
            st = self.conv_stats.setdefault(group, [0, 0])
            st[0] += 2 * (x.numel() + res.numel()) + (2 * x.numel() if out is not None else 4 * head_out.numel()) + (x.numel() // 2 if pool else 0)
            st[1] += 2 * 2 * 9 * 32 * 32 * n * H * W
        hd = self.out
        with timer:
            _lib.check(self.lib.aliby_nn_conv3x3_pair_bf16(
                self.h, _ptr(x), _ptr(self._pack(ua)), _ptr(self._pack(ub)), _ptr(out) if out is not None else 0, _ptr(ua.scale), _ptr(sa),
                self._sps(sa), _ptr(bias_a), _ptr(ub.scale), _ptr(sb), self._sps(sb), _ptr(bias_b), _ptr(res), n, H, W,
                _ptr(pooled) if pooled is not None else 0,
                _ptr(hd.scale) if head_out is not None else 0, _ptr(hd.shift) if head_out is not None else 0,
                _ptr(self.out_w) if head_out is not None else 0, _ptr(hd.bias) if head_out is not None else 0,
                self.out_w.shape[0] if head_out is not None else 0, _ptr(head_out) if head_out is not None else 0, _stream_ptr()))
        return (out, pooled) if pool else out

    def _first_pair(self, tiles, d):
        """The first layer, the block's second unit and its 1x1 projection in one launch (aliby_nn_first_pair_bf16): neither c0 nor
        the raw bf16 copy of the tiles is written to HBM.  Same bits as first_conv -> _unit_proj."""
        n, cin, H, W = tiles.shape
        u, proj = d["u"], d["proj"]
        wpk1 = self._pack(u[1])
        if proj.wpk is None:
            proj.wpk = torch.empty(32 * 16, dtype=torch.bfloat16, device="cuda")
            _lib.check(self.lib.aliby_nn_pack_conv1x1_bf16(self.h, _ptr(proj.w32), 32, proj.w32.shape[1], 16, _ptr(proj.wpk), _stream_ptr()))
        out = self._new(n, 32, H, W)
        group = "conv3x3_mfma_first_pair"
        timer = self.eng.timed(group)
        if timer.active:
            st = self.conv_stats.setdefault(group, [0, 0])
            st[0] += 4 * tiles.numel() + 2 * out.numel()
            st[1] += 2 * (9 * cin + 9 * 32 + cin) * 32 * n * H * W
        with timer:
            _lib.check(self.lib.aliby_nn_first_pair_bf16(
                self.h, _ptr(tiles), n, cin, H, W, _ptr(u[0].scale), _ptr(u[0].shift), _ptr(self.first_w), _ptr(wpk1), _ptr(u[1].scale),
                _ptr(d["shift1_b0"]), _ptr(d["pb1"]), _ptr(proj.wpk), _ptr(out), _stream_ptr()))
        return out

    def _unit_head(self, x, unit, shift, bias, res, y):
        """The network's last unit with the output head in its epilogue (aliby_nn_conv3x3_head_bf16): y float32 [N,O,H,W]
        is written from the accumulators and the unit's own bf16 output — which nothing else reads — is not written."""
        n, cin, H, W = x.shape
        cout = unit.w32.shape[0]
        if unit.wpk is None:
            unit.wpk = {}
        key = (0, cin, 0, cout)
        if key not in unit.wpk:
            pk = torch.empty(cout * cin * 9, dtype=torch.bfloat16, device="cuda")
            _lib.check(self.lib.aliby_nn_pack_conv3x3_bf16(self.h, _ptr(unit.w32), cout, cin, cin, _ptr(pk), _stream_ptr()))
            unit.wpk[key] = pk
        sh = unit.shift if shift is None else shift
        group = "conv3x3_mfma_head"  # its own timing group: fewer bytes and more epilogue arithmetic than the plain unit
        timer = self.eng.timed(group)
        if timer.active:
            st = self.conv_stats.setdefault(group, [0, 0])
            st[0] += 2 * (x.numel() + res.numel()) + 4 * y.numel()
            st[1] += 2 * (9 * cin + y.shape[1]) * cout * n * H * W
        with timer:
            _lib.check(self.lib.aliby_nn_conv3x3_head_bf16(
                self.h, _ptr(x), _ptr(unit.wpk[key]), 0, _ptr(unit.scale), _ptr(sh), self._sps(sh), _ptr(bias), _ptr(res), 0, n, H, W,
                cin, cout, _ptr(self.out.scale), _ptr(self.out.shift), _ptr(self.out_w), _ptr(self.out.bias), self.out_w.shape[0], _ptr(y),
                _stream_ptr()))

    def _unit_proj(self, x, unit, shift, bias, x_in, proj):
        """conv3x3(relu(scale*x + shift)) + bias + proj(x_in) in ONE launch (aliby_nn_conv3x3_proj_bf16): the residual
        block's 1x1 projection of its raw input is a few extra k-steps, its output never touches HBM."""
        n, cin, H, W = x.shape
        cout = unit.w32.shape[0]
        pc_pad = 16 if cin == 32 else 32
        if unit.wpk is None:
            unit.wpk = {}
        key = (0, cin, 0, cout)
        if key not in unit.wpk:
            pk = torch.empty(cout * cin * 9, dtype=torch.bfloat16, device="cuda")
            _lib.check(self.lib.aliby_nn_pack_conv3x3_bf16(self.h, _ptr(unit.w32), cout, cin, cin, _ptr(pk), _stream_ptr()))
            unit.wpk[key] = pk
        if proj.wpk is None:
            proj.wpk = torch.empty(cout * pc_pad, dtype=torch.bfloat16, device="cuda")
            _lib.check(self.lib.aliby_nn_pack_conv1x1_bf16(self.h, _ptr(proj.w32), cout, proj.w32.shape[1], pc_pad, _ptr(proj.wpk), _stream_ptr()))
        sh = unit.shift if shift is None else shift
        out = self._new(n, cout, H, W)
        group = "conv3x3_mfma"
        timer = self.eng.timed(group)
        if timer.active:
            st = self.conv_stats.setdefault(group, [0, 0])
            st[0] += 2 * (x.numel() + out.numel() + x_in.numel())
            st[1] += 2 * (9 * cin + x_in.shape[1]) * cout * n * H * W
        with timer:
            _lib.check(self.lib.aliby_nn_conv3x3_proj_bf16(
                self.h, _ptr(x), _ptr(unit.wpk[key]), _ptr(out), _ptr(unit.scale), _ptr(sh), self._sps(sh), _ptr(bias), n, H, W, cin, cout,
                _ptr(x_in), _ptr(proj.wpk), x_in.shape[1], _stream_ptr()))
        return out

    def _launch_deep(self, x, unit, out, shift, bias, res, res_up, in_up):
        """One launch of the K-loop kernel (aliby_nn_conv3x3_deep_bf16): fp32 accumulation over the whole 9 * CIN reduction."""
        n, cin, cout = x.shape[0], x.shape[1], out.shape[1]
        H, W = out.shape[2], out.shape[3]
        sh = unit.shift if shift is None else shift
        group = "conv3x3_mfma_deep"
        timer = self.eng.timed(group)
        if timer.active:  # algorithmic bytes: input + residual read once, output written once (no partial sums any more)
            st = self.conv_stats.setdefault(group, [0, 0])
            st[0] += 2 * (x.numel() + out.numel() + (res.numel() if res is not None else 0))
            st[1] += 2 * 9 * cin * cout * n * H * W
        with timer:
            _lib.check(self.lib.aliby_nn_conv3x3_deep_bf16(
                self.h, _ptr(x), _ptr(self._pack(unit)), _ptr(out), _ptr(unit.scale), _ptr(sh), self._sps(sh),
                _ptr(bias) if bias is not None else 0, _ptr(res) if res is not None else 0, 1 if res_up else 0, n, H, W, cin, cout,
                1 if in_up else 0, _stream_ptr()))

    def _launch_unit(self, x, unit, kslice, nslice, out, shift, bias, res, res_up, in_up, pooled):
        n, ctot, cout_tot = x.shape[0], x.shape[1], out.shape[1]
        (k0, k1), (n0, n1) = kslice, nslice
        cin, cout = k1 - k0, n1 - n0
        H, W = out.shape[2], out.shape[3]
        assert pooled is None or (cin, cout, bool(in_up)) in _POOL_SHAPES
        if unit.wpk is None:
            unit.wpk = {}
        key = (k0, k1, n0, n1)
        if key not in unit.wpk:
            w = unit.w32[n0:n1, k0:k1].contiguous()
            pk = torch.empty(cout * cin * 9, dtype=torch.bfloat16, device="cuda")
            _lib.check(self.lib.aliby_nn_pack_conv3x3_bf16(self.h, _ptr(w), cout, w.shape[1], cin, _ptr(pk), _stream_ptr()))
            unit.wpk[key] = pk
        sh = unit.shift if shift is None else shift
        sh = sh[..., k0:k1]
        scale = unit.scale[k0:k1]
        # two timing groups: the 32/64-output-channel launches are HBM-bound (144-288 FLOP/B), the 128-output-channel
        # ones of the deep levels (576+ FLOP/B) are bound by the matrix cores
        group = "conv3x3_mfma_deep" if cout >= 128 else "conv3x3_mfma"
        timer = self.eng.timed(group)
        if timer.active:  # algorithmic bytes / flops of exactly the launches that are bracketed with events
            px_out = n * H * W
            st = self.conv_stats.setdefault(group, [0, 0])
            st[0] += 2 * (x.numel() * cin // ctot + px_out * cout + (res.numel() * cout // cout_tot if res is not None else 0)
                          + (px_out // 4 * cout if pooled is not None else 0))
            st[1] += 2 * 9 * cin * cout * px_out
        with timer:
            _lib.check(self.lib.aliby_nn_conv3x3_bf16(
                self.h, _ptr(x), _ptr(unit.wpk[key]), _ptr(out), _ptr(scale), _ptr(sh), self._sps(sh),
                _ptr(bias[n0:n1]) if bias is not None else 0, _ptr(res) if res is not None else 0, 1 if res_up else 0, n, H, W, cin, cout,
                1 if in_up else 0, ctot, k0, cout_tot, n0, _ptr(pooled) if pooled is not None else 0, _stream_ptr()))

    def _style_shift(self, unit):
        lo, hi = unit.style_slice
        return self._style_all[:, lo:hi]

    @staticmethod
    def _sps(sh):
        """shift_per_sample argument of the kernels: 0 shared, else the row stride in floats."""
        return 0 if sh.ndim == 1 else sh.stride(0)

    def _down_mfma(self, i, d, x_raw, x_act, x1=None):
        """Residual down block on the MFMA unit: 4 launches, no pointwise passes (x1 given: the first two units already ran)."""
        u = d["u"]
        if x1 is not None:
            pool = i + 1 < len(self.down) and u[3].w32.shape[0] <= 128
            if self.fused_pair and tuple(u[2].w32.shape[:2]) == (32, 32) and tuple(u[3].w32.shape[:2]) == (32, 32):
                return self._pair(x1, u[2], u[3], None, None, u[2].bias, u[3].bias, x1, pool=pool)
            c2 = self._unit(x1, u[2], bias=u[2].bias)
            return self._unit(c2, u[3], bias=u[3].bias, res=x1, pool=pool)
        fuse_proj = (u[1].w32.shape[1], u[1].w32.shape[0]) in ((32, 32), (64, 64))  # projection rides in conv1's launch
        p = None if fuse_proj else self._proj(x_raw, d["proj"])
        if i == 0:  # 2 -> 32 channels: K = 18 is too thin for the matrix cores; c0 comes from the first-layer kernel (or
            c0 = x_act if self._first_c0 else self._conv(x_act, u[0])  # MIOpen), its bias rides in the next shift
            sh1 = d["shift1_b0"]
        else:
            c0 = self._unit(x_raw, u[0], bias=u[0].bias)
            sh1 = None
        if fuse_proj:
            x1 = self._unit_proj(c0, u[1], sh1, d["pb1"], x_raw, d["proj"])
        else:
            x1 = self._unit(c0, u[1], shift=sh1, bias=d["pb1"], res=p)
        pool = i + 1 < len(self.down) and u[3].w32.shape[0] <= 128
        if self.fused_pair and tuple(u[2].w32.shape[:2]) == (32, 32) and tuple(u[3].w32.shape[:2]) == (32, 32):
            return self._pair(x1, u[2], u[3], None, None, u[2].bias, u[3].bias, x1, pool=pool)
        c2 = self._unit(x1, u[2], bias=u[2].bias)
        return self._unit(c2, u[3], bias=u[3].bias, res=x1, pool=pool)  # (x2, maxpool(x2))

    def _up_mfma(self, d, x, skip, style, up=True, head_out=None):
        u = d["u"]
        p_low = self._proj(x, d["proj"])  # 1x1 at the low resolution, read through the upsample as a residual
        c0s = self._unit(x, u[0], bias=u[0].bias, res=skip, in_up=up)  # wider than one launch holds: split along K / N
        sh = [self._style_shift(k) for k in u[1:]]  # [N,C] views of the batched style projection
        x1 = self._unit(c0s, u[1], shift=sh[0], bias=d["pb1"], res=p_low, res_up=up)
        # (round 2: with the head summed on the vector unit in its consumers' epilogue the pair measured 1139 us against 454 + 610 for
        # conv2 and the unit-with-head launch; round 3: the head is two MFMA k-steps (csrc/nn_conv.hip, head_apply), the pair with
        # the head measures 944 us against 412 + 512: a tie, so the two launches stay the default — ALIBY_NET_PAIR_HEAD=1 for the pair)
        pair_shapes = tuple(u[2].w32.shape[:2]) == (32, 32) and tuple(u[3].w32.shape[:2]) == (32, 32)
        if self.fused_pair and pair_shapes and (head_out is None or self.pair_head):
            return self._pair(x1, u[2], u[3], sh[1], sh[2], u[2].bias, u[3].bias, x1, head_out=head_out)
        c2 = self._unit(x1, u[2], shift=sh[1], bias=u[2].bias)
        if head_out is not None:
            self._unit_head(c2, u[3], sh[2], u[3].bias, x1, head_out)
            return None
        return self._unit(c2, u[3], shift=sh[2], bias=u[3].bias, res=x1)

    # -------------------------------------------------------------------------------- forward
    @torch.no_grad()
    def __call__(self, tiles: torch.Tensor, out: torch.Tensor | None = None):
        """tiles float32 [N, cin, H, W] (contiguous NCHW) -> (y float32 [N,3,H,W], style float32 [N,256]).
        `out`, when given, is the contiguous float32 [N,3,H,W] buffer y is written into (no copy afterwards)."""
        n, cin, H, W = tiles.shape
        d0 = self.down[0]
        self._first_c0 = 0 in self.mfma_levels and cin <= 2
        u0 = d0["u"]
        x1_first = None
        if (self._first_c0 and self.fused_first and tuple(u0[1].w32.shape[:2]) == (32, 32) and u0[0].w32.shape[0] == 32
                and d0["proj"].w32.shape[0] == 32):
            x1_first = self._first_pair(tiles.contiguous(), d0)
        raw = self._new(n, 8, H, W) if x1_first is None else None
        if x1_first is not None:
            act = None
        elif self._first_c0:  # first-layer kernel: `act` IS c0 = conv3x3(relu(bn(x))) (32 channels), no MIOpen call
            act = self._new(n, 32, H, W)
            with self.eng.timed("first_conv"):
                _lib.check(self.lib.aliby_nn_first_conv_bf16(self.h, _ptr(tiles), n, cin, H, W, _ptr(d0["u"][0].scale), _ptr(d0["u"][0].shift),
                                                             _ptr(self.first_w), _ptr(raw), _ptr(act), _stream_ptr()))
        else:
            act = self._new(n, 8, H, W)
            _lib.check(self.lib.aliby_nn_tiles_to_nhwc8_bf16(self.h, _ptr(tiles), n, cin, H, W, _ptr(d0["u"][0].scale),
                                                             _ptr(d0["u"][0].shift), _ptr(raw), _ptr(act), _stream_ptr()))
        feats, pooled = [], None
        x_raw, x_act = raw, act
        for i, d in enumerate(self.down):
            u = d["u"]
            if i > 0:
                x_raw = pooled if pooled is not None else F.max_pool2d(feats[-1], 2, 2)
                pooled = None
            if i in self.mfma_levels:
                x2 = self._down_mfma(i, d, x_raw, x_act, x1=x1_first if i == 0 else None)
                if isinstance(x2, tuple):
                    x2, pooled = x2  # the block's last convolution also wrote the next level's input
                feats.append(x2)
                continue
            if i > 0:
                _, x_act = self._fused(x_raw, act=u[0])
            p = self._conv(x_raw, d["proj"], pad=0)
            c0 = self._conv(x_act, u[0])
            _, a1 = self._fused(c0, act=u[1], bias=u[0].bias)
            c1 = self._conv(a1, u[1])
            x1, a2 = self._fused(p, c1, want_sum=True, act=u[2], bias=d["pb1"])
            c2 = self._conv(a2, u[2])
            _, a3 = self._fused(c2, act=u[3], bias=u[2].bias)
            c3 = self._conv(a3, u[3])
            x2, _ = self._fused(x1, c3, want_sum=True, bias=u[3].bias)
            feats.append(x2)
        deep = feats[-1]
        style = torch.empty((n, deep.shape[1]), dtype=torch.float32, device="cuda")
        self._style_all = torch.empty((n, self.style_b.numel()), dtype=torch.float32, device="cuda")  # [N, sum C]
        with self.eng.timed("style"):
            _lib.check(self.lib.aliby_nn_style_bf16(self.h, _ptr(deep), n, deep.shape[2], deep.shape[3], deep.shape[1], _ptr(self.style_w),
                                                    _ptr(self.style_b), self.style_b.numel(), _ptr(style), _ptr(self._style_all), _stream_ptr()))
        x, up = feats[-1], False
        y = out if out is not None else torch.empty((n, self.out_w.shape[0], H, W), dtype=torch.float32, device="cuda")
        assert y.is_contiguous() and y.dtype == torch.float32 and tuple(y.shape) == (n, self.out_w.shape[0], H, W)
        for i in range(len(self.up) - 1, -1, -1):
            d = self.up[i]
            u = d["u"]
            skip = feats[i]
            if i in self.mfma_levels:
                # the last unit carries the output head in its epilogue (ALIBY_NET_FUSED_HEAD=0: separate k_out_head launch)
                fuse_head = (i == 0 and self.fused_head and u[3].w32.shape[0] == 32 and u[3].w32.shape[1] == 32
                             and self.out_w.shape[0] <= 3)
                x = self._up_mfma(d, x, skip, style, up, head_out=y if fuse_head else None)
                if fuse_head:
                    return y, style
                up = True
                continue
            p_low = self._conv(x, d["proj"], pad=0)                 # at x's resolution; read through the upsample below
            _, a0 = self._fused(x, act=u[0], upA=up)
            c0 = self._conv(a0, u[0])
            sh = [self._style_shift(k) for k in u[1:]]  # [N,C] views of the batched style projection
            _, a1 = self._fused(c0, skip, act=u[1], shift=sh[0], bias=u[0].bias)
            c1 = self._conv(a1, u[1])
            x1, a2 = self._fused_up_sum(p_low, c1, up, u[2], sh[1], d["pb1"])
            c2 = self._conv(a2, u[2])
            _, a3 = self._fused(c2, act=u[3], shift=sh[2], bias=u[2].bias)
            c3 = self._conv(a3, u[3])
            x, _ = self._fused(x1, c3, want_sum=True, bias=u[3].bias)
            up = True
        if x.shape[1] == 32:
            with self.eng.timed("out_head"):
                _lib.check(self.lib.aliby_nn_out_head_bf16(self.h, _ptr(x), _ptr(self.out.scale), _ptr(self.out.shift), _ptr(self.out_w),
                                                           _ptr(self.out.bias), n, H, W, 32, self.out_w.shape[0], _ptr(y), _stream_ptr()))
        else:
            _, a = self._fused(x, act=self.out)
            yb = self._conv(a, self.out, pad=0)  # [N,3,H,W] channels_last == NHWC with 3 channels
            _lib.check(self.lib.aliby_nn_nhwc_to_nchw_f32(self.h, _ptr(yb), n, H, W, yb.shape[1], yb.shape[1], _ptr(self.out.bias),
                                                          _ptr(y), _stream_ptr()))
        return y, style

    def _fused_up_sum(self, p_low, c1, up, unit, shift, bias):
        """x1 = upsample?(p_low) + c1 ; a2 = relu(bn(x1) + style shift).  A must be the full-resolution operand
        for the output shape, so the (possibly low-res) projection goes in slot B."""
        n, c, H, W = c1.shape
        S, T = self._new(n, c, H, W), self._new(n, c, H, W)
        timer = self.eng.timed("fused_pointwise")
        if timer.active:
            self.bytes_moved += 2 * (c1.numel() + p_low.numel() + 2 * c1.numel())
        with timer:
          _lib.check(self.lib.aliby_nn_fused_act_bf16(
            self.h, _ptr(c1), _ptr(p_low), _ptr(S), _ptr(T), _ptr(bias), _ptr(unit.scale), _ptr(shift), n, H, W, c, 0,
            1 if up else 0, 1, self._sps(shift),
            _stream_ptr()))
        return S, T
