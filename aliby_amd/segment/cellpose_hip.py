"""
`CellposeModel` for MI355X: the object `dispatch_segmenter(kind="cellpose")` builds
(reference: `CellposeModel(gpu=..., device=...)` at src/aliby/segment/dispatch.py:161,171-175 and its
`.eval(...)` call at 208-215).

Per batch of tiles [F,Y,X]:
    normalize99 (HIP) -> zero-pad + 224-px overlapped tiles (HIP) -> residual U-Net forward (PyTorch-ROCm,
    the only torch compute) -> taper-weighted blending + un-pad (HIP) -> dynamics: flow following,
    seeds, labels, flow-error QC, hole filling / small-mask removal (HIP).

`flows_override(img_u16 [F,Y,X]) -> (dP [F,2,Y,X], cellprob [F,Y,X])` replaces the network's output
(the network still runs when `run_network_with_override=True`): Cellpose weights cannot be fetched
offline, so tests and bench.py feed analytic flows derived from the synthetic
ground truth into the same dynamics.
"""

from __future__ import annotations

import os
import warnings

import numpy as np
import torch

from aliby_amd import _lib
from aliby_amd.extraction.engine import FeatureEngine, _ptr, _stream_ptr
from aliby_amd.segment import dynamics


def tile_starts(L, bsize=224, tile_overlap=0.1):
    tile_overlap = min(0.5, max(0.05, tile_overlap))
    b = min(bsize, L)
    n = 1 if L <= bsize else int(np.ceil((1.0 + 2 * tile_overlap) * L / bsize))
    return np.linspace(0, L - b, n).astype(np.int32), int(b)


def pad_amounts(Ly, Lx, div=16, extra=1):
    out = []
    for L in (Ly, Lx):
        lpad = int(div * np.ceil(L / div) - L)
        out += [extra * div // 2 + lpad // 2, extra * div // 2 + lpad - lpad // 2]
    return out  # ypad1, ypad2, xpad1, xpad2


def taper_mask(by, bx, sig=7.5):
    """cellpose's blending weights of a by x bx tile: the central crop of a square sigmoid window of side max(224, by, bx)."""
    bs = max(224, by, bx)
    xm = np.arange(bs)
    xm = np.abs(xm - xm.mean())
    m = 1 / (1 + np.exp((xm - (bs / 2 - 20)) / sig))

    def crop(b):
        return m[bs // 2 - b // 2 : bs // 2 + b // 2 + b % 2]

    return (crop(bx) * crop(by)[:, np.newaxis]).astype(np.float32)


class CellposeModel:
    def __init__(self, gpu=True, device=None, pretrained_model=None, net=None, net_dtype=None, seed=0,
                 flows_override=None, run_network_with_override=False, bsize=224, tile_overlap=0.1, batch_size=288,
                 use_bfloat16=True, model_type=None, diam_mean=30.0, nchan=2, fused=True):
        """net_dtype: "bfloat16" (the default) runs the network on the hand-written MFMA kernels (segment/fused_unet.py); "float32" /
        "float16" run the same module through PyTorch's own convolutions (MIOpen: a numerical reference, tens of seconds of kernel
        search at the first batch, several times slower).  `use_bfloat16` is cellpose's spelling of the same switch — the reference
        builds `CellposeModel(gpu=..., device=...)` with cellpose 4's defaults (dispatch.py:168-172), where it is True — and applies
        when net_dtype is not given.  batch_size: 224-pixel tiles per network forward (cellpose's eval default is 8, for GPUs with
        a few GB): 288 tiles keep every launch of the forward above 4000 workgroup tiles and cost ~15 GB of the 288 GB — measured
        574 / 594 / 607 FOV tiles/s end to end at 64 / 128 / 288."""
        if net_dtype is None:
            net_dtype = "bfloat16" if use_bfloat16 else "float32"
        if model_type is not None:
            # cellpose resolves a model NAME by downloading its checkpoint; there is no such store here
            raise ValueError(f"model_type={model_type!r}: built-in models are files cellpose downloads — pass the checkpoint's path "
                             "as pretrained_model")
        if int(nchan) != 2:
            raise NotImplementedError(f"nchan={nchan}: the network built here is CPnet's two-channel residual U-Net")
        self.diam_mean = float(diam_mean)
        if not gpu or not torch.cuda.is_available():
            raise _lib.AlibyHipError("CellposeModel (HIP) needs a GPU: there is no CPU fallback in this build")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.eng = FeatureEngine(self.device.index if self.device.index is not None else torch.cuda.current_device())
        self.bsize, self.tile_overlap, self.batch_size = int(bsize), float(tile_overlap), int(batch_size)
        self.net_dtype = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}[net_dtype]
        self.flows_override = flows_override
        self.run_network_with_override = run_network_with_override
        self.pretrained = pretrained_model is not None or net is not None
        if net is None:
            from aliby_amd.segment.unet import build_network

            net = build_network(seed=seed, pretrained_model=pretrained_model, device=self.device)
        # MIOpen exhaustive find per convolution shape (one-off cost at the first batch): -27 % on the forward
        torch.backends.cudnn.benchmark = True
        self.net = net.to(self.device).eval()
        self.fused = None
        if self.net_dtype == torch.bfloat16 and fused:
            from aliby_amd.segment.fused_unet import FusedUNet

            self.fused = FusedUNet(self.net, self.eng)  # keeps fp32 master weights, folds BN, bf16 execution
        if self.net_dtype != torch.float32:
            self.net = self.net.to(self.net_dtype)
        self.net = self.net.to(memory_format=torch.channels_last)
        if not self.pretrained and flows_override is None:
            warnings.warn("CellposeModel: no pretrained weights available offline; the network is randomly "
                          "initialised and its masks are not meaningful (SURVEY.md §0.5).")
        self._geom_cache = {}
        # Optional hipGraph replay of the network forward (~230 dependent launches per batch; the whole forward is kernels
        # of this library on one stream with static shapes): captured once per batch shape, only when no per-kernel event
        # timing is requested.  OFF by default: measured on config 2 it is 4 % SLOWER than eager launches (453 vs 473
        # tiles/s) — the launches are long enough that the host stays ahead, and replay needs the batch copied into and
        # out of the graph's static buffers.  ALIBY_NET_GRAPH=1 turns it on; a refused capture falls back to eager.
        self.use_graph = os.environ.get("ALIBY_NET_GRAPH", "0") == "1"
        self._graphs = {}

    def _forward_batch(self, tiles_b: torch.Tensor, out_b: torch.Tensor) -> None:
        """One network forward on a batch of tiles, written into out_b; graph replay when allowed."""
        if not (self.use_graph and self.eng.profile is None):
            self.fused(tiles_b, out=out_b)
            return
        key = tuple(tiles_b.shape)
        entry = self._graphs.get(key)
        if entry is None:
            static_in, static_out = torch.empty_like(tiles_b), torch.empty_like(out_b)
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):  # warm-up: weight packing, function attributes, allocator pools
                    static_in.copy_(tiles_b)
                    self.fused(static_in, out=static_out)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self.fused(static_in, out=static_out)
                entry = (graph, static_in, static_out)
            except Exception as exc:  # capture refused: stay eager for this model
                warnings.warn(f"network graph capture failed ({exc!r}); running eager")
                self.use_graph = False
                torch.cuda.synchronize()
                self.fused(tiles_b, out=out_b)
                return
            self._graphs[key] = entry
        graph, static_in, static_out = entry
        static_in.copy_(tiles_b)
        graph.replay()
        out_b.copy_(static_out)

    # ------------------------------------------------------------------ reference-side helpers
    def select_and_project(self, pixels, channel: int) -> torch.Tensor:
        """pixels [F,C,Z,Y,X] (host uint16 array or device tensor) -> device uint16 [F,Y,X]:
        channel select + max over Z (dispatch.py:192,199-206)."""
        if isinstance(pixels, torch.Tensor):
            dev = pixels.to(self.device)
        else:
            pixels = np.ascontiguousarray(pixels)
            if pixels.dtype in (np.uint8, np.bool_):
                pixels = pixels.astype(np.uint16)
            elif pixels.dtype != np.uint16:
                pixels = pixels.astype(np.float32)
            dev = torch.from_numpy(pixels).to(self.device)
        dev = dev.contiguous()
        F, C, Z, Y, X = dev.shape
        if dev.dtype != torch.uint16:
            # float stacks (not produced by any of the reference's fixtures, which are uint16: SURVEY §3.3): the select and
            # the Z maximum are two strided reads, done with device tensor ops rather than a dedicated kernel
            dev = dev.to(torch.float32)
            return (dev[:, int(channel)].amax(dim=1) if Z > 1 else dev[:, int(channel), 0]).contiguous()
        out = torch.empty((F, Y, X), dtype=torch.uint16, device=self.device)
        with self.eng.timed("select_project"):
            _lib.check(self.eng.lib.aliby_select_project_u16(self.eng.ctx.handle, _ptr(dev), F, C, Z, Y, X, int(channel),
                                                             _ptr(out), _stream_ptr()))
        return out

    def max_project_and_relabel(self, labels: torch.Tensor) -> torch.Tensor:
        """labels [F,Y,X] -> [Y,X]: max over axis 0 then relabel_sequential (dispatch.py:218-223)."""
        F, Y, X = labels.shape
        out = torch.empty((1, Y, X), dtype=torch.uint16, device=labels.device)
        from aliby_amd import _lib as L

        _lib.check(self.eng.lib.aliby_reduce_z(self.eng.ctx.handle, _ptr(labels.contiguous()), L.U16, 1, F, Y * X, L.RED_MAX,
                                               _ptr(out), L.U16, _stream_ptr()))
        self.eng.relabel_sequential(out)
        return out[0]

    def count_labels(self, labels: torch.Tensor) -> int:
        mx = np.zeros(1, np.int32)
        lab = labels.reshape(1, -1, labels.shape[-1])
        _lib.check(self.eng.lib.aliby_label_max(self.eng.ctx.handle, _ptr(lab), 1, lab.shape[1], lab.shape[2], _ptr(mx),
                                                _stream_ptr()))
        return int(mx[0])

    # ------------------------------------------------------------------------------ network leg
    def _geometry(self, Y, X, bsize=None, tile_overlap=None):
        bsize = self.bsize if bsize is None else int(bsize)
        tile_overlap = self.tile_overlap if tile_overlap is None else float(tile_overlap)
        key = (Y, X, bsize, tile_overlap)
        if key not in self._geom_cache:
            yp1, yp2, xp1, xp2 = pad_amounts(Y, X)
            Ly, Lx = Y + yp1 + yp2, X + xp1 + xp2
            ys, by = tile_starts(Ly, bsize, tile_overlap)
            xs, bx = tile_starts(Lx, bsize, tile_overlap)
            self._geom_cache[key] = dict(
                ypad1=yp1, xpad1=xp1, Ly=Ly, Lx=Lx, by=by, bx=bx, ny=len(ys), nx=len(xs),
                ys=torch.from_numpy(ys).to(self.device), xs=torch.from_numpy(xs).to(self.device),
                taper=torch.from_numpy(taper_mask(by, bx)).to(self.device),
            )
        return self._geom_cache[key]

    def normalize(self, img_u16: torch.Tensor) -> torch.Tensor:
        F, Y, X = img_u16.shape
        if img_u16.dtype != torch.uint16:
            return self._normalize_float(img_u16)
        out = torch.empty((F, Y, X), dtype=torch.float32, device=self.device)
        pct = torch.empty((F, 2), dtype=torch.float64, device=self.device)
        with self.eng.timed("normalize99"):
            _lib.check(self.eng.lib.aliby_normalize99_u16(self.eng.ctx.handle, _ptr(img_u16), F, Y, X, 1.0, 99.0, _ptr(out),
                                                          _ptr(pct), _stream_ptr()))
        return out

    def _normalize_float(self, img: torch.Tensor) -> torch.Tensor:
        """normalize99 for float images: exact order statistics by a device sort per image, numpy.percentile's linear
        interpolation in float64, then the same (x - p1) / (p99 - p1) -> float32 rule as the uint16 kernel."""
        F, Y, X = img.shape
        flat = img.reshape(F, -1).to(torch.float64)
        srt = torch.sort(flat, dim=1).values
        n = srt.shape[1]
        pct = []
        for q in (1.0, 99.0):
            pos = (n - 1) * (q / 100.0)
            lo = int(np.floor(pos))
            hi = min(lo + 1, n - 1)
            a, b = srt[:, lo], srt[:, hi]
            pct.append(a + (b - a) * (pos - lo))
        x01, x99 = pct[0][:, None], pct[1][:, None]
        out = torch.where(x99 - x01 > 1e-3, (flat - x01) / (x99 - x01), torch.zeros_like(flat))
        return out.to(torch.float32).reshape(F, Y, X)

    def run_network(self, img_u16: torch.Tensor, normalize: bool = True, bsize=None, tile_overlap=None, batch_size=None):
        """uint16 [F,Y,X] -> (dP float32 [F,2,Y,X], cellprob float32 [F,Y,X]) through the U-Net.
        bsize / tile_overlap / batch_size: this call's tile geometry and tiles per forward (default: the model's)."""
        F, Y, X = img_u16.shape
        g = self._geometry(Y, X, bsize, tile_overlap)
        batch_size = self.batch_size if batch_size is None else max(1, int(batch_size))
        lib, h = self.eng.lib, self.eng.ctx.handle
        # normalize=False: the raw values go to the network as float32, as cellpose does with its `normalize` switch off
        norm = self.normalize(img_u16) if normalize else (
            img_u16.to(torch.int32).to(torch.float32) if img_u16.dtype == torch.uint16 else img_u16.to(torch.float32)).contiguous()
        ntiles = F * g["ny"] * g["nx"]
        tiles = torch.empty((ntiles, 2, g["by"], g["bx"]), dtype=torch.float32, device=self.device)
        with self.eng.timed("make_tiles"):
            _lib.check(lib.aliby_make_tiles(h, _ptr(norm), F, Y, X, g["ypad1"], g["xpad1"], g["Ly"], g["Lx"], g["by"],
                                            g["bx"], g["ny"], g["nx"], _ptr(g["ys"]), _ptr(g["xs"]), 2, _ptr(tiles),
                                            _stream_ptr()))
        yt = torch.empty((ntiles, 3, g["by"], g["bx"]), dtype=torch.float32, device=self.device)
        with self.eng.timed("unet_forward"), torch.no_grad():
            for i in range(0, ntiles, batch_size):
                if self.fused is not None and g["by"] % 8 == 0 and g["bx"] % 8 == 0:
                    self._forward_batch(tiles[i : i + batch_size], yt[i : i + batch_size])  # written in place
                else:
                    xb = tiles[i : i + batch_size].to(self.net_dtype).contiguous(memory_format=torch.channels_last)
                    yb, _ = self.net(xb)
                    yt[i : i + batch_size] = yb.to(torch.float32)
        dP = torch.empty((F, 2, Y, X), dtype=torch.float32, device=self.device)
        prob = torch.empty((F, Y, X), dtype=torch.float32, device=self.device)
        with self.eng.timed("average_tiles"):
            _lib.check(lib.aliby_average_tiles(h, _ptr(yt), F, Y, X, g["ypad1"], g["xpad1"], g["Ly"], g["Lx"], g["by"],
                                               g["bx"], g["ny"], g["nx"], _ptr(g["ys"]), _ptr(g["xs"]), _ptr(g["taper"]),
                                               _ptr(dP), _ptr(prob), _stream_ptr()))
        return dP, prob

    # ------------------------------------------------------------------------------------- eval
    # cellpose's other `eval` keywords (3.1 / 4.0 signatures) and the values at which they change nothing here: a caller who
    # passes one of these goes through; any other value asks for something that is not built and raises instead of being dropped
    _EVAL_NEUTRAL = {
        "resample": (True,), "channels": (None, [0, 0], (0, 0)), "channel_axis": (None,), "invert": (False,), "rescale": (None, 1, 1.0),
        "diameter": (None, 0, 0.0), "anisotropy": (None,), "flow3D_smooth": (0,), "augment": (False,), "compute_masks": (True,),
        "progress": (None,), "interp": (True,), "dP_smooth": (0,),
    }

    def eval(self, x, do_3D=False, stitch_threshold=0.0, normalize=True, z_axis=None, niter=None,
             flow_threshold=0.4, cellprob_threshold=0.0, min_size=15, max_size_fraction=0.4, bsize=None, tile_overlap=None,
             batch_size=None, **other):
        """x: uint16 [F,Y,X] (device tensor or host array) -> (masks, flows, styles) like cellpose.
        masks is a device uint16 tensor, [Y,X] when F == 1 ("Cellpose squeezes dims"), else [F,Y,X].
        bsize / tile_overlap / batch_size: cellpose's per-call tile size, overlap and tiles per forward (default: the model's own
        224 / 0.1 / 288; results do not depend on batch_size).  Keywords of cellpose's eval that are not built (diameter / rescale
        resizing, invert, augment, ...) raise NotImplementedError unless they carry their neutral value; unknown ones TypeError."""
        if other.get("diameter") not in (None, 0):
            # diameter == the model's diam_mean (30 px for the cyto family, the checkpoint's own value when it carries one) is a
            # rescale factor of one; any other diameter would resize the image before the network, which is not built
            mean = (getattr(getattr(self, "net", None), "diam", None) or {}).get("diam_mean") or self.diam_mean
            if abs(float(other["diameter"]) - float(mean)) < 1e-6:
                other = {**other, "diameter": None}
        for k, v in other.items():
            if k not in self._EVAL_NEUTRAL:
                raise TypeError(f"eval() got an unexpected keyword argument {k!r}")
            if not any(v is n or (n is not None and not isinstance(v, bool) and v == n) or (isinstance(n, bool) and v is n)
                       for n in self._EVAL_NEUTRAL[k]):
                raise NotImplementedError(f"eval({k}={v!r}) is not built: only {self._EVAL_NEUTRAL[k]} (no effect) are accepted")
        if do_3D:
            raise NotImplementedError("do_3D: hand the planes of the stack to eval as a batch and stitch them (segment/dispatch.py does)")
        if not isinstance(x, torch.Tensor):
            x = np.ascontiguousarray(x)
            x = torch.from_numpy(x if x.dtype == np.uint16 else x.astype(np.uint16 if x.dtype in (np.uint8, np.bool_) else np.float32))
        x = x.to(self.device).contiguous()
        if x.ndim == 2:
            x = x[None]
        if isinstance(normalize, dict):
            # cellpose's option dict: the reference passes dict(norm3D=False) on its 3-D branch (dispatch.py:196) = every plane
            # normalised on its own, which is what a batch of planes gets here (so norm3D is moot for 2-D input).  Keys at the
            # values that leave cellpose's default normalisation (1st / 99th percentile) unchanged pass; anything else is not built.
            neutral = {"lowhigh": (None,), "percentile": (None, (1, 99), [1, 99], (1.0, 99.0), [1.0, 99.0]), "sharpen_radius": (0,),
                       "smooth_radius": (0,), "tile_norm_blocksize": (0,), "invert": (False,)}
            other_opts = {k: v for k, v in normalize.items() if k not in ("norm3D", "normalize", "tile_norm_smooth3D")
                          and not (k in neutral and any(v is n or (n is not None and not isinstance(v, bool) and v == n) for n in neutral[k]))}
            if other_opts:
                raise NotImplementedError(f"normalize options {other_opts} are not built (percentile normalisation per plane is)")
            normalize = bool(normalize.get("normalize", True))
        dP = prob = None
        if self.flows_override is None or self.run_network_with_override:
            dP, prob = self.run_network(x, normalize=bool(normalize), bsize=bsize, tile_overlap=tile_overlap, batch_size=batch_size)
        if self.flows_override is not None:
            dP, prob = self.flows_override(x)
        labels, counts = dynamics.masks_from_flows(
            self.eng, dP, prob, niter=200 if niter is None else niter, cellprob_threshold=cellprob_threshold,
            flow_threshold=flow_threshold, min_size=min_size, max_size_fraction=max_size_fraction,
        )
        dynamics._mark("eval:dynamics returned")
        self.last_counts = counts
        masks = labels[0] if labels.shape[0] == 1 else labels
        return masks, [None, dP, prob], None
