"""
TCZYX -> FCZYX tile stager on the GPU, behind the reference Tiler's interface.

Mirrors src/aliby/tile/tiler.py: `dispatch_tiler` (56-72), `Tiler.from_image` (251-271),
`run_tp`/`_run_tp` (393-448), `get_fczyx` (309-333), `get_tp_channel` (335-366),
`if_out_of_bounds_pad` (601-650), `get_center` (699-718).  The crop / median-pad arithmetic runs in
aliby_amd/csrc/stager.hip through the C ABI (aliby_crop_pad_u16); the returned NumPy block is also
registered in aliby_amd.devcache so that segment/extract steps reuse the device copy.

Drift estimation (`find_drift`, phase cross-correlation) runs on the GPU when `calculate_drift` is set
(aliby_amd/tile/drift.py).  With `tile_size` set, tile centres come from trap detection on the first frame
(`segment_traps`, aliby_amd/tile/traps.py) unless `trap_locations=[(y,x), ...]` are given.
"""

from __future__ import annotations

import logging
import warnings
from functools import partial
from time import perf_counter

import numpy as np

from aliby_amd import devcache
from aliby_amd.tile.tiles import TileLocations

TILER_DEFAULTS = {"tile_size": 117, "ref_channel": 0, "ref_z": 0, "track_drift": True}


class TilerParameters:
    """Defaults of the reference's TilerParameters (tiler.py:45-53)."""

    _defaults = dict(TILER_DEFAULTS)

    def __init__(self, **kwargs):
        params = dict(self._defaults)
        params.update(kwargs)
        for k, v in params.items():
            setattr(self, k, v)
        self._keys = list(params)

    def to_dict(self):
        return {k: getattr(self, k) for k in self._keys}

    @classmethod
    def default(cls, **kwargs):
        return cls(**kwargs)


from aliby_amd.io.image import ImageArray, dispatch_image  # noqa: E402,F401  (re-exported: pipe_core imports them from here)


def dispatch_tiler(kind, kwargs: dict):
    """Returns a constructor that needs an Image (tiler.py:56-72).  kind="crop" is not built."""
    keys = set(TILER_DEFAULTS)
    tiler_kwargs = {k: v for k, v in kwargs.items() if k in keys}
    extra = {k: v for k, v in kwargs.items() if k not in keys}
    if kind == "crop":
        raise NotImplementedError("CropTiler (tiler.py:138-189) is outside the hot path (SURVEY §2 row 4)")
    return partial(Tiler.from_image, parameters=TilerParameters(**tiler_kwargs), **extra)


def _plane_of(pixels, tp, c, z):
    """One YX plane of a TCZYX stack: a device tensor stays one (the drift estimate runs on the GPU anyway), anything else
    becomes a NumPy array as in the reference (tiler.py:296-303)."""
    plane = pixels[tp, c, z]
    return plane if type(plane).__module__.startswith("torch") else np.asarray(plane)


def get_center(pixels_shape):
    yx = tuple(pixels_shape[-2:])
    return TileLocations.from_tiler_init((tuple(s // 2 for s in yx),), max_size=yx)


class Tiler:
    def __init__(self, pixels, meta, parameters, tile_locs=None, trap_locations=None, **kwargs):
        self._parameters = parameters
        for k, v in parameters.to_dict().items():
            setattr(self, k, v)
        self.pixels = pixels
        # 8-bit sources (uint8 / bool) are stored as uint16 on the device; the mark travels with the step's pixel arrays so that the
        # texture kernel takes their grey level unchanged, as skimage.util.img_as_ubyte does for uint8 (engine.to_device_planes)
        self.eight_bit = str(getattr(pixels, "dtype", "")) in ("uint8", "bool", "torch.uint8", "torch.bool")
        # float stacks (already normalised data): float32 on the device, monotile positions only; the runner keeps such positions
        # on its host-typed tile path, like the 8-bit ones
        self.float_source = str(getattr(pixels, "dtype", "")).replace("torch.", "") in ("float16", "float32", "float64")
        self.meta = meta
        self.channels = list(range(pixels.shape[-4]))
        if self.tile_size is not None:
            idx = parameters.ref_channel
            if isinstance(idx, str):
                idx = self.channels.index(idx)
            self.ref_channel_index = idx
        self.tile_locs = tile_locs
        self._trap_locations = trap_locations
        self.tile_size = self.tile_size or tuple(self.pixels.shape[-2:])
        if isinstance(self.tile_size, int):
            self.tile_size = (self.tile_size, self.tile_size)
        self.no_processed = 0
        self._dev_stack = {}  # tp -> device [C,Z,Y,X] (keeps the last two, like load_image's lru_cache(2))
        self._engine = None
        self._ingest_stream = self._ingest_pool = self._ingest_pending = None
        self._crop_cache = None
        self._upload_into = {}  # tp -> device buffer [C,Z,Y,X] the stack of that timepoint must land in (set by the batched runner)
        # The reference reads `calculate_drift` as an attribute a caller sets after construction (tiler.py:428-431); a
        # pipeline dict has no way to do that, so the step parameter of the same name is accepted here.
        if "calculate_drift" in kwargs:
            self.calculate_drift = bool(kwargs["calculate_drift"])

    @classmethod
    def from_image(cls, image, parameters, **kwargs):
        return cls(image.data, image.meta, parameters, **kwargs)

    @property
    def parameters(self):
        return self._parameters

    @property
    def shape(self):
        return self.pixels.shape

    # ------------------------------------------------------------------ protocol
    def run_tp(self, tp: int, **kwargs):
        t1 = perf_counter()
        out = self._run_tp(tp, **kwargs)
        logging.getLogger("aliby").debug(f"Tiler.run_tp took {(perf_counter() - t1):.4f}s")
        return out

    def run_tp_device(self, tp: int):
        """`run_tp` whose "pixels" stay on the device (a uint16 tensor [F,C,Z,h,w]) — what the position-batched runner
        (aliby_amd/runner.py) hands to the segment / extract steps.  Tiles that the reference would return as float NaN
        blocks (> 25 % outside the frame) have no device form: those timepoints come back through the host path."""
        return self._run_tp(tp, device=True)

    def find_drift(self, tp: int):
        """Translational drift of frame `tp` against frame `tp - 1` on the reference channel / z plane
        (tiler.py:284-307): phase cross-correlation on the GPU (aliby_amd/tile/drift.py)."""
        from aliby_amd.tile.drift import phase_cross_correlation

        ref_z = getattr(self, "ref_z", 0)
        prev_tp = max(0, tp - 1)
        drift = phase_cross_correlation(_plane_of(self.pixels, prev_tp, self.ref_channel_index, ref_z),
                                        _plane_of(self.pixels, tp, self.ref_channel_index, ref_z))
        if 0 < tp < len(self.tile_locs.drifts):
            self.tile_locs.drifts[tp] = drift.tolist()
        else:
            self.tile_locs.drifts.append(drift.tolist())

    def _run_tp(self, tp: int, device: bool = False):
        if self.no_processed == 0:
            if hasattr(self, "ref_channel_index"):
                self.tile_locs = self._areas_of_interest()
            else:
                self.tile_locs = get_center(self.pixels.shape)
        n_drifts = len(self.tile_locs.drifts)
        if self.no_processed != n_drifts:
            warnings.warn("Tiler: the number of processed tiles and the number of drifts calculated do not match.")
            self.no_processed = n_drifts
        if not hasattr(self, "calculate_drift"):
            self.calculate_drift = False
        if self.calculate_drift:
            self.find_drift(tp)
        else:
            drift = [0.0, 0.0]
            if 0 < tp < len(self.tile_locs.drifts):
                self.tile_locs.drifts[tp] = drift
            else:
                self.tile_locs.drifts.append(drift)
        self.no_processed = tp + 1
        if device:
            dev, flags = self.get_fczyx_device(tp)
            if not flags.any():
                return {"drift": self.tile_locs.to_dict(tp), "pixels": dev}
        return {"drift": self.tile_locs.to_dict(tp), "pixels": self.get_fczyx(tp)}

    def _areas_of_interest(self):
        """set_areas_of_interest (tiler.py:653-696): trap detection on the first frame of the reference channel
        (aliby_amd/tile/traps.py), centres too close to an edge dropped; explicit `trap_locations=[(y, x), ...]` take
        the detector's place when given."""
        shape = self.pixels.shape[-2:]
        tmin = min(self.tile_size)
        if min(shape) // 2 > tmin // 2:
            half, max_size = tmin // 2, min(shape)
            if self._trap_locations is not None:
                found = self._trap_locations
            else:
                from aliby_amd.tile.traps import segment_traps

                try:
                    initial = _plane_of(self.pixels, 0, self.ref_channel_index, getattr(self, "ref_z", 0))
                    if not isinstance(initial, np.ndarray):
                        initial = initial.cpu().numpy()  # (a stack that already lives on the device)
                    found = segment_traps(initial, tmin)
                except Exception as e:
                    warnings.warn(f"Trap detection failed ({e}), falling back to center tile.")
                    return get_center(self.pixels.shape)
            locs = [[int(a), int(b)] for a, b in found if half < a < max_size - half and half < b < max_size - half]
            return TileLocations.from_tiler_init(locs, self.tile_size, max_size)
        return get_center(self.pixels.shape)

    # --------------------------------------------------------------------- pixels
    def _device_stack(self, tp: int):
        import torch

        if tp in self._dev_stack:
            return self._dev_stack[tp]
        target = self._upload_into.pop(tp, None)
        dev = self._ingest(tp, target)
        if dev is not None:
            pass
        else:
            dev = self._upload(self.pixels[tp], target)
        if len(self._dev_stack) >= 2:
            self._dev_stack.pop(next(iter(self._dev_stack)))
        self._dev_stack[tp] = dev
        return dev

    def set_upload_buffer(self, tp: int, buffer) -> None:
        """The stack of timepoint `tp` is to be uploaded into `buffer` (a device uint16 tensor [C,Z,Y,X]): the position-batched
        runner hands every position a slice of one [B,C,Z,Y,X] block, so the batch is contiguous without a gather copy."""
        self._upload_into[tp] = buffer

    def _upload(self, block, target=None):
        import torch

        if hasattr(block, "compute"):
            block = block.compute(scheduler="synchronous")
        if isinstance(block, torch.Tensor):
            if block.dtype in (torch.uint8, torch.bool):
                block = block.to(torch.int32).to(torch.uint16)
            elif block.dtype.is_floating_point:
                block = block.to(torch.float32)
            if target is not None and tuple(target.shape) == tuple(block.shape) and block.dtype == target.dtype:
                target.copy_(block, non_blocking=True)
                return target
            return block.cuda()
        block = np.ascontiguousarray(block)
        if block.dtype in (np.uint8, np.bool_):
            block = block.astype(np.uint16)
        elif block.dtype.kind == "f":
            block = block.astype(np.float32)
        if block.dtype not in (np.uint16, np.float32):
            raise NotImplementedError(
                f"stager handles uint16, 8-bit and float stacks (the reference's fixtures are uint16, SURVEY §3.3); got {block.dtype}"
            )
        src = torch.from_numpy(block)
        # page-locked host memory goes up asynchronously on the current stream at PCIe rate (pageable memory is staged by the
        # runtime at a quarter of it); torch's host allocator keeps a pinned block alive until the copy has run
        if target is not None and tuple(target.shape) == tuple(src.shape) and target.dtype == src.dtype:
            target.copy_(src, non_blocking=src.is_pinned())
            return target
        return src.cuda(non_blocking=src.is_pinned())

    def _ingest(self, tp: int, target=None):
        """File-backed stacks (aliby_amd/io/image.py): decode + upload through csrc/ingest.hip on a side stream, and
        start decoding the next time point on a helper thread while this one is being processed."""
        import torch

        pixels = self.pixels
        if not hasattr(pixels, "read_device") or np.dtype(pixels.dtype) != np.uint16:
            return None
        from aliby_amd.extraction.engine import FeatureEngine

        if self._engine is None:
            self._engine = FeatureEngine()
        if self._ingest_stream is None:
            from concurrent.futures import ThreadPoolExecutor

            self._ingest_stream = torch.cuda.Stream()
            self._ingest_pool = ThreadPoolExecutor(max_workers=1)
        ctx, stream, device = self._engine.ctx.handle, self._ingest_stream.cuda_stream, torch.cuda.current_device()
        pending, self._ingest_pending = self._ingest_pending, None
        dev = None
        if pending is not None:
            got = pending[1].result()  # the helper owns the staging block until it is done
            if pending[0] == tp:
                dev = got
        if dev is None:
            dev = pixels.read_device(tp, ctx, stream, target, device)
        if dev is not None and tp + 1 < pixels.shape[0]:
            self._ingest_pending = (tp + 1, self._ingest_pool.submit(pixels.read_device, tp + 1, ctx, stream, None, device))
        return dev

    def rects(self, tp: int) -> np.ndarray:
        """[F,4] (y0, x0, h, w) from Tile.as_range (first axis = rows, tiles.py:151-166)."""
        rows = []
        for tile in self.tile_locs:
            ys, xs = tile.as_range(tp)
            rows.append((ys.start, xs.start, ys.stop - ys.start, xs.stop - xs.start))
        return np.asarray(rows, dtype=np.int32).reshape(-1, 4)

    def get_fczyx_device(self, tp: int):
        """(device uint16 [F,C,Z,h,w], nan_flags[F]) — crop + median pad on the GPU."""
        import torch

        from aliby_amd import _lib
        from aliby_amd.extraction.engine import FeatureEngine, _ptr, _stream_ptr

        if self._engine is None:
            self._engine = FeatureEngine()
        rects = self.rects(tp)
        # the engine asks for the tiles of a timepoint once per consumer (its own step, then every segment step through
        # passed_methods: pipe_core.py:211-215); the crop of a (tp, windows) pair is done once
        key = (tp, rects.tobytes())
        if self._crop_cache is not None and self._crop_cache[0] == key:
            return self._crop_cache[1], self._crop_cache[2]
        stack = self._device_stack(tp)
        C, Z, Y, X = stack.shape
        F = len(rects)
        h, w = (int(rects[0, 2]), int(rects[0, 3])) if F else (0, 0)
        flags = np.zeros(max(F, 1), np.int32)
        if F == 1 and tuple(rects[0]) == (0, 0, Y, X):
            # monotile window = the whole frame: the stack itself is the tile block, no copy
            self._crop_cache = (key, stack[None], flags[:F])
            return stack[None], flags[:F]
        if stack.dtype != torch.uint16:
            raise NotImplementedError(f"tiles cropped out of {stack.dtype} stacks are not built (the stager crops uint16; a float "
                                      "stack goes through as ONE tile: tile_size=None)")
        out = torch.empty((F, C, Z, h, w), dtype=torch.uint16, device=stack.device)
        if F:
            _lib.check(
                self._engine.lib.aliby_crop_pad_u16(
                    self._engine.ctx.handle, _ptr(stack), C, Z, Y, X, _ptr(rects), F, h, w, _ptr(out), _ptr(flags),
                    _stream_ptr(),
                )
            )
        self._crop_cache = (key, out, flags[:F])
        return out, flags[:F]

    def get_fczyx(self, tp: int, drift: bool = True) -> np.ndarray:
        dev, flags = self.get_fczyx_device(tp)
        host = dev.cpu().numpy()
        if flags.any():
            # if_out_of_bounds_pad returns float NaN tiles; np.stack then upcasts the block (tiler.py:642-650)
            host = host.astype(np.float64)
            host[flags.astype(bool)] = np.nan
            return host
        if self.eight_bit:
            host = host.astype(np.uint8)  # (the dtype the reference's tiler hands on)
        elif self.float_source and str(getattr(self.pixels, "dtype", "")).replace("torch.", "") == "float64":
            host = host.astype(np.float64)
        return devcache.attach(host, dev, kind="pixels", eight_bit=self.eight_bit)

    def get_tp_channel(self, tp: int, c: int, drift: bool = True) -> np.ndarray:
        return self.get_fczyx(tp)[:, c]

    def get_tile_data(self, tile_id: int, tp: int, c: int) -> np.ndarray:
        return self.get_fczyx(tp)[tile_id, c]
