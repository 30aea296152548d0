"""
Metric registry: instruction -> HIP kernel launch(es) writing a block of columns.

Plays the role of the reference's CELL_FUNS / REDUCTION_FUNS registries
(src/extraction/core/functions/loaders.py:28-79,110-127): names are the cp_measure feature names
bound at loaders.py:71-77 plus the in-repo cell.py metrics (loaders.py:19-25).  A name that is not
registered raises KeyError, as `CELL_FUNS[metric]` does (extract.py:147-153).
"""

from __future__ import annotations

import torch

from aliby_amd import _lib
from aliby_amd.extraction import features as feat
from aliby_amd.extraction.engine import _ptr, _stream_ptr

_RED = {"max": _lib.RED_MAX, "add": _lib.RED_ADD, "div": _lib.RED_DIV}


class PlaneCache:
    """z-reduced planes [F,C,Y,X], one reduction per distinct red_z instead of one per call
    (the reference redoes it for every object x instruction, extract.py:105-107)."""

    def __init__(self, eng, planes):
        self.eng = eng
        self.tensor, self.dtype = planes  # [F,C,Z,Y,X]
        if self.tensor.ndim != 5:
            raise Exception(f"pixels must be [F,C,Z,Y,X], got {tuple(self.tensor.shape)}")
        self._cache = {}

    def get(self, red_z):
        if red_z in self._cache:
            return self._cache[red_z]
        if red_z not in _RED:
            # np.mean / np.median / None are not ufuncs: distributors.py:20-24 raises
            raise Exception(f"{red_z} is an invalid reducer.")
        F, C, Z, Y, X = self.tensor.shape
        op = _RED[red_z]
        if Z == 1 and op == _lib.RED_MAX:
            out, dt = self.tensor.reshape(F, C, Y, X), self.dtype
        else:
            # np.add.reduce(uint16) is exact uint64 and np.divide.reduce float64 (distributors.py:22-24); the feature kernels
            # read uint16 or float32 planes.  Sums of uint16 are exact in float32 while Z * 65535 < 2^24 (Z <= 256): beyond
            # that the plane would silently lose bits, so it is refused.  "div" results carry float32 rounding (6e-8
            # relative, three orders inside the 1e-4 feature tolerance).
            integer = self.dtype in (_lib.U16, _lib.U8W)
            if op == _lib.RED_ADD and integer and Z * 65535 >= (1 << 24):
                raise NotImplementedError(f"reduce_z 'add' over Z={Z} uint16 planes exceeds the exact float32 range (Z <= 256)")
            dt = self.dtype if op == _lib.RED_MAX else _lib.F32
            # (the maximum over z of 8-bit values is still 8-bit: the code U8W stays on the plane for the texture kernel)
            out = torch.empty((F, C, Y, X), dtype=torch.uint16 if dt != _lib.F32 else torch.float32,
                              device=self.tensor.device)
            with self.eng.timed("reduce_z"):
                _lib.check(
                    self.eng.lib.aliby_reduce_z(self.eng.ctx.handle, _ptr(self.tensor), _lib.U16 if integer else _lib.F32,
                                                F * C, Z, Y * X, op, _ptr(out), _lib.U16 if dt != _lib.F32 else _lib.F32,
                                                _stream_ptr())
                )
        self._cache[red_z] = (out, dt)
        return out, dt


def _launch_intensity(eng, labels, table, plane, dt, ch, out, col0, kw):
    eng.intensity(labels, plane, dt, ch, table, out, col0, edge_measurements=kw.get("edge_measurements", True))


def _launch_sizeshape(eng, labels, table, plane, dt, ch, out, col0, kw):
    eng.sizeshape(labels, table, out, col0)


def _launch_feret(eng, labels, table, plane, dt, ch, out, col0, kw):
    eng.feret(labels, table, out, col0)


def _launch_zernike(eng, labels, table, plane, dt, ch, out, col0, kw):
    eng.zernike(labels, None, 0, 0, table, out, col0, weighted=False)


def _launch_radial_zernikes(eng, labels, table, plane, dt, ch, out, col0, kw):
    eng.zernike(labels, plane, dt, ch, table, out, col0, weighted=True)


def _launch_texture(eng, labels, table, plane, dt, ch, out, col0, kw):
    eng.texture(labels, plane, dt, ch, table, out, col0, scale=kw.get("scale", 3), gray_levels=kw.get("gray_levels", 256))


def _launch_radial_distribution(eng, labels, table, plane, dt, ch, out, col0, kw):
    eng.radial_distribution(labels, plane, dt, ch, table, out, col0, bin_count=kw.get("bin_count", 4),
                            scaled=kw.get("scaled", True), maximum_radius=kw.get("maximum_radius", 100))


def _launch_granularity(eng, labels, table, plane, dt, ch, out, col0, kw, **_):
    if kw.get("image_mask", "frame") != "frame":
        # the reference measures one object at a time on a single-object label image (extract.py:147-153): with the objects as
        # the image mask, every object would get its own background and spectrum images.  Not built.
        raise NotImplementedError("granularity(image_mask='objects') through the extraction tree is not built: the batched "
                                  "kernels share one image-level spectrum per tile (image_mask='frame', CellProfiler's default)")
    eng.granularity(labels, plane, dt, ch, table, out, col0, **{k: kw[k] for k in _GRANULARITY_KWARGS if k in kw})


_GRANULARITY_KWARGS = ("subsample_size", "image_sample_size", "element_size", "granular_spectrum_length", "image_mask", "mask_order")

# name -> {names(kw) -> list[str] | None (scalar), launch, needs_pixels}
MONO = {
    "intensity": dict(names=lambda kw: feat.intensity_names(kw.get("edge_measurements", True)),
                      launch=_launch_intensity, needs_pixels=True),
    "sizeshape": dict(names=lambda kw: feat.sizeshape_names(), launch=_launch_sizeshape, needs_pixels=False),
    "feret": dict(names=lambda kw: feat.feret_names(), launch=_launch_feret, needs_pixels=False),
    "zernike": dict(names=lambda kw: feat.zernike_names(), launch=_launch_zernike, needs_pixels=False),
    "radial_zernikes": dict(names=lambda kw: feat.radial_zernike_names(), launch=_launch_radial_zernikes,
                            needs_pixels=True),
}
MULTI = {name: dict(names=(lambda kw, _n=name: list(feat.COLOC[_n]))) for name in feat.COLOC}

# cell.py metrics (loaders.py:19-25): scalar results, one column each; every metric of a (channel, red_z)
# comes from ONE launch of k_cell whose 17-column block is cached in `cell_cache`.
_CELL_MASK_ONLY = ("area", "centroid_x", "centroid_y", "conical_volume", "eccentricity", "spherical_volume", "volume")
_CELL_PIXELS = ("mean", "median", "std", "total", "total_squared", "max2p5pc", "max5px_median", "moment_of_inertia")


def _make_cell_launch(metric):
    def launch(eng, labels, table, plane, dt, ch, out, col0, kw, cell_cache=None):
        key = (None if plane is None else (plane.data_ptr(), ch))
        if cell_cache is None:
            cell_cache = {}
        if key not in cell_cache:
            cell_cache[key] = eng.cell_metrics(labels, plane, dt, ch, table)
        j = eng.CELL_COLUMNS.index(metric)
        out[:, col0] = cell_cache[key][:, j]

    return launch


for _m in _CELL_MASK_ONLY:
    MONO[_m] = dict(names=lambda kw: None, launch=_make_cell_launch(_m), needs_pixels=False, cell=True)
for _m in _CELL_PIXELS:
    MONO[_m] = dict(names=lambda kw: None, launch=_make_cell_launch(_m), needs_pixels=True, cell=True)


def _launch_ratio(eng, labels, table, plane, dt, ch, out, col0, kw):
    # cell.ratio needs a [Y, X, 2] image (cell.py:270); the extraction path hands every metric ONE z-reduced plane, so the
    # reference returns NaN for every object here.  The two-channel form is FeatureEngine.cell_ratio / functions.ratio.
    out[:, col0] = float("nan")


MONO["ratio"] = dict(names=lambda kw: None, launch=_launch_ratio, needs_pixels=True)


def register_optional(eng_cls):
    """Families whose kernels are built in later commits register themselves when the engine has them."""
    if hasattr(eng_cls, "texture"):
        MONO["texture"] = dict(names=lambda kw: feat.texture_names(kw.get("scale", 3), kw.get("gray_levels", 256)),
                               launch=_launch_texture, needs_pixels=True)
    if hasattr(eng_cls, "granularity"):
        # `cell` = run on the main stream after the fan-out has joined: the call iterates to a fixed point and waits on its
        # stream in between
        MONO["granularity"] = dict(names=lambda kw: feat.granularity_names(kw.get("granular_spectrum_length", 16)),
                                   launch=_launch_granularity, needs_pixels=True, cell=True)
    if hasattr(eng_cls, "radial_distribution"):
        MONO["radial_distribution"] = dict(names=lambda kw: feat.radial_distribution_names(kw.get("bin_count", 4), kw.get("scaled", True)),
                                           launch=_launch_radial_distribution, needs_pixels=True)


class _FanOut:
    """Per-object kernels are one small workgroup per object: latency-bound and far from filling 256 CUs, so
    independent instructions (different channels / families / channel pairs) go to a few side HIP streams and run
    concurrently.  fork(): side streams wait for the main stream; join(): the main stream waits for them.  Only
    taken when every object window is LDS-resident (the global-scratch variants share the context's scratch)."""

    def __init__(self, eng, table, out):
        import os

        n = int(os.environ.get("ALIBY_FEATURE_STREAMS", "4"))
        small = table.n_obj > 0 and table.max_h * table.max_w <= 4096
        self.enabled = n > 1 and small and out.is_cuda
        self.out = out
        self.k = 0
        if self.enabled:
            pool = eng.__dict__.setdefault("_side_streams", [])
            while len(pool) < n:
                pool.append(torch.cuda.Stream())
            self.streams = pool[:n]
            self.used = set()

    def fork(self):
        if self.enabled:
            ev = torch.cuda.Event()
            ev.record()
            for st in self.streams:
                st.wait_event(ev)

    def next_stream(self):
        if not self.enabled:
            return _NULL_CTX
        i = self.k % len(self.streams)
        self.k += 1
        self.used.add(i)
        self.out.record_stream(self.streams[i])
        return torch.cuda.stream(self.streams[i])

    def join(self):
        if self.enabled:
            main = torch.cuda.current_stream()
            for i in sorted(self.used):
                ev = torch.cuda.Event()
                ev.record(self.streams[i])
                main.wait_event(ev)
            self.used.clear()


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL_CTX = _NullCtx()


# the functions the reference takes from cp_measure (get_core_measurements / get_correlation_measurements, loaders.py:71-77): only
# their entries of cp_measure_kwargs are ever looked at there; an entry named after one of the in-repo functions is ignored
CP_MEASURE_NAMES = ("intensity", "sizeshape", "zernike", "feret", "texture", "radial_distribution", "radial_zernikes", "granularity",
                    "pearson", "manders_fold", "rwc", "costes")
# per-feature keywords (cp_measure_kwargs / cp_measure_feature_kwargs of the builder) that the kernels implement
BUILT_KWARGS = {
    "intensity": ("edge_measurements",),
    "texture": ("scale", "gray_levels"),
    "radial_distribution": ("bin_count", "scaled", "maximum_radius"),
    "granularity": ("subsample_size", "image_sample_size", "element_size", "granular_spectrum_length", "image_mask", "mask_order"),
    "manders_fold": ("thr",),
    "rwc": ("thr",),
    "costes": ("scale_max",),
}


def evaluate(eng, labels, table, planes, instructions, cp_measure_kwargs, multi=False):
    """Run every instruction over every object; returns (matrix [n_obj, n_cols] on device, blocks)."""
    register_optional(type(eng))
    blocks, specs = [], []
    col = 0
    for inst in instructions:
        metric = inst[-1]
        kw = dict(cp_measure_kwargs.get(metric, {}))
        reg = MULTI if (multi and inst[1] == "None") else MONO
        if metric not in reg:
            raise KeyError(metric)
        # the reference forwards these to the cp_measure function of that name (loaders.py:71-77), where an unknown keyword is an
        # error; here a keyword the kernels do not implement must not be dropped either
        unbuilt = sorted(set(kw) - set(BUILT_KWARGS.get(metric, ()))) if metric in CP_MEASURE_NAMES else ()
        if unbuilt:
            raise NotImplementedError(f"cp_measure_kwargs[{metric!r}]: {unbuilt} not built (built: {sorted(BUILT_KWARGS.get(metric, ()))})")
        names = reg[metric]["names"](kw)
        blocks.append((col, names))
        specs.append((inst, reg[metric], kw, col, 1 if names is None else len(names)))
        col += 1 if names is None else len(names)
    out = eng.new_output(table.n_obj, col)
    cache = PlaneCache(eng, planes) if planes is not None else None

    fan = _FanOut(eng, table, out)
    if multi:
        # one launch per (pair, red_z): every requested colocalisation metric of that pair together
        groups = {}
        for inst, reg, kw, col0, _ in specs:
            (ch0, ch1), red_ch, red_z, metric = inst[0], inst[1], inst[2], inst[3]
            if red_ch != "None":
                raise NotImplementedError(
                    "channel-combining multi instructions (extract.py:227-235) are not built; "
                    "the builder only emits red_ch='None' (pipe_builder.py:33-43)"
                )
            # kwargs are per feature name, as the reference bakes them into one partial per name (loaders.py:71-77):
            # manders_fold's thr must not leak into rwc.  The kernel takes one thr per launch, so metrics of a pair whose
            # thresholds differ go to separate launches (sub-group key = the thr the launch will carry).
            thr = float(kw.get("thr", 15.0)) if metric in ("manders_fold", "rwc") else None
            scale_max = float(kw.get("scale_max", 255.0)) if metric == "costes" else None
            groups.setdefault(((ch0, ch1), red_z), []).append((metric, col0, thr, scale_max))
        split = {}
        for key, entries in groups.items():
            thrs = sorted({t for _, _, t, _ in entries if t is not None})
            scale_max = next((sm for _, _, _, sm in entries if sm is not None), 255.0)
            if len(thrs) <= 1:
                split[(key, 0)] = dict(cols={m: c for m, c, _, _ in entries}, thr=thrs[0] if thrs else 15.0, scale_max=scale_max)
            else:  # threshold-free metrics ride with the first threshold's launch
                for k, t in enumerate(thrs):
                    cols = {m: c for m, c, tt, _ in entries if tt == t or (tt is None and k == 0)}
                    split[(key, k)] = dict(cols=cols, thr=t, scale_max=scale_max)
        groups = {(pair_z, k): g for (pair_z, k), g in split.items()}
        if cache is None:
            raise Exception("pixels are required for colocalisation instructions")
        for (((ch0, ch1), red_z), _k), g in groups.items():  # shared inputs first, on the main stream: z-reduction, rank planes
            plane, dt = cache.get(red_z)
            if "rwc" in g["cols"]:
                eng.rank_planes(labels, plane, dt, table, (ch0, ch1))
        # every pair that shares a z-reduction and its parameters goes into one launch (csrc/feat_coloc.hip, k_coloc_pairs)
        import os

        batches = {}
        for (((ch0, ch1), red_z), _k), g in groups.items():
            batches.setdefault((red_z, g["thr"], g["scale_max"]), []).append(((ch0, ch1), g["cols"]))
        left = []
        for (red_z, thr, scale_max), pairs in batches.items():
            plane, dt = cache.get(red_z)
            if os.environ.get("ALIBY_COLOC_PAIRS", "1") == "0" or len(pairs) < 2 or not eng.coloc_pairs(
                    labels, plane, dt, pairs, table, out, thr=thr, scale_max=scale_max):
                left.extend((pair, red_z, cols, thr, scale_max) for pair, cols in pairs)
        if left:
            fan.fork()
            for (ch0, ch1), red_z, cols, thr, scale_max in left:
                plane, dt = cache.get(red_z)
                with fan.next_stream():
                    eng.coloc(labels, plane, dt, ch0, ch1, table, out, cols, thr=thr, scale_max=scale_max)
            fan.join()
        return out, blocks

    # shared, pixel-independent inputs first, on the main stream (each is cached on the object table)
    for inst, reg, kw, col0, ncols in specs:
        metric = inst[-1]
        if inst[0] != "None" and reg["needs_pixels"] and cache is not None and inst[1] in _RED:
            cache.get(inst[1])
        if metric in ("zernike", "radial_zernikes") and table.n_obj:
            eng.mec(labels, table)
        if metric == "radial_distribution" and table.n_obj:
            eng.radial_geometry(labels, table, kw.get("bin_count", 4), None if kw.get("scaled", True) else kw.get("maximum_radius", 100))
    fan.fork()
    done = {}  # (metric, kwargs) of pixel-independent families already computed -> first column
    copies, after = [], []
    zern_groups = {}  # red_z -> [(channel, first column)] of the radial_zernikes instructions
    sizeshape_col0 = next((c0 for inst, reg, kw, c0, _ in specs if inst[-1] == "sizeshape"), None)
    cell_cache = {}
    for spec in specs:
        inst, reg, kw, col0, ncols = spec
        ch, red_z, metric = inst[0], inst[1], inst[-1]
        if reg.get("cell"):
            after.append(spec)  # cell.py metrics share a cached block: main stream, after the join
            continue
        if ch == "None" or not reg["needs_pixels"]:
            if ch != "None" and cache is not None:
                cache.get(red_z)  # the reference would still reduce (and raise on a bad reducer)
            key = (metric, tuple(sorted(kw.items())))
            if metric == "feret" and sizeshape_col0 is not None:
                # the two Feret diameters are columns of the sizeshape block (one hull kernel writes both): copy them
                ss = feat.sizeshape_names()
                for j, name in enumerate(feat.feret_names()):
                    copies.append((col0 + j, sizeshape_col0 + ss.index(name), 1))
            elif key in done:
                # e.g. "feret"/"zernike" listed under every channel: same labels, same numbers
                copies.append((col0, done[key], ncols))
            else:
                with fan.next_stream():
                    reg["launch"](eng, labels, table, None, 0, None, out, col0, kw)
                done[key] = col0
        else:
            if cache is None:
                raise Exception("pixels are required for this instruction")
            plane, dt = cache.get(red_z)
            if metric == "radial_zernikes" and hasattr(eng, "radial_zernikes_multi") and table.n_obj:
                zern_groups.setdefault(red_z, []).append((ch, col0))  # channels of one plane block share a launch (below)
                continue
            with fan.next_stream():
                reg["launch"](eng, labels, table, plane, dt, ch, out, col0, kw)
    for red_z, members in zern_groups.items():
        plane, dt = cache.get(red_z)
        with fan.next_stream():
            eng.radial_zernikes_multi(labels, plane, dt, [c for c, _ in members], table, out, [c0 for _, c0 in members])
    fan.join()
    for col0, src, ncols in copies:
        out[:, col0 : col0 + ncols] = out[:, src : src + ncols]
    for inst, reg, kw, col0, ncols in after:
        ch, red_z = inst[0], inst[1]
        extra = {"cell_cache": cell_cache}
        if ch == "None" or not reg["needs_pixels"]:
            if ch != "None" and cache is not None:
                cache.get(red_z)
            reg["launch"](eng, labels, table, None, 0, None, out, col0, kw, **extra)
        else:
            if cache is None:
                raise Exception("pixels are required for this instruction")
            plane, dt = cache.get(red_z)
            reg["launch"](eng, labels, table, plane, dt, ch, out, col0, kw, **extra)
    return out, blocks
