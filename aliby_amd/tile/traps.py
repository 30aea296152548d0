"""
ALCATRAS trap detection on the GPU, behind the reference's two functions.

Mirrors src/aliby/tile/process_traps.py: `segment_traps(image, tile_size, downscale=0.4, disk_radius_frac=0.01,
square_size=3, min_frac_tilesize=0.3, **identify_traps_kwargs)` (24-137) and `identify_trap_locations(image,
trap_template, optimize_scale=True, downscale=0.35, trap_size=None)` (140-218) — same arguments, same return value (an
integer array of (row, col) pairs, strongest match first), same `Exception("No valid tiles found.")`, same second attempt
without down-scaling when fewer than 30 traps are found.

Every scikit-image call of that file is one kernel of aliby_amd/csrc/traps.hip (float64, the arithmetic of scikit-image
0.18.3 as restated and pinned in oracle/traps_restated.py); the few scalars in between (Otsu's threshold from a 256-bin
histogram, percentiles, medians for the `median` padding, the greedy spacing of a dozen peaks) are taken with torch on
the device or on the handful of values copied back.  Nothing here runs per tile: one frame per position.
"""

from __future__ import annotations

import numpy as np
import torch

from aliby_amd import _lib
from aliby_amd.extraction.engine import FeatureEngine, _ptr, _stream_ptr

REFERENCE_TILE_SIZE = 117  # imaging_specifications["tile_size"], global_settings.py:18

_engine = None


def _eng():
    global _engine
    if _engine is None:
        _engine = FeatureEngine()
    return _engine


def _call(name, *args):
    eng = _eng()
    _lib.check(getattr(eng.lib, name)(eng.ctx.handle, *args, _stream_ptr()))


def _f64(x):
    if isinstance(x, torch.Tensor):
        return x.to(device="cuda", dtype=torch.float64).contiguous()
    return torch.from_numpy(np.ascontiguousarray(x)).cuda().to(torch.float64)


# ------------------------------------------------------------------------------------------------ resampling
def _gaussian(image, sigmas, truncate_int: bool):
    out = image
    for axis, sigma in enumerate(sigmas):
        if sigma <= 1e-15:
            continue
        radius = int(4.0 * sigma + 0.5)
        x = np.arange(-radius, radius + 1)
        w = np.exp(-0.5 / (sigma * sigma) * x**2)
        w = torch.from_numpy(w / w.sum()).cuda()
        nxt = torch.empty_like(out)
        _call("aliby_trap_gauss1d", _ptr(out), _ptr(nxt), out.shape[0], out.shape[1], axis, _ptr(w), radius, int(truncate_int))
        out = nxt
    return out


def _warp(image, matrix6, out_shape, mode: int, cval: float = 0.0):
    out = torch.empty(out_shape, dtype=torch.float64, device="cuda")
    m = np.ascontiguousarray(matrix6, dtype=np.float64)
    _call("aliby_trap_warp", _ptr(image), image.shape[0], image.shape[1], _ptr(out), out_shape[0], out_shape[1], m.ctypes.data,
          mode, float(cval))
    return out


def resize(image, out_shape, integer_input: bool = False):
    """transform.resize(order=1, mode='reflect', anti_aliasing=True) of scikit-image 0.18.3."""
    factors = np.asarray(image.shape, np.float64) / np.asarray(out_shape, np.float64)
    sm = _gaussian(image, np.maximum(0, (factors - 1) / 2), integer_input)
    if integer_input:
        sm = sm * (1.0 / 65535.0)
    shift = factors * 0.5 - 0.5
    return _warp(sm, [factors[1], 0.0, shift[1], 0.0, factors[0], shift[0]], tuple(int(v) for v in out_shape), 1)


def rescale(image, scale: float, integer_input: bool = False):
    return resize(image, tuple(int(v) for v in np.round(scale * np.asarray(image.shape, np.float64))), integer_input)


def rotate(image, angle_deg: float, cval: float):
    rows, cols = image.shape
    cx, cy = cols / 2.0 - 0.5, rows / 2.0 - 0.5
    t = np.deg2rad(angle_deg)
    c, s = np.cos(t), np.sin(t)
    return _warp(image, [c, -s, cx - c * cx + s * cy, s, c, cy - s * cx - c * cy], (rows, cols), 0, cval)


# ------------------------------------------------------------------------------------------------ segmentation
def _histogram256(flat):
    """np.histogram(x, bins=256) on the device, including its edge corrections."""
    n = 256
    first, last = float(flat.min()), float(flat.max())
    edges = np.linspace(first, last, n + 1)
    e = torch.from_numpy(edges).cuda()
    idx = ((flat - first) * (n / (last - first))).to(torch.int64)
    idx[idx == n] -= 1
    idx = idx - (flat < e[idx]).to(torch.int64)
    idx = idx + ((flat >= e[idx + 1]) & (idx != n - 1)).to(torch.int64)
    return torch.bincount(idx, minlength=n).cpu().numpy(), edges


def threshold_otsu(image):
    flat = image.reshape(-1)
    if bool((flat == flat[0]).all()):
        return float(flat[0])
    counts, edges = _histogram256(flat)
    centers = (edges[:-1] + edges[1:]) / 2.0
    counts = counts.astype(np.float64)
    w1 = np.cumsum(counts)
    w2 = np.cumsum(counts[::-1])[::-1]
    m1 = np.cumsum(counts * centers) / w1
    m2 = (np.cumsum((counts * centers)[::-1]) / w2[::-1])[::-1]
    return float(centers[int(np.argmax(w1[:-1] * w2[1:] * (m1[:-1] - m2[1:]) ** 2))])


def _closing(bw_u8, k: int):
    H, W = bw_u8.shape
    # scipy's grey_dilation moves an even footprint's origin by one (window [-(k/2)+1, k/2]); grey_erosion does not
    lo, hi = (-(k // 2), k // 2) if k % 2 else (-(k // 2) + 1, k // 2)
    dil, out = torch.empty_like(bw_u8), torch.empty_like(bw_u8)
    _call("aliby_trap_morph", _ptr(bw_u8), _ptr(dil), H, W, lo, hi, 1)
    _call("aliby_trap_morph", _ptr(dil), _ptr(out), H, W, -hi, -lo, 0)
    return out


def _regions(bw_u8):
    """label + clear_border + regionprops: list of dicts in label order, border-touching components removed."""
    H, W = bw_u8.shape
    lab = torch.empty((H, W), dtype=torch.int32, device="cuda")
    _call("aliby_trap_label", _ptr(bw_u8), H, W, _ptr(lab))
    sums = torch.zeros((H * W, 7), dtype=torch.int64, device="cuda")
    _call("aliby_trap_region_sums", _ptr(lab), H, W, _ptr(sums))
    rows = sums[sums[:, 0] > 0].cpu().numpy().astype(np.float64)
    out = []
    for n, sy, sx, syy, sxx, sxy, border in rows:
        if border:
            continue
        cy, cx = sy / n, sx / n
        a, c, b = syy / n - cy * cy, sxx / n - cx * cx, sxy / n - cy * cx
        l1 = (a + c) / 2 + np.sqrt(4 * b * b + (a - c) ** 2) / 2
        out.append({"centroid": (cy, cx), "major_axis_length": 4 * np.sqrt(l1), "area": int(n)})
    return out


def trap_regions(image, tile_size, downscale=0.4, disk_radius_frac=0.01, square_size=3, min_frac_tilesize=0.3):
    """process_traps.py:66-104: entropy image -> Otsu -> closing -> clear_border -> label -> candidate regions."""
    image_np = image if isinstance(image, np.ndarray) else None
    integer = image_np is not None and np.issubdtype(image_np.dtype, np.integer) or (
        isinstance(image, torch.Tensor) and not image.dtype.is_floating_point)
    sf = tile_size / REFERENCE_TILE_SIZE
    disk_radius_frac *= sf
    min_frac_tilesize *= sf
    square_size = int(square_size * sf)
    dev = _f64(image)
    H, W = dev.shape
    if downscale != 1:
        img = rescale(dev, downscale, integer_input=integer)
        u8 = torch.clamp(torch.round(img * 255.0), 0, 255).to(torch.uint8)
    elif integer:
        img = dev
        u8 = torch.floor(dev / 256.0).to(torch.uint8)  # img_as_ubyte(uint16): floor_divide by 2**8
    else:
        img = dev
        u8 = torch.clamp(torch.round(dev * 255.0), 0, 255).to(torch.uint8)
    radius = int(min(disk_radius_frac * x for x in img.shape))
    ent = torch.empty(u8.shape, dtype=torch.float64, device="cuda")
    _call("aliby_trap_entropy", _ptr(u8), u8.shape[0], u8.shape[1], radius, _ptr(ent))
    if downscale != 1:
        ent = rescale(ent, 1 / downscale)
    thresh = threshold_otsu(ent)
    bw = _closing((ent > thresh).to(torch.uint8).contiguous(), square_size)
    regs = _regions(bw)
    half = tile_size // 2
    valid = [
        r for r in regs
        if (min_frac_tilesize * tile_size < r["major_axis_length"] < tile_size)
        and (half < r["centroid"][0] < H - half - 1)
        and (half < r["centroid"][1] < W - half - 1)
    ]
    return {"entropy": ent, "otsu": thresh, "bw": bw, "regions": regs, "valid": valid, "disk_radius": radius}


# ------------------------------------------------------------------------------------------------ template matching
def _median(t, dim):
    s, _ = torch.sort(t, dim=dim)
    n = t.shape[dim]
    mid = s.select(dim, n // 2)
    return mid if n % 2 else (s.select(dim, n // 2 - 1) + mid) / 2.0


def _pad_median(image, th: int, tw: int):
    col = _median(image, 0)
    tall = torch.cat([col.expand(th, -1), image, col.expand(th, -1)], 0)
    row = _median(tall, 1)[:, None]
    return torch.cat([row.expand(-1, tw), tall, row.expand(-1, tw)], 1).contiguous()


def match_template(image, template):
    """feature.match_template(image, template, pad_input=True, mode='median')."""
    th, tw = template.shape
    H, W = image.shape
    P = _pad_median(image, th, tw)
    template = template.contiguous()
    t_mean = template.mean()
    t_ssd = float(((template - t_mean) ** 2).sum())
    out = torch.empty((H, W), dtype=torch.float64, device="cuda")
    _call("aliby_trap_match_template", _ptr(P), P.shape[0], P.shape[1], _ptr(template), th, tw, _ptr(out), H, W, float(t_mean), t_ssd)
    return out


def _percentile(t, q: float) -> float:
    s, _ = torch.sort(t.reshape(-1))
    pos = q / 100.0 * (s.numel() - 1)
    lo = int(np.floor(pos))
    hi = min(lo + 1, s.numel() - 1)
    a, b = float(s[lo]), float(s[hi])
    frac = pos - lo
    return b - (b - a) * (1 - frac) if frac >= 0.5 else a + (b - a) * frac


def peak_local_max(image, min_distance: int, exclude_border: int) -> np.ndarray:
    H, W = image.shape
    tmp, mx = torch.empty_like(image), torch.empty_like(image)
    _call("aliby_trap_maxfilter1d", _ptr(image), _ptr(tmp), H, W, 0, min_distance)
    _call("aliby_trap_maxfilter1d", _ptr(tmp), _ptr(mx), H, W, 1, min_distance)
    mask = (image == mx) & (image > image.min())
    if exclude_border:
        mask[:exclude_border] = False
        mask[-exclude_border:] = False
        mask[:, :exclude_border] = False
        mask[:, -exclude_border:] = False
    coords = torch.nonzero(mask)
    values = image[mask].cpu().numpy()
    coords = coords.cpu().numpy()[np.argsort(-values, kind="stable")]
    kept = []
    for c in coords:
        if all(np.max(np.abs(c - k)) >= min_distance for k in kept):
            kept.append(c)
    return np.array(kept, dtype=np.int64).reshape(-1, 2)


# ------------------------------------------------------------------------------------------------ the two functions
def identify_trap_locations(image, trap_template, optimize_scale=True, downscale=0.35, trap_size=None):
    if trap_size is None:
        trap_size = trap_template.shape[0]
    img = rescale(_f64(image), downscale)
    template = rescale(_f64(trap_template), downscale)
    med = float(_median(img.reshape(-1), 0))
    scores = {}
    for rotation in (0, 90, 180, 270):
        scores[rotation] = _percentile(match_template(img, rotate(template, rotation, med)) ** 2, 99.9)
    best_rotation = max(scores, key=scores.get)
    template = rotate(template, best_rotation, med)
    if optimize_scale:
        best, matched = None, None
        for scale in np.linspace(0.5, 2, 10):
            m = match_template(img, rescale(template, scale)) ** 2
            score = _percentile(m, 99.9)
            if best is None or score > best:
                best, matched = score, m
    else:
        matched = match_template(img, template)
    return peak_local_max(rescale(matched, 1 / downscale), int(trap_size * 0.70), trap_size // 3)


def segment_traps(image, tile_size, downscale=0.4, disk_radius_frac=0.01, square_size=3, min_frac_tilesize=0.3,
                  **identify_traps_kwargs):
    if isinstance(image, torch.Tensor):
        image = image.cpu().numpy()
    image = np.asarray(image)
    found = trap_regions(image, tile_size, downscale, disk_radius_frac, square_size, min_frac_tilesize)
    if not found["valid"]:
        raise Exception("No valid tiles found.")
    centroids = np.array([r["centroid"] for r in found["valid"]]).round().astype(int)
    lo, hi = tile_size // 2, -(tile_size // -2)
    dev = _f64(image)
    # candidate templates are tile_size x tile_size slices; .astype(int) truncates a float frame like the reference
    mean_template = torch.stack([torch.trunc(dev[y - lo : y + hi, x - lo : x + hi]) for y, x in centroids]).mean(0)
    traps = identify_trap_locations(image, mean_template, **identify_traps_kwargs)
    traps_retry = []
    if len(traps) < 30 and downscale != 1:
        print("Tiler:TrapIdentification: Trying again.")
        traps_retry = segment_traps(image, tile_size, downscale=1)
    return traps if len(traps_retry) < len(traps) else traps_retry
