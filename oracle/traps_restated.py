"""
TEST INFRASTRUCTURE ONLY — CPU restatement of ALCATRAS trap detection.

Follows src/aliby/tile/process_traps.py: `segment_traps` (24-137) and `identify_trap_locations` (140-218), and, for
every scikit-image call those make, the behaviour of scikit-image 0.18.3 (the version this container can run; uv.lock
pins 0.26.0 — the differences that matter are noted where they occur):

* `transform.rescale` / `resize` / `rotate` / `warp` (order 1): Gaussian pre-filter with sigma = (factor - 1) / 2 in
  the *input's dtype* (so the uint16 frame is truncated after each axis, as `scipy.ndimage` does), then bilinear
  sampling at `factor * (i + 0.5) - 0.5` with mirrored borders (`reflect`) or a constant;
* `filters.rank.entropy` on `img_as_ubyte`, footprint clipped at the borders; `threshold_otsu` (256 bins);
  `closing(square(k))`, `clear_border`, `label` (8-connected), `regionprops` centroid / major axis;
* `feature.match_template(pad_input=True, mode="median")` and `feature.peak_local_max(min_distance, exclude_border)`.

Pinned by tests/golden/reference_traps.json: the reference functions themselves, imported from /root/reference and run
in the container's conda interpreter on `aliby_amd.synth.trap_image` (tests/golden/make_golden.py --stage traps).
"""

from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi
from scipy.signal import fftconvolve

REFERENCE_TILE_SIZE = 117  # global_settings.py:18


# ------------------------------------------------------------------------------------------------ resampling
def gaussian_kernel(sigma: float, truncate: float = 4.0):
    radius = int(truncate * sigma + 0.5)
    x = np.arange(-radius, radius + 1)
    w = np.exp(-0.5 / (sigma * sigma) * x**2)
    return w / w.sum(), radius


def _mirror_index(i, n):
    """scipy 'mirror' (d c b | a b c d | c b a): reflect about the centre of the edge pixels."""
    if n == 1:
        return np.zeros_like(i)
    period = 2 * (n - 1)
    i = np.abs(i) % period
    return np.where(i >= n, period - i, i)


def gaussian_filter(image, sigmas, integer_output: bool):
    """scipy.ndimage.gaussian_filter(mode='mirror'), axis by axis; an integer input keeps its dtype between and after
    the passes, by truncation (ni_support's line-buffer copy is a C cast)."""
    out = image.astype(np.float64)
    for axis, sigma in enumerate(sigmas):
        if sigma <= 1e-15:
            continue
        w, r = gaussian_kernel(float(sigma))
        n = out.shape[axis]
        acc = np.zeros_like(out)
        base = np.arange(n)
        for k in range(-r, r + 1):
            acc += w[k + r] * np.take(out, _mirror_index(base + k, n), axis=axis)
        out = np.trunc(acc) if integer_output else acc
    return out


def warp_bilinear(image, a, b, out_shape, mode: str, cval: float = 0.0, matrix=None):
    """skimage `_warp_fast` (order 1).  Either the metric map row = a[0] * r + b[0], col = a[1] * c + b[1], or a full
    2x3 `matrix` acting on (col, row).  mode 'reflect' mirrors indices, 'constant' reads `cval` outside."""
    H, W = image.shape
    rr, cc = np.mgrid[0 : out_shape[0], 0 : out_shape[1]].astype(np.float64)
    if matrix is None:
        y = a[0] * rr + b[0]
        x = a[1] * cc + b[1]
    else:
        x = matrix[0, 0] * cc + matrix[0, 1] * rr + matrix[0, 2]
        y = matrix[1, 0] * cc + matrix[1, 1] * rr + matrix[1, 2]
    y0, x0 = np.floor(y), np.floor(x)
    y1, x1 = np.ceil(y), np.ceil(x)
    dy, dx = y - y0, x - x0

    def px(r, c):
        r, c = r.astype(np.int64), c.astype(np.int64)
        if mode == "reflect":
            return image[_mirror_index(r, H), _mirror_index(c, W)]
        inside = (r >= 0) & (r < H) & (c >= 0) & (c < W)
        return np.where(inside, image[np.clip(r, 0, H - 1), np.clip(c, 0, W - 1)], cval)

    top = (1 - dx) * px(y0, x0) + dx * px(y0, x1)
    bottom = (1 - dx) * px(y1, x0) + dx * px(y1, x1)
    return (1 - dy) * top + dy * bottom


def resize(image, out_shape, integer_input: bool = False):
    """transform.resize(order=1, mode='reflect', anti_aliasing=True); an integer frame is smoothed in its own dtype
    and then scaled to [0, 1] by img_as_float (x * (1 / 65535))."""
    factors = np.asarray(image.shape, np.float64) / np.asarray(out_shape, np.float64)
    sm = gaussian_filter(image, np.maximum(0, (factors - 1) / 2), integer_output=integer_input)
    if integer_input:
        sm = sm * (1.0 / 65535.0)
    return warp_bilinear(sm, factors, factors * 0.5 - 0.5, out_shape, "reflect")


def rescale(image, scale: float, integer_input: bool = False):
    out_shape = tuple(int(v) for v in np.round(scale * np.asarray(image.shape, np.float64)))
    return resize(image, out_shape, integer_input)


def rotate(image, angle_deg: float, cval: float):
    """transform.rotate(resize=False, order=1, mode='constant'): output (r, c) reads R (c - cx, r - cy) + centre."""
    rows, cols = image.shape
    cx, cy = cols / 2.0 - 0.5, rows / 2.0 - 0.5
    t = np.deg2rad(angle_deg)
    c, s = np.cos(t), np.sin(t)
    m = np.array([[c, -s, cx - c * cx + s * cy], [s, c, cy - s * cx - c * cy]])
    return warp_bilinear(image, None, None, image.shape, "constant", cval, matrix=m)


# ------------------------------------------------------------------------------------------------ segmentation
def disk(radius: int):
    yy, xx = np.mgrid[-radius : radius + 1, -radius : radius + 1]
    return (yy * yy + xx * xx) <= radius * radius


def rank_entropy(img_u8, radius: int):
    """filters.rank.entropy: Shannon entropy (bits) of the grey levels under the footprint, image pixels only."""
    H, W = img_u8.shape
    fp = disk(radius)
    offs = [(dy - radius, dx - radius) for dy in range(2 * radius + 1) for dx in range(2 * radius + 1) if fp[dy, dx]]
    pad = np.full((H + 2 * radius, W + 2 * radius), -1, np.int16)
    pad[radius : radius + H, radius : radius + W] = img_u8
    stack = np.stack([pad[radius + dy : radius + dy + H, radius + dx : radius + dx + W] for dy, dx in offs])  # [K,H,W]
    valid = stack >= 0
    pop = valid.sum(0).astype(np.float64)
    out = np.zeros((H, W))
    for level in np.unique(img_u8):
        cnt = ((stack == level) & valid).sum(0).astype(np.float64)
        p = cnt / pop
        with np.errstate(divide="ignore", invalid="ignore"):
            out -= np.where(cnt > 0, p * np.log(p) / 0.6931471805599453, 0.0)
    return out


def threshold_otsu(image, nbins: int = 256):
    flat = image.ravel()
    if np.all(flat == flat[0]):
        return flat[0]
    counts, edges = np.histogram(flat, bins=nbins)
    centers = (edges[:-1] + edges[1:]) / 2.0
    counts = counts.astype(np.float64)
    w1 = np.cumsum(counts)
    w2 = np.cumsum(counts[::-1])[::-1]
    m1 = np.cumsum(counts * centers) / w1
    m2 = (np.cumsum((counts * centers)[::-1]) / w2[::-1])[::-1]
    var12 = w1[:-1] * w2[1:] * (m1[:-1] - m2[1:]) ** 2
    return centers[int(np.argmax(var12))]


def closing_square(bw, k: int):
    """morphology.closing(bool, square(k)): k x k maximum then minimum, borders replicated (ndimage 'reflect')."""
    fp = np.ones((k, k), bool)
    out = ndi.grey_dilation(bw.astype(np.uint8), footprint=fp, mode="reflect")
    return ndi.grey_erosion(out, footprint=fp, mode="reflect").astype(bool)


EIGHT = np.ones((3, 3), bool)


def clear_border(bw):
    lab, _ = ndi.label(bw, structure=EIGHT)
    edge = np.zeros_like(bw)
    edge[0, :] = edge[-1, :] = edge[:, 0] = edge[:, -1] = True
    touching = np.unique(lab[edge & (lab > 0)])
    return bw & ~np.isin(lab, touching)


def regions(bw):
    """label (8-connected, raster order of first pixel) + regionprops centroid, major_axis_length, area."""
    lab, n = ndi.label(bw, structure=EIGHT)
    out = []
    for k in range(1, n + 1):
        ys, xs = np.nonzero(lab == k)
        cy, cx = ys.mean(), xs.mean()
        a = ((ys - cy) ** 2).mean()  # mu20 / area along rows
        c = ((xs - cx) ** 2).mean()
        b = ((ys - cy) * (xs - cx)).mean()
        l1 = (a + c) / 2 + np.sqrt(4 * b * b + (a - c) ** 2) / 2
        out.append({"centroid": (cy, cx), "major_axis_length": 4 * np.sqrt(l1), "area": len(ys)})
    return out


# ------------------------------------------------------------------------------------------------ template matching
def pad_median(image, widths):
    """np.pad(mode='median'): axis 0 first (column medians), then axis 1 (medians of the rows of the result so far,
    taken over the original columns)."""
    h, w = widths
    col = np.median(image, axis=0)
    tall = np.concatenate([np.tile(col, (h, 1)), image, np.tile(col, (h, 1))], 0)
    row = np.median(tall, axis=1)[:, None]
    return np.concatenate([np.tile(row, (1, w)), tall, np.tile(row, (1, w))], 1)


def match_template(image, template):
    """feature.match_template(image, template, pad_input=True, mode='median')."""
    th, tw = template.shape
    H, W = image.shape
    P = pad_median(image.astype(np.float64), (th, tw))

    def window_sum(a):
        s = np.cumsum(a, axis=0)
        s = s[th:-1] - s[: -th - 1]
        s = np.cumsum(s, axis=1)
        return s[:, tw:-1] - s[:, : -tw - 1]

    ws, ws2 = window_sum(P), window_sum(P * P)
    t_mean = template.mean()
    t_ssd = np.sum((template - t_mean) ** 2)
    xcorr = fftconvolve(P, template[::-1, ::-1], mode="valid")[1:-1, 1:-1]
    num = xcorr - ws * t_mean
    den = np.sqrt(np.maximum((ws2 - ws * ws / (th * tw)) * t_ssd, 0))
    resp = np.zeros_like(xcorr)
    ok = den > np.finfo(np.float64).eps
    resp[ok] = num[ok] / den[ok]
    d0, d1 = (th - 1) // 2, (tw - 1) // 2
    return resp[d0 : d0 + H, d1 : d1 + W]


def peak_local_max(image, min_distance: int, exclude_border: int):
    """feature.peak_local_max (0.18.3 defaults): strict window maxima above image.min(), borders excluded, highest first,
    then Chebyshev spacing >= min_distance enforced greedily."""
    size = 2 * min_distance + 1
    mx = ndi.maximum_filter(image, size=size, mode="constant")
    mask = (image == mx) & (image > image.min())
    if exclude_border:
        mask[:exclude_border] = mask[-exclude_border:] = False
        mask[:, :exclude_border] = mask[:, -exclude_border:] = False
    coords = np.transpose(np.nonzero(mask))
    order = np.argsort(-image[mask], kind="stable")
    coords = coords[order]
    kept = []
    for c in coords:
        if all(np.max(np.abs(c - k)) >= min_distance for k in kept):
            kept.append(c)
    return np.array(kept, dtype=np.int64).reshape(-1, 2)


# ------------------------------------------------------------------------------------------------ the two functions
def identify_trap_locations(image, trap_template, optimize_scale: bool = True, downscale: float = 0.35, trap_size=None,
                            details: dict | None = None):
    if trap_size is None:
        trap_size = trap_template.shape[0]
    img = rescale(image.astype(np.float64), downscale)
    template = rescale(np.asarray(trap_template, np.float64), downscale)
    med = float(np.median(img))
    scores = {}
    for rotation in (0, 90, 180, 270):
        scores[rotation] = np.percentile(match_template(img, rotate(template, rotation, med)) ** 2, 99.9)
    best_rotation = max(scores, key=scores.get)
    template = rotate(template, best_rotation, med)
    if optimize_scale:
        matches = {}
        for scale in np.linspace(0.5, 2, 10):
            matches[scale] = match_template(img, rescale(template, scale)) ** 2
        best_scale = max(matches, key=lambda s: np.percentile(matches[s], 99.9))
        matched = matches[best_scale]
    else:
        best_scale = 1.0
        matched = match_template(img, template)
    if details is not None:
        details.update(best_rotation=best_rotation, best_scale=float(best_scale), rotation_scores=scores,
                       small=img, template_small=rescale(np.asarray(trap_template, np.float64), downscale))
    return peak_local_max(rescale(matched, 1 / downscale), int(trap_size * 0.70), trap_size // 3)


def trap_regions(image, tile_size, downscale=0.4, disk_radius_frac=0.01, square_size=3, min_frac_tilesize=0.3):
    """First half of segment_traps (process_traps.py:66-104): entropy image -> candidate regions."""
    sf = tile_size / REFERENCE_TILE_SIZE
    disk_radius_frac *= sf
    min_frac_tilesize *= sf
    square_size = int(square_size * sf)
    if downscale != 1:
        img = rescale(image, downscale, integer_input=np.issubdtype(image.dtype, np.integer))
        u8 = np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8)
    else:
        img = image
        u8 = (image >> 8).astype(np.uint8) if image.dtype == np.uint16 else np.clip(np.rint(image * 255.0), 0, 255).astype(np.uint8)
    radius = int(min(disk_radius_frac * x for x in img.shape))
    ent = rank_entropy(u8, radius)
    if downscale != 1:
        ent = rescale(ent, 1 / downscale)
    thresh = threshold_otsu(ent)
    bw = closing_square(ent > thresh, square_size)
    regs = regions(clear_border(bw))
    half = tile_size // 2
    valid = [
        r for r in regs
        if (min_frac_tilesize * tile_size < r["major_axis_length"] < tile_size)
        and (half < r["centroid"][0] < image.shape[0] - half - 1)
        and (half < r["centroid"][1] < image.shape[1] - half - 1)
    ]
    return {"entropy": ent, "otsu": thresh, "bw": bw, "regions": regs, "valid": valid, "disk_radius": radius}


def segment_traps(image, tile_size, downscale=0.4, details: dict | None = None, **kwargs):
    found = trap_regions(image, tile_size, downscale=downscale, **kwargs)
    if not found["valid"]:
        raise Exception("No valid tiles found.")
    centroids = np.array([r["centroid"] for r in found["valid"]]).round().astype(int)
    lo, hi = tile_size // 2, -(tile_size // -2)
    templates = [image[y - lo : y + hi, x - lo : x + hi] for y, x in centroids]
    mean_template = np.stack(templates).astype(int).mean(axis=0)
    if details is not None:
        details.update(found, centroids=centroids, template=mean_template)
    traps = identify_trap_locations(image, mean_template, details=details)
    retry = []
    if len(traps) < 30 and downscale != 1:
        retry = segment_traps(image, tile_size, downscale=1)
    return traps if len(retry) < len(traps) else retry
