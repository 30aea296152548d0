"""
TEST INFRASTRUCTURE — CPU restatement of the reference's extraction orchestration, in the
reference's own structure: label image -> (N,Y,X) bool stack -> one metric call per
(object x instruction) on a full-frame binary mask, z-reduction redone per call.

Follows (restated, not copied):
  flatten / kv                    src/extraction/extract.py:33-74
  measure / measure_mono          src/extraction/extract.py:77-153
  measure_multi                   src/extraction/extract.py:200-237
  process_tree_masks              src/extraction/extract.py:240-301
  extract_tree / _multi           src/extraction/extract.py:304-453 (ncores=None serial path)
  transform_2d_to_3d              src/agora/utils/masks.py:5-37
  reduce_z / load_redfuns         src/extraction/core/functions/distributors.py:6-24, loaders.py:110-127
  wrap_cp_measure_features        src/extraction/core/functions/loaders.py:135-150
  wrap_cp_corr_features           src/extraction/core/functions/loaders.py:153-167
  format_extraction               src/extraction/extract.py:520-599

This is also the `cpu_baseline` leg of bench.py (kind "port").
"""

from __future__ import annotations

from itertools import product

import numpy as np

from oracle import cell_metrics
from oracle import cp_measure_restated as cpm

REDUCTION_FUNS = {
    "max": np.maximum,
    "mean": np.mean,
    "median": np.median,
    "div": np.divide,
    "add": np.add,
    "None": None,
}


def flatten(tree, prefix=()):
    """Nested dict -> {path tuple: leaf}; insertion order preserved."""
    flat = {}
    for key, val in tree.items():
        if isinstance(val, dict):
            flat.update(flatten(val, (*prefix, key)))
        else:
            flat[(*prefix, key)] = val
    return flat


def kv(flat):
    """{path: [leaf, ...]} -> [(*path, leaf), ...]."""
    return [(*path, leaf) for path, leaves in flat.items() for leaf in leaves]


def transform_2d_to_3d(masks):
    labels = np.arange(1, masks.max() + 1)
    return np.equal.outer(labels, masks)


def reduce_z(pixels, fun, axis=0):
    if isinstance(fun, np.ufunc):
        return fun.reduce(pixels, axis=axis)
    raise Exception(f"{fun} is an invalid reducer.")


def load_cellfuns(cp_measure_kwargs=None):
    """CELL_FUNS registry: in-repo cell.py metrics + cp_measure features with per-feature kwargs."""
    cp_measure_kwargs = dict(cp_measure_kwargs or {})
    funs = {}
    for name, f in cell_metrics.ONE_ARG.items():
        funs[name] = (lambda mask, pixels, _f=f: _f(mask))
    funs.update(cell_metrics.TWO_ARG)
    for name, f in cpm.get_core_measurements().items():
        kw = dict(cp_measure_kwargs.get(name, {}))
        funs[name] = (lambda mask, pixels, _f=f, _kw=kw: _f(mask.astype(np.uint16), pixels, **_kw))
    if hasattr(cpm, "get_correlation_measurements"):
        for name, f in cpm.get_correlation_measurements().items():
            kw = dict(cp_measure_kwargs.get(name, {}))
            funs[name] = (lambda mask, p1, p2, _f=f, _kw=kw: _f(p1, p2, mask, **_kw))
    return funs


def measure(mask, pixels, reduction, metric):
    if pixels is not None:
        pixels = reduce_z(pixels, reduction)
    return metric(mask, pixels)


def measure_mono(tileid_x, masks, pixels, cell_funs):
    (tile_i, mask_label), (ch, red_z, metric) = tileid_x
    return measure(
        masks[tile_i][mask_label - 1],
        pixels[tile_i, ch] if ch != "None" else None,
        REDUCTION_FUNS[red_z],
        cell_funs[metric],
    )


def measure_multi(tileid_x, masks, pixels, cell_funs):
    (tile_i, mask_i), ((ch0, ch1), red_ch, red_z, metric) = tileid_x
    if red_ch == "None":
        pix = pixels[tile_i, [ch0, ch1]]
        pix = reduce_z(pix, REDUCTION_FUNS[red_z], axis=1)
        return cell_funs[metric](masks[tile_i][mask_i - 1], *pix)
    new_pixels = reduce_z(
        np.stack((pixels[tile_i, ch0], pixels[tile_i, ch1])), REDUCTION_FUNS[red_ch], axis=0
    )[np.newaxis, ...]
    # literal: the combined stack [1,Z,Y,X] is indexed [tile_i, 0] downstream, as in the reference
    return measure_mono(((tile_i, mask_i), (0, red_z, metric)), masks, new_pixels, cell_funs)


def extract_tree(tileid_instructions, masks, pixels, cp_measure_kwargs=None, limit=None):
    funs = load_cellfuns(cp_measure_kwargs)
    result = []
    if len(tileid_instructions):
        binmasks = [transform_2d_to_3d(m) if len(m) else None for m in masks]
        for k, t in enumerate(tileid_instructions):
            if limit is not None and k >= limit:
                break
            result.append(measure_mono(t, binmasks, pixels, funs))
    return result


def extract_tree_multi(tileid_instructions, masks, pixels, cp_measure_kwargs=None, limit=None):
    funs = load_cellfuns(cp_measure_kwargs)
    result = []
    if len(tileid_instructions):
        binmasks = [transform_2d_to_3d(m) for m in masks]
        for k, t in enumerate(tileid_instructions):
            if limit is not None and k >= limit:
                break
            result.append(measure_multi(t, binmasks, pixels, funs))
    return result


def process_tree_masks(tree, masks, pixels, measure_fn, cp_measure_kwargs=None, limit=None, max_objects=None, objects=None):
    if not isinstance(masks, list):
        masks = [masks]
    instructions = kv(flatten(tree))
    ind_masks = []
    for tile_i, m in enumerate(masks):
        if len(m):
            for mask_i in range(1, int(m.max()) + 1):
                ind_masks.append((tile_i, mask_i))
    if max_objects is not None:  # bounded sample for the cpu_baseline leg of bench.py
        ind_masks = ind_masks[:max_objects]
    if objects is not None:  # fixed subsample [(tile, label), ...] for full-size parity tests
        wanted = set(map(tuple, objects))
        ind_masks = [o for o in ind_masks if o in wanted]
    tileid_instructions = tuple(product(ind_masks, instructions))
    result = measure_fn(tileid_instructions, masks, pixels, cp_measure_kwargs=cp_measure_kwargs, limit=limit)
    return tileid_instructions, result


def relabel_sequential(plane):
    """skimage.segmentation.relabel_sequential(plane): (relabelled, {new: old}); ascending original labels -> 1..K."""
    uniq = np.unique(plane)
    uniq = uniq[uniq != 0]
    fwd = {int(v): k + 1 for k, v in enumerate(uniq)}
    out = np.zeros(plane.shape, np.int64)
    for old, new in fwd.items():
        out[plane == old] = new
    return out, {0: 0, **{new: old for old, new in fwd.items()}}


def process_tree_masks_overlap(tree, masks, pixels, cp_measure_kwargs=None):
    """extract.py:456-517 + extract_tree(overlap=True) 304-359 + measure_mono_overlap 156-197, with the object of
    (tile, stack, k) taken as the k-th label of the RELABELLED plane (see aliby_amd.extraction.extract: the shipped code
    indexes the un-relabelled stack with relabelled ids).  Returns (tileid_instructions, results, inverse_mappings)."""
    if not isinstance(masks, list):
        masks = [masks]
    instructions = kv(flatten(tree))
    funs = load_cellfuns(cp_measure_kwargs)
    tsm, inverse, relabelled = [], {}, {}
    for tile_i, stack in enumerate(masks):
        for stack_i, plane in enumerate(np.asarray(stack)):
            rel, inv = relabel_sequential(plane)
            relabelled[(tile_i, stack_i)] = rel
            inverse[(tile_i, stack_i)] = inv
            tsm.extend((tile_i, stack_i, k) for k in sorted(inv) if k > 0)
    tileid_instructions = tuple(product(tsm, instructions))
    result = []
    for (tile_i, stack_i, k), (ch, red_z, metric) in tileid_instructions:
        mask = relabelled[(tile_i, stack_i)] == k
        result.append(measure(mask, pixels[tile_i, ch] if ch != "None" else None, REDUCTION_FUNS[red_z], funs[metric]))
    return tileid_instructions, result, inverse


def format_extraction_records(instructions_result):
    """Long records (tile, label, metric, value) exactly as format_extraction builds them
    (extract.py:534-572) — the pivot itself is product code and is tested against this."""
    rows = []
    for inst, metrics in zip(*instructions_result, strict=True):
        tileid, label = inst[0][0], inst[0][-1]
        branch = "/".join(str(x) for x in inst[1])
        if isinstance(metrics, (int, float)):
            rows.append((tileid, label, f"{branch}/{inst[1][-1]}", metrics))
        elif isinstance(metrics, dict):
            for k, values in metrics.items():
                for v in values:
                    rows.append((tileid, label, f"{branch}/{k}", v))
        elif isinstance(metrics, np.ndarray):
            for (r, c), v in np.ndenumerate(metrics):
                rows.append((r, 0, f"X_{c}", v))
        else:
            raise Exception(f"the metrics are in an invalid value: {type(metrics)}.")
    return rows
