"""
Instruction tree -> per-object feature matrix on the GPU, behind the reference's extraction API.

Mirrors the call surface of src/extraction/extract.py:
  flatten / kv                 33-74    (instruction tree -> [(ch, red_z, metric), ...])
  process_tree_masks           240-301  (objects x instructions product, returns (tileid_instructions, results))
  extract_tree                 304-375  (single-channel metrics)
  extract_tree_multi           378-453  (channel-pair metrics)
  format_extraction            520-599  (long -> wide pivot, sorted columns)

What changes underneath: the reference explodes labels to an (N,Y,X) bool stack
(agora/utils/masks.py:35-37) and calls one metric per (object x instruction) on a full frame,
redoing the z-reduction every time (extract.py:105-107).  Here labels stay a label image in HBM,
`aliby_object_table` builds a compact (tile, label) table, every distinct (channel, red_z) plane is
reduced once, and each metric family is one kernel launch over all objects of all tiles.  `results`
is a list-like (`DeviceResults`) that yields the same dict-per-(object, instruction) the reference
returns, backed by the dense [n_objects, n_columns] float64 matrix the kernels wrote.

`ncores` / `progress_bar` are accepted and ignored (SURVEY.md §8b "Threading").
"""

from __future__ import annotations

from collections.abc import Sequence
from functools import reduce
from itertools import product

import numpy as np
import pyarrow as pa

from aliby_amd import devcache
from aliby_amd.extraction import features as feat

# reducers the reference registers (extraction/core/functions/loaders.py:110-127)
REDUCTION_FUNS = {"max": "max", "mean": None, "median": None, "div": "div", "add": "add", "None": None}

# metric name -> number of output keys is resolved in aliby_amd.extraction.families
from aliby_amd.extraction import families  # noqa: E402


class LazyProduct:
    """tuple(product(objects, instructions)) without materialising it (process_tree_masks' first return value)."""

    def __init__(self, objects, instructions):
        self.objects, self.instructions = objects, instructions

    def __len__(self):
        return len(self.objects) * len(self.instructions)

    def __iter__(self):
        return iter(product(self.objects, self.instructions))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return tuple(self[j] for j in range(*i.indices(len(self))))
        if i < 0:
            i += len(self)
        o, k = divmod(i, len(self.instructions))
        return (self.objects[o], self.instructions[k])


PRODUCT_TYPES = (tuple, list, LazyProduct)  # containers of (object, instruction) pairs the columnar pivot accepts


def flatten(d: dict, pref=()) -> dict:
    """Nested dict -> {path: leaf} (extract.py:33-57)."""
    return reduce(
        lambda acc, kv_: (
            {**acc, **flatten(kv_[1], (*pref, kv_[0]))} if isinstance(kv_[1], dict) else {**acc, (*pref, kv_[0]): kv_[1]}
        ),
        d.items(),
        {},
    )


def kv(flat: dict) -> list:
    """{path: leaves} -> [(*path, leaf), ...] (extract.py:60-74)."""
    return [(*k, leaf) for k, leaves in flat.items() for leaf in leaves]


class DeviceResults(Sequence):
    """List-like of per-(object, instruction) results backed by the dense feature matrix.

    `self[i]` is what the reference's measure_fn returns for `tileid_instructions[i]`: a dict
    {cp_measure key: float64 array of length 1} or, for the in-repo cell.py metrics, a Python float.
    """

    def __init__(self, matrix, objects, instructions, blocks, pairs=None):
        self._matrix = matrix           # np.float64 [n_obj, n_cols], or a handle with .get() while the rows are in flight
        self.objects = objects          # [(tile, label), ...] row order
        self.instructions = instructions  # distinct instruction tuples, column-block order
        self.blocks = blocks            # per instruction: (col_start, [key, ...] or None for scalar)
        self._pairs = pairs             # optional explicit [(row, inst_index)] when not a full product

    @property
    def matrix(self):
        if not isinstance(self._matrix, np.ndarray):
            self._matrix = self._matrix.get()  # rows downloaded on a side stream (aliby_amd/runner.py): wait for them now
        return self._matrix

    def __len__(self):
        if self._pairs is not None:
            return len(self._pairs)
        return len(self.objects) * len(self.instructions)

    def _pair(self, i):
        if self._pairs is not None:
            return self._pairs[i]
        return divmod(i, len(self.instructions))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        row, k = self._pair(i)
        start, keys = self.blocks[k]
        if keys is None:
            return float(self.matrix[row, start])
        return {key: self.matrix[row, start + j : start + j + 1] for j, key in enumerate(keys)}

    def column_names(self):
        names = []
        for inst, (start, keys) in zip(self.instructions, self.blocks):
            branch = "/".join(str(x) for x in inst)
            if keys is None:
                names.append(f"{branch}/{inst[-1]}")
            else:
                names.extend(f"{branch}/{key}" for key in keys)
        return names


def _stack_masks(masks):
    """list of [Y,X] label images (one per tile) -> (device uint16 [F,Y,X]).  Reuses device copies."""
    import torch

    from aliby_amd.extraction.engine import to_device_u16

    devs = []
    for m in masks:
        hit = devcache.lookup(m) if isinstance(m, np.ndarray) else None
        if hit is not None:
            devs.append(hit[0].reshape(-1, *hit[0].shape[-2:]))
        elif isinstance(m, torch.Tensor):
            devs.append(to_device_u16(m).reshape(-1, *m.shape[-2:]))
        else:
            m = np.asarray(m)
            if m.ndim != 2:
                raise Exception(f"each tile's mask must be a 2-D label image, got shape {m.shape}")
            devs.append(to_device_u16(m[None]))
    if len(devs) == 1:
        return devs[0]
    return torch.cat(devs, 0)


def _device_pixels(pixels):
    """[F,C,Z,Y,X] -> (device tensor, dtype code)."""
    import torch

    from aliby_amd import _lib
    from aliby_amd.extraction.engine import to_device_planes

    if isinstance(pixels, np.ndarray):
        hit = devcache.lookup(pixels)
        if hit is not None:
            if hit[0].dtype == torch.uint16:
                return hit[0], _lib.U8W if hit[1].get("eight_bit") else _lib.U16
            return hit[0], _lib.F32
    return to_device_planes(pixels)


def _objects_of(masks):
    objs = []
    for tile_i, m in enumerate(masks):
        if len(m):
            for lab in range(1, int(m.max()) + 1):
                objs.append((tile_i, lab))
    return objs


def _split(tileid_instructions):
    """Recover distinct objects / instructions (first-seen order) and whether it is a full product."""
    if isinstance(tileid_instructions, LazyProduct):  # (process_tree_masks hands the product over unmaterialised)
        return list(tileid_instructions.objects), list(tileid_instructions.instructions), True
    objects = list(dict.fromkeys(t[0] for t in tileid_instructions))
    instructions = list(dict.fromkeys(t[1] for t in tileid_instructions))
    full = len(tileid_instructions) == len(objects) * len(instructions)
    return objects, instructions, full


class InverseMapping(dict):
    """new (sequential) label -> original label of one (tile, stack) plane: what skimage's relabel_sequential returns as its
    third value, as far as the reference uses it (`.in_values`, `[label]`; extract.py:496-503, 628)."""

    @property
    def in_values(self):
        return np.asarray(list(self.keys()), dtype=np.int64)

    @property
    def out_values(self):
        return np.asarray(list(self.values()), dtype=np.int64)


def relabel_planes(masks):
    """Overlapping-mask stacks, one [S,Y,X] integer array per tile (BABY's layered masks) -> (relabelled planes uint16
    [P,Y,X] with labels 1..K per plane in ascending order of the original labels, tile index of every plane,
    {(tile, stack): InverseMapping}).  Host side: the stacks arrive as NumPy arrays and a trap tile is ~14 k pixels."""
    planes, tile_of, inverse = [], [], {}
    for tile_i, stack in enumerate(masks):
        stack = np.asarray(stack)
        if stack.ndim == 2:
            stack = stack[None]
        if stack.ndim != 3:
            raise Exception(f"each tile's overlapping masks must be a [stack, Y, X] array, got shape {stack.shape}")
        for stack_i, plane in enumerate(stack):
            uniq = np.unique(plane)
            uniq = uniq[uniq != 0]
            if uniq.size and (uniq.min() < 0 or uniq.size >= 65535):
                raise OverflowError(f"tile {tile_i}, stack {stack_i}: labels outside the uint16 range")
            inverse[(tile_i, stack_i)] = InverseMapping({0: 0, **{k + 1: int(v) for k, v in enumerate(uniq)}})
            planes.append(np.searchsorted(uniq, plane).astype(np.uint16) + (plane != 0).astype(np.uint16) if uniq.size
                          else np.zeros(plane.shape, np.uint16))
            tile_of.append(tile_i)
    return planes, tile_of, inverse


def _run_overlap(tileid_instructions, masks, pixels, cp_measure_kwargs):
    """extract_tree(overlap=True): every (tile, stack) plane is one label image of the batched object table; the pixels of
    a plane are its tile's.  Rows come out in (tile, stack, label) order, the order process_tree_masks_overlap enumerates."""
    if not len(tileid_instructions):
        return []
    import torch

    from aliby_amd.extraction.engine import FeatureEngine

    eng = FeatureEngine()
    planes_host, tile_of, inverse = relabel_planes(masks)
    labels = _stack_masks([p for p in planes_host])
    table = eng.object_table(labels)
    planes = None
    if pixels is not None:
        px, dt = _device_pixels(pixels)
        planes = (px.index_select(0, torch.as_tensor(tile_of, device=px.device)), dt)  # [P,C,Z,Y,X]: plane p sees its tile's pixels
    instructions = list(dict.fromkeys(t[1] for t in tileid_instructions))
    matrix_dev, blocks = families.evaluate(eng, labels, table, planes, instructions, cp_measure_kwargs or {}, multi=False)
    matrix = eng.to_host(matrix_dev)
    stack_of, seen = [], {}
    for p, t in enumerate(tile_of):
        stack_of.append(seen.get(t, 0))
        seen[t] = stack_of[-1] + 1
    row_objects = [(tile_of[int(p)], stack_of[int(p)], int(l)) for p, l in zip(table.host["tile"], table.host["label"])]
    row_of = {o: i for i, o in enumerate(row_objects)}
    inst_index = {inst: k for k, inst in enumerate(instructions)}
    try:
        pairs = [(row_of[tuple(t[0])], inst_index[t[1]]) for t in tileid_instructions]
    except KeyError as e:
        raise IndexError(f"no object {e.args[0]} (tile, stack, label) in the overlapping masks") from None
    res = DeviceResults(matrix, row_objects, instructions, blocks, pairs=pairs)
    res.inverse_mappings = inverse
    return res


def _run(tileid_instructions, masks, pixels, cp_measure_kwargs, multi):
    if not len(tileid_instructions):
        return []
    from aliby_amd.extraction.engine import FeatureEngine

    eng = FeatureEngine()
    objects, instructions, full = _split(tileid_instructions)
    labels = _stack_masks(masks)
    table = eng.object_table(labels)
    planes = None
    if pixels is not None:
        planes = _device_pixels(pixels)
    matrix_dev, blocks = families.evaluate(
        eng, labels, table, planes, instructions, cp_measure_kwargs or {}, multi=multi
    )
    matrix = eng.to_host(matrix_dev)
    # rows of the matrix follow the table: every label 1..max of every tile
    row_of = {}
    offs = table.offsets
    for (tile_i, lab) in objects:
        row_of[(tile_i, lab)] = int(offs[tile_i]) + lab - 1
    all_objects = [(int(t), int(l)) for t, l in zip(table.host["tile"], table.host["label"])]
    if full and objects == all_objects:
        return DeviceResults(matrix, objects, instructions, blocks)
    inst_index = {inst: k for k, inst in enumerate(instructions)}
    pairs = [(row_of[t[0]], inst_index[t[1]]) for t in tileid_instructions]
    return DeviceResults(matrix, all_objects, instructions, blocks, pairs=pairs)


def extract_tree(tileid_instructions, masks, pixels, ncores=False, progress_bar=False, overlap=False,
                 cp_measure_kwargs=None):
    """Single-channel features for every (object, instruction) (extract.py:304-375)."""
    if overlap:
        # measure_mono_overlap (extract.py:156-197): object (tile, stack, label) of a [stack, Y, X] mask array per tile
        return _run_overlap(tileid_instructions, masks, pixels, cp_measure_kwargs)
    return _run(tileid_instructions, masks, pixels, cp_measure_kwargs, multi=False)


def extract_tree_multi(tileid_instructions, masks, pixels, ncores=None, progress_bar=False, cp_measure_kwargs=None):
    """Channel-pair features (extract.py:378-453)."""
    assert isinstance(masks, list) or masks.ndim >= 3, "Masks dimensions < 2. It should include batch/tile dimension."
    return _run(tileid_instructions, masks, pixels, cp_measure_kwargs, multi=True)


def process_tree_masks(tree, masks, pixels, measure_fn, ncores=None, progress_bar=False, cp_measure_kwargs=None):
    """Objects x instructions product, then `measure_fn` (extract.py:240-301)."""
    if not isinstance(masks, list):
        masks = [masks]
    instructions = kv(flatten(tree))
    ind_masks = _objects_of(masks)
    # the measuring function sees the product as a sequence that knows its two factors (18 k pairs for a 256-object position:
    # recovering objects and instructions from the materialised tuple cost 3 of a position's 18 ms); the caller gets the tuple
    # the reference returns
    lazy = LazyProduct(ind_masks, instructions)
    extra = {}
    if cp_measure_kwargs is not None:
        extra["cp_measure_kwargs"] = cp_measure_kwargs
    result = measure_fn(lazy, masks, pixels, ncores=ncores, progress_bar=progress_bar, **extra)
    return tuple(lazy), result


def process_tree_masks_overlap(tree, masks, pixels, measure_fn, ncores=None, progress_bar=False, overlap=True,
                               cp_measure_kwargs=None):
    """Overlapping masks (BABY's layered output): `masks[tile]` is a [stack, Y, X] integer array whose planes hold objects
    that may overlap ACROSS planes (extract.py:456-517).  Per (tile, stack) plane the labels are made sequential
    (skimage `relabel_sequential`, ascending original label) and every (tile, stack, sequential label) is measured with every
    instruction.  Returns (tileid_instructions, results) like the reference; `results.inverse_mappings` holds the
    {(tile, stack): new label -> original label} maps that `format_extraction_overlap` needs as its third item.

    One documented difference: the reference's `extract_tree(overlap=True)` builds its boolean masks from the UN-relabelled
    stack (`transform_2d_to_3d(mask)`, extract.py:348) and indexes them with the relabelled id (`masks[tile][label - 1, stack]`,
    extract.py:193), which is the same object only when a plane's labels are already 1..K; here the k-th label of the plane
    (ascending) is object k whatever the original numbering — identical to the reference on sequential planes."""
    if not isinstance(masks, list):
        masks = [masks]
    instructions = kv(flatten(tree))
    _, _, inverse = relabel_planes(masks)
    tile_stack_mask = [(tile_i, stack_i, int(mask_i)) for (tile_i, stack_i), inv in inverse.items()
                       for mask_i in inv.in_values[inv.in_values > 0]]
    tileid_instructions = tuple(product(tile_stack_mask, instructions))
    extra = {}
    if cp_measure_kwargs is not None:
        extra["cp_measure_kwargs"] = cp_measure_kwargs
    result = measure_fn(tileid_instructions, masks, pixels, ncores=ncores, progress_bar=progress_bar, **extra)
    return tileid_instructions, result


def format_extraction_overlap(instructions_result) -> pa.Table:
    """(instructions, results, inverse_mappings) -> wide table keyed by (tile, ORIGINAL label) (extract.py:602-682): the
    label column is `inverse_mappings[tile, stack][label]`, columns are sorted, `tile` / `label` come back as
    `metadata_tile` / `metadata_label`."""
    inverse_mappings = instructions_result[-1]
    rows, metrics_seen = {}, set()
    for inst, metrics in zip(*instructions_result[:2], strict=True):
        tileid, stack_id, label = inst[0]
        branch = "/".join(str(x) for x in inst[1])
        original = inverse_mappings[tileid, stack_id][label]
        row = rows.setdefault((tileid, original), {"tile": tileid, "label": original})
        if isinstance(metrics, (int, float)):
            row[f"{branch}/{inst[1][-1]}"] = metrics
            metrics_seen.add(f"{branch}/{inst[1][-1]}")
        elif isinstance(metrics, dict):
            for k, values in metrics.items():
                for value in values:
                    row[f"{branch}/{k}"] = value
                    metrics_seen.add(f"{branch}/{k}")
        elif isinstance(metrics, list):
            for value in metrics:
                row[f"{branch}/{inst[1][-1]}"] = value
                metrics_seen.add(f"{branch}/{inst[1][-1]}")
    names = sorted(metrics_seen)
    out = {"metadata_tile": [r["tile"] for r in rows.values()], "metadata_label": [r["label"] for r in rows.values()]}
    for m in names:
        out[m] = [r.get(m, None) for r in rows.values()]
    return pa.Table.from_pydict(out)


# --------------------------------------------------------------------------------------------
# long -> wide
# --------------------------------------------------------------------------------------------


_LAYOUTS: dict = {}


def _dense_layout(results: DeviceResults):
    """(sorted metric names, matrix column of each) for a results layout; the same for every position of a run, so it is
    worked out once.  Duplicate metric names collapse to the last writer, as the reference's dict pivot does."""
    # the key names are part of the key: cp_measure kwargs rename columns without changing their count (texture `scale` 3 vs 5)
    key = (tuple(results.instructions), tuple((s, None if k is None else tuple(k)) for s, k in results.blocks))
    hit = _LAYOUTS.get(key)
    if hit is None:
        names = results.column_names()
        flat_cols = []
        for inst, (start, keys) in zip(results.instructions, results.blocks):
            flat_cols.extend(range(start, start + (1 if keys is None else len(keys))))
        last = {}
        for j in sorted(range(len(names)), key=names.__getitem__):
            last[names[j]] = flat_cols[j]
        ordered = sorted(last)
        fields = [pa.field("tile", pa.int64()), pa.field("label", pa.int64())] + [pa.field(n, pa.float64()) for n in ordered]
        hit = _LAYOUTS[key] = (pa.schema(fields), np.asarray([last[n] for n in ordered], dtype=np.intp))
        if len(_LAYOUTS) > 64:
            _LAYOUTS.pop(next(iter(_LAYOUTS)))
    return hit


def _format_dense(instructions, results: DeviceResults) -> pa.Table:
    """Columnar equivalent of the reference pivot for a full objects x instructions product:
    rows in first-seen (tile, label) order, metric columns sorted (extract.py:574-596).  One transposed copy of the feature
    matrix, then every Arrow column is a window of that buffer (no per-value Python work)."""
    schema, take = _dense_layout(results)
    ready = getattr(results, "_transposed", None)
    if ready is not None:
        block = ready.get()  # already column-sorted and transposed on the device (aliby_amd/runner.py)
        n = block.shape[1]
    else:
        matrix = results.matrix
        n = matrix.shape[0]
        block = np.ascontiguousarray(matrix[:, take].T)  # [n_cols, n_rows], C order: row j = column j of the table
    buf = pa.py_buffer(block)
    f64 = pa.float64()
    objs = np.asarray(results.objects, dtype=np.int64).reshape(n, 2)
    arrays = [pa.array(np.ascontiguousarray(objs[:, 0])), pa.array(np.ascontiguousarray(objs[:, 1]))]
    step = n * 8
    arrays.extend(pa.Array.from_buffers(f64, n, [None, buf.slice(j * step, step)]) for j in range(block.shape[0]))
    return pa.Table.from_arrays(arrays, schema=schema)


_Dense = object()  # marker: "the instructions are the full product of results.objects x results.instructions"


def format_extraction(instructions_result) -> pa.Table:
    """(instructions, results) -> wide pyarrow table (extract.py:520-599)."""
    if isinstance(instructions_result, (tuple, list)) and len(instructions_result) == 2:
        inst, res = instructions_result
        if inst is _Dense and isinstance(res, DeviceResults) and res._pairs is None:
            return _format_dense(None, res) if len(res) else pa.table({"tile": pa.array([], pa.int64()), "label": pa.array([], pa.int64())})
        if isinstance(res, DeviceResults) and res._pairs is None and isinstance(inst, PRODUCT_TYPES):
            if len(inst) != len(res):
                raise ValueError("zip() argument 2 is shorter than argument 1" if len(res) < len(inst)
                                 else "zip() argument 2 is longer than argument 1")
            if len(res):
                return _format_dense(inst, res)
    formatted = {k: [] for k in ("tile", "label", "metric", "value")}
    for inst, metrics in zip(*instructions_result, strict=True):
        tileid, label = inst[0][0], inst[0][-1]
        branch = "/".join(str(x) for x in inst[1])
        if isinstance(metrics, (int, float)):
            formatted["tile"].append(tileid)
            formatted["label"].append(label)
            formatted["metric"].append(f"{branch}/{inst[1][-1]}")
            formatted["value"].append(metrics)
        elif isinstance(metrics, dict):
            for k, values in metrics.items():
                for value in values:
                    formatted["value"].append(value)
                    formatted["tile"].append(tileid)
                    formatted["label"].append(label)
                    formatted["metric"].append(f"{branch}/{k}")
        elif isinstance(metrics, np.ndarray):
            for (r, c), value in np.ndenumerate(metrics):
                formatted["tile"].append(r)
                formatted["label"].append(0)
                formatted["metric"].append(f"X_{c}")
                formatted["value"].append(value)
        else:
            raise Exception(
                f"the metrics are in an invalid value: {type(metrics)}. Valid values are int/float, dict or numpy array."
            )
    pivoted = {}
    for t, lbl, m, v in zip(formatted["tile"], formatted["label"], formatted["metric"], formatted["value"], strict=True):
        row = pivoted.setdefault((t, lbl), {"tile": t, "label": lbl})
        row[m] = v
    metrics_list = sorted(set(formatted["metric"]))
    out = {"tile": [], "label": []}
    for m in metrics_list:
        out[m] = []
    for row in pivoted.values():
        out["tile"].append(row["tile"])
        out["label"].append(row["label"])
        for m in metrics_list:
            out[m].append(row.get(m, None))
    return pa.Table.from_pydict(out)
