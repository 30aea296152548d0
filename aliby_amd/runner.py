"""
Position-batched runner behind the step API: many positions per device step.

The reference's caller pattern (examples/01_cell_painting_tiff.py:108-144) is one `run_pipeline_and_post` call per position,
fanned out over loky workers; each call walks `pipeline_step` (pipe_core.py:162-251) for its own position and ends with
`get_profiles_from_state` + one parquet file (pipe_core.py:401-413).  On a GPU one 1024x1024 FOV per launch leaves most of the
256 CUs idle, so `run_positions` keeps that per-position contract — the same pipeline dicts in, the same
`profiles/<name>.parquet` and `steps/<name>/<step>/<tp:04d>.npz` out, byte for byte what N single calls write — but walks B
positions in lockstep:

    ingest thread   tile step of batch k+1: decode / upload every position's stack into one [B,C,Z,Y,X] block   (H2D stream)
    main thread     segment_* : ONE select/normalise/network/dynamics pass over all tiles of the batch            (compute stream)
                    extract_* / extractmulti_* : ONE object table + one launch per feature family for the batch
    writer pool     per position: rows -> Arrow table (columnar fast path), join, parquet (zstd); masks -> .npz    (host threads)

Every step still goes through the step objects `init_step` builds (the tiler's `get_fczyx_device`, the segmenter's `.batch`,
the extraction families), state keeps the reference's layout {"tps", "data", "fn"} per position, and `save` / `save_interval`
/ `retain` / resume-by-skip behave as in `_run_pipeline_and_post_impl`.  Positions whose pipelines differ in structure, or step
kinds without a batched form (track, anything custom), run per position inside the same loop.

Multi-GPU: `run_positions` shards positions `i % world == rank` (aliby_amd/parallel.py); no data-path
collective, the optional end-of-run gather of rows is `parallel.gather_rows`.
"""

from __future__ import annotations

import threading
from concurrent.futures import ThreadPoolExecutor
from itertools import product
from pathlib import Path

import numpy as np
import pyarrow.parquet

from aliby_amd import devcache, pipe_core
from aliby_amd.extraction import extract as ex
from aliby_amd.io.write import dispatch_write_fn


class _Position:
    def __init__(self, index, pipeline, name, output_path):
        self.index, self.pipeline, self.name = index, pipeline, name
        self.steps_dir = Path(output_path) / "steps" / name
        self.profiles_file = Path(output_path) / "profiles" / f"{name}.parquet"
        self.engine = None
        self.state = None
        self.pending = []  # futures of this position's file writes


def _signature(pipeline: dict):
    """Positions can share device steps when their pipelines have the same shape: step names in order, ntps, wiring."""
    return (tuple(pipeline["steps"]), pipeline.get("ntps", 1), repr(sorted(pipeline.get("passed_data", {}).items())),
            repr(sorted(pipeline.get("passed_methods", {}).items())))


class _SharedSteps:
    """Step objects that hold no per-position state (segmenters: the network and its workspaces; extract partials) are built
    once per distinct parameter dict and shared by every position, instead of once per position as N single calls would."""

    def __init__(self, init_step_fn):
        self.init_step_fn = init_step_fn
        self._made = []  # [(step_name, parameters snapshot, fn)]

    def get(self, step_name, parameters, other):
        if step_name.startswith("tile") or step_name.startswith("track"):
            return self.init_step_fn(step_name, parameters, other)  # per-position state (the image, the running labels)
        for name, params, fn in self._made:
            if name == step_name and params == parameters:
                return fn
        fn = self.init_step_fn(step_name, parameters, other)
        self._made.append((step_name, dict(parameters), fn))
        return fn


class _LazyRows:
    """Feature rows of one position inside a batch matrix that is still on its way to the host."""

    def __init__(self, download, index, lo, hi):
        self.download, self.index, self.lo, self.hi = download, index, lo, hi
        self._rows = None
        self._lock = threading.Lock()

    def get(self):
        with self._lock:
            if self._rows is None:
                self._rows = self.download.wait(spin=False)[self.index][self.lo : self.hi]
                self.download = None
            return self._rows


class _Product:
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


ex.PRODUCT_TYPES = (tuple, list, _Product)


class BatchRunner:
    def __init__(self, init_step_fn, writers: int = 8):
        self.shared = _SharedSteps(init_step_fn)
        self.pool = ThreadPoolExecutor(max_workers=max(1, writers), thread_name_prefix="aliby-writer")
        self.ingest = ThreadPoolExecutor(max_workers=1, thread_name_prefix="aliby-ingest")
        self.timing = {}
        self._tables = {}

    # ------------------------------------------------------------------------------------------------ tile step
    def _tile_batch(self, batch, name, tp):
        """Tile step of every position -> list of step results; monotile positions of equal shape land in ONE device block."""
        import torch

        tilers = []
        for pos in batch:
            if name not in pos.state["fn"]:
                pos.state["fn"][name] = self.shared.get(name, pos.pipeline["steps"][name], pos.state["fn"])
            tilers.append(pos.state["fn"][name])
        wanted = batch[0].pipeline.get("save") or []
        if name in wanted or not all(hasattr(t, "run_tp_device") for t in tilers):
            return [pipe_core.run_step(t, tp=tp) for t in tilers]  # host path: the reference's own container types
        return [t.run_tp_device(tp) for t in tilers]

    # --------------------------------------------------------------------------------------------- segment step
    def _segment_batch(self, batch, name, tp):
        fns = []
        for pos in batch:
            if name not in pos.state["fn"]:
                pos.state["fn"][name] = self.shared.get(name, pos.pipeline["steps"][name], pos.state["fn"])
            fns.append(pos.state["fn"][name])
        blocks = []
        for pos in batch:
            kwargs = pos.engine._inputs_for(name, pos.state)
            if kwargs:
                return None  # a segmenter fed by passed_data: not the builder's wiring, run per position
            spec = pos.pipeline.get("passed_methods", {}).get(name)
            if spec is None:
                return None
            owner, method = spec
            tiler = pos.state["fn"][owner]
            if method == "get_fczyx" and hasattr(tiler, "get_fczyx_device"):
                dev, flags = tiler.get_fczyx_device(tp)
                if flags.any():
                    return None
                blocks.append(dev)
            else:
                blocks.append(getattr(tiler, method)(tp))
        if not all(f is fns[0] for f in fns) or not hasattr(fns[0], "batch"):
            return None
        return fns[0].batch(blocks)

    # --------------------------------------------------------------------------------------------- extract steps
    def _extract_batch(self, batch, name, tp, multi):
        import torch

        from aliby_amd.extraction import families
        from aliby_amd.extraction.engine import FeatureEngine

        params = [pos.pipeline["steps"][name] for pos in batch]
        if not all(p.get("tree") == params[0].get("tree") and p.get("kwargs", {}) == params[0].get("kwargs", {}) for p in params):
            return None
        for pos in batch:
            if name not in pos.state["fn"]:
                pos.state["fn"][name] = self.shared.get(name, pos.pipeline["steps"][name], pos.state["fn"])
        inputs = [pos.engine._inputs_for(name, pos.state) for pos in batch]
        if not all(set(i) == {"masks", "pixels"} for i in inputs):
            return None
        tree = params[0]["tree"]
        cp_kwargs = params[0].get("kwargs", {}).get("cp_measure_kwargs") or {}
        instructions = ex.kv(ex.flatten(tree))
        labels, pixels, tiles_of = [], [], []
        for inp in inputs:
            masks = inp["masks"] if isinstance(inp["masks"], list) else [inp["masks"]]
            lab = ex._stack_masks(masks)
            px, dt = ex._device_pixels(inp["pixels"])
            if lab.shape[0] != px.shape[0] and lab.shape[0] != 1:
                return None
            labels.append(lab)
            pixels.append((px[: lab.shape[0]], dt))
            tiles_of.append(lab.shape[0])
        if len({(p.shape[1:], d, p.dtype) for p, d in pixels}) != 1 or len({l.shape[1:] for l in labels}) != 1:
            return None
        eng = FeatureEngine()
        lab_all = labels[0] if len(labels) == 1 else _cat(labels)
        px_all = pixels[0][0] if len(pixels) == 1 else _cat([p for p, _ in pixels])
        # extract_<obj> and extractmulti_<obj> of one timepoint measure the same label block: one object table for both
        key = (tuple(l.data_ptr() for l in labels), tuple(lab_all.shape))
        hit = self._tables.get(key)
        if hit is None:
            hit = self._tables[key] = (eng.object_table(lab_all), labels)  # (the label tensors are kept so the key stays theirs)
        table = hit[0]
        matrix, blocks = families.evaluate(eng, lab_all, table, (px_all, pixels[0][1]), instructions, cp_kwargs, multi=multi)
        download = eng.to_host_async((matrix,), slot=None)
        out, t0 = [], 0
        for pos, nt in zip(batch, tiles_of):
            lo, hi = int(table.offsets[t0]), int(table.offsets[t0 + nt])
            rows = table.host[lo:hi]
            objects = [(int(t) - t0, int(l)) for t, l in zip(rows["tile"], rows["label"])]
            res = ex.DeviceResults(_LazyRows(download, 0, lo, hi), objects, instructions, blocks)
            out.append((_Product(objects, instructions), res))
            t0 += nt
        return out

    # --------------------------------------------------------------------------------------------------- the loop
    def _save(self, pos, step_name, result, tp):
        wanted = pos.pipeline.get("save") or []
        every = pos.pipeline.get("save_interval", 1)
        if wanted and every > 0 and tp % every == 0 and step_name in wanted:
            pos.pending.append(self.pool.submit(dispatch_write_fn(step_name), result, steps_dir=pos.steps_dir, subpath=step_name, tp=tp))

    def run_batch(self, batch):
        """All timepoints of a batch of positions with one signature.  Returns one future per position -> (profiles, {})."""
        steps = batch[0].pipeline["steps"]
        ntps = batch[0].pipeline.get("ntps", 1)
        for pos in batch:
            pos.engine = pipe_core.Engine(pos.pipeline, pos.steps_dir, self.shared.get)
            pos.state = pos.engine.fresh_state(steps)
        for tp in range(ntps):
            self._tables = {}
            for name in steps:
                results = None
                if len(batch) > 0:
                    if name.startswith("tile"):
                        results = self._tile_batch(batch, name, tp)
                    elif name.startswith("segment"):
                        results = self._segment_batch(batch, name, tp)
                    elif name.startswith("extract_"):
                        results = self._extract_batch(batch, name, tp, multi=False)
                    elif name.startswith("extractmulti_"):
                        results = self._extract_batch(batch, name, tp, multi=True)
                if results is None:  # no batched form (or not applicable): the engine's own per-position path for this step
                    results = []
                    for pos in batch:
                        pos.state["data"].setdefault(name, [])
                        if name not in pos.state["fn"]:
                            pos.state["fn"][name] = self.shared.get(name, pos.pipeline["steps"][name], pos.state["fn"])
                        results.append(pipe_core.run_step(pos.state["fn"][name], *pos.engine._method_args(name, pos.state, tp), tp=tp,
                                                          **pos.engine._inputs_for(name, pos.state)))
                for pos, result in zip(batch, results):
                    pos.state["data"].setdefault(name, [])
                    self._save(pos, name, result, tp)
                    pos.state["data"][name].append(result)
                    pos.state["tps"][name] = tp + 1
            for pos in batch:
                pos.engine._end_of_timepoint(pos.state)
        return [self.pool.submit(self._finish, pos) for pos in batch]

    def _finish(self, pos):
        profiles = pipe_core.get_profiles_from_state(pos.state, pos.pipeline)
        pos.profiles_file.parent.mkdir(parents=True, exist_ok=True)
        pyarrow.parquet.write_table(profiles, pos.profiles_file, compression="zstd")
        for f in pos.pending:
            f.result()
        pos.state = pos.engine = None  # releases the device blocks of this position
        return profiles, {}

    def close(self):
        self.pool.shutdown(wait=True)
        self.ingest.shutdown(wait=True)


def _cat(tensors):
    """Concatenate along axis 0; free when the pieces are consecutive views of one allocation (the batched tile step)."""
    import torch

    first = tensors[0]
    base = getattr(first, "_base", None)
    ok = base is not None
    if ok:
        expect = first.data_ptr()
        for t in tensors:
            if getattr(t, "_base", None) is not base or t.data_ptr() != expect or not t.is_contiguous():
                ok = False
                break
            expect += t.numel() * t.element_size()
    if ok:
        n = sum(t.shape[0] for t in tensors)
        return torch.as_strided(first, (n, *first.shape[1:]), first.stride())
    return torch.cat(tensors, 0)


def run_positions(pipelines, names, output_path, overwrite: bool = True, batch_size: int = 16, init_step_fn=None,
                  writers: int = 8, shard: bool = True):
    """`run_pipeline_and_post` for many positions: pipelines[i] / names[i] -> profiles/<names[i]>.parquet (+ step outputs).

    Returns a list aligned with `pipelines`: (pyarrow.Table, {}) for the positions this rank processed, (None, None) for
    positions skipped by resume (`overwrite=False` and the parquet exists) or owned by another rank (`shard=True` under
    torch.distributed.run: positions i % world == rank, examples/01:100-104's round-robin)."""
    from aliby_amd import parallel

    if init_step_fn is None:
        from aliby_amd.pipe import init_step as init_step_fn
    if len(pipelines) != len(names):
        raise ValueError("pipelines and names must have the same length")
    rank, world, _ = parallel.rank_world()
    mine = parallel.positions_for_rank(len(pipelines), rank, world) if shard else list(range(len(pipelines)))
    out = [(None, None)] * len(pipelines)
    todo = []
    for i in mine:
        pipe_core.validate_pipeline(pipelines[i])
        pos = _Position(i, pipelines[i], names[i], output_path)
        if not overwrite and pos.profiles_file.exists():
            pipe_core.logger.info(f"Skipping {names[i]}")
            continue
        todo.append(pos)
    runner = BatchRunner(init_step_fn, writers=writers)
    futures = []
    try:
        k = 0
        while k < len(todo):
            sig = _signature(todo[k].pipeline)
            batch = [todo[k]]
            while len(batch) < batch_size and k + len(batch) < len(todo) and _signature(todo[k + len(batch)].pipeline) == sig:
                batch.append(todo[k + len(batch)])
            futures.extend(zip(batch, runner.run_batch(batch)))
            k += len(batch)
        for pos, fut in futures:
            out[pos.index] = fut.result()
    finally:
        runner.close()
    return out
