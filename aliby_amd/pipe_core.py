"""
Segmenter-agnostic pipeline engine with the reference's observable behaviour.

What the reference's engine does (src/aliby/pipe_core.py): steps are initialised lazily by an injected
`init_step_fn` (54-92, 182-183), each timepoint wires `passed_data` / `passed_methods` (188-215), runs the
step (147-154), writes the outputs listed in "save" every `save_interval` (219-228), appends to
`state["data"]` (230), drops tile pixels at the end of the tp (238-242) and trims histories per "retain"
(245-249); `validate_pipeline` (254-365) rejects malformed dicts; `_run_pipeline_and_post_impl` (381-450)
writes `profiles/<name>.parquet` and skips positions already done; `get_profiles_from_state` (453-512)
pivots every extract step's output, adds metadata columns, concatenates per step prefix and joins.

This module keeps those behaviours (same state layout `{"tps", "data", "fn"}`, same messages, same files)
behind a small `Engine` class.  Remote Nahual steps and trackastra global steps are out of scope
 and raise.
"""

from __future__ import annotations

import logging
import threading
from functools import partial
from pathlib import Path
from typing import Callable

import numpy
import pyarrow as pa
import pyarrow.parquet

from aliby_amd.extraction.extract import (
    extract_tree,
    extract_tree_multi,
    format_extraction,
    process_tree_masks,
    process_tree_masks_overlap,
)
from aliby_amd.io.write import dispatch_write_fn, write_profiles
from aliby_amd.tile.tiler import dispatch_image, dispatch_tiler

logger = logging.getLogger("aliby")
_META_KEYS = ("tp", "tile", "object", "label")


def configure_logging(file):
    """DEBUG file sink like the reference's loguru one (pipe_core.py:37-46), on stdlib logging."""
    from logging.handlers import RotatingFileHandler

    Path(file).parent.mkdir(parents=True, exist_ok=True)
    sink = RotatingFileHandler(file, maxBytes=10 * 1024 * 1024, backupCount=5)
    sink.setFormatter(logging.Formatter("%(asctime)s | %(levelname)-8s | %(name)s:%(funcName)s:%(lineno)d - %(message)s"))
    logger.handlers = [sink]
    logger.setLevel(logging.DEBUG)


# ------------------------------------------------------------------------------------------------
# step initialisers shared by the pipelines' init_step dispatchers
# ------------------------------------------------------------------------------------------------
def _need(parameters: dict, key: str, step_name: str):
    if key not in parameters:
        raise ValueError(f"Step '{step_name}' is missing required '{key}'.")
    return parameters[key]


def _init_tile(step_name: str, parameters: dict) -> Callable:
    image_kwargs = parameters.pop("image_kwargs", None)
    if image_kwargs is None:
        raise ValueError(f"Step '{step_name}' is missing required 'image_kwargs'.")
    if "source" not in image_kwargs:
        raise ValueError(f"Step '{step_name}' 'image_kwargs' is missing required 'source'.")
    make_tiler = dispatch_tiler(parameters.pop("kind", None), parameters)
    return make_tiler(dispatch_image(source=image_kwargs["source"])(**image_kwargs))


def _init_extract(step_name: str, parameters: dict, *, overlap: bool) -> Callable:
    tree = _need(parameters, "tree", step_name)
    extra = parameters.get("kwargs", {})
    if overlap:
        return partial(process_tree_masks_overlap, measure_fn=partial(extract_tree, overlap=True), tree=tree, **extra)
    return partial(process_tree_masks, measure_fn=extract_tree, tree=tree, **extra)


def _init_extract_multi(step_name: str, parameters: dict) -> Callable:
    tree = _need(parameters, "tree", step_name)
    return partial(process_tree_masks, measure_fn=extract_tree_multi, tree=tree, **parameters.get("kwargs", {}))


def _init_nahual(step_name: str, parameters: dict) -> Callable:
    if parameters.get("address") is None:
        raise ValueError(f"If using Nahual you must have an address, currently it is None in step '{step_name}'")
    raise NotImplementedError(f"'{step_name}': Nahual remote steps are out of scope (SURVEY §2 rows 6/16)")


def run_step(step, *args, **kwargs):
    """Objects with `run_tp` get the time point; plain callables do not (pipe_core.py:147-154)."""
    runner = getattr(step, "run_tp", None)
    if runner is not None:
        return runner(*args, **kwargs)
    kwargs.pop("tp", None)
    return step(*args, **kwargs)


# ------------------------------------------------------------------------------------------------
# validation
# ------------------------------------------------------------------------------------------------
def _is_count(x) -> bool:
    return isinstance(x, int) and not isinstance(x, bool)


def _check_links(pipeline: dict, steps: dict) -> dict:
    passed_data = pipeline.get("passed_data")
    if not isinstance(passed_data, dict):
        raise ValueError("Pipeline must contain a 'passed_data' dictionary.")
    for consumer, deps in passed_data.items():
        if not isinstance(deps, (list, tuple)):
            raise TypeError(f"'passed_data' dependencies for step '{consumer}' must be a sequence.")
        for dep in deps:
            if not isinstance(dep, (list, tuple)) or len(dep) < 2:
                raise ValueError(f"Invalid dependency format in 'passed_data' for '{consumer}': {dep}")
            if dep[1] not in steps:
                raise ValueError(f"Step '{consumer}' expects data from '{dep[1]}', but '{dep[1]}' is not defined in 'steps'.")
    methods = pipeline.get("passed_methods", {})
    if not isinstance(methods, dict):
        raise TypeError("'passed_methods' must be a dictionary.")
    for consumer, dep in methods.items():
        if not isinstance(dep, (list, tuple)) or len(dep) < 2:
            raise ValueError(f"Invalid method dependency format for '{consumer}': {dep}")
        if dep[0] not in steps:
            raise ValueError(f"Step '{consumer}' expects a method from '{dep[0]}', but '{dep[0]}' is not defined in 'steps'.")
    return passed_data


def _check_outputs(pipeline: dict, steps: dict, passed_data: dict) -> None:
    save = pipeline.get("save")
    if save is not None:
        if not isinstance(save, (list, tuple, set)):
            raise TypeError("'save' must be a sequence of step names.")
        known = set(steps) | set(pipeline.get("global_steps", {}))
        for name in save:
            if name not in known:
                raise ValueError(f"Step '{name}' listed in 'save' is not defined in the pipeline 'steps' or 'global_steps'.")
    if "save_interval" in pipeline and not (_is_count(pipeline["save_interval"]) and pipeline["save_interval"] >= 1):
        raise ValueError(f"'save_interval' must be a positive int, got {pipeline['save_interval']!r}.")
    retain = pipeline.get("retain", {})
    if not isinstance(retain, dict):
        raise TypeError("'retain' must be a dictionary mapping step name to int or 'all'.")
    tracked = {dep[1] for consumer, deps in passed_data.items() if consumer.startswith("track") for dep in deps}
    for name, keep in retain.items():
        if name not in steps:
            raise ValueError(f"'retain' references step '{name}' not defined in 'steps'.")
        if keep != "all" and not (_is_count(keep) and keep >= 0):
            raise ValueError(f"'retain[{name}]' must be a non-negative int or 'all', got {keep!r}.")
        if name in tracked and isinstance(keep, int) and keep < 2:
            raise ValueError(
                f"'retain[{name}]' = {keep} is too small; per-tp 'track' step reads the last 2 timepoints of '{name}'."
            )


def validate_pipeline(pipeline: dict) -> None:
    if not isinstance(pipeline, dict):
        raise TypeError("Pipeline configuration must be a dictionary.")
    steps = pipeline.get("steps")
    if not isinstance(steps, dict):
        raise ValueError("Pipeline must contain a 'steps' dictionary mapping step names to parameters.")
    passed_data = _check_links(pipeline, steps)
    _check_outputs(pipeline, steps, passed_data)
    for name, params in steps.items():
        if not isinstance(params, dict):
            raise TypeError(f"Parameters for step '{name}' must be a dictionary.")
        if name.startswith("nahual") and "address" not in params:
            raise ValueError(f"Nahual-deployed step '{name}' must provide an 'address' parameter.")
    if pipeline.get("global_steps"):
        if "global_passed_data" not in pipeline:
            raise ValueError("Pipeline defines 'global_steps' but is missing 'global_passed_data'.")
        if not isinstance(pipeline["global_passed_data"], dict):
            raise TypeError("'global_passed_data' must be a dictionary.")
        # well-formed but out of scope here (trackastra via Nahual): refused before any timepoint is
        # processed, on the first run and on a resume alike
        raise NotImplementedError("global steps (trackastra via Nahual) are out of scope (SURVEY §2 row 16)")


# ------------------------------------------------------------------------------------------------
# the engine
# ------------------------------------------------------------------------------------------------
class Engine:
    """Runs a pipeline dict one timepoint at a time; `state` has the reference's layout."""

    def __init__(self, pipeline: dict, steps_dir, init_step_fn: Callable):
        self.pipeline, self.steps_dir, self.init_step_fn = pipeline, steps_dir, init_step_fn

    @staticmethod
    def fresh_state(steps: dict) -> dict:
        return {"tps": {name: 0 for name in steps}, "data": {}, "fn": {}}

    def _inputs_for(self, step_name: str, state: dict) -> dict:
        kwargs = {}
        for kwd, producer, *alias in self.pipeline["passed_data"].get(step_name, ()):
            history = state["data"].get(producer, [])
            if not len(history):
                continue
            if step_name == "track" and kwd == "masks":
                # the tracker reads the last two timepoints, regrouped per tile
                value = [[tp_tiles[i] for tp_tiles in history[-2:]] for i in range(len(history[-1]))]
            else:
                value = history[-1]
                if isinstance(value, dict):
                    value = value[kwd]
            kwargs[alias[0] if alias else kwd] = value
        return kwargs

    def _method_args(self, step_name: str, state: dict, tp: int) -> tuple:
        spec = self.pipeline.get("passed_methods", {}).get(step_name)
        if spec is None or not step_name.startswith("segment"):
            return ()
        owner, method = spec
        return (getattr(state["fn"][owner], method)(tp),)

    def _maybe_save(self, step_name: str, result, tp: int) -> None:
        wanted = self.pipeline.get("save") or []
        every = self.pipeline.get("save_interval", 1)
        if wanted and every > 0 and tp % every == 0 and step_name in wanted:
            print(f"Saving {step_name} to {self.steps_dir}")
            dispatch_write_fn(step_name)(result, steps_dir=self.steps_dir, subpath=step_name, tp=tp)

    def _end_of_timepoint(self, state: dict) -> None:
        for name, history in state["data"].items():
            if name.startswith("tile") and history and isinstance(history[-1], dict):
                history[-1].pop("pixels", None)  # pixels are only consumed inside their own tp
        for name, history in state["data"].items():
            keep = self.pipeline.get("retain", {}).get(name, "all")
            if isinstance(keep, int) and keep >= 0 and len(history) > keep:
                del history[: len(history) - keep]

    def step(self, state: dict | None) -> dict:
        steps = self.pipeline["steps"]
        if not state:
            state = self.fresh_state(steps)
        tp = next(iter(state["tps"].values()))
        for name, parameters in steps.items():
            state["data"].setdefault(name, [])
            if name not in state["fn"]:
                state["fn"][name] = self.init_step_fn(name, parameters, state["fn"])
            result = run_step(state["fn"][name], *self._method_args(name, state, tp), tp=tp, **self._inputs_for(name, state))
            self._maybe_save(name, result, tp)
            state["data"][name].append(result)
            state["tps"][name] = tp + 1
        self._end_of_timepoint(state)
        return state


def pipeline_step(pipeline: dict, state: dict | None, steps_dir, init_step_fn: Callable) -> dict:
    """One timepoint (function form kept for callers of the reference's API)."""
    return Engine(pipeline, steps_dir, init_step_fn).step(state)


def run_pipeline_return_state(pipeline: dict, steps_dir, init_step_fn: Callable) -> dict:
    validate_pipeline(pipeline)
    engine, state = Engine(pipeline, steps_dir, init_step_fn), {}
    for _ in range(pipeline.get("ntps", 1)):
        state = engine.step(state)
    return state


# Pipelines of one process run one at a time on the device: a segmenter's model (weights, workspaces) is shared by every caller
# with the same parameters (segment/dispatch.py), and so are the dynamics' workspace and the runner's arenas.  Re-entrant: the
# position-batched runner holds it for a whole run_positions call and runs fallback steps under it.
DEVICE_LOCK = threading.RLock()


def _run_pipeline_and_post_impl(pipeline: dict, pipeline_name: str, output_path, overwrite: bool = True, *,
                                init_step_fn: Callable, post_state_hook: Callable | None = None):
    """One position: `profiles/<name>.parquet` (zstd), `steps/<name>/...`; an existing parquet is skipped
    unless `overwrite` (resume-by-skip)."""
    output_path = Path(output_path)
    profiles_file = output_path / "profiles" / f"{pipeline_name}.parquet"
    validate_pipeline(pipeline)  # before the resume check: a bad / out-of-scope dict fails the same way on every run
    if not overwrite and profiles_file.exists():
        logger.info(f"Skipping {pipeline_name}")
        return None, None
    # While the call runs, what the process allocated before it is out of the cycle collector's sight: a position allocates
    # enough (18 k (object, instruction) pairs for 256 nuclei) to trigger full collections, and a full collection over a heap with
    # torch, pyarrow and the caller's data in it is ~100 ms against the ~18 ms a position takes.  (Nothing is disabled: the
    # collector keeps running over what the call itself allocates.  ALIBY_MANAGE_GC=0: hands off.)
    import gc
    import os

    frozen = gc.isenabled() and gc.get_freeze_count() == 0 and os.environ.get("ALIBY_MANAGE_GC", "1") != "0"
    if frozen:
        gc.freeze()
    try:
        with DEVICE_LOCK:  # (one pipeline at a time per process: callers on several threads share the GPU and the segmenters' models)
            state = run_pipeline_return_state(pipeline, output_path / "steps" / pipeline_name, init_step_fn)
        profiles = get_profiles_from_state(state, pipeline)
        profiles_file.parent.mkdir(parents=True, exist_ok=True)
        write_profiles(profiles, profiles_file)
        if post_state_hook is not None:
            post_state_hook(state, pipeline, output_path, pipeline_name)
        return profiles, {}
    finally:
        if frozen:
            gc.unfreeze()


# ------------------------------------------------------------------------------------------------
# profile table
# ------------------------------------------------------------------------------------------------
def _empty_profiles() -> pa.Table:
    fields = [("metadata_tile", pa.int64()), ("metadata_label", pa.int64()), ("metadata_object", pa.string()),
              ("metadata_tp", pa.int64())]
    return pa.Table.from_pylist([], schema=pa.schema([pa.field(n, t) for n, t in fields]))


def _wide_table(step_name: str, tp: int, output) -> pa.Table | None:
    if isinstance(output, numpy.ndarray):  # arbitrary embedders: one (instructions, metrics) pair
        output = ((("__", "__"),), (output,))
    table = format_extraction(output)
    table = table.rename_columns([{"tile": "metadata_tile", "label": "metadata_label"}.get(c, c) for c in table.column_names])
    if not len(table):
        return None
    table = table.append_column("metadata_object", pa.array([step_name.split("_")[-1]] * len(table), pa.string()))
    return table.append_column("metadata_tp", pa.array([tp] * len(table), pa.uint16()))


def get_profiles_from_state(state: dict, pipeline: dict) -> pa.Table:
    """Pivot every (extract step, tp), concatenate per step prefix ("extract", "extractmulti", ...) and
    join the prefixes on the metadata columns."""
    by_prefix: dict[str, list] = {}
    for name in pipeline["steps"]:
        if not (name.startswith("extract") or name.startswith("nahual_embed")):
            continue
        bucket = by_prefix.setdefault(name.split("_")[0], [])
        for tp, output in enumerate(state["data"][name]):
            table = _wide_table(name, tp, output)
            if table is not None:
                bucket.append(table)
    merged = [pa.concat_tables(tables) for tables in by_prefix.values() if tables]
    if not merged:
        return _empty_profiles()
    profiles = merged[0]
    for other in merged[1:]:
        profiles = _join_on_metadata(profiles, other)
    return profiles


def _join_on_metadata(left: pa.Table, right: pa.Table, strict: bool = False) -> pa.Table:
    """`left.join(right, keys=metadata_*)` (pipe_core.py:499-510; pyarrow's default left outer join).  Both sides come from
    the same object table, so their key columns are normally equal row for row: the join is then the left table plus the
    right table's other columns, without building an Acero plan for a thousand-column table (tens of ms per position).
    Anything else goes through pyarrow's join."""
    keys = [f"metadata_{k}" for k in _META_KEYS]
    # (column lists taken once: `table[name]` and `table.column_names` cost a pass over a thousand fields each)
    lnames, rnames = left.column_names, right.column_names
    lcols, rcols = left.columns, right.columns
    lpos, rpos = {n: i for i, n in enumerate(lnames)}, {n: i for i, n in enumerate(rnames)}
    keyset = set(keys)
    if left.num_rows == right.num_rows and all(k in lpos and k in rpos for k in keys) and all(
            lcols[lpos[k]].equals(rcols[rpos[k]]) for k in keys) and not (set(lnames) & set(rnames)) - keyset:
        names = list(lnames)
        arrays = list(lcols)
        for name, col in zip(rnames, rcols):
            if name not in keyset:
                names.append(name)
                arrays.append(col)
        return pa.Table.from_arrays(arrays, names=names)
    if strict:  # the caller relies on the row order of the left table: no plan-ordered join
        return None
    return left.join(right, keys=keys)


# ------------------------------------------------------------------------------------------------
# reading step outputs back
# ------------------------------------------------------------------------------------------------
def get_step_output(state_data: dict, fetchers, steps_dir=None) -> numpy.ndarray:
    gathered = []
    for fetcher in fetchers:
        if callable(fetcher):
            gathered.append(fetcher(state_data))
        elif not isinstance(fetcher, str):
            raise Exception(f"Invalid type, expected Callable or string, got {type(fetcher)}")
        elif fetcher.startswith("from_disk:"):
            if steps_dir is None:
                raise ValueError("from_disk fetcher requires steps_dir; pass it through get_step_output(..., steps_dir=...)")
            gathered.append(_load_per_tp_masks(Path(steps_dir) / fetcher.removeprefix("from_disk:")))
        else:
            gathered.append([entry[0] for entry in state_data[fetcher]])  # monotile
    return numpy.asarray(gathered)


def _load_per_tp_masks(step_dir) -> list:
    """Per-tp `.npz` written by io.write.write_ndarray: `tile_i` keys (dict results) or a single `arr_0`."""
    files = sorted(Path(step_dir).glob("*.npz"))
    if not files:
        raise FileNotFoundError(f"No per-tp .npz files found under {step_dir}; ensure this step is listed in pipeline['save'].")
    masks = []
    for path in files:
        with numpy.load(path) as npz:
            keys = list(npz.keys())
            if "tile_0" in keys:
                masks.append(npz["tile_0"])
            elif keys == ["arr_0"]:
                masks.append(npz["arr_0"][0])
            else:
                raise ValueError(f"Unrecognised .npz layout in {path}: keys={keys}")
    return masks
