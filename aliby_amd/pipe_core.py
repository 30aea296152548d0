"""
Segmenter-agnostic pipeline engine with the reference's semantics.

Mirrors src/aliby/pipe_core.py: `_init_tile` / `_init_extract` / `_init_extract_multi` (54-92),
`run_step` (147-154), `pipeline_step` (162-251), `validate_pipeline` (254-365),
`run_pipeline_return_state` (368-378), `_run_pipeline_and_post_impl` (381-450),
`get_profiles_from_state` (453-512), `get_step_output` / `_load_per_tp_masks` (515-571).

Remote Nahual steps and trackastra global steps are out of scope and raise.
"""

from __future__ import annotations

import logging
from functools import partial
from itertools import cycle
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
from aliby_amd.io.write import dispatch_write_fn
from aliby_amd.tile.tiler import dispatch_image, dispatch_tiler

logger = logging.getLogger("aliby")


def configure_logging(file):
    """File sink at DEBUG, like the reference's loguru sink (pipe_core.py:37-46), via stdlib logging."""
    from logging.handlers import RotatingFileHandler

    Path(file).parent.mkdir(parents=True, exist_ok=True)
    handler = RotatingFileHandler(file, maxBytes=10 * 1024 * 1024, backupCount=5)
    handler.setFormatter(logging.Formatter("%(asctime)s | %(levelname)-8s | %(name)s:%(funcName)s:%(lineno)d - %(message)s"))
    logger.handlers = [handler]
    logger.setLevel(logging.DEBUG)


# ---------------------------------------------------------------------------------------------
# step initialisers
# ---------------------------------------------------------------------------------------------


def _init_tile(step_name: str, parameters: dict) -> Callable:
    image_kwargs = parameters.pop("image_kwargs", None)
    if image_kwargs is None:
        raise ValueError(f"Step '{step_name}' is missing required 'image_kwargs'.")
    if "source" not in image_kwargs:
        raise ValueError(f"Step '{step_name}' 'image_kwargs' is missing required 'source'.")
    constructor = dispatch_tiler(parameters.pop("kind", None), parameters)
    image = dispatch_image(source=image_kwargs["source"])(**image_kwargs)
    return constructor(image)


def _init_extract(step_name: str, parameters: dict, *, overlap: bool) -> Callable:
    if "tree" not in parameters:
        raise ValueError(f"Step '{step_name}' is missing required 'tree'.")
    process, measure_fn = process_tree_masks, extract_tree
    if overlap:
        process, measure_fn = process_tree_masks_overlap, partial(extract_tree, overlap=True)
    return partial(process, measure_fn=measure_fn, tree=parameters["tree"], **parameters.get("kwargs", {}))


def _init_extract_multi(step_name: str, parameters: dict) -> Callable:
    if "tree" not in parameters:
        raise ValueError(f"Step '{step_name}' is missing required 'tree'.")
    return partial(process_tree_masks, measure_fn=extract_tree_multi, tree=parameters["tree"],
                   **parameters.get("kwargs", {}))


def _init_nahual(step_name: str, parameters: dict) -> Callable:
    if parameters.get("address") is None:
        raise ValueError(f"If using Nahual you must have an address, currently it is None in step '{step_name}'")
    raise NotImplementedError(f"'{step_name}': Nahual remote steps are out of scope (SURVEY §2 rows 6/16)")


def run_step(step, *args, **kwargs):
    if hasattr(step, "run_tp"):
        return step.run_tp(*args, **kwargs)
    kwargs.pop("tp", None)
    return step(*args, **kwargs)


# ---------------------------------------------------------------------------------------------
# per-timepoint loop
# ---------------------------------------------------------------------------------------------


def pipeline_step(pipeline: dict, state: dict | None, steps_dir, init_step_fn: Callable) -> dict:
    """One timepoint: init-once, wire passed_data / passed_methods, run, save, retain."""
    steps = pipeline["steps"]
    passed_methods = pipeline.get("passed_methods", {})
    if not state:
        state = {"tps": dict(zip(steps, cycle([0]))), "data": {}, "fn": {}}
    tp = next(iter(state["tps"].values()))

    for step_name, parameters in steps.items():
        state["data"].setdefault(step_name, [])
        if step_name not in state["fn"]:
            state["fn"][step_name] = init_step_fn(step_name, parameters, state["fn"])
        step = state["fn"][step_name]

        passed_data = {}
        for kwd, from_step, *varname in pipeline["passed_data"].get(step_name, {}):
            history = state["data"].get(from_step, [])
            argname = varname[0] if varname else kwd
            if len(history):
                if step_name == "track" and kwd == "masks":
                    passed_data[argname] = [
                        [tp_tiles[tile] for tp_tiles in history[-2:]] for tile in range(len(history[-1]))
                    ]
                else:
                    last = history[-1]
                    if isinstance(last, dict):
                        last = last[kwd]
                    passed_data[argname] = last

        args = ()
        method_spec = passed_methods.get(step_name)
        if method_spec is not None and step_name.startswith("segment"):
            source_step, method = method_spec
            args = (getattr(state["fn"][source_step], method)(tp),)

        result = run_step(step, *args, tp=tp, **passed_data)

        to_write = pipeline.get("save") or []
        interval = pipeline.get("save_interval", 1)
        if bool(to_write) and interval > 0 and (tp % interval) == 0 and step_name in to_write:
            print(f"Saving {step_name} to {steps_dir}")
            dispatch_write_fn(step_name)(result, steps_dir=steps_dir, subpath=step_name, tp=tp)

        state["data"][step_name].append(result)
        state["tps"][step_name] = tp + 1

    for step_name, history in state["data"].items():
        if step_name.startswith("tile") and history:
            entry = history[-1]
            if isinstance(entry, dict) and "pixels" in entry:
                del entry["pixels"]

    for step_name, history in state["data"].items():
        keep = pipeline.get("retain", {}).get(step_name, "all")
        if isinstance(keep, int) and keep >= 0 and len(history) > keep:
            del history[: len(history) - keep]
    return state


def validate_pipeline(pipeline: dict) -> None:
    if not isinstance(pipeline, dict):
        raise TypeError("Pipeline configuration must be a dictionary.")
    if "steps" not in pipeline or not isinstance(pipeline["steps"], dict):
        raise ValueError("Pipeline must contain a 'steps' dictionary mapping step names to parameters.")
    steps = pipeline["steps"]
    if "passed_data" not in pipeline or not isinstance(pipeline["passed_data"], dict):
        raise ValueError("Pipeline must contain a 'passed_data' dictionary.")
    passed_data = pipeline["passed_data"]
    for target, deps in passed_data.items():
        if not isinstance(deps, (list, tuple)):
            raise TypeError(f"'passed_data' dependencies for step '{target}' must be a sequence.")
        for dep in deps:
            if not isinstance(dep, (list, tuple)) or len(dep) < 2:
                raise ValueError(f"Invalid dependency format in 'passed_data' for '{target}': {dep}")
            if dep[1] not in steps:
                raise ValueError(
                    f"Step '{target}' expects data from '{dep[1]}', but '{dep[1]}' is not defined in 'steps'."
                )
    passed_methods = pipeline.get("passed_methods", {})
    if not isinstance(passed_methods, dict):
        raise TypeError("'passed_methods' must be a dictionary.")
    for target, dep in passed_methods.items():
        if not isinstance(dep, (list, tuple)) or len(dep) < 2:
            raise ValueError(f"Invalid method dependency format for '{target}': {dep}")
        if dep[0] not in steps:
            raise ValueError(
                f"Step '{target}' expects a method from '{dep[0]}', but '{dep[0]}' is not defined in 'steps'."
            )
    save = pipeline.get("save")
    if save is not None:
        if not isinstance(save, (list, tuple, set)):
            raise TypeError("'save' must be a sequence of step names.")
        for step in save:
            if step not in steps and step not in pipeline.get("global_steps", {}):
                raise ValueError(
                    f"Step '{step}' listed in 'save' is not defined in the pipeline 'steps' or 'global_steps'."
                )
    if "save_interval" in pipeline:
        si = pipeline["save_interval"]
        if not isinstance(si, int) or isinstance(si, bool) or si < 1:
            raise ValueError(f"'save_interval' must be a positive int, got {si!r}.")
    retain = pipeline.get("retain", {})
    if not isinstance(retain, dict):
        raise TypeError("'retain' must be a dictionary mapping step name to int or 'all'.")
    for step_name, keep in retain.items():
        if step_name not in steps:
            raise ValueError(f"'retain' references step '{step_name}' not defined in 'steps'.")
        if keep != "all" and not (isinstance(keep, int) and not isinstance(keep, bool) and keep >= 0):
            raise ValueError(f"'retain[{step_name}]' must be a non-negative int or 'all', got {keep!r}.")
        track_reads = any(
            dep[1] == step_name for target, deps in passed_data.items() if target.startswith("track") for dep in deps
        )
        if track_reads and isinstance(keep, int) and keep < 2:
            raise ValueError(
                f"'retain[{step_name}]' = {keep} is too small; per-tp 'track' step "
                f"reads the last 2 timepoints of '{step_name}'."
            )
    for k, params in steps.items():
        if not isinstance(params, dict):
            raise TypeError(f"Parameters for step '{k}' must be a dictionary.")
        if k.startswith("nahual") and "address" not in params:
            raise ValueError(f"Nahual-deployed step '{k}' must provide an 'address' parameter.")
    if pipeline.get("global_steps", {}):
        if "global_passed_data" not in pipeline:
            raise ValueError("Pipeline defines 'global_steps' but is missing 'global_passed_data'.")
        if not isinstance(pipeline["global_passed_data"], dict):
            raise TypeError("'global_passed_data' must be a dictionary.")


def run_pipeline_return_state(pipeline: dict, steps_dir, init_step_fn: Callable) -> dict:
    validate_pipeline(pipeline)
    state = {}
    for _ in range(pipeline.get("ntps", 1)):
        state = pipeline_step(pipeline, state, steps_dir, init_step_fn)
    return state


def _run_pipeline_and_post_impl(pipeline: dict, pipeline_name: str, output_path, overwrite: bool = True, *,
                                init_step_fn: Callable, post_state_hook: Callable | None = None):
    """Run one position; write `profiles/<name>.parquet` (zstd) and `steps/<name>/...`; resume by skip."""
    output_path = Path(output_path)
    steps_dir = output_path / "steps" / pipeline_name
    profiles_file = output_path / "profiles" / f"{pipeline_name}.parquet"
    profiles, post_results = None, None
    if overwrite or not profiles_file.exists():
        state = run_pipeline_return_state(pipeline, steps_dir, init_step_fn)
        profiles = get_profiles_from_state(state, pipeline)
        profiles_file.parent.mkdir(parents=True, exist_ok=True)
        pyarrow.parquet.write_table(profiles, profiles_file, compression="zstd")
        if post_state_hook is not None:
            post_state_hook(state, pipeline, output_path, pipeline_name)
        post_results = {}
        if pipeline.get("global_steps"):
            raise NotImplementedError("global steps (trackastra via Nahual) are out of scope (SURVEY §2 row 16)")
    else:
        logger.info(f"Skipping {pipeline_name}")
    return profiles, post_results


def get_profiles_from_state(state: dict, pipeline: dict) -> pa.Table:
    """format_extraction per (extract step, tp) -> metadata columns -> concat per prefix -> join."""
    profiles = pa.Table.from_pylist(
        [],
        schema=pa.schema([
            pa.field("metadata_tile", pa.int64()),
            pa.field("metadata_label", pa.int64()),
            pa.field("metadata_object", pa.string()),
            pa.field("metadata_tp", pa.int64()),
        ]),
    )
    feature_steps = [s for s in pipeline["steps"] if s.startswith("extract") or s.startswith("nahual_embed")]
    data = {s.split("_")[0]: [] for s in feature_steps}
    for ext_step in feature_steps:
        prefix = ext_step.split("_")[0]
        for tp, ext_output in enumerate(state["data"][ext_step]):
            if isinstance(ext_output, numpy.ndarray):
                ext_output = ((("__", "__"),), (ext_output,))
            table = format_extraction(ext_output)
            rename = {"tile": "metadata_tile", "label": "metadata_label"}
            table = table.rename_columns([rename.get(c, c) for c in table.column_names])
            if len(table):
                table = table.append_column(
                    "metadata_object", pa.array([ext_step.split("_")[-1]] * len(table), pa.string())
                )
                table = table.append_column("metadata_tp", pa.array([tp] * len(table), pa.uint16()))
                data[prefix].append(table)
    wide = [pa.concat_tables(tabs) for tabs in data.values() if len(tabs)]
    if wide:
        profiles = wide[0]
        for table in wide[1:]:
            profiles = profiles.join(table, keys=[f"metadata_{k}" for k in ("tp", "tile", "object", "label")])
    return profiles


def get_step_output(state_data: dict, fetchers, steps_dir=None) -> numpy.ndarray:
    combined = []
    for fetcher in fetchers:
        if isinstance(fetcher, str):
            if fetcher.startswith("from_disk:"):
                if steps_dir is None:
                    raise ValueError("from_disk fetcher requires steps_dir; pass it through get_step_output(..., steps_dir=...)")
                out = _load_per_tp_masks(Path(steps_dir) / fetcher.removeprefix("from_disk:"))
            else:
                out = [x[0] for x in state_data[fetcher]]
        elif callable(fetcher):
            out = fetcher(state_data)
        else:
            raise Exception(f"Invalid type, expected Callable or string, got {type(fetcher)}")
        combined.append(out)
    return numpy.asarray(combined)


def _load_per_tp_masks(step_dir: Path) -> list:
    files = sorted(Path(step_dir).glob("*.npz"))
    if not files:
        raise FileNotFoundError(
            f"No per-tp .npz files found under {step_dir}; ensure this step is listed in pipeline['save']."
        )
    masks = []
    for f in files:
        with numpy.load(f) as npz:
            keys = list(npz.keys())
            if "tile_0" in keys:
                masks.append(npz["tile_0"])
            elif keys == ["arr_0"]:
                masks.append(npz["arr_0"][0])
            else:
                raise ValueError(f"Unrecognised .npz layout in {f}: keys={keys}")
    return masks
