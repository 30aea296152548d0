"""
Cellpose + cp_measure pipeline on MI355X: the plug-in point.

Same seam as the reference (src/aliby/pipe.py:47-77): `init_step` picks the step implementation from
the step-name prefix and `run_pipeline_and_post` is the engine bound to it.  A pipeline dict built by
`build_pipeline_steps` (ours or the reference's) runs unchanged.
"""

from __future__ import annotations

from functools import partial
from typing import Callable

from aliby_amd.pipe_core import (
    _init_extract,
    _init_extract_multi,
    _init_nahual,
    _init_tile,
    _run_pipeline_and_post_impl,
)
from aliby_amd.segment.dispatch import dispatch_segmenter


def _init_segment_cellpose(step_name: str, parameters: dict, other_steps: dict) -> Callable:
    seg_kwargs = parameters.get("segmenter_kwargs", {})
    if "channel_to_segment" not in parameters:
        raise ValueError(f"Step '{step_name}' is missing required 'channel_to_segment'.")
    return dispatch_segmenter(channel_to_segment=parameters["channel_to_segment"], **seg_kwargs)


def _init_track_cellpose(step_name: str, parameters: dict, other_steps: dict) -> Callable:
    raise NotImplementedError(
        "the reference's 'stitch' tracker is broken/deprecated (src/aliby/track/trackers.py:11,75,87; SURVEY §2 row 15)"
    )


_PREFIXES = (
    ("tile", lambda s, p, o: _init_tile(s, p)),
    ("segment", _init_segment_cellpose),
    ("track", _init_track_cellpose),
    ("extract_", lambda s, p, o: _init_extract(s, p, overlap=False)),
    ("extractmulti_", lambda s, p, o: _init_extract_multi(s, p)),
    ("nahual_embed", lambda s, p, o: _init_nahual(s, p)),
    ("nahual_track", lambda s, p, o: _init_nahual(s, p)),
)


def init_step(step_name: str, parameters: dict, other_steps: dict | None = None) -> Callable:
    """Set up any step of the cellpose pipeline; first matching prefix wins (pipe.py:56-72)."""
    other_steps = {} if other_steps is None else other_steps
    for prefix, init in _PREFIXES:
        if step_name.startswith(prefix):
            return init(step_name, parameters, other_steps)
    raise ValueError(f"Invalid step name {step_name=}")


run_pipeline_and_post = partial(_run_pipeline_and_post_impl, init_step_fn=init_step, post_state_hook=None)
