"""
Cellpose + cp_measure pipeline on MI355X: the plug-in point.

Same seam as the reference (src/aliby/pipe.py:47-77): `init_step` picks the step implementation from
the step-name prefix and `run_pipeline_and_post` is the engine bound to it.  A pipeline dict built by
`build_pipeline_steps` (ours or the reference's) runs unchanged.
"""

from __future__ import annotations

from functools import partial
from typing import Callable

from aliby_amd import pipe_core as core
from aliby_amd.segment.dispatch import dispatch_segmenter


def _segmenter(step_name: str, parameters: dict, _initialised: dict) -> Callable:
    """Cellpose on the HIP path; `segmenter_kwargs` go to the dispatcher untouched."""
    channel = parameters.get("channel_to_segment", parameters)
    if channel is parameters:
        raise ValueError(f"Step '{step_name}' is missing required 'channel_to_segment'.")
    return dispatch_segmenter(channel_to_segment=channel, **parameters.get("segmenter_kwargs", {}))


def _tracker(step_name: str, parameters: dict, _initialised: dict) -> Callable:
    """`dispatch_tracker(**parameters)` like the reference (pipe.py:41-44); the `stitch` kind is the HIP IoU stitcher."""
    from aliby_amd.track.stitch import dispatch_tracker

    return dispatch_tracker(**parameters)


# first matching prefix wins, in the reference's order (pipe.py:56-72)
_PREFIXES = (
    ("tile", lambda name, params, done: core._init_tile(name, params)),
    ("segment", _segmenter),
    ("track", _tracker),
    ("extract_", lambda name, params, done: core._init_extract(name, params, overlap=False)),
    ("extractmulti_", lambda name, params, done: core._init_extract_multi(name, params)),
    ("nahual_embed", lambda name, params, done: core._init_nahual(name, params)),
    ("nahual_track", lambda name, params, done: core._init_nahual(name, params)),
)


def init_step(step_name: str, parameters: dict, other_steps: dict | None = None) -> Callable:
    """Set up any step of the cellpose pipeline from its name prefix."""
    for prefix, make in _PREFIXES:
        if step_name.startswith(prefix):
            return make(step_name, parameters, other_steps or {})
    raise ValueError(f"Invalid step name {step_name=}")


run_pipeline_and_post = partial(core._run_pipeline_and_post_impl, init_step_fn=init_step, post_state_hook=None)
