"""
Pipeline-definition builder for the Cellpose + cp_measure pipeline.

API-identical to the reference's `build_pipeline_steps` (src/aliby/pipe_builder.py:46-167) and
`_create_extract_multich_tree` (19-43): same keyword arguments, same defaults, same resulting dict
(steps / passed_data / passed_methods / save / save_interval), so a pipeline dict built by either
can be run by `aliby_amd.pipe.run_pipeline_and_post`.
"""

from __future__ import annotations

from itertools import combinations
from typing import Sequence

DEFAULT_FEATURES = ("radial_zernikes", "intensity", "feret", "texture", "radial_distribution", "zernike")
COLOC_METRICS = ("pearson", "costes", "manders_fold", "rwc")


def _step_kwargs(extract_ncores, cp_measure_feature_kwargs):
    kwargs = {"ncores": extract_ncores}
    if cp_measure_feature_kwargs:
        kwargs["cp_measure_kwargs"] = dict(cp_measure_feature_kwargs)
    return kwargs


def _create_extract_multich_tree(channels, extract_ncores, cp_measure_feature_kwargs=None) -> dict:
    """{(c0,c1): {"None": {"max": [pearson, costes, manders_fold, rwc]}}} for every channel pair."""
    tree = {pair: {"None": {"max": list(COLOC_METRICS)}} for pair in combinations(channels, r=2)}
    return {"tree": tree, "kwargs": _step_kwargs(extract_ncores, cp_measure_feature_kwargs)}


def build_pipeline_steps(
    channels_to_segment: dict[str, int] | None = None,
    channels_to_extract: Sequence[int] | None = None,
    features_to_extract: Sequence[str] = DEFAULT_FEATURES,
    extract_ncores: int | None = None,
    nahual_addresses=None,
    steps_to_write: Sequence[str] | None = None,
    trackastra_address: str | None = None,
    trackastra_parameters: dict | None = None,
    cp_measure_feature_kwargs: dict[str, dict] | None = None,
) -> dict:
    if channels_to_segment is None:
        channels_to_segment = {"nuclei": 1, "cell": 0}
    kind = "cellpose" if nahual_addresses is None else "nahual_cellpose"
    if channels_to_extract is None:
        channels_to_extract = list(channels_to_segment.values())
    objects = list(channels_to_segment)

    steps = {"tile": {"tile_size": None}}
    for obj, ch in channels_to_segment.items():
        steps[f"segment_{obj}"] = {"segmenter_kwargs": {"kind": kind}, "channel_to_segment": ch}

    # the single-channel spec is shared by every object set (as in the reference, where both
    # extract_<obj> entries point at the same dict)
    mono = {"tree": {"None": {"None": ("sizeshape",)}},
            "kwargs": _step_kwargs(extract_ncores, cp_measure_feature_kwargs)}
    for ch in channels_to_extract:
        mono["tree"][ch] = {"max": features_to_extract}
    multi = _create_extract_multich_tree(channels_to_extract, extract_ncores, cp_measure_feature_kwargs)
    for prefix, spec in (("extract", mono), ("extractmulti", multi)):
        if len(spec):
            for obj in objects:
                steps[f"{prefix}_{obj}"] = spec

    pipeline = {
        "steps": steps,
        "passed_data": {
            f"extract{m}_{obj}": [("masks", f"segment_{obj}"), ("pixels", "tile")]
            for obj in objects
            for m in ("", "multi")
        },
        "passed_methods": {f"segment_{obj}": ("tile", "get_fczyx") for obj in objects},
        "save": [f"segment_{obj}" for obj in objects],
        "save_interval": 1,
    }
    if steps_to_write is not None:
        pipeline["save"] = list(steps_to_write)
    if trackastra_address is not None:
        raise NotImplementedError("nahual_trackastra global steps are remote services (SURVEY §2 row 16): out of scope")
    return pipeline
