"""
Segmenter dispatch: builds the `segment(pixels, **kw) -> uint16 labels` step callable.

Mirrors the `kind="cellpose"` branch of src/aliby/segment/dispatch.py:158-234 (`_to_uint16_labels`
14-19): channel select (192), Z max-projection or squeeze (199-206), `model.eval(..., do_3D=False,
stitch_threshold=0.0, normalize=True, z_axis=None)` (208-215), 3-D result -> max over axis 0 +
relabel_sequential (218-223), `>= 65535` -> OverflowError (230-233), cast to uint16.
The Nahual kinds are remote RPC to other people's servers: out of scope, they raise.

`model` is aliby_amd.segment.cellpose_hip.CellposeModel: PyTorch-ROCm only for the network forward,
everything around it (normalisation, tiling/blending, flow following, labelling, QC) in HIP kernels.
"""

from __future__ import annotations

import numpy as np

from aliby_amd import devcache


def _to_uint16_labels(labels: np.ndarray) -> np.ndarray:
    if labels.size and labels.max() >= np.iinfo(np.uint16).max:
        raise OverflowError(f"Segmentation produced {labels.max()} labels; uint16 cast unsafe.")
    return labels.astype(np.uint16, copy=False)


def dispatch_segmenter(kind: str, channel_to_segment: int, address: str = None, **kwargs) -> callable:
    if kind in ("nahual_baby", "nahual_cellpose", "nahual_spotiflow") or (kind or "").startswith("nahual"):
        raise NotImplementedError(f"segmenter kind '{kind}' is a remote Nahual service (SURVEY §2 row 6): out of scope")
    if kind != "cellpose":
        raise Exception(f"Invalid segmentation method {kind}")

    from aliby_amd.segment.cellpose_hip import CellposeModel

    # Extension (not in the reference): per_tile=True returns one label image per tile, the container BABY's parser hands
    # the reference's engine (list of [Y,X] masks, extract.py:271-281), instead of collapsing the tile axis as a Z axis
    # (dispatch.py:218-223: what the reference's cellpose branch does to a multi-tile batch).  The trap-tile time-lapse
    # workload (BASELINE config 4: tiles -> segment -> track -> extract) needs it; default False = reference behaviour.
    per_tile = bool(kwargs.get("per_tile", False))
    setup_params = dict(kwargs.get("setup_params", {}))
    gpu = setup_params.pop("gpu", True)
    device = setup_params.pop("device", None)
    model = CellposeModel(gpu=gpu, device=device, **setup_params)

    def segment(pixels, do_3D: bool = False, stitch_threshold=None, **kw):
        """Assumes FCZYX pixels.  Returns uint16 labels [Y,X] (monotile), as the reference does."""
        dev_hit = devcache.lookup(pixels) if isinstance(pixels, np.ndarray) else None
        z_size = pixels.shape[2]
        if pixels.ndim > 5:
            pixels = pixels[0]
            dev_hit = None
        if do_3D and z_size > 1:
            raise NotImplementedError("3-D Cellpose (do_3D) is beyond what the pipeline wires (SURVEY §8d C5 note)")
        src = dev_hit[0] if dev_hit is not None else pixels
        plane = model.select_and_project(src, channel_to_segment)  # device [F,Y,X], max over Z if Z>1
        result = model.eval(plane, do_3D=False, stitch_threshold=0.0, normalize=True, z_axis=None, **kw)
        labels_dev = result[0]  # device uint16 [F,Y,X] (or [Y,X] when F==1: "Cellpose squeezes dims")
        if per_tile:
            stack = labels_dev if labels_dev.ndim == 3 else labels_dev[None]
            if max(model.last_counts, default=0) >= np.iinfo(np.uint16).max:
                raise OverflowError(f"Segmentation produced {max(model.last_counts)} labels; uint16 cast unsafe.")
            host = stack.cpu().numpy()
            return [devcache.attach(host[k], stack[k], kind="labels") for k in range(host.shape[0])]
        if labels_dev.ndim == 3:
            # reference: labels.max(axis=0) then relabel_sequential (dispatch.py:218-223)
            labels_dev = model.max_project_and_relabel(labels_dev)
        elif not 1 < labels_dev.ndim < 4:
            raise Exception(f"Segmentation yielded {labels_dev.ndim} dimensions instead of 3")
        n_labels = model.count_labels(labels_dev)
        if n_labels >= np.iinfo(np.uint16).max:
            raise OverflowError(f"Segmentation produced {n_labels} labels; uint16 cast unsafe.")
        host = labels_dev.cpu().numpy()
        return devcache.attach(host, labels_dev, kind="labels")

    segment.model = model
    return segment
