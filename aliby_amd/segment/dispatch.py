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
from aliby_amd import trace as _trace


def _to_uint16_labels(labels: np.ndarray) -> np.ndarray:
    if labels.size and labels.max() >= np.iinfo(np.uint16).max:
        raise OverflowError(f"Segmentation produced {labels.max()} labels; uint16 cast unsafe.")
    return labels.astype(np.uint16, copy=False)


def _as_one_block(blocks):
    """Device blocks that are consecutive slices of one allocation (the batched runner uploads them that way) -> one tensor
    over all of them, else None."""
    import torch

    first = blocks[0]
    if len(blocks) < 2 or not all(isinstance(b, torch.Tensor) and b.is_contiguous() and b.dtype == first.dtype
                                  and b.shape[1:] == first.shape[1:] for b in blocks):
        return None
    expect = first.data_ptr()
    for b in blocks:
        if b.data_ptr() != expect or b.untyped_storage().data_ptr() != first.untyped_storage().data_ptr():
            return None
        expect += b.numel() * b.element_size()
    return torch.as_strided(first, (sum(b.shape[0] for b in blocks), *first.shape[1:]), first.stride())


_MODELS: list = []  # [(gpu, device, setup parameters, CellposeModel)], most recently used last


def _model_for(gpu, device, setup_params):
    """The CellposeModel for these parameters, built once per process: the reference builds one per position
    (init_step -> dispatch_segmenter -> CellposeModel(...), pipe.py:47-77), which costs this build ~50 ms of weight packing and
    workspace allocation per position — three quarters of a 1024^2 position's run_pipeline_and_post call.  The model keeps no
    state between evals; the last 4 parameter sets are kept, `aliby_amd.runner.release_pinned()` drops them."""
    from aliby_amd.segment.cellpose_hip import CellposeModel

    def same(a, b):
        try:
            return bool(a == b)
        except (ValueError, RuntimeError, TypeError):  # (arrays / tensors among the values)
            return False

    for k in range(len(_MODELS) - 1, -1, -1):
        g, d, sp, model = _MODELS[k]
        if g == gpu and d == device and same(sp, setup_params):
            _MODELS.append(_MODELS.pop(k))
            return model
    model = CellposeModel(gpu=gpu, device=device, **setup_params)
    _MODELS.append((gpu, device, dict(setup_params), model))
    del _MODELS[:-4]
    return model


def dispatch_segmenter(kind: str, channel_to_segment: int, address: str = None, **kwargs) -> callable:
    if kind in ("nahual_baby", "nahual_cellpose", "nahual_spotiflow") or (kind or "").startswith("nahual"):
        raise NotImplementedError(f"segmenter kind '{kind}' is a remote Nahual service (SURVEY §2 row 6): out of scope")
    if kind != "cellpose":
        raise Exception(f"Invalid segmentation method {kind}")

    # Extension (not in the reference): per_tile=True returns one label image per tile, the container BABY's parser hands
    # the reference's engine (list of [Y,X] masks, extract.py:271-281), instead of collapsing the tile axis as a Z axis
    # (dispatch.py:218-223: what the reference's cellpose branch does to a multi-tile batch).  The trap-tile time-lapse
    # workload (BASELINE config 4: tiles -> segment -> track -> extract) needs it; default False = reference behaviour.
    per_tile = bool(kwargs.get("per_tile", False))
    setup_params = dict(kwargs.get("setup_params", {}))
    gpu = setup_params.pop("gpu", True)
    device = setup_params.pop("device", None)
    model = _model_for(gpu, device, setup_params)

    def _device_block(pixels):
        """host array / device tensor [F,C,Z,Y,X] of any real dtype -> device tensor the model takes (uint16 stays uint16,
        uint8 / bool are widened losslessly, everything else becomes float32: dispatch.py:179-215 accepts any dtype)."""
        import torch

        hit = devcache.lookup(pixels) if isinstance(pixels, np.ndarray) else None
        if hit is not None:
            return hit[0]
        if isinstance(pixels, torch.Tensor):
            t = pixels.to(model.device)
        else:
            a = np.ascontiguousarray(pixels)
            if a.dtype in (np.uint8, np.bool_):
                a = a.astype(np.uint16)
            elif a.dtype != np.uint16:
                a = a.astype(np.float32)
            t = torch.from_numpy(a).to(model.device)
        if t.dtype == torch.uint8 or t.dtype == torch.bool:
            t = t.to(torch.int32).to(torch.uint16)
        elif t.dtype not in (torch.uint16, torch.float32):
            t = t.to(torch.float32)
        return t

    def _labels_of(blocks, kw):
        """One model.eval over every tile of every block: list of device blocks [F_i,C,Z,Y,X] -> labels [sum F_i,Y,X], counts."""
        import torch

        whole = _as_one_block(blocks)
        if whole is not None:
            plane = model.select_and_project(whole, channel_to_segment)
        else:
            planes = [model.select_and_project(b, channel_to_segment) for b in blocks]  # [F_i,Y,X], max over Z if Z>1
            plane = planes[0] if len(planes) == 1 else torch.cat(planes, 0)
        result = model.eval(plane, do_3D=False, stitch_threshold=0.0, normalize=kw.pop("normalize", True), z_axis=None, **kw)
        labels = result[0]
        _trace.mark("segment:eval returned")
        return (labels if labels.ndim == 3 else labels[None]), np.asarray(model.last_counts)

    def _finish(stack, counts, host_stack=None, ready=None):
        """The reference's post-processing of one position's label stack [F,Y,X] (dispatch.py:216-234) -> step result.
        host_stack: the same stack already on the host (the batched entry downloads every position's labels in one copy)."""
        if per_tile:
            if counts.size and counts.max() >= np.iinfo(np.uint16).max:
                raise OverflowError(f"Segmentation produced {counts.max()} labels; uint16 cast unsafe.")
            host = stack.cpu().numpy() if host_stack is None else host_stack
            return [devcache.attach(host[k], stack[k], kind="labels", ready=ready, max_label=int(counts[k])) for k in range(host.shape[0])]
        if stack.shape[0] > 1:
            # reference: a 3-D result is collapsed, labels.max(axis=0) then relabel_sequential (dispatch.py:218-223)
            labels_dev = model.max_project_and_relabel(stack)
            n_labels = model.count_labels(labels_dev)
        else:
            labels_dev = stack[0]  # "Cellpose squeezes dims"
            n_labels = int(counts[0]) if counts.size else 0
        if n_labels >= np.iinfo(np.uint16).max:
            raise OverflowError(f"Segmentation produced {n_labels} labels; uint16 cast unsafe.")
        if host_stack is not None and stack.shape[0] == 1:
            return devcache.attach(host_stack[0], labels_dev, kind="labels", ready=ready, max_label=n_labels)
        return devcache.attach(labels_dev.cpu().numpy(), labels_dev, kind="labels", max_label=n_labels)

    def segment(pixels, do_3D: bool = False, stitch_threshold=None, **kw):
        """Assumes FCZYX pixels.  Returns uint16 labels [Y,X] (monotile), as the reference does."""
        z_size = pixels.shape[2]
        if pixels.ndim > 5:
            pixels = pixels[0]
        if do_3D and z_size > 1:
            return _segment_volume(pixels, dict(kw))
        stack, counts = _labels_of([_device_block(pixels)], dict(kw))
        return _finish(stack, counts)

    def _segment_volume(pixels, kw):
        """The reference's do_3D branch (dispatch.py:193-198, 216-223): the stack goes to the model with z_axis = 1,
        stitch_threshold = 0.01, normalize = dict(norm3D=False); whatever 3-D labels come back are collapsed with max(axis=0) and
        relabelled, so the step's result is 2-D again.  Here (round 3, an extension — cellpose itself is not vendored and its
        do_3D mode, flows from three orthogonal passes, is not built): every plane is segmented as a 2-D image (normalised on its
        own = norm3D False), the planes are stitched along Z by IoU >= 0.01 (cellpose's stitch3D rule, aliby_track_stitch), and
        the volume labels are kept on `segment.last_volume` = (uint16 device tensor [F,Z,Y,X], objects per stack) for 3-D
        features (FeatureEngine.intensity3d) before the reference's collapse."""
        import torch

        block = _device_block(pixels)  # [F,C,Z,Y,X]
        F, _, Z, Y, X = block.shape
        planes = block[:, channel_to_segment].reshape(F * Z, Y, X).contiguous()
        kw.pop("normalize", None)
        result = model.eval(planes, do_3D=False, stitch_threshold=0.0, normalize=dict(norm3D=False), z_axis=None, **kw)
        labels = result[0] if result[0].ndim == 3 else result[0][None]
        volume, counts = model.eng.stitch_planes(labels.view(F, Z, Y, X), threshold=0.01)
        segment.last_volume = (volume, counts)
        out = [_finish(volume[f], np.asarray([counts[f]])) for f in range(F)]
        return out[0] if F == 1 else out

    def segment_batch(blocks, pinned_alloc=None, **kw):
        """Position-batched form used by aliby_amd.runner: `blocks` = one FCZYX block per position (device tensors or host
        arrays); every tile of every position goes through ONE network / dynamics pass, then each position's stack gets the
        reference's post-processing on its own.  Returns one step result per position, identical to `segment(block)`."""
        devs = [_device_block(b) for b in blocks]
        stack, counts = _labels_of(devs, dict(kw))
        _trace.mark("segment:labels_of returned")
        # every position's labels in ONE download through a page-locked buffer (N pageable copies cost a sync each)
        import torch

        # ... and without waiting for it: the next consumers (extract, track) read the DEVICE labels through devcache; whoever
        # needs the host bytes (the .npz writer threads, a caller after the run) goes through devcache.wait_ready first
        # (a fresh page-locked buffer costs a hipHostMalloc of ~100 MB per batch on the launch thread: the runner lends its arena)
        pinned = (pinned_alloc(tuple(stack.shape), stack.dtype) if pinned_alloc is not None
                  else torch.empty(tuple(stack.shape), dtype=stack.dtype, pin_memory=True))  # (its NumPy views keep it alive)
        _trace.mark("segment:labels ready")
        # on a side stream: 134 MB for 64 frames is ~3 ms of PCIe time the feature kernels would otherwise queue behind
        side = getattr(segment_batch, "_copy_stream", None)
        if side is None:
            side = segment_batch._copy_stream = torch.cuda.Stream()
        done = torch.cuda.Event()
        done.record()
        side.wait_event(done)
        with torch.cuda.stream(side):
            pinned.copy_(stack, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(side)
        stack.record_stream(side)
        host_all = pinned.numpy()
        _trace.mark("segment:download queued")
        out, k = [], 0
        for d in devs:
            f = d.shape[0]
            out.append(_finish(stack[k : k + f], counts[k : k + f], host_all[k : k + f], ready))
            k += f
        return out

    segment.model = model
    segment.batch = segment_batch
    segment.channel_to_segment = channel_to_segment
    segment.per_tile = per_tile
    return segment
