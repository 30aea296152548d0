"""
ctypes binding of libaliby_hip.so (include/aliby_hip.h).

There is no CPU fallback: importing this module without the built library, or
creating a context without a visible MI355X, raises.  `oracle/` is never
imported from here.
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["ALIBY_HIP_LIB"]) if os.environ.get("ALIBY_HIP_LIB") else _HERE / "libaliby_hip.so"  # (the override: diagnostic builds)

OK, ERR_INVALID, ERR_OVERFLOW, ERR_HIP, ERR_TOO_LARGE, ERR_UNSUPPORTED = range(6)
U16, F32 = 0, 1
U8W = 4  # uint8 / bool pixels in uint16 storage: texture's grey level is the value itself (ALIBY_U8W)
U64, F64 = 2, 3  # aliby_reduce_z output dtypes (NumPy's result types for uint16 add / divide)
RED_MAX, RED_ADD, RED_DIV = 0, 1, 2


class AlibyHipError(RuntimeError):
    pass


class aliby_object(C.Structure):
    _fields_ = [
        ("tile", C.c_int32),
        ("label", C.c_int32),
        ("y0", C.c_int32),
        ("x0", C.c_int32),
        ("y1", C.c_int32),
        ("x1", C.c_int32),
        ("area", C.c_int32),
        ("pad_", C.c_int32),
    ]


class aliby_pq_column(C.Structure):
    _fields_ = [("name", C.c_char_p), ("type", C.c_int32), ("reserved", C.c_int32)]


class aliby_npy_member(C.Structure):
    _fields_ = [("name", C.c_char_p), ("descr", C.c_char_p), ("shape", C.POINTER(C.c_int64)), ("data", C.c_void_p),
                ("ndim", C.c_int32), ("itemsize", C.c_int32)]


PQ_F64, PQ_I64, PQ_U16, PQ_STR = range(4)

OBJECT_DTYPE = np.dtype(
    [("tile", "<i4"), ("label", "<i4"), ("y0", "<i4"), ("x0", "<i4"), ("y1", "<i4"), ("x1", "<i4"),
     ("area", "<i4"), ("pad_", "<i4")]
)

_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
_ip = C.POINTER(C.c_int32)

# name -> (restype, argtypes); mirrors include/aliby_hip.h one to one
_SIGNATURES = {
    "aliby_abi_version": (_i, []),
    "aliby_last_error": (C.c_char_p, []),
    "aliby_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "aliby_ctx_destroy": (_i, [_vp]),
    "aliby_device_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_sz), C.c_char_p, _i]),
    "aliby_malloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "aliby_free": (_i, [_vp, _vp]),
    "aliby_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "aliby_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "aliby_memset": (_i, [_vp, _vp, _i, _sz, _vp]),
    "aliby_stream_sync": (_i, [_vp, _vp]),
    "aliby_crop_pad_u16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "aliby_reduce_z": (_i, [_vp, _vp, _i, _sz, _i, _sz, _i, _vp, _i, _vp]),
    "aliby_label_max": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "aliby_object_table": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "aliby_relabel_sequential": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "aliby_select_project_u16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "aliby_normalize99_u16": (_i, [_vp, _vp, _i, _i, _i, C.c_double, C.c_double, _vp, _vp, _vp]),
    "aliby_make_tiles": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "aliby_average_tiles": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "aliby_nn_fused_act_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "aliby_nn_conv3x3_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "aliby_nn_conv3x3_deep_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "aliby_nn_conv3x3_deep16_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "aliby_nn_pack_conv3x3_deep16_bf16": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "aliby_nn_maxpool2_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "aliby_debug_conv_deep_trace": (_i, [_vp, _vp]),
    "aliby_track_stitch": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, _vp, _vp, _vp]),
    "aliby_debug_conv_trace": (_i, [_vp, _vp]),
    "aliby_features_coloc_pairs": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _i, C.c_double, C.c_double, _vp, _vp, _vp]),
    "aliby_trap_gauss1d": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "aliby_trap_warp": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, C.c_double, _vp]),
    "aliby_trap_entropy": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "aliby_trap_morph": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "aliby_trap_label": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "aliby_trap_region_sums": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "aliby_trap_match_template": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _i, C.c_double, C.c_double, _vp]),
    "aliby_trap_maxfilter1d": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "aliby_tiff_probe": (_i, [C.c_char_p, _vp, C.c_char_p, _i]),
    "aliby_ingest_tiff_planes": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _i, _i, _vp]),
    "aliby_ingest_inflate": (_i, [_i, _vp, _sz, _vp, _sz, _vp]),
    "aliby_nn_pack_conv3x3_bf16": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "aliby_nn_pack_conv1x1_bf16": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "aliby_nn_conv3x3_proj_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "aliby_nn_conv3x3_head_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "aliby_nn_conv3x3_pair_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                                        _i, _vp, _vp]),
    "aliby_nn_first_pair_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "aliby_nn_conv1x1_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "aliby_nn_first_conv_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "aliby_nn_style_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "aliby_nn_out_head_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "aliby_nn_nhwc_to_nchw_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "aliby_nn_tiles_to_nhwc8_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "aliby_masks_workspace_bytes": (_sz, [_i, _i, _i]),
    "aliby_masks_from_flows": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, C.c_float, C.c_float, _i, C.c_float, _vp, _sz,
                                    _vp, _vp, _vp, _vp]),
    "aliby_features_intensity": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "aliby_features_sizeshape": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "aliby_features_feret": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp]),
    "aliby_object_mec": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "aliby_features_zernike": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp]),
    "aliby_features_radial_zernikes_multi": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _vp]),
    "aliby_features_texture": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "aliby_radial_geometry": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "aliby_features_radial_distribution": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp]),
    "aliby_features_cell_ratio": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "aliby_features_trap_background": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "aliby_granularity_workspace_bytes": (_sz, [_i, _i, _i, _i, C.c_double, C.c_double]),
    "aliby_features_granularity": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, C.c_double, C.c_double, _i, _i, _i,
                                        _vp, _sz, _vp, _i, _i, _vp]),
    "aliby_radial_geometry_unscaled": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, C.c_double, _vp, _vp]),
    "aliby_features_radial_distribution_rings": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "aliby_features_cell": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "aliby_features_coloc": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _i,
                                  C.c_double, C.c_double, _vp, _vp, _vp]),
    "aliby_labels_apply_lut": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "aliby_features_intensity3d": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp]),
    "aliby_parquet_write": (_i, [C.c_char_p, _vp, _i, _vp, _i, _vp, _vp, _i]),
    "aliby_npz_write": (_i, [C.c_char_p, _vp, _i, _i]),
    "aliby_host_codecs": (_i, [C.POINTER(_i), C.POINTER(_i)]),
    "aliby_host_copy": (_i, [_vp, _vp, _sz, _i]),
    "aliby_object_ranks": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp]),
}

_lib = None


def exported_symbols() -> list[str]:
    return sorted(_SIGNATURES)


def load() -> C.CDLL:
    """Load the shared library (once).  torch is imported first so that its
    bundled HIP runtime (same SONAME, libamdhip64.so.7) is the one both share."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise AlibyHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C aliby_amd/csrc). There is no CPU fallback for the HIP path."
        )
    try:
        import torch  # noqa: F401  (loads libamdhip64 before our DT_NEEDED resolves)
    except Exception:  # pragma: no cover - torch is part of the image
        pass
    lib = C.CDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library drift apart
        fn.restype = res
        fn.argtypes = args
    if lib.aliby_abi_version() != 1:
        raise AlibyHipError("libaliby_hip.so ABI version mismatch")
    _lib = lib
    return lib


_fast = None


def load_fast() -> C.PyDLL:
    """The same library through ctypes.PyDLL: calls keep the interpreter lock.  For the short, asynchronous launch functions
    of the network (~600 calls per 64-position batch): a CDLL call drops the lock and must win it back from the writer threads
    every time, which showed up as idle gaps on the device.  Never use it for a call that blocks (stream waits, file writes)."""
    global _fast
    if _fast is None:
        load()
        lib = C.PyDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2))
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _fast = lib
    return _fast


def check(rc: int) -> None:
    """Map C status codes to the exception types the reference raises."""
    if rc == OK:
        return
    msg = load().aliby_last_error().decode("utf-8", "replace")
    if rc == ERR_INVALID:
        raise ValueError(msg)
    if rc == ERR_OVERFLOW:
        raise OverflowError(msg)
    if rc == ERR_UNSUPPORTED:
        raise Exception(msg)
    raise AlibyHipError(msg)


class Context:
    """Owns an aliby_ctx*; one per (thread, GPU): see default_context (one process per GPU, SURVEY §8b "Threading")."""

    def __init__(self, device: int = 0):
        lib = load()
        h = _vp()
        check(lib.aliby_ctx_create(device, C.byref(h)))
        self.handle = h
        self.lib = lib
        self.device = device

    def info(self) -> dict:
        cu, lds, hbm = _i(), _i(), _sz()
        name = C.create_string_buffer(128)
        check(self.lib.aliby_device_info(self.handle, C.byref(cu), C.byref(lds), C.byref(hbm), name, 128))
        return dict(cu_count=cu.value, lds_bytes=lds.value, hbm_bytes=hbm.value, name=name.value.decode())

    def close(self):
        if getattr(self, "handle", None):
            self.lib.aliby_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


_tls = threading.local()


def default_context(device: int | None = None) -> Context:
    """The calling THREAD's context for `device`.  A context owns a scratch block that several entry points use for a few
    words each (tile rectangles of the stager, per-frame offsets of the object table and the dynamics' flow QC): the
    position-batched runner stages the next batch on an ingest thread while the launch thread is inside the dynamics, and with
    one context per device the stager's rectangles could land on the offsets a queued kernel was about to read.  One context
    per (thread, device); it goes away with its thread."""
    if device is None:
        import torch

        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    have = _tls.__dict__.setdefault("ctx", {})
    if device not in have:
        have[device] = Context(device)
    return have[device]
