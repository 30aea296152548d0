"""
Image ingest on the CPU: the native TIFF decoder against files written by two independent encoders
(tests/golden/make_tiff_fixtures.py), the lazy TCZYX view and `adjust_dimensions`, the Image / Dataset classes on the
reference's own naming schemes (tests/common.py REGEX_PARAMETERS), and zarr v2 / v3 stores written by hand.
The decoder is host code inside libaliby_hip.so and needs no GPU.
"""

import gzip
import json
import shutil
import zlib
from pathlib import Path

import numpy as np
import pytest

from aliby_amd import _lib
from aliby_amd.io import dataset as ds
from aliby_amd.io import image as im

TIFFS = Path(__file__).parent / "golden" / "tiff"
EXPECTED = np.load(TIFFS / "expected.npz")

CASES = {
    "plain_strips7": "noisy", "deflate_pred": "noisy", "deflate_pred_big": "big", "zstd_tiles16": "noisy",
    "plain_tiles32x16": "noisy", "packbits_u8": "bytes", "bigendian_deflate_pred": "noisy", "bigendian_plain": "noisy",
    "bigtiff_f32": "floats", "signed_deflate": "signed", "pages6_deflate": "stack", "imagej_t2z3": "stack",
    "pil_lzw": "noisy", "pil_lzw_big": "big", "pil_lzw_pred_big": "big", "pil_lzw_smooth": "smooth",
    "pil_deflate": "noisy", "pil_packbits_u8": "bytes",
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_tiff_decoder_matches_the_encoders_input(name):
    want = EXPECTED[CASES[name]]
    info = im.tiff_info(TIFFS / f"{name}.tif")
    assert (info["height"], info["width"]) == want.shape[-2:]
    assert info["dtype"] == want.dtype
    assert info["pages"] == (want.shape[0] if want.ndim == 3 else 1)
    got = np.asarray(im.tiff_stack([str(TIFFS / f"{name}.tif")]))[0]
    assert got.dtype == want.dtype
    np.testing.assert_array_equal(got.reshape(want.shape), want)


def test_tiff_first_sample_of_chunky_rgb():
    got = np.asarray(im.tiff_stack([str(TIFFS / "rgb_chunky.tif")]))[0]
    np.testing.assert_array_equal(got, EXPECTED["rgb"][..., 0])


def test_tiff_threads_and_page_order():
    stack = im.tiff_stack([str(TIFFS / "pages6_deflate.tif")])
    src = stack.source
    out = np.zeros((6, 40, 52), np.uint16)
    for threads in (1, 4):
        out[:] = 0
        src.decode([5, 0, 3, 3, 1, 2], out.ctypes.data, n_threads=threads)
        np.testing.assert_array_equal(out, EXPECTED["stack"][[5, 0, 3, 3, 1, 2]])


def test_tiff_errors(tmp_path):
    bad = tmp_path / "not_a_tiff.tif"
    bad.write_bytes(b"PK\x03\x04 definitely something else")
    with pytest.raises(Exception, match="byte-order mark"):
        im.tiff_info(bad)
    cut = tmp_path / "cut.tif"
    cut.write_bytes((TIFFS / "deflate_pred.tif").read_bytes()[:600])
    with pytest.raises(Exception):
        np.asarray(im.tiff_stack([str(cut)]))
    with pytest.raises(Exception, match="cannot open"):
        im.tiff_info(tmp_path / "missing.tif")
    # second file of a list with another geometry
    with pytest.raises(Exception, match="geometry differs"):
        np.asarray(im.tiff_stack([str(TIFFS / "pil_lzw.tif"), str(TIFFS / "pil_lzw_big.tif")]))


# ------------------------------------------------------------------------------------- lazy view / adjust_dimensions
def test_lazy_array_indexing_matches_numpy():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 60000, (3, 2, 4, 6, 5)).astype(np.uint16)
    lazy = im.LazyArray(im.ArraySource(a))
    for index in [1, (2, 1), (slice(1, 3), 0, slice(None), 2), (0, 1, 3, slice(2, 5), slice(1, 4)), (Ellipsis, 2), (-1, -1, -1)]:
        np.testing.assert_array_equal(lazy[index], a[index])
    moved = lazy.moveaxis([2, 0, 1, 3, 4], range(5))
    np.testing.assert_array_equal(np.asarray(moved), np.moveaxis(a, [2, 0, 1, 3, 4], range(5)))
    np.testing.assert_array_equal(moved[3, 1], np.moveaxis(a, [2, 0, 1, 3, 4], range(5))[3, 1])
    with pytest.raises(IndexError):
        lazy[3]


@pytest.mark.parametrize(
    "shape, capture, want_shape",
    [
        ((5, 6), "YX", (1, 1, 1, 5, 6)),
        ((3, 5, 6), "CYX", (1, 3, 1, 5, 6)),
        ((3, 5, 6), "ZYX", (1, 1, 3, 5, 6)),
        ((2, 3, 5, 6), "YX", (1, 2, 3, 5, 6)),  # unnamed leading axes take the missing names from the end (C, Z)
        ((4, 2, 3, 5, 6), "CYX", (4, 3, 2, 5, 6)),  # unnamed (T, Z) then C
        ((1, 3, 5, 6), "WCYX", (1, 3, 1, 5, 6)),  # W is of size one and goes
        ((2, 5, 6, 3), "TYXC", (2, 3, 1, 5, 6)),
    ],
)
def test_adjust_dimensions_like_the_reference(shape, capture, want_shape):
    a = np.arange(int(np.prod(shape)), dtype=np.uint16).reshape(shape)
    out = im.adjust_dimensions(a, capture, "TCZYX")
    assert out.shape == want_shape
    # same rule written with NumPy only (image.py:527-599)
    order = capture
    if a.ndim > len(order):
        missing = [d for d in "TCZYX" if d not in order]
        order = "".join(missing[-(a.ndim - len(order)):]) + order
    b = a
    for i in range(len(order) - 1, -1, -1):
        if order[i] not in "TCZYX":
            b = np.squeeze(b, i)
            order = order[:i] + order[i + 1:]
    for d in sorted(d for d in "TCZYX" if d not in order):
        b = b[..., None]
        order += d
    b = np.moveaxis(b, [order.index(d) for d in "TCZYX"], range(5))
    np.testing.assert_array_equal(np.asarray(out), b)


def test_adjust_dimensions_refuses_to_drop_a_real_axis():
    with pytest.raises(AssertionError, match="must be 1 to be squeezed"):
        im.adjust_dimensions(np.zeros((2, 5, 6), np.uint16), "WYX", "TCZYX")


# ------------------------------------------------------------------------------------- Image / Dataset classes
def test_dispatch_image_types(tmp_path):
    assert im.dispatch_image({"path": ["a.tif", "b.tif"]}) is im.ImageList
    assert im.dispatch_image(["a.tif", "b.tif"]) is im.ImageList
    assert im.dispatch_image({"path": "/path.zarr", "key": "1"}) is im.ImageZarr
    assert im.dispatch_image("*.tif") is im.ImageList
    assert im.dispatch_image("img.tif") is im.ImageMultiTiff
    assert im.dispatch_image(str(tmp_path)) is im.ImageDir
    assert im.dispatch_image(np.zeros((1, 1, 1, 2, 2), np.uint16)) is im.ImageArray
    assert im.dispatch_image("stack.npy") is im.ImageArray


def _cell_painting_tree(root: Path):
    """crop_cellpainting_256-like names: <plate>__<well>__<field>__<channel>.tif"""
    sources = ["plain_strips7", "pil_lzw", "deflate_pred", "zstd_tiles16", "bigendian_plain"]
    truth = {}
    k = 0
    for well in ("A01", "B02"):
        for field in "12":
            for channel in ("DNA", "ER", "RNA"):
                shutil.copy(TIFFS / f"{sources[k % 5]}.tif", root / f"plate__{well}__{field}__{channel}.tif")
                truth[(well, field, channel)] = EXPECTED["noisy"]
                k += 1
    return truth


def test_dataset_dir_and_image_list_cell_painting(tmp_path):
    _cell_painting_tree(tmp_path)
    regex, order = ".*__([A-Z][0-9]{2})__([0-9])__([A-Za-z]+).tif", "WFC"
    dataset = ds.dispatch_dataset(tmp_path, regex=regex, capture_order=order)
    assert isinstance(dataset, ds.DatasetDir)
    positions = dataset.get_position_ids()
    # multisort's last key wins (dataset.py:186-190): field of view is the primary key, well the secondary
    assert [p["key"] for p in positions] == ["A01__1", "B02__1", "A01__2", "B02__2"]
    assert [Path(f).name for f in positions[1]["path"]] == [f"plate__B02__1__{c}.tif" for c in ("DNA", "ER", "RNA")]
    img = im.ImageList(source=positions[0]["path"], regex=regex, capture_order=order)
    data = img.get_data_lazy()
    assert data.shape == (1, 3, 1, 40, 52) and data.dtype == np.uint16
    assert img.dimorder == "TCZYX" and img.name == tmp_path.name
    assert img.meta["size_C"] == 3 and img.meta["size_Y"] == 40
    for c in range(3):
        np.testing.assert_array_equal(data[0, c, 0], EXPECTED["noisy"])
    assert img.image_id == im.calculate_checksum(positions[0]["path"])
    # wildcard source
    wild = im.ImageList(source=str(tmp_path / "plate__B02__2__*.tif"), regex=regex, capture_order=order)
    assert wild.data.shape == (1, 3, 1, 40, 52)
    # a file too many for the captured dimensions
    with pytest.raises(Exception, match="does not match the expected one"):
        im.ImageList(source=positions[0]["path"] + positions[2]["path"][:1], regex=regex, capture_order=order).dimorder_d


def test_image_list_timeseries_layout(tmp_path):
    """crop_timeseries_alcatras-like names: <pos>/<expt>_<tp 6 digits>_<channel>_<z>.tif, capture order FTCZ."""
    regex, order = ".*/([^/]+)/.+_([0-9]{6})_([A-Za-z0-9]+)_(?:.*_)?([0-9]+).tif", "FTCZ"
    truth = {}
    for pos in ("pos001", "pos002"):
        (tmp_path / pos).mkdir()
        for t in range(3):
            for ch in ("Brightfield", "GFP"):
                for z in range(2):
                    plane = (EXPECTED["noisy"] + 1000 * t + 100 * (ch == "GFP") + 10 * z).astype(np.uint16)
                    # uncompressed little-endian strips written by hand would bypass the decoders: reuse a fixture's
                    # header instead and patch the pixel block (plain_strips7 is uncompressed, data in one run)
                    raw = bytearray((TIFFS / "plain_strips7.tif").read_bytes())
                    at = raw.find(EXPECTED["noisy"].tobytes()[:64])
                    assert at > 0
                    raw[at: at + plane.nbytes] = plane.tobytes()
                    (tmp_path / pos / f"expt_{t:06d}_{ch}_{z + 1:03d}.tif").write_bytes(bytes(raw))
                    truth[(pos, t, ch, z)] = plane
    positions = ds.DatasetDir(tmp_path, regex=regex, capture_order=order).get_position_ids()
    assert [p["key"] for p in positions] == ["pos001", "pos002"]
    img = im.ImageList(source=positions[1]["path"], regex=regex, capture_order=order)
    data = img.data
    assert data.shape == (3, 2, 2, 40, 52)
    for t in range(3):
        for c, ch in enumerate(("Brightfield", "GFP")):
            for z in range(2):
                np.testing.assert_array_equal(data[t, c, z], truth[("pos002", t, ch, z)])
    np.testing.assert_array_equal(data[2], np.stack([[truth[("pos002", 2, ch, z)] for z in range(2)] for ch in ("Brightfield", "GFP")]))


def test_image_multi_tiff_and_dir(tmp_path):
    img = im.ImageMultiTiff(TIFFS / "imagej_t2z3.tif", capture_order="TZYX")
    assert img.data.shape == (2, 1, 3, 40, 52)
    assert img.meta == {"size_T": 2, "size_C": 1, "size_Z": 3, "size_Y": 40, "size_X": 52}
    np.testing.assert_array_equal(np.asarray(img.data)[:, 0], EXPECTED["stack"].reshape(2, 3, 40, 52))
    img = im.ImageMultiTiff(TIFFS / "pages6_deflate.tif", capture_order="CYX")
    assert img.data.shape == (1, 6, 1, 40, 52)
    for t in range(2):
        for c in range(3):
            shutil.copy(TIFFS / "pages6_deflate.tif", tmp_path / "skip.tif")  # not .tiff: ignored
            shutil.copy(TIFFS / ("pil_lzw.tif" if (t + c) % 2 else "plain_strips7.tif"), tmp_path / f"img_T{t:03d}_C{c:02d}.tiff")
    d = im.ImageDir(tmp_path)
    assert d.meta == {"size_T": 2, "size_C": 3}
    assert d.data.shape == (2, 3, 1, 40, 52)
    np.testing.assert_array_equal(d.data[1, 2, 0], EXPECTED["noisy"])


def _write_zarr_v2(root: Path, key: str, a: np.ndarray, chunks, compressor):
    arr = root / key
    arr.mkdir(parents=True)
    (root / ".zgroup").write_text(json.dumps({"zarr_format": 2}))
    meta = {"zarr_format": 2, "shape": list(a.shape), "chunks": list(chunks), "dtype": a.dtype.str, "order": "C",
            "fill_value": 7, "filters": None, "compressor": compressor}
    (arr / ".zarray").write_text(json.dumps(meta))
    grid = [range(-(-s // c)) for s, c in zip(a.shape, chunks)]
    import itertools

    for idx in itertools.product(*grid):
        if idx == (0,) * a.ndim and a.ndim == 3:
            continue  # a missing chunk reads as fill_value
        block = np.full(chunks, 7, a.dtype)
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, a.shape))
        block[tuple(slice(0, s.stop - s.start) for s in sl)] = a[sl]
        raw = block.tobytes()
        if compressor and compressor["id"] == "zlib":
            raw = zlib.compress(raw, 3)
        elif compressor and compressor["id"] == "gzip":
            raw = gzip.compress(raw, 3)
        (arr / ".".join(map(str, idx))).write_bytes(raw)


@pytest.mark.parametrize("compressor", [None, {"id": "zlib", "level": 3}, {"id": "gzip", "level": 3}])
def test_image_zarr_v2(tmp_path, compressor):
    a = np.random.default_rng(5).integers(0, 65535, (3, 50, 70)).astype(np.uint16)
    store = tmp_path / "plate.zarr"
    _write_zarr_v2(store, "A01_1", a, (2, 32, 32), compressor)
    positions = ds.DatasetZarr(store).get_position_ids()
    assert positions == [{"path": store, "key": "A01_1"}]
    img = im.ImageZarr(source=positions[0])
    data = img.get_data_lazy()
    assert data.shape == (1, 3, 1, 50, 70) and img.name == "/A01_1" and img.dimorder == "TCZYX"
    want = a.copy()
    want[:2, :32, :32] = 7  # the chunk that is not on disk
    np.testing.assert_array_equal(np.asarray(data)[0, :, 0], want)
    np.testing.assert_array_equal(data[0, 2, 0, 30:40, 60:], want[2, 30:40, 60:])


def test_image_zarr_v3_and_unsupported_codec(tmp_path):
    a = np.random.default_rng(6).integers(0, 65535, (2, 40, 48)).astype(np.uint16)
    arr = tmp_path / "s.zarr" / "pos"
    arr.mkdir(parents=True)
    meta = {"zarr_format": 3, "node_type": "array", "shape": list(a.shape), "data_type": "uint16",
            "chunk_grid": {"name": "regular", "configuration": {"chunk_shape": [1, 40, 32]}},
            "chunk_key_encoding": {"name": "default", "configuration": {"separator": "/"}}, "fill_value": 0,
            "codecs": [{"name": "bytes", "configuration": {"endian": "big"}}, {"name": "gzip", "configuration": {"level": 2}}]}
    (arr / "zarr.json").write_text(json.dumps(meta))
    for c in range(2):
        for x in range(2):
            block = np.zeros((1, 40, 32), ">u2")
            part = a[c: c + 1, :, 32 * x: 32 * x + 32]
            block[:, :, : part.shape[2]] = part
            path = arr / "c" / str(c) / "0" / str(x)
            path.parent.mkdir(parents=True, exist_ok=True)
            path.write_bytes(gzip.compress(block.tobytes(), 2))
    img = im.ImageZarr(source={"path": tmp_path / "s.zarr", "key": "pos"}, capture_order="CYX")
    np.testing.assert_array_equal(np.asarray(img.data)[0, :, 0], a)
    meta["codecs"][1] = {"name": "sharding_indexed", "configuration": {}}
    (arr / "zarr.json").write_text(json.dumps(meta))
    with pytest.raises(NotImplementedError, match="sharding_indexed"):
        im.ImageZarr(source={"path": tmp_path / "s.zarr", "key": "pos"}, capture_order="CYX").data


def test_inflate_entry_reports_corrupt_streams():
    import ctypes as C

    lib = _lib.load()
    src = np.frombuffer(b"this is not a zlib stream at all", np.uint8)
    out = np.zeros(64, np.uint8)
    got = C.c_size_t(0)
    with pytest.raises(Exception, match="Deflate"):
        _lib.check(lib.aliby_ingest_inflate(0, src.ctypes.data, src.size, out.ctypes.data, out.size, C.byref(got)))



# ------------------------------------------------------------------------------------- Blosc frames (zarr v2's default compressor)
BLOSC = Path(__file__).parent / "golden" / "blosc"


def _blosc_decode(frame: np.ndarray, room: int):
    import ctypes as C

    frame = np.ascontiguousarray(frame)
    out = np.zeros(max(room, 1), np.uint8)
    got = C.c_size_t(0)
    _lib.check(_lib.load().aliby_ingest_inflate(2, frame.ctypes.data if frame.size else out.ctypes.data, frame.size, out.ctypes.data, room,
                                               C.byref(got)))
    return out[: got.value]


def test_blosc_frames_decode_to_the_encoders_input():
    """Every frame c-blosc 1.21.0 (through imagecodecs, tests/golden/make_blosc_fixtures.py) wrote — blosclz / lz4 / lz4hc / zlib /
    zstd x no / byte / bit shuffle x item sizes 1, 2, 4 x one and several blocks, compressible, incompressible (memcpyed) and
    empty inputs — comes back as the bytes that went in.  The decoder (csrc/ingest.hip) shares no code with c-blosc."""
    n = 0
    with np.load(BLOSC / "frames.npz") as z:
        for k in z.files:
            if k.startswith("f_"):
                want = z["x_" + k[2:].split("__")[0]]
                got = _blosc_decode(z[k], want.size)
                assert got.size == want.size and np.array_equal(got, want), k
                n += 1
    assert n >= 150


def test_blosc_hostile_frames_are_refused_not_fatal():
    with np.load(BLOSC / "frames.npz") as z:
        frame, room = z["f_u16_noisy__lz4__s1__b4096"].copy(), z["x_u16_noisy"].size
    import struct

    def patched(off, value, fmt="<I"):
        b = bytearray(frame.tobytes())
        struct.pack_into(fmt, b, off, value)
        return np.frombuffer(bytes(b), np.uint8)

    first_block = struct.unpack_from("<I", frame.tobytes(), 16)[0]
    for bad, why in [(frame[:10], "shorter"), (patched(4, 0xFFFFFFFF), "room"), (patched(8, 0), "block size"), (patched(12, 0xFFFFFFFF), "truncated"),
                     (patched(16, 0xFFFFFFF0), "past the frame"), (patched(first_block, 0x7FFFFFFF), "past the frame"),
                     (patched(0, 9, "<B"), "version"), (patched(2, 0x41, "<B"), "codec"), (frame[: frame.size // 2], "truncated")]:
        with pytest.raises(Exception, match=why):
            _blosc_decode(bad, room)
    # a corrupted stream is either refused or decodes to something else: never more than the room it was given
    rng = np.random.default_rng(0)
    for _ in range(200):
        b = frame.copy()
        b[rng.integers(16, b.size, 4)] = rng.integers(0, 256, 4)
        try:
            assert _blosc_decode(b, room).size <= room
        except Exception:
            pass


def test_image_zarr_blosc_group_config5_shape():
    """ImageZarr (image.py:236-264) over a Blosc-compressed zarr v2 group shaped like BASELINE config 5."""
    img = im.ImageZarr(source={"path": BLOSC / "c5.zarr", "key": "0"}, capture_order="TCZYX")
    with np.load(BLOSC / "c5_expected.npz") as z:
        want = z["pixels"]
    data = img.get_data_lazy()
    assert data.shape == want.shape == (1, 2, 8, 128, 160) and img.dimorder == "TCZYX"
    np.testing.assert_array_equal(np.asarray(data), want)
    np.testing.assert_array_equal(data[0, 1, 3, 17:90, 5:61], want[0, 1, 3, 17:90, 5:61])


# ------------------------------------------------------------------------------------- hostile directory fields
def _patch_tag(data: bytes, tag: int, value: int, dtype: int | None = None) -> bytes:
    """Classic little-endian TIFF: overwrite the inline value (and optionally the type) of `tag` in the first IFD."""
    import struct

    b = bytearray(data)
    assert b[:2] == b"II" and struct.unpack_from("<H", b, 2)[0] == 42
    off = struct.unpack_from("<I", b, 4)[0]
    n = struct.unpack_from("<H", b, off)[0]
    for e in range(n):
        eo = off + 2 + 12 * e
        if struct.unpack_from("<H", b, eo)[0] == tag:
            if dtype is not None:
                struct.pack_into("<H", b, eo + 2, dtype)
            struct.pack_into("<I", b, eo + 8, value & 0xFFFFFFFF)
            return bytes(b)
    raise KeyError(tag)


_HOSTILE = [
    ("plain_strips7", 277, 0, None),            # SamplesPerPixel 0: was a null-pointer memcpy
    ("plain_strips7", 277, 0xFFFF, None),
    ("plain_strips7", 277, 0xFFFFFFFF, 4),
    ("plain_tiles32x16", 322, 0x40000000, 4),   # TileWidth 2^30: was std::bad_alloc -> abort
    ("plain_tiles32x16", 323, 0x40000000, 4),
    ("plain_tiles32x16", 322, 0, None),
    ("plain_strips7", 258, 0, None),
]


@pytest.mark.parametrize("name,tag,value,dtype", _HOSTILE)
def test_tiff_hostile_fields_are_refused_not_fatal(tmp_path, name, tag, value, dtype):
    """One-field corruptions of the directory must come back as an error code through the C ABI: run in a child process so
    that a crash (SIGSEGV / std::terminate) is a test failure rather than the end of the test session."""
    import subprocess
    import sys

    good = (TIFFS / f"{name}.tif").read_bytes()
    bad = tmp_path / "bad.tif"
    bad.write_bytes(_patch_tag(good, tag, value, dtype))
    info = im.tiff_info(TIFFS / f"{name}.tif")
    code = (
        "import sys, numpy as np, ctypes as C\n"
        f"sys.path.insert(0, {str(Path(__file__).resolve().parents[1])!r})\n"
        "from aliby_amd import _lib\n"
        "lib = _lib.load()\n"
        f"path = {str(bad)!r}.encode()\n"
        "info = np.zeros(12, np.int64); desc = C.create_string_buffer(256)\n"
        "rc0 = lib.aliby_tiff_probe(path, info.ctypes.data, desc, 256)\n"
        f"w, h, bps = {info['width']}, {info['height']}, {info['bits'] // 8}\n"
        "dst = np.zeros(w * h * bps, np.uint8)\n"
        "paths = (C.c_char_p * 1)(path); pages = np.zeros(1, np.int32)\n"
        "rc1 = lib.aliby_ingest_tiff_planes(None, paths, pages.ctypes.data, 1, w, h, bps, dst.ctypes.data, dst.size, 0, 2, None)\n"
        "print('RC', rc0, rc1)\n"
    )
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, (run.returncode, run.stderr[-800:])
    rc0, rc1 = (int(x) for x in run.stdout.split("RC")[1].split())
    assert rc0 != 0 and rc1 != 0  # both entry points refuse the page


def test_randomised_tiff_round_trips(monkeypatch, capsys):
    """tests/fuzz/fuzz_tiff.py (5400 seeds by hand): planes of random size (down to 1 x 1) and dtype, uncompressed or Deflate, random
    strip heights, written by the minimal baseline writer and read back through ImageMultiTiff."""
    import runpy
    import sys

    script = Path(__file__).resolve().parent / "fuzz" / "fuzz_tiff.py"
    monkeypatch.setattr(sys, "argv", [str(script), "0", "60"])
    monkeypatch.chdir(script.parents[2])
    runpy.run_path(str(script), run_name="__main__")
    assert "60 seeds ok" in capsys.readouterr().out

