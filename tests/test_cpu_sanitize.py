"""
Sanitizer run of the host-side TIFF decoder (CPU build only: the GPU pool runs no sanitizers, so this file is listed in
.gpurunignore and never travels to the GPU box).
"""

from pathlib import Path

TIFFS = Path(__file__).parent / "golden" / "tiff"

def test_tiff_decoder_survives_corrupted_files_under_asan(tmp_path):
    """The host-side decoder built with AddressSanitizer + UBSan (CPU build: GPU sanitizers are not available on the pool)
    over every fixture and 60 corrupted copies of each (truncations, byte flips in the header / anywhere): it must decode
    or refuse, never touch memory it does not own."""
    import os
    import subprocess

    root = Path(__file__).resolve().parents[1]
    exe = tmp_path / "tiff_fuzz"
    src = [root / "aliby_amd" / "csrc" / "ingest.hip", root / "aliby_amd" / "csrc" / "ctx.hip", root / "tests" / "tools" / "tiff_fuzz_harness.cpp"]
    build = subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                            "--offload-arch=gfx950", "-I", str(root / "include"), "-I", str(root / "aliby_amd" / "csrc"), *map(str, src),
                            "-o", str(exe), "-lz", "-ldl", "-lpthread"], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    fixtures = sorted(str(p) for p in TIFFS.glob("*.tif"))
    run = subprocess.run([str(exe), str(tmp_path / "case.tif"), "60", *fixtures], capture_output=True, text=True, timeout=600,
                         env={**os.environ, "ASAN_OPTIONS": "detect_leaks=0"})
    assert run.returncode == 0 and "no memory error" in run.stdout, (run.stdout[-500:], run.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr


def test_blosc_decoder_survives_corrupted_frames_under_asan(tmp_path):
    """The Blosc-1 frame decoder (zarr v2's default compressor; csrc/ingest.hip) under AddressSanitizer + UBSan: every fixture
    frame, 14 directed header patches and 120 random corruptions of each, decoded into a buffer of exactly the declared size."""
    import os
    import subprocess

    import numpy as np

    root = Path(__file__).resolve().parents[1]
    exe = tmp_path / "blosc_fuzz"
    src = [root / "aliby_amd" / "csrc" / "ingest.hip", root / "aliby_amd" / "csrc" / "ctx.hip", root / "tests" / "tools" / "blosc_fuzz_harness.cpp"]
    build = subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                            "--offload-arch=gfx950", "-I", str(root / "include"), "-I", str(root / "aliby_amd" / "csrc"), *map(str, src),
                            "-o", str(exe), "-lz", "-ldl", "-lpthread"], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    frames = []
    with np.load(root / "tests" / "golden" / "blosc" / "frames.npz") as z:
        for k in z.files:
            if k.startswith("f_") and "empty" not in k:
                path = tmp_path / (k + ".blosc")
                path.write_bytes(z[k].tobytes())
                frames.append(str(path))
    run = subprocess.run([str(exe), "120", *frames], capture_output=True, text=True, timeout=900,
                         env={**os.environ, "ASAN_OPTIONS": "detect_leaks=0"})
    assert run.returncode == 0 and "no memory error" in run.stdout, (run.stdout[-500:], run.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr
