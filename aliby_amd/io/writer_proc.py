"""
Parquet writer process: `python -m aliby_amd.io.writer_proc`, driven over stdin / stdout by aliby_amd/runner.py.

Why a process: writing a thousand-column table with pyarrow holds the interpreter lock for most of its ~20 ms, so writer
THREADS stop scaling at three (measured on the MI355X box: 21 -> 7 ms per file at 4 threads, still 7 ms at 16), while the
device produces a position every ~2 ms.  Processes scale (2 ms per file at 12).  The table travels by reference: the parent
puts one Arrow IPC stream per device batch into /dev/shm, a worker maps it (zero-copy), slices its position's rows and makes
the reference's call (pipe_core.py:412-413: parquet, zstd) through `aliby_amd.io.write.write_profiles`.

Protocol, one JSON object per line: {"ipc": path, "parts": [[row0, n], ...], "out": path} -> {"ok": true} | {"ok": false, "error": "..."}.
The worker imports pyarrow only (no torch, no GPU) and exits when stdin closes.
"""

from __future__ import annotations

import json
import sys
from pathlib import Path


def main() -> int:
    import pyarrow as pa

    from aliby_amd.io.write import write_profiles

    opened: dict = {}  # ipc path -> table (memory-mapped); the two most recent batches
    out = sys.stdout
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        try:
            task = json.loads(line)
            table = opened.get(task["ipc"])
            if table is None:
                source = pa.memory_map(task["ipc"], "r")
                table = pa.ipc.open_stream(source).read_all()
                while len(opened) >= 2:
                    opened.pop(next(iter(opened)))
                opened[task["ipc"]] = table
            parts = [table.slice(lo, n) for lo, n in task["parts"]]
            profiles = parts[0] if len(parts) == 1 else pa.concat_tables(parts)
            target = Path(task["out"])
            target.parent.mkdir(parents=True, exist_ok=True)
            write_profiles(profiles, target)
            del parts, profiles
            out.write('{"ok": true}\n')
        except Exception as exc:  # reported to the parent, which raises it in the caller's thread
            out.write(json.dumps({"ok": False, "error": f"{type(exc).__name__}: {exc}"}) + "\n")
        out.flush()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
