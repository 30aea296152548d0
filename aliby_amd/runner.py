"""
Position-batched runner behind the step API: many positions per device step.

The reference's caller pattern (examples/01_cell_painting_tiff.py:108-144) is one `run_pipeline_and_post` call per position,
fanned out over loky workers; each call walks `pipeline_step` (pipe_core.py:162-251) for its own position and ends with
`get_profiles_from_state` + one parquet file (pipe_core.py:401-413).  On a GPU one 1024x1024 FOV per launch leaves most of the
256 CUs idle, so `run_positions` keeps that per-position contract — the same pipeline dicts in, the same
`profiles/<name>.parquet` and `steps/<name>/<step>/<tp:04d>.npz` out, byte for byte what N single calls write — but walks B
positions in lockstep:

    ingest thread   tile step of batch k+1: decode / upload every position's stack into one [B,C,Z,Y,X] block   (H2D stream)
    main thread     segment_* : ONE select/normalise/network/dynamics pass over all tiles of the batch            (compute stream)
                    extract_* / extractmulti_* : ONE object table + one launch per feature family for the batch
    writer pool     per position: rows -> Arrow table (columnar fast path), join, parquet (zstd); masks -> .npz    (host threads)

Every step still goes through the step objects `init_step` builds (the tiler's `get_fczyx_device`, the segmenter's `.batch`,
the extraction families), state keeps the reference's layout {"tps", "data", "fn"} per position, and `save` / `save_interval`
/ `retain` / resume-by-skip behave as in `_run_pipeline_and_post_impl`.  Positions whose pipelines differ in structure, or step
kinds without a batched form (track, anything custom), run per position inside the same loop.

Multi-GPU: `run_positions` shards positions `i % world == rank` (aliby_amd/parallel.py); no data-path
collective, the optional end-of-run gather of rows is `parallel.gather_rows`.
"""

from __future__ import annotations

import functools
import os
import threading
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import pyarrow.parquet

from aliby_amd import devcache, pipe_core, trace
from aliby_amd.extraction import extract as ex
from aliby_amd.io.write import _same_columns, dispatch_write_fn, table_layout, write_parquet_native, write_profiles


class _Position:
    def __init__(self, index, pipeline, name, output_path):
        self.index, self.pipeline, self.name = index, pipeline, name
        self.steps_dir = Path(output_path) / "steps" / name
        self.profiles_file = Path(output_path) / "profiles" / f"{name}.parquet"
        self.engine = None
        self.state = None
        self.pending = []  # futures of this position's file writes


def _signature(pipeline: dict):
    """Positions can share device steps when their pipelines have the same shape: step names in order, ntps, wiring, and what
    is saved (a saved tile step takes the host path, so positions that differ in `save` do not share a batch)."""
    return (tuple(pipeline["steps"]), pipeline.get("ntps", 1), repr(sorted(pipeline.get("passed_data", {}).items())),
            repr(sorted(pipeline.get("passed_methods", {}).items())), tuple(sorted(pipeline.get("save") or ())),
            pipeline.get("save_interval", 1))


_STEPS: list = []  # [(init_step_fn, step_name, parameters snapshot, fn)], most recently used last


def _same_parameters(a, b) -> bool:
    """a == b for step parameter dicts; values that do not compare to a plain bool (arrays, tensors) count as different."""
    try:
        return bool(a == b)
    except (ValueError, RuntimeError, TypeError):
        return False


def _shape_hint(pipeline: dict):
    """Shape of a position's frames when its first tile step is fed from an array (anything with a .shape): positions of different
    frame sizes cannot share a network / dynamics pass.  None for sources that have to be opened to know (files, zarr)."""
    for name, params in pipeline["steps"].items():
        if name.startswith("tile") and isinstance(params, dict):
            source = (params.get("image_kwargs") or {}).get("source")
            shape = getattr(source, "shape", None)
            return tuple(shape) if shape is not None else None
    return None


class _SharedSteps:
    """Step objects that hold no per-position state (segmenters: the network and its workspaces; extract partials) are built
    once per distinct parameter dict and shared by every position, instead of once per position as N single calls would —
    and kept for the next run_positions call of the process (a segmenter is ~50 ms of weight packing and workspace allocation:
    a plate processed as one call per well would pay it per well).  The last 8 are kept; `release_pinned()` drops them."""

    def __init__(self, init_step_fn):
        self.init_step_fn = init_step_fn

    def get(self, step_name, parameters, other):
        if step_name.startswith("tile") or step_name.startswith("track"):
            return self.init_step_fn(step_name, parameters, other)  # per-position state (the image, the running labels)
        for k in range(len(_STEPS) - 1, -1, -1):
            maker, name, params, fn = _STEPS[k]
            if maker is self.init_step_fn and name == step_name and _same_parameters(params, parameters):
                _STEPS.append(_STEPS.pop(k))
                return fn
        fn = self.init_step_fn(step_name, parameters, other)
        _STEPS.append((self.init_step_fn, step_name, dict(parameters), fn))
        del _STEPS[:-8]
        return fn


class _LazyRows:
    """Feature rows of one position inside a batch matrix that is still on its way to the host.  copy=True: the rows are
    taken OUT of the download buffer (a page-locked arena the runner reuses) because they outlive the batch."""

    def __init__(self, download, index, lo, hi, copy=False):
        self.download, self.index, self.lo, self.hi, self.copy = download, index, lo, hi, copy
        self._rows = None
        self._lock = threading.Lock()

    def get(self):
        with self._lock:
            if self._rows is None:
                rows = self.download.wait(spin=False)[self.index][self.lo : self.hi]  # (None : None = the whole block)
                if self.copy and rows.flags.c_contiguous and rows.nbytes >= (8 << 20):
                    # ~130 MB per batch, with every writer thread of the batch waiting for the table built on it: 4 threads
                    from aliby_amd import _lib

                    out = np.empty_like(rows)
                    _lib.check(_lib.load().aliby_host_copy(out.ctypes.data, rows.ctypes.data, rows.nbytes, 4))
                    rows = out
                elif self.copy:
                    rows = np.array(rows)
                self._rows = rows
                self.download = None
            return self._rows


class _Arena:
    """Page-locked memory of one batch in flight: bump-allocated by the batch, handed back when the batch's last position is on
    disk.  hipHostMalloc of the ~400 MB a 64-position batch downloads (labels, feature rows twice) takes tens of ms on the
    launch thread, so nothing is ever given back: a request that does not fit the chunks the arena has gets a new chunk, and
    the next batch — the same requests in the same order — walks through the same chunks.  Three arenas go round."""

    def __init__(self, ring):
        self.ring, self.left = ring, 0
        self.chunks, self.chunk, self.used = [], 0, 0  # page-locked uint8 tensors / the one being filled / bytes used of it

    def alloc(self, shape, dtype):
        import torch

        n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        while True:
            if self.chunk == len(self.chunks):
                trace.mark(f"arena: new chunk for {n} bytes")
                self.chunks.append(torch.empty(int(n * 1.25) + (1 << 20), dtype=torch.uint8, pin_memory=True))
                self.used = 0
            start = (self.used + 255) & ~255
            if start + n <= self.chunks[self.chunk].numel():
                self.used = start + n
                return self.chunks[self.chunk][start : start + n].view(dtype).view(shape)
            self.chunk, self.used = self.chunk + 1, 0

    def retire(self):
        """One position of the batch is done; the last one frees the arena."""
        with self.ring.lock:
            self.left -= 1
            if self.left > 0:
                return
            self.chunk = self.used = 0
            self.ring.free.append(self)
            self.ring.lock.notify()


class _ArenaRing:
    def __init__(self, n=3):
        self.n = n
        self.lock = threading.Condition()
        self.free = [_Arena(self) for _ in range(n)]
        self.waited = 0.0

    def acquire(self, positions):
        import time

        t0 = time.perf_counter()
        with self.lock:
            while not self.free:
                self.lock.wait()
            arena = self.free.pop(0)
        self.waited += time.perf_counter() - t0  # back-pressure: the launch thread was ahead of the writers by n batches
        arena.left = positions
        return arena


_RINGS: dict = {}


def _shared_ring():
    """The page-locked arenas outlive a run_positions call: hipHostMalloc of ~400 MB takes tens of ms on the launch thread, and
    a job of a few batches would pay it three times per call.  One ring per (process, device); `release_pinned()` frees them."""
    key = None
    try:
        import torch

        key = torch.cuda.current_device() if torch.cuda.is_available() else None
    except ImportError:
        pass
    ring = _RINGS.get(key)
    if ring is None or len(ring.free) != ring.n:  # (a ring with arenas still out belongs to a run that failed: start afresh)
        ring = _RINGS[key] = _ArenaRing(3)
    ring.waited = 0.0
    return ring


def release_pinned():
    """Give back what `run_positions` keeps between calls: the page-locked arenas and the shared step objects (segmenters with
    their device workspaces)."""
    _RINGS.clear()
    _STEPS.clear()
    from aliby_amd.segment import dispatch

    dispatch._MODELS.clear()


_Product = ex.LazyProduct
_ABLATE = os.environ.get("ALIBY_ABLATE", "")


class _Phase:
    """with-block that adds its wall time (device drained on both sides) to measure[phase]; a no-op when not measuring."""

    def __init__(self, store, phase):
        self.store, self.phase = store, phase

    def __enter__(self):
        if self.store is not None:
            import time

            _sync()
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if self.store is not None:
            import time

            _sync()
            self.store[self.phase] = self.store.get(self.phase, 0.0) + time.perf_counter() - self.t0
        return False


def _sync():
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except ImportError:
        pass


class _WriterProcess:
    """One `python -m aliby_amd.io.writer_proc` child, owned by one writer thread (see that module for the why)."""

    def __init__(self):
        import subprocess
        import sys

        root = str(Path(__file__).resolve().parents[1])
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
        self.proc = subprocess.Popen([sys.executable, "-m", "aliby_amd.io.writer_proc"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                     text=True, bufsize=1, env=env)

    def write(self, ipc, parts, out):
        import json

        self.proc.stdin.write(json.dumps({"ipc": ipc, "parts": parts, "out": str(out)}) + "\n")
        self.proc.stdin.flush()
        reply = self.proc.stdout.readline()
        if not reply:
            raise RuntimeError(f"parquet writer process died (exit code {self.proc.poll()})")
        reply = json.loads(reply)
        if not reply.get("ok"):
            raise RuntimeError(f"parquet writer process: {reply.get('error')}")

    def hang_up(self):
        try:
            self.proc.stdin.close()  # the worker's read loop ends
        except Exception:
            pass

    def close(self):
        self.hang_up()
        try:
            self.proc.wait()  # (a plain waitpid: wait(timeout=...) polls with sleeps)
        except Exception:
            self.proc.kill()


class BatchRunner:
    def __init__(self, init_step_fn, writers: int = 8, measure: bool = False, writer_processes: int = 0):
        self.shared = _SharedSteps(init_step_fn)
        self.pool = ThreadPoolExecutor(max_workers=max(1, writers), thread_name_prefix="aliby-writer")
        self.writer_processes = int(writer_processes) if os.access("/dev/shm", os.W_OK) else 0
        # one proxy thread per writer process: it hands a position to its process and sleeps on the pipe until the file is there
        self.proxies = ThreadPoolExecutor(max_workers=self.writer_processes, thread_name_prefix="aliby-proxy") if self.writer_processes else None
        self._procs, self._procs_lock, self._local = [], threading.Lock(), threading.local()
        self._ipc_files, self._ipc_seq = {}, 0
        self.thread_seconds = {}  # what the writer threads spent their time on (summed over threads)
        self.ingest = ThreadPoolExecutor(max_workers=1, thread_name_prefix="aliby-ingest")
        self.measure = {} if measure else None  # phase -> seconds, every phase synchronised (diagnostic mode)
        self._tables = {}
        self._early = {}  # extractmulti step -> (tp, results, event): launched beside its extract step
        self._multi_stream = None
        self._side_multi = os.environ.get("ALIBY_MULTI_STREAM", "1") != "0"
        self._dense = {}  # extract step -> whole-batch results of the batch in flight
        self._h2d_stream = None
        self._ring = _shared_ring()
        self._arena = None  # of the batch the launch thread is working on
        self._submits = []  # writer tasks not handed over yet (see _queue)
        self._defer = os.environ.get("ALIBY_DEFER_SUBMITS", "1") != "0"
        trace.BEFORE_BLOCK.append(self.flush_submits)

    def _tick(self, what, t0):
        import time

        dt = time.perf_counter() - t0
        with self._procs_lock:
            self.thread_seconds[what] = self.thread_seconds.get(what, 0.0) + dt

    def _timed(self, phase):
        return _Phase(self.measure, phase)

    # ------------------------------------------------------------------------------------------------ tile step
    def _tile_batch(self, batch, name, tp):
        """Tile step of every position -> list of step results; monotile positions of equal shape land in ONE device block."""
        import torch

        tilers = []
        for pos in batch:
            if name not in pos.state["fn"]:
                pos.state["fn"][name] = self.shared.get(name, pos.pipeline["steps"][name], pos.state["fn"])
            tilers.append(pos.state["fn"][name])
        # (8-bit sources: the mark that texture needs travels with the host arrays' registrations, tile/tiler.py; float sources keep
        # their own dtype there too — both stay on the host-typed path)
        on_device = all(hasattr(t, "run_tp_device") and not getattr(t, "eight_bit", False) and not getattr(t, "float_source", False)
                        for t in tilers)
        if any(name in (pos.pipeline.get("save") or []) for pos in batch) or not on_device:
            return [pipe_core.run_step(t, tp=tp) for t in tilers]  # host path: the reference's own container types
        # monotile positions of one shape: every stack is uploaded into its slice of ONE [B,C,Z,Y,X] block, so the batch the
        # segment / extract steps see is contiguous (no gather copy) and the identity window needs no crop either
        shapes = {tuple(t.shape[1:]) for t in tilers}
        if (len(tilers) > 1 and len(shapes) == 1 and all(hasattr(t, "set_upload_buffer") and not hasattr(t, "ref_channel_index")
                                                         and str(getattr(t.pixels, "dtype", "")) == "uint16" for t in tilers)):
            block = torch.empty((len(tilers), *next(iter(shapes))), dtype=torch.uint16, device="cuda")
            for i, t in enumerate(tilers):
                t.set_upload_buffer(tp, block[i])
        return [t.run_tp_device(tp) for t in tilers]

    # --------------------------------------------------------------------------------------------- segment step
    def _segment_batch(self, batch, name, tp):
        fns = []
        for pos in batch:
            if name not in pos.state["fn"]:
                pos.state["fn"][name] = self.shared.get(name, pos.pipeline["steps"][name], pos.state["fn"])
            fns.append(pos.state["fn"][name])
        blocks = []
        for pos in batch:
            kwargs = pos.engine._inputs_for(name, pos.state)
            if kwargs:
                return None  # a segmenter fed by passed_data: not the builder's wiring, run per position
            spec = pos.pipeline.get("passed_methods", {}).get(name)
            if spec is None:
                return None
            owner, method = spec
            tiler = pos.state["fn"][owner]
            if method == "get_fczyx" and hasattr(tiler, "get_fczyx_device"):
                dev, flags = tiler.get_fczyx_device(tp)
                if flags.any():
                    return None
                blocks.append(dev)
            else:
                blocks.append(getattr(tiler, method)(tp))
        if not all(f is fns[0] for f in fns) or not hasattr(fns[0], "batch"):
            return None
        if len({tuple(b.shape[1:]) for b in blocks}) != 1:
            return None  # frames of different sizes: one network / dynamics pass cannot hold them, the positions go one by one
        trace.mark("segment:blocks ready")
        return fns[0].batch(blocks, pinned_alloc=self._arena.alloc if self._arena is not None else None)

    # --------------------------------------------------------------------------------------------- extract steps
    def _extract_batch(self, batch, name, tp, multi, early=False):
        import torch

        from aliby_amd.extraction import families
        from aliby_amd.extraction.engine import FeatureEngine

        if multi and not early:
            hit = self._early.pop(name, None)  # launched beside its extract_<obj> step (below)
            if hit is not None and hit[0] == tp:
                torch.cuda.current_stream().wait_event(hit[2])
                return hit[1]
        params = [pos.pipeline["steps"][name] for pos in batch]
        if not all(p.get("tree") == params[0].get("tree") and p.get("kwargs", {}) == params[0].get("kwargs", {}) for p in params):
            return None
        for pos in batch:
            if name not in pos.state["fn"]:
                pos.state["fn"][name] = self.shared.get(name, pos.pipeline["steps"][name], pos.state["fn"])
        inputs = [pos.engine._inputs_for(name, pos.state) for pos in batch]
        if not all(set(i) == {"masks", "pixels"} for i in inputs):
            return None
        tree = params[0]["tree"]
        cp_kwargs = params[0].get("kwargs", {}).get("cp_measure_kwargs") or {}
        instructions = ex.kv(ex.flatten(tree))
        labels, pixels, tiles_of, known = [], [], [], []
        for inp in inputs:
            masks = inp["masks"] if isinstance(inp["masks"], list) else [inp["masks"]]
            lab = ex._stack_masks(masks)
            for m in masks:  # the segmenter registers each frame's object count (= its largest label) with the device copy
                hit = devcache.lookup(m) if isinstance(m, np.ndarray) else None
                known.append(hit[1].get("max_label") if hit is not None and hit[0].ndim == 2 else None)
            px, dt = ex._device_pixels(inp["pixels"])
            if lab.shape[0] != px.shape[0] and lab.shape[0] != 1:
                return None
            labels.append(lab)
            pixels.append((px[: lab.shape[0]], dt))
            tiles_of.append(lab.shape[0])
        if len({(p.shape[1:], d, p.dtype) for p, d in pixels}) != 1 or len({l.shape[1:] for l in labels}) != 1:
            return None
        eng = FeatureEngine()
        lab_all = labels[0] if len(labels) == 1 else _cat(labels)
        px_all = pixels[0][0] if len(pixels) == 1 else _cat([p for p, _ in pixels])
        # extract_<obj> and extractmulti_<obj> of one timepoint measure the same label block: one object table for both
        key = (tuple(l.data_ptr() for l in labels), tuple(lab_all.shape))
        hit = self._tables.get(key)
        if hit is None:
            mx = known if len(known) == lab_all.shape[0] and all(k is not None for k in known) else None
            hit = self._tables[key] = (eng.object_table(lab_all, max_labels=mx), labels)  # (the label tensors are kept so the key stays theirs)
        table = hit[0]
        # extractmulti_<obj> right behind extract_<obj> (the builder's order, pipe_builder.py:60-75) reads the same masks and
        # pixels: its launches (rank planes + one colocalisation launch for all pairs, ~3 ms alone on the device) go out FIRST,
        # on a stream of their own, and run beside the per-channel families instead of after them; the step itself then finds
        # its result here.  Only LDS-resident object windows (the large-object variants share the context's scratch block).
        steps = list(batch[0].pipeline["steps"])
        partner = "extractmulti_" + name[len("extract_"):] if name.startswith("extract_") else None
        if (not multi and self._side_multi and self.measure is None and partner is not None and table.n_obj > 0
                and table.max_h * table.max_w <= 4096 and steps.index(name) + 1 < len(steps) and steps[steps.index(name) + 1] == partner):
            if self._multi_stream is None:
                self._multi_stream = torch.cuda.Stream()
            side, main = self._multi_stream, torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                got = self._extract_batch(batch, partner, tp, multi=True, early=True)
                done = torch.cuda.Event()
                done.record(side)
            if got is not None:
                self._early[partner] = (tp, got, done)
        matrix, blocks = families.evaluate(eng, lab_all, table, (px_all, pixels[0][1]), instructions, cp_kwargs, multi=multi)
        out, t0, bounds, every = [], 0, [], []
        for pos, nt in zip(batch, tiles_of):
            lo, hi = int(table.offsets[t0]), int(table.offsets[t0 + nt])
            rows = table.host[lo:hi]
            objects = list(zip((rows["tile"] - t0).tolist(), rows["label"].tolist()))
            bounds.append((lo, hi))
            every.append(objects)
            t0 += nt
        # The table's column order (metric names sorted) is known from the layout alone: the device hands over the rows
        # twice — as they are (per-position results, read on demand) and column-sorted + transposed, so that every Arrow
        # column of the batch table is a window of the downloaded block and the host never transposes 100+ MB per batch.
        # (Only one-timepoint runs pivot the batch at once: a time-lapse downloads its rows once per timepoint, as they are.)
        alloc = self._arena.alloc if self._arena is not None else None
        if batch[0].pipeline.get("ntps", 1) == 1 and tp == 0:
            # (tile, label) of every row of the batch, tiles counted from each position's first: an int64 [n, 2] array, not
            # 16 k tuples — _format_dense wants exactly that array
            flat = np.empty((bounds[-1][1] if bounds else 0, 2), np.int64)
            t0 = 0
            for (lo, hi), nt in zip(bounds, tiles_of):
                flat[lo:hi, 0] = table.host["tile"][lo:hi] - t0
                flat[lo:hi, 1] = table.host["label"][lo:hi]
                t0 += nt
            whole = ex.DeviceResults(None, flat, instructions, blocks)
            _, take = ex._dense_layout(whole)
            sorted_t = matrix.index_select(1, torch.as_tensor(take, device=matrix.device)).t().contiguous() if matrix.shape[0] else matrix.t()
            download = eng.to_host_async((matrix, sorted_t), slot=None, alloc=alloc)
            whole._matrix = _LazyRows(download, 0, 0, len(flat))
            whole._transposed = _LazyRows(download, 1, None, None, copy=self._arena is not None)  # (the returned tables live on)
            self._dense[name] = dict(results=whole, bounds=bounds)  # what _profiles_for_batch needs to pivot the whole batch at once
        else:
            download = eng.to_host_async((matrix,), slot=None, alloc=alloc)
        for objects, (lo, hi) in zip(every, bounds):
            out.append((_Product(objects, instructions), ex.DeviceResults(_LazyRows(download, 0, lo, hi), objects, instructions, blocks)))
        return out

    # --------------------------------------------------------------------------------------------------- the loop
    def _save(self, pos, step_name, result, tp):
        wanted = pos.pipeline.get("save") or []
        every = pos.pipeline.get("save_interval", 1)
        if wanted and every > 0 and tp % every == 0 and step_name in wanted and _ABLATE != "files":
            if self.measure is not None:
                with self._timed("write: step outputs (.npz, zlib)"):
                    dispatch_write_fn(step_name)(result, steps_dir=pos.steps_dir, subpath=step_name, tp=tp)  # (_timed drained the device)
                return
            # (a dict result is written as it is NOW: the engine drops the tile step's "pixels" at the end of the timepoint,
            # pipe_core.py:238-242, while the writer thread may still be serialising it)
            snap = dict(result) if isinstance(result, dict) else result
            self._queue(self.pool, pos.pending, self._write_step, dispatch_write_fn(step_name), snap, pos.steps_dir, step_name, tp)

    def _queue(self, pool, sink, fn, *args):
        """pool.submit(fn, *args), later: tasks for the writer threads are handed over in bursts of a whole batch, and every
        woken thread starts with Python (paths, ctypes structures, table slices) that competes with the launch thread for the
        interpreter lock — measured as 13 ms between the segmenter's last synchronisation and the first feature kernel of a
        64-position batch.  So they wait in `_submits` until the launch thread is about to sleep in the next long device wait
        (trace.about_to_block, called by the dynamics) or the run ends.  Returns a Future that mirrors the real one."""
        from concurrent.futures import Future

        if self.measure is not None or not self._defer:
            fut = pool.submit(fn, *args)
            if sink is not None:
                sink.append(fut)
            return fut
        proxy = Future()

        def go():
            def done(real):
                exc = real.exception()
                if exc is not None:
                    proxy.set_exception(exc)
                else:
                    proxy.set_result(real.result())

            pool.submit(fn, *args).add_done_callback(done)

        self._submits.append(go)
        if sink is not None:
            sink.append(proxy)
        return proxy

    def flush_submits(self):
        self._flushed = True
        todo, self._submits = self._submits, []
        for go in todo:
            go()

    def _write_step(self, fn, result, steps_dir, step_name, tp):
        import time

        t0 = time.perf_counter()
        trace.mark("w:npz begin")
        _wait_host(result)  # the batched segmenter downloads labels asynchronously
        fn(result, steps_dir=steps_dir, subpath=step_name, tp=tp)
        trace.mark("w:npz end")
        self._tick("step outputs (.npz)", t0)

    def prepare(self, batch):
        """States of a batch and the first timepoint of its leading tile step, on the caller's thread: run_positions calls this
        for batch k+1 on the ingest thread while batch k computes; uploads go to a side stream, `ready` marks their end."""
        steps = batch[0].pipeline["steps"]
        trace.mark("prepare:begin (ingest thread)")
        for pos in batch:
            pos.engine = pipe_core.Engine(pos.pipeline, pos.steps_dir, self.shared.get)
            pos.state = pos.engine.fresh_state(steps)
        first = next(iter(steps))
        batch[0].prefetched = None
        if not first.startswith("tile"):
            return
        try:
            import torch

            cuda = torch.cuda.is_available()
        except ImportError:
            cuda = False
        if not cuda:
            batch[0].prefetched = (first, self._tile_batch(batch, first, 0), None)
            return
        if self._h2d_stream is None:
            self._h2d_stream = torch.cuda.Stream()
        with self._timed("tile: ingest + H2D"):
            with torch.cuda.stream(self._h2d_stream):
                results = self._tile_batch(batch, first, 0)
                ready = torch.cuda.Event()
                ready.record(self._h2d_stream)
        batch[0].prefetched = (first, results, ready)
        if trace.MARKS is not None:
            st = torch.cuda.memory_stats()
            trace.mark(f"prepare:end (ingest thread) device mallocs so far {st.get('num_device_alloc', 0)}, reserved {st.get('reserved_bytes.all.current', 0) >> 20} MiB")
        else:
            trace.mark("prepare:end (ingest thread)")

    def run_batch(self, batch):
        """All timepoints of a batch of positions with one signature.  Returns one future per position -> (profiles, {})."""
        import contextlib

        steps = batch[0].pipeline["steps"]
        ntps = batch[0].pipeline.get("ntps", 1)
        if batch[0].state is None:
            self.prepare(batch)
        pre, batch[0].prefetched = getattr(batch[0], "prefetched", None), None
        self._dense = {}
        arena = None
        self._flushed = False
        try:
            import torch

            if torch.cuda.is_available():
                if not self._ring.free:
                    self.flush_submits()  # (the tasks that will free an arena must be running before we wait for one)
                arena = self._ring.acquire(len(batch))  # blocks while three batches are still being written: back-pressure
        except ImportError:
            pass
        self._arena = arena
        trace.mark("run_batch:arena acquired")
        for pos in batch:
            pos.arena = arena
        for tp in range(ntps):
            self._tables, self._early = {}, {}
            for name in steps:
                results = None
                phase = ("tile: ingest + H2D" if name.startswith("tile") else "segment: project + normalise + network + dynamics + labels D2H"
                         if name.startswith("segment") else "extract: object table + feature kernels + rows D2H"
                         if name.startswith("extract") else f"step {name}")
                trace.mark(f"{name}:begin")
                with self._timed(phase) if not (tp == 0 and pre is not None and pre[0] == name) else contextlib.nullcontext():
                    results = self._run_step_batched(batch, name, tp, pre)
                trace.mark(f"{name}:returned")
                for pos, result in zip(batch, results):
                    pos.state["data"].setdefault(name, [])
                    pos.state["data"][name].append(result)
                    pos.state["tps"][name] = tp + 1
                for pos, result in zip(batch, results):
                    self._save(pos, name, result, tp)  # (queued: see _queue)
            for pos in batch:
                pos.engine._end_of_timepoint(pos.state)
        trace.mark("batch:steps done")
        dense, self._dense = self._dense, {}
        names = [n for n in steps if n.startswith("extract") or n.startswith("nahual_embed")]
        whole = None
        if ntps == 1 and names and all(n in dense for n in names):
            whole = _Once(lambda: self._profiled_pivot(batch, names, dense))  # one pivot + join for the batch, on a writer thread
        if not self._flushed:
            self.flush_submits()  # no long device wait in this batch's steps (no segmenter): nothing to hide the hand-over behind
        if self.measure is not None:
            return [_Done(self._finish(pos, whole, k)) for k, pos in enumerate(batch)]
        if whole is not None and self.proxies is not None and self.writer_processes:
            self._queue(self.pool, None, self._warm, whole)  # pivot + IPC export start on a writer thread right away
            return [self._queue(self.proxies, None, self._finish, pos, whole, k) for k, pos in enumerate(batch)]
        return [self._queue(self.pool, None, self._finish, pos, whole, k) for k, pos in enumerate(batch)]

    @staticmethod
    def _warm(whole):
        got = whole.get()
        if got is not None and got[1] is not None:
            got[1].get()

    def _profiled_pivot(self, batch, names, dense):
        if not os.environ.get("ALIBY_PROFILE_PIVOT"):
            return self._profiles_for_batch(batch, names, dense)
        import cProfile
        import pstats
        import sys

        prof = cProfile.Profile()
        out = prof.runcall(self._profiles_for_batch, batch, names, dense)
        self._pivots = getattr(self, "_pivots", 0) + 1
        if self._pivots == 6:  # (one steady-state batch is enough)
            pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(30)
        return out

    def _profiles_for_batch(self, batch, names, dense):
        """get_profiles_from_state (pipe_core.py:453-512) for every position of a one-timepoint batch at once: each extract
        step's rows are pivoted ONCE for the batch (they sit in one matrix), the step prefixes are joined ONCE, and a position's
        table is a zero-copy slice — the per-column Python work of a thousand-column table is paid per batch, not per position.
        Returns None when the batch does not have the shape this relies on (the per-position path then does the work)."""
        with self._timed("profiles: rows -> Arrow table, join"):
            by_prefix = {}
            for n in names:
                by_prefix.setdefault(n.split("_")[0], []).append(n)
            groups = list(by_prefix.values())
            if len({len(g) for g in groups}) != 1:
                return None
            wide = {n: pipe_core._wide_table(n, 0, (ex._Dense, dense[n]["results"])) for n in names}
            joined = []
            for i in range(len(groups[0])):
                members = [g[i] for g in groups]
                if any(dense[m]["bounds"] != dense[members[0]]["bounds"] for m in members) or any(wide[m] is None for m in members):
                    return None
                table = wide[members[0]]
                for m in members[1:]:
                    table = pipe_core._join_on_metadata(table, wide[m], strict=True)
                    if table is None:
                        return None
                joined.append((table, dense[members[0]]["bounds"]))
            n = len(batch)
            ipc = _Once(lambda: self._export_ipc(joined, n)) if self.writer_processes else None
            # where every column of the batch tables lives (address of row 0, item width): a position's file is then encoded
            # natively from row windows of these buffers, on a writer thread with the interpreter lock released
            layouts = None
            if not self.writer_processes and os.environ.get("ALIBY_NATIVE_WRITERS", "1") != "0":
                layouts = [table_layout(t) for t, _ in joined]
                if any(lay is None for lay in layouts) or any(not _same_columns(t.schema, joined[0][0].schema) for t, _ in joined[1:]):
                    layouts = None
            return joined, ipc, layouts

    def _export_ipc(self, joined, n_positions):
        """The batch's tables as ONE Arrow IPC stream in /dev/shm for the writer processes; -> (path, row offset of each table)."""
        import pyarrow as pa

        schema = joined[0][0].schema
        if any(not t.schema.equals(schema) for t, _ in joined[1:]):
            return None
        with self._procs_lock:
            self._ipc_seq += 1
            path = f"/dev/shm/aliby_{os.getpid()}_{id(self) & 0xffffff:x}_{self._ipc_seq}.arrow"
            self._ipc_files[path] = n_positions  # positions still to be written from this file
        offsets, row = [], 0
        with pa.OSFile(path, "wb") as sink, pa.ipc.new_stream(sink, schema) as writer:
            for t, _ in joined:
                offsets.append(row)
                writer.write_table(t)
                row += t.num_rows
        return path, offsets

    def _release_ipc(self, path):
        with self._procs_lock:
            left = self._ipc_files.get(path, 0) - 1
            if left > 0:
                self._ipc_files[path] = left
                return
            self._ipc_files.pop(path, None)
        try:
            os.unlink(path)
        except OSError:
            pass

    def _my_process(self):
        proc = getattr(self._local, "proc", None)
        if proc is None:
            proc = self._local.proc = _WriterProcess()
            with self._procs_lock:
                self._procs.append(proc)
        return proc

    def _run_step_batched(self, batch, name, tp, pre):
        results = None
        if tp == 0 and pre is not None and pre[0] == name:
            import torch

            if pre[2] is not None:
                torch.cuda.current_stream().wait_event(pre[2])  # the uploads of the ingest thread
                main = torch.cuda.current_stream()
                for r in pre[1]:
                    px = r.get("pixels") if isinstance(r, dict) else None
                    if isinstance(px, torch.Tensor):
                        px.record_stream(main)
            return pre[1]
        if name.startswith("tile"):
            results = self._tile_batch(batch, name, tp)
        elif name.startswith("segment"):
            results = self._segment_batch(batch, name, tp)
        elif name.startswith("extract_"):
            results = self._extract_batch(batch, name, tp, multi=False)
        elif name.startswith("extractmulti_"):
            results = self._extract_batch(batch, name, tp, multi=True)
        if results is None:  # no batched form (or not applicable): the engine's own per-position path for this step
            results = []
            for pos in batch:
                pos.state["data"].setdefault(name, [])
                if name not in pos.state["fn"]:
                    pos.state["fn"][name] = self.shared.get(name, pos.pipeline["steps"][name], pos.state["fn"])
                args, kwargs = pos.engine._method_args(name, pos.state, tp), pos.engine._inputs_for(name, pos.state)
                _wait_host((args, kwargs))  # a step without a batched form reads its inputs on the host, now: labels the batched
                # segmenter is still downloading must have landed (the reference hands over finished arrays, pipe_core.py:188-205)
                results.append(pipe_core.run_step(pos.state["fn"][name], *args, tp=tp, **kwargs))
        return results

    def _finish(self, pos, whole=None, k=0):
        try:
            return self._finish_position(pos, whole, k)
        finally:
            if getattr(pos, "arena", None) is not None:  # also when a write failed: the arena must go back to the ring
                pos.arena.retire()
                pos.arena = None

    def _finish_position(self, pos, whole=None, k=0):
        import time

        t0 = time.perf_counter()
        trace.mark("w:finish begin")
        got = whole.get() if whole is not None else None
        trace.mark("w:pivot there")
        self._tick("batch pivot (one thread works, the others wait)", t0)
        joined, ipc, layouts = got if got is not None else (None, None, None)
        if _ABLATE == "files":  # diagnostic: no per-position work at all (what the launch thread does when left alone)
            pos.state = pos.engine = None
            return None, {}
        written = False
        if joined is not None:
            import pyarrow as pa

            t0 = time.perf_counter()
            parts = [t.slice(b[k][0], b[k][1] - b[k][0]) for t, b in joined if b[k][1] > b[k][0]]
            profiles = (parts[0] if len(parts) == 1 else pa.concat_tables(parts)) if parts else pipe_core._empty_profiles()
            self._tick("position's table: slices of the batch table", t0)
            if layouts is not None and parts:
                t0 = time.perf_counter()
                with self._timed("write: parquet (zstd)"):
                    pos.profiles_file.parent.mkdir(parents=True, exist_ok=True)
                    write_parquet_native(pos.profiles_file, [(lay, b[k][0], b[k][1] - b[k][0]) for lay, (t, b) in zip(layouts, joined)
                                                             if b[k][1] > b[k][0]])
                self._tick("parquet, native encoder on this thread", t0)
                written = True
            t0 = time.perf_counter()
            ipc = ipc.get() if ipc is not None else None
            self._tick("IPC export to /dev/shm (one thread works, the others wait)", t0)
            if ipc is not None:
                try:
                    if parts:
                        rows = [[off + b[k][0], b[k][1] - b[k][0]] for (t, b), off in zip(joined, ipc[1]) if b[k][1] > b[k][0]]
                        t0 = time.perf_counter()
                        with self._timed("write: parquet (zstd)"):
                            self._my_process().write(ipc[0], rows, pos.profiles_file)
                        self._tick("parquet in a writer process", t0)
                        written = True
                finally:
                    self._release_ipc(ipc[0])
        else:
            with self._timed("profiles: rows -> Arrow table, join"):
                profiles = pipe_core.get_profiles_from_state(pos.state, pos.pipeline)
        if not written:
            t0 = time.perf_counter()
            with self._timed("write: parquet (zstd)"):
                pos.profiles_file.parent.mkdir(parents=True, exist_ok=True)
                write_profiles(profiles, pos.profiles_file)
            self._tick("parquet in this process", t0)
        t0 = time.perf_counter()
        for f in pos.pending:
            f.result()
        self._tick("waiting for this position's step outputs", t0)
        t0 = time.perf_counter()
        pos.state = pos.engine = None  # releases the device blocks of this position
        self._tick("releasing the position's state", t0)
        trace.mark("w:finish end")
        return profiles, {}

    def close(self):
        self.flush_submits()
        if self.flush_submits in trace.BEFORE_BLOCK:
            trace.BEFORE_BLOCK.remove(self.flush_submits)
        if self.proxies is not None:
            self.proxies.shutdown(wait=True)
        self.pool.shutdown(wait=True)
        self.ingest.shutdown(wait=True)
        for proc in self._procs:
            proc.hang_up()  # all of them first: they exit side by side
        for proc in self._procs:
            proc.close()
        for path in list(self._ipc_files):  # only after an error: every file is normally released by its last position
            try:
                os.unlink(path)
            except OSError:
                pass


class _Once:
    """A value computed by whichever thread asks first; the others wait for it."""

    def __init__(self, fn):
        self.fn, self.lock, self.done, self.value = fn, threading.Lock(), False, None

    def get(self):
        with self.lock:
            if not self.done:
                self.value, self.done, self.fn = self.fn(), True, None
            return self.value


class _Done:
    def __init__(self, value):
        self.value = value

    def result(self):
        return self.value


def _wait_host(item):
    """Block until every NumPy array inside `item` (lists / tuples / dicts, nested) holds its final bytes."""
    if isinstance(item, np.ndarray):
        devcache.wait_ready(item)
    elif isinstance(item, (list, tuple)):
        for x in item:
            _wait_host(x)
    elif isinstance(item, dict):
        for x in item.values():
            _wait_host(x)


def _cat(tensors):
    """Concatenate along axis 0; free when the pieces are consecutive views of one allocation (the batched tile step)."""
    import torch

    first = tensors[0]
    ok = len(tensors) > 1 and all(t.is_contiguous() and t.dtype == first.dtype and t.shape[1:] == first.shape[1:] for t in tensors)
    if ok:
        expect = first.data_ptr()
        for t in tensors:
            if t.data_ptr() != expect or t.untyped_storage().data_ptr() != first.untyped_storage().data_ptr():
                ok = False
                break
            expect += t.numel() * t.element_size()
    if ok:
        n = sum(t.shape[0] for t in tensors)
        return torch.as_strided(first, (n, *first.shape[1:]), first.stride())
    return torch.cat(tensors, 0)


_RUN_LOCK = pipe_core.DEVICE_LOCK  # (shared with single run_pipeline_and_post calls of other threads)


def _run_positions(pipelines, names, output_path, overwrite: bool = True, batch_size: int = 64, init_step_fn=None,
                   writers: int | None = None, shard: bool = True, measure: bool = False, stats: dict | None = None,
                   writer_processes: int | bool | None = None, switch_interval: float | None = 2e-4):
    """`run_pipeline_and_post` for many positions: pipelines[i] / names[i] -> profiles/<names[i]>.parquet (+ step outputs).

    batch_size: positions per device batch (64: 1600 network tiles of a 1024^2 plate per batch, ~0.7 GB of pixels on the device and
    three 0.4 GB page-locked arenas on the host; the bench line is measured at this size).
    Returns a list aligned with `pipelines`: (pyarrow.Table, {}) for the positions this rank processed, (None, None) for
    positions skipped by resume (`overwrite=False` and the parquet exists) or owned by another rank (`shard=True` under
    torch.distributed.run: positions i % world == rank, examples/01:100-104's round-robin).
    measure=True (diagnostic): every phase runs synchronised and inline, and the return value is {phase: ms per position}.
    switch_interval: while the call runs, the interpreter's thread switch interval is lowered to this (process-wide, restored on
    return; None leaves it alone) — see the comment at its use."""
    from aliby_amd import parallel

    if init_step_fn is None:
        from aliby_amd.pipe import init_step as init_step_fn
    if len(pipelines) != len(names):
        raise ValueError("pipelines and names must have the same length")
    if len(set(names)) != len(names):  # (two positions of one name would write the same files from two writer threads)
        dup = sorted({n for n in names if list(names).count(n) > 1})[:5]
        raise ValueError(f"position names must be unique, got {dup} more than once")
    # The cycle collector, first thing (the comment at `collect` below says why): even the loop over the positions right here
    # allocates enough to trigger a full collection over everything the caller holds — 135 us per position instead of 5.
    import gc

    manage_gc = gc.isenabled() and not measure and os.environ.get("ALIBY_MANAGE_GC", "1") != "0"
    if manage_gc:
        gc.freeze()  # (no collection first: a full one is 50-100 ms with torch and pyarrow imported, and unfreeze undoes this)
        gc.disable()
    try:
        return _run_positions_body(pipelines, names, output_path, overwrite, batch_size, init_step_fn, writers, shard, measure, stats,
                                   writer_processes, switch_interval, manage_gc)
    finally:
        if manage_gc:
            gc.enable()
            gc.unfreeze()


def _run_positions_body(pipelines, names, output_path, overwrite, batch_size, init_step_fn, writers, shard, measure, stats,
                        writer_processes, switch_interval, manage_gc):
    from aliby_amd import parallel

    t_enter = __import__("time").perf_counter()
    rank, world, _ = parallel.rank_world()
    mine = parallel.positions_for_rank(len(pipelines), rank, world) if shard else list(range(len(pipelines)))
    out = [(None, None)] * len(pipelines)
    todo = []
    for i in mine:
        pipe_core.validate_pipeline(pipelines[i])
        pos = _Position(i, pipelines[i], names[i], output_path)
        if not overwrite and pos.profiles_file.exists():
            pipe_core.logger.info(f"Skipping {names[i]}")
            continue
        todo.append(pos)
    t_validated = __import__("time").perf_counter()
    # positions that can share device steps — same signature, and the same frame shape where the source says it without being
    # read (arrays) — are batched together wherever they stand in the list; groups in order of first appearance
    groups: dict = {}
    for pos in todo:
        groups.setdefault((_signature(pos.pipeline), _shape_hint(pos.pipeline)), []).append(pos)
    batches = [members[k : k + batch_size] for members in groups.values() for k in range(0, len(members), batch_size)]
    from aliby_amd import hostinfo

    cores = hostinfo.usable_cores()
    if writer_processes is None or writer_processes is True:
        # parquet goes to processes once the job is large enough to pay for their start-up (an interpreter + a pyarrow import
        # each).  Half the share each for the parquet processes and the writer threads (.npz compression, pivot, IPC export):
        # measured on the 16-core share of a one-GPU box, (threads, processes) = (4, 12) 264, (6, 10) 274, (8, 8) 314,
        # (10, 6) 245 positions/s (scripts/api_sweep.sh)
        # (round 3: the native encoders run on the writer threads with the interpreter lock released, which beats the processes —
        # they stay available for tables the native encoder does not cover: ALIBY_WRITER_PROCS=n or writer_processes=n)
        big = writer_processes is True or (len(todo) >= 4 * batch_size and not measure)
        default_procs = max(1, cores // 2) if os.environ.get("ALIBY_NATIVE_WRITERS", "1") == "0" else 0
        writer_processes = int(os.environ.get("ALIBY_WRITER_PROCS", default_procs)) if big else 0
    if measure:
        writer_processes = 0
    if writers is None:
        writers = int(os.environ.get("ALIBY_WRITERS", max(2, cores - writer_processes) if writer_processes else max(2, cores - 4)))
    t_batched = __import__("time").perf_counter()
    runner = BatchRunner(init_step_fn, writers=writers, measure=measure, writer_processes=writer_processes)
    t_runner = __import__("time").perf_counter()
    futures = []
    measured_from = 0
    try:
        device = None
        try:
            import torch

            device = torch.cuda.current_device() if torch.cuda.is_available() else None
        except ImportError:
            pass

        def prepare(batch):
            if device is not None:
                torch.cuda.set_device(device)  # (the current device is per thread)
            runner.prepare(batch)

        import sys
        import time

        # writer threads run Python between their GIL-free stretches; with the default 5 ms switch interval the launch thread
        # can wait that long for every hand-over, which shows up as idle gaps on the device
        interval = sys.getswitchinterval()
        if os.environ.get("ALIBY_SWITCH_INTERVAL"):
            switch_interval = float(os.environ["ALIBY_SWITCH_INTERVAL"])
        if switch_interval is not None:
            sys.setswitchinterval(min(interval, switch_interval))
        nxt = None
        clock = {"wait_ingest_s": 0.0, "device_steps_s": 0.0, "drain_writers_s": 0.0, "gc_s": 0.0}
        # The cycle collector starts when allocation counts say so — in the middle of the ~25 ms per batch in which the launch
        # thread is what the device waits for — and a full collection walks everything torch / pyarrow / numpy ever imported:
        # measured as one 100 ms hole in every few batches (value_api 455 -> 477 tiles/s without it).  While the call runs the
        # objects that exist now are frozen out of the collector's sight, automatic collection is off, and the launch thread
        # collects by hand each time it is about to sleep behind a whole network forward (trace.about_to_block), or at the end
        # of a batch that had no such wait.  ALIBY_MANAGE_GC=0 leaves the collector alone.
        import gc

        collected = [False, 0]  # collected during this batch / collections so far

        def collect():
            # the young generations every batch; everything every 16th batch, after which the survivors (the tables and states the
            # call will return) are frozen too: a full collection otherwise walks all results so far — 3.7 ms per batch at 16
            # batches, 10 ms at 64, quadratic over a long run
            t0 = time.perf_counter()
            collected[1] += 1
            if collected[1] % 16:
                gc.collect(1)
            else:
                gc.collect()
                gc.freeze()
            collected[0] = True
            clock["gc_s"] += time.perf_counter() - t0

        if manage_gc:
            trace.BEFORE_BLOCK.append(collect)
        clock["before_first_batch_s"] = time.perf_counter() - t_enter
        clock["of_which_validate_s"], clock["of_which_batching_s"], clock["of_which_runner_s"] = (
            t_validated - t_enter, t_batched - t_validated, t_runner - t_batched)
        for b, batch in enumerate(batches):
            t0 = time.perf_counter()
            if measure:
                if b == 1:
                    runner.measure.clear()  # the first batch carried the one-off costs (weights packed, workspaces, arenas)
                    measured_from = sum(len(x) for x in batches[1:])
                runner.prepare(batch)
            else:
                (nxt or runner.ingest.submit(prepare, batch)).result()
                nxt = runner.ingest.submit(prepare, batches[b + 1]) if b + 1 < len(batches) else None
            t1 = time.perf_counter()
            collected[0] = False
            futures.extend(zip(batch, runner.run_batch(batch)))
            if manage_gc and not collected[0]:
                collect()
            t2 = time.perf_counter()
            clock["wait_ingest_s"] += t1 - t0
            clock["device_steps_s"] += t2 - t1
        runner.flush_submits()
        t0 = time.perf_counter()
        for pos, fut in futures:
            out[pos.index] = fut.result()
        clock["drain_writers_s"] = time.perf_counter() - t0
        if stats is not None:
            if trace.MARKS is not None:
                stats["trace"] = [(lab, round(t, 5)) for lab, t in trace.MARKS]
                trace.MARKS.clear()
            clock["of_which_waiting_for_a_free_arena_s"] = runner._ring.waited
            stats.update({k: round(v, 4) for k, v in clock.items()}, batches=len(batches), writers=writers,
                         writer_processes=int(runner.writer_processes),
                         writer_thread_seconds={k: round(v, 3) for k, v in runner.thread_seconds.items()})
    finally:
        t_close = __import__("time").perf_counter()
        runner.close()
        if stats is not None:
            stats["close_s"] = round(__import__("time").perf_counter() - t_close, 4)
        try:
            if manage_gc and collect in trace.BEFORE_BLOCK:
                trace.BEFORE_BLOCK.remove(collect)
            sys.setswitchinterval(interval)
        except NameError:
            pass
    if measure:
        n = max(measured_from if len(batches) > 1 else len(todo), 1)
        return {phase: round(1e3 * sec / n, 4) for phase, sec in runner.measure.items()}
    return out


@functools.wraps(_run_positions)
def run_positions(*args, **kwargs):
    with _RUN_LOCK:
        return _run_positions(*args, **kwargs)


run_positions.__doc__ = (_run_positions.__doc__ or "") + """

    One call at a time per process: the calls of a process share the GPU, the step objects (segmenters with their device
    workspaces), the page-locked arenas and process-wide interpreter settings; a second caller waits for the first."""
