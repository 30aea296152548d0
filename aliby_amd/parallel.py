"""
Multi-GPU layer: one process per GPU, positions sharded round-robin, one gather of profile rows.

The reference's only parallelism is data-parallel over independent positions (loky worker per position,
examples/01_cell_painting_tiff.py:141-144; Nahual addresses round-robined at :100-104).  There is no
data-path collective: each rank runs the whole hot path on its own positions.  The single exchange is
the end-of-run gather of the per-object rows to rank 0: an all_gather of row counts, then
a padded gather of float64[rows_max, n_cols] + int64[rows_max, 4] metadata.  Backend "nccl" is RCCL over
xGMI on MI355X (direct, link-parallel gather — no ring); "gloo" covers the CPU tests.
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist


def rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def positions_for_rank(n_positions: int, rank: int, world: int) -> list[int]:
    """Round-robin i % world == rank; time-lapse positions stay whole on one rank (pipe_core.py:195-200)."""
    return [i for i in range(n_positions) if i % world == rank]


def init(backend: str | None = None):
    rank, world, local = rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device(f"cuda:{local}")
        dist.init_process_group(backend, **kw)
    return rank, world, local


def gather_rows(values: torch.Tensor, meta: torch.Tensor, dst: int = 0, group=None, always_collective: bool = False):
    """values float64 [n_i, n_cols], meta int64 [n_i, k] on every rank -> concatenated (values, meta) in
    rank order on `dst`, (None, None) elsewhere.  Single process: returns the inputs (always_collective=True runs the
    collectives on a one-rank group all the same: the test that the RCCL calls themselves work on a one-GPU machine)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not always_collective):
        return values, meta
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    out_dev = values.device
    if dist.get_backend(group) == "gloo" and values.is_cuda:  # gloo moves host memory; RCCL moves device memory directly
        values, meta = values.cpu(), meta.cpu()
    dev = values.device
    n_local = torch.tensor([values.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts) if counts else 0
    n_cols, n_meta = values.shape[1], meta.shape[1]
    pv = torch.zeros((n_max, n_cols), dtype=values.dtype, device=dev)
    pm = torch.zeros((n_max, n_meta), dtype=meta.dtype, device=dev)
    pv[: values.shape[0]] = values
    pm[: meta.shape[0]] = meta
    if rank == dst:
        bv = [torch.empty_like(pv) for _ in range(world)]
        bm = [torch.empty_like(pm) for _ in range(world)]
    else:
        bv = bm = None
    dist.gather(pv, bv, dst=dst, group=group)
    dist.gather(pm, bm, dst=dst, group=group)
    if rank != dst:
        return None, None
    return (torch.cat([b[:c] for b, c in zip(bv, counts)], 0).to(out_dev), torch.cat([b[:c] for b, c in zip(bm, counts)], 0).to(out_dev))


def barrier():
    if dist.is_initialized():
        dist.barrier()


def run_positions(pipelines, names, output_path, **kwargs):
    """Many positions behind the step API, B per device step, sharded i % world == rank: see aliby_amd/runner.py."""
    from aliby_amd.runner import run_positions as _run

    return _run(pipelines, names, output_path, **kwargs)
