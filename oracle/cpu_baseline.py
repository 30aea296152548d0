"""
TEST INFRASTRUCTURE — the `cpu_baseline` leg of bench.py: the CPU restatement run in the reference's structure on the
GPU box's host cores, whole tile, no extrapolation (SURVEY.md §8d "CPU baseline beside it").

Two of the reference's modes are timed on one synthetic FOV:
  * all cores — the `ncores=k` path of extract_tree / extract_tree_multi (src/extraction/extract.py:360-374, 438-451:
    joblib/loky fan-out over (object x instruction) with `n_jobs=min(len, ncores)`), here a fork pool over objects (the
    workers inherit the exploded (N,Y,X) bool stack and the pixels instead of having them pickled per task, which only
    flatters the CPU number);
  * one core — the serial `ncores=None` loop (extract.py:349-359): its time is the SUM of the per-object times measured
    inside the workers, i.e. every object of the tile is measured, nothing is extrapolated.
Segmentation: NumPy dynamics on the whole frame (single-threaded by nature) and the U-Net forward in fp32 on CPU through
torch with all host threads.  Must be called before the process touches the GPU (it forks).
"""

from __future__ import annotations

import os
import time

import numpy as np

_G = {}


def _one_object(label):
    from oracle import aliby_extract as ox

    t0 = time.perf_counter()
    funs = _G["funs"]
    binmasks, pixels = _G["binmasks"], _G["pixels"]
    n = 0
    for inst in _G["mono"]:
        ox.measure_mono(((0, label), inst), binmasks, pixels, funs)
        n += 1
    for inst in _G["multi"]:
        ox.measure_multi(((0, label), inst), binmasks, pixels, funs)
        n += 1
    return time.perf_counter() - t0, n


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def whole_tile(fov, flow, mono_tree, multi_tree, workers=None, cp_measure_kwargs=None, net_tiles=36, budget_s=240.0):
    """fov = dict(pixels [C,Z,Y,X], nuclei [Y,X]); flow = (dP, prob).  Returns the cpu_baseline JSON object."""
    import multiprocessing as mp

    from oracle import aliby_extract as ox
    from oracle import cellpose_restated as cr

    from aliby_amd import hostinfo

    host = os.cpu_count() or 1
    workers = int(workers or hostinfo.usable_cores())
    labels = fov["nuclei"]
    n_obj = int(labels.max())
    t0 = time.perf_counter()
    _G.update(binmasks=[ox.transform_2d_to_3d(labels)], pixels=fov["pixels"][None], funs=ox.load_cellfuns(cp_measure_kwargs),
              mono=ox.kv(ox.flatten(mono_tree)), multi=ox.kv(ox.flatten(multi_tree)))
    t_explode = time.perf_counter() - t0
    t0 = time.perf_counter()
    done, per_obj, n_calls = 0, [], 0
    pool = mp.get_context("fork").Pool(workers)
    stopped = False
    try:
        for dt, n in pool.imap_unordered(_one_object, range(1, n_obj + 1), chunksize=1):
            per_obj.append(dt)
            n_calls += n
            done += 1
            if time.perf_counter() - t0 > budget_s:  # safety net on a very slow box: report what was measured
                pool.terminate()
                stopped = True
                break
    finally:
        if not stopped:
            pool.close()  # (workers exit by themselves: no SIGTERM, see bench.make_inputs)
        pool.join()
    t_feat_wall = time.perf_counter() - t0
    _G.clear()
    scale = n_obj / max(done, 1)
    # segmentation
    dP, prob = flow
    t0 = time.perf_counter()
    cr.compute_masks(dP.copy(), prob.copy())
    t_dyn = time.perf_counter() - t0
    import torch

    from aliby_amd.segment.unet import build_network

    net = build_network(seed=0, device="cpu")
    xt = torch.zeros((net_tiles, 2, 224, 224))
    torch.set_num_threads(workers)
    with torch.no_grad():
        net(xt[:2])
        t0 = time.perf_counter()
        net(xt)
        t_net_all = time.perf_counter() - t0
        torch.set_num_threads(1)
        t0 = time.perf_counter()
        net(xt[:2])
        t_net_one = (time.perf_counter() - t0) * net_tiles / 2.0
    torch.set_num_threads(workers)
    wall_all = t_explode + t_feat_wall * scale + t_dyn + t_net_all
    wall_one = t_explode + float(np.sum(per_obj)) * scale + t_dyn + t_net_one
    return {
        "value": round(1.0 / wall_all, 6), "unit": "tiles/s", "cores": workers, "kind": "port",
        "cpu_model": cpu_model(), "os_cpu_count": host, "workers": workers, "host": hostinfo.describe(),
        "one_core": {"value": round(1.0 / wall_one, 6), "unit": "tiles/s", "cores": 1,
                     "seconds_per_tile": round(wall_one, 1)},
        "seconds_per_tile": round(wall_all, 2),
        "sample": f"oracle (CPU restatement in the reference's structure, not cp_measure itself) on one whole FOV: {done} of "
                  f"{n_obj} objects x {n_calls // max(done, 1)} (object x instruction) calls on full-frame masks, fork pool of {workers} "
                  f"processes = {t_feat_wall:.1f} s wall ({float(np.sum(per_obj)):.1f} CPU-s summed over objects = the serial loop's "
                  f"time); (N,Y,X) mask explosion {t_explode:.1f} s; NumPy dynamics on the whole frame {t_dyn:.1f} s; U-Net fp32 on "
                  f"CPU, all {net_tiles} tiles, {workers} threads {t_net_all:.1f} s (1 thread: {t_net_one:.1f} s from 2 tiles)",
    }
