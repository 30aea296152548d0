"""
bench.py — whole-job throughput of the segmentation + cp_measure hot path on synthetic TCZYX stacks.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched under
torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.

Workload = config 2 of BASELINE.json (the configuration the metric is quoted on): 1024x1024, 5-channel
Cell Painting synthetic FOVs (~250 nuclei each), Cellpose nuclei segmentation on the DNA channel + the
full default cp_measure feature list + sizeshape + 10 channel pairs x 4 colocalisation metrics.
`--config {1,2,4,5}` selects another BASELINE configuration (2 is the default and the one the metric is quoted on).
A "step" is one pass of the hot path over one batch of B FOVs per GPU, inputs resident in HBM:

    stage (crop/pad)                                                   HIP
    segment: Z-project -> normalize99 -> 224-px tiles                  HIP
             -> residual U-Net forward (bf16, every conv hand-written) HIP (MFMA)
             -> taper blend                                            HIP
             -> dynamics (flow following, seeds, labels, QC, fill)     HIP
    extract: object table -> every feature family -> D2H of the rows   HIP

Cellpose's pretrained weights cannot be fetched offline: the network runs with
fixed-seed random weights (its full cost is paid inside the timed region, its output is discarded) and
the dynamics are fed analytic network-scale flows derived from the synthetic ground truth, so masks and
feature workloads are realistic and checkable.  FOVs are independent: ranks shard them with no
data-path collective ("scaling": "weak"; `--total-fovs N` fixes the job instead: N FOVs split over the ranks,
"scaling": "strong"); the only exchange is the final gather of profile rows.

Besides `value` (the HBM-resident rate the contract asks for) the line carries `value_api`: the same workload THROUGH the
step API — pinned host arrays -> build_pipeline_steps() dicts -> aliby_amd.parallel.run_positions (Tiler -> segment -> extract ->
get_profiles_from_state -> parquet + mask .npz on disk), B positions per device step — and `api_split_ms_per_fov`, where that
time goes (H2D, kernels, table assembly, file writes), from a synchronised pass of the same runner.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def build_trees(channels, seg_channel=0, features=None):
    """The builder's trees (pipe_builder.py:108-129) for one object set."""
    from aliby_amd.extraction import families
    from aliby_amd.extraction.engine import FeatureEngine
    from aliby_amd.pipe_builder import build_pipeline_steps

    families.register_optional(FeatureEngine)
    kw = {} if features is None else {"features_to_extract": tuple(features)}
    pipe = build_pipeline_steps(channels_to_segment={"nuclei": seg_channel}, channels_to_extract=list(channels), **kw)
    mono = pipe["steps"]["extract_nuclei"]["tree"]
    multi = pipe["steps"].get("extractmulti_nuclei", {}).get("tree", {})
    missing = sorted({m for v in mono.values() for vv in v.values() for m in vv if m not in families.MONO}
                     | {m for v in multi.values() for vv in v.values() for vvv in vv.values() for m in vvv
                        if m not in families.MULTI})
    assert not missing, f"families not built: {missing}"
    return mono, multi


BSIZE = 224  # network tile size of this run (--bsize)


def alg_bytes(kernel, B, C, Z, Y, X, n_obj, n_tiles_net):
    """Algorithmic bytes of one launch of a kernel group (every input byte touched once per
    stage, outputs once)."""
    P = Y * X
    table = {
        "stage_crop_pad": 2 * B * C * Z * P * 2,
        "reduce_z": B * C * (Z + 1) * P * 2,
        "select_project": B * Z * P * 2 + B * P * 2,
        "normalize99": B * P * 2 * 2 + B * P * 4,
        "make_tiles": B * P * 4 + n_tiles_net * 2 * BSIZE * BSIZE * 4,
        "average_tiles": n_tiles_net * 3 * BSIZE * BSIZE * 4 + B * 3 * P * 4,
        "dynamics": B * (3 * P * 4 + P * 2),
        "object_table": B * P * 2 + n_obj * 32,  # (one pass: the largest labels come from the segmenter's counts)
        "intensity": B * P * 2 * 2 + n_obj * 21 * 8,
        "sizeshape": B * P * 2 + n_obj * 78 * 8,
        "feret": B * P * 2 + n_obj * 2 * 8,
        "mec": B * P * 2 + n_obj * 32,
        "zernike": B * P * 2 + n_obj * 30 * 8,
        # every extracted channel in one launch (aliby_features_radial_zernikes_multi): labels once + C pixel planes
        "radial_zernikes": B * P * 2 * (1 + C) + n_obj * 60 * 8 * C if C >= 2 else B * P * 2 * 2 + n_obj * 60 * 8,
        "texture": B * P * 2 * 2 + n_obj * 52 * 8,
        "radial_geometry": B * P * 2 + B * P,
        "radial_distribution": B * P * (2 + 1 + 2) + n_obj * 12 * 8,
        "coloc": B * P * 2 * 3 + n_obj * 8 * 8,
        "ranks": B * P * (2 + 2 + 4),
        "track_stitch": 2 * B * P * 2,
    }
    return float(table.get(kernel, B * P * 2 * 2))


# BASELINE.json configs -> what one bench step is.  `tile` = network tile; `per_step` = FOV stacks per step.
CONFIGS = {
    1: dict(C=2, Z=1, size=512, seg_channel=1, n_target=60, features=("intensity",), channels=(0,), label="C1"),
    2: dict(C=5, Z=1, size=1024, seg_channel=0, n_target=250, features=None, channels=None, label="C2"),
    4: dict(C=1, Z=5, size=512, seg_channel=0, n_target=None, features=(), channels=(), label="C4"),
    5: dict(C=2, Z=32, size=512, seg_channel=0, n_target=100, features=("intensity",), channels=None, label="C5"),
}


def _make_one(job):
    from aliby_amd import synth

    config, fov, size, C, Z, n_target = job
    f = synth.make_fov(config, fov, shape=(size, size), n_channels=C, n_z=Z, n_target=n_target)
    dP, prob = synth.analytic_flows(f["nuclei"])
    return dict(pixels=f["pixels"], nuclei=f["nuclei"], dP=dP, prob=prob)


def under_profiler() -> bool:
    """rocprofv3 preloads its tool library into the program it starts, and (with --pmc) that library initialises the GPU before
    main() runs: such a process must not fork (gpurun_out/prof_r02h.log: a fork-pool child carrying the tool's signal handlers
    was SIGTERM'd by Pool.terminate() and took the profiler down with it)."""
    return any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_PATH"))


def make_inputs(config, cfg, fov_ids, size, procs, cache=None):
    """Distinct synthetic FOVs, generated by a pool of host processes BEFORE this process touches the GPU.  `cache`: an .npz
    the set is loaded from when it exists and written to when it does not (scripts/*.sh generate it in an unprofiled command
    and hand it to the profiled one, which then neither forks nor spends minutes generating on one core)."""

    jobs = [(config, i, size, cfg["C"], cfg["Z"], cfg["n_target"]) for i in fov_ids]
    keys = ("pixels", "nuclei", "dP", "prob")
    if cache and os.path.exists(cache):
        with np.load(cache) as z:
            if list(z["jobs"].tolist()) == [list(map(lambda v: -1 if v is None else v, j)) for j in jobs]:
                return [{k: z[f"{k}_{i}"] for k in keys} for i in range(len(jobs))]
    out = _make_inputs(jobs, 1 if under_profiler() else procs)
    if cache:
        np.savez(cache, jobs=np.array([[-1 if v is None else v for v in j] for j in jobs]),
                 **{f"{k}_{i}": o[k] for i, o in enumerate(out) for k in keys})
    return out


def _make_inputs(jobs, procs):
    import multiprocessing as mp

    if procs <= 1 or len(jobs) <= 1:
        return [_make_one(j) for j in jobs]
    # close + join, not the context manager's terminate() (see under_profiler)
    pool = mp.get_context("fork").Pool(min(procs, len(jobs)))
    try:
        return pool.map(_make_one, jobs, chunksize=1)
    finally:
        pool.close()
        pool.join()


def launch_ranks(n, argv):
    """`python bench.py --gpus N` from a plain environment (no RANK): start N rank processes — one per GPU, this same file under
    torch.distributed.run, rendezvous on 127.0.0.1 — wait for them and relay rank 0's JSON line.  This parent never touches the
    GPU (no torch.cuda, no libaliby_hip.so): the ranks are fresh children, nothing that has initialised HIP is exec'ed or forked.
    Each rank gets an equal share of the host cores this process may use (ALIBY_HOST_CORES)."""
    import socket
    import subprocess

    from aliby_amd import hostinfo

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("ALIBY_HOST_CORES", str(max(1, hostinfo.usable_cores(default_cap=16 * n) // n)))
    env.setdefault("OMP_NUM_THREADS", env["ALIBY_HOST_CORES"])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:  # rank 0 prints the one JSON line; anything else a rank writes to stdout is passed through to stderr
        if out.startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc != 0 or line is None:
        sys.stderr.write(f"bench.py: the {n}-rank job failed (exit code {rc}, JSON line {'missing' if line is None else 'present'})\n")
        sys.exit(rc or 1)
    print(line)
    sys.exit(0)


def rehearse(args):
    """`--rehearse`: the N>1 control flow without any device work — rendezvous, barrier, max-over-ranks timing, the final gather
    of stand-in rows — so that the launcher and the collectives can be tested on a machine without a GPU (gloo).  The line it
    prints has `value: null`: it is not a measurement."""
    import torch

    from aliby_amd import parallel

    rank, world, _ = parallel.rank_world()
    backend = os.environ.get("ALIBY_DIST_BACKEND", "nccl")
    parallel.init(backend if world > 1 else None)
    dist = torch.distributed if world > 1 else None
    parallel.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    parallel.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n = 3 + rank
    vals = torch.full((n, 4), float(rank), dtype=torch.float64)
    meta = torch.tensor([[i * world + rank, i, 0, 0] for i in range(n)], dtype=torch.int64)
    gv, _ = parallel.gather_rows(vals, meta)
    if rank == 0:
        print(json.dumps({"metric": "FOV tiles/sec (whole node)", "value": None, "unit": "tiles/s", "n_gpus": world, "ranks": world,
                          "backend": (dist.get_backend() if dist is not None else None), "steps": 0, "warmup": 0,
                          "ms_per_step": round(1e3 * float(t.item()), 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "data": "none (--rehearse: launcher + collectives only, no device work)",
                          "config": {"workload": "rehearsal", "gathered_rows": int(gv.shape[0]),
                                     "host_cores_per_rank": int(os.environ.get("ALIBY_HOST_CORES", 0))}}))
    if dist is not None:
        dist.destroy_process_group()


def build_rooflines(prof, model, args, B, C, Z, Y, X, n_obj, n_tiles_net, steps):
    """The roofline objects of a bench line from the per-group HIP-event timings (`prof`): the dominant hand-written group against
    HBM peak (algorithmic bytes per launch / average launch time), the deep K-loop conv group against the dense MFMA peak, every
    group's HBM fraction, and the whole network's rate."""
    class _T:  # (the code below was written against `table.n_obj`)
        pass

    table = _T()
    table.n_obj = n_obj
    roof = roof_deep = None
    fracs = {}
    if prof:  # (--no-kernel-timing: nothing was bracketed)
        # ---- roofline of the dominant hand-written kernel group ----------------------------------------------
        hip_groups = {k: v for k, v in prof.items() if k not in ("unet_forward", "rows_d2h")}
        dominant = max(hip_groups, key=lambda k: hip_groups[k]["ms_total"])
        launches = hip_groups[dominant]["launches"]
        avg_ms = hip_groups[dominant]["ms_total"] / launches
        ab = alg_bytes(dominant, B, C, Z, Y, X, table.n_obj, n_tiles_net)
        timed_launches = hip_groups[dominant].get("timed_launches", launches)
        if dominant == "fused_pointwise":  # exact: operands read once + results written once, summed over the timed launches
            ab = model.fused.bytes_moved / timed_launches
        if dominant.startswith("conv3x3_mfma"):  # exact: input + residual read once, output (+ pooled) written once, per timed launch
            ab = model.fused.conv_stats[dominant][0] / timed_launches
        achieved = ab / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
        # FETCH_SIZE doubled on gfx950): measured offline, committed under profiles/, see profiles/pmc_traffic.json
        traffic = None
        pmc_file = ROOT / "profiles" / "pmc_traffic.json"
        if pmc_file.exists():
            traffic = json.loads(pmc_file.read_text()).get(dominant, {}).get("hbm_bytes_per_launch")
        roof = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "alg_bytes_per_launch": ab,
                "avg_launch_ms": round(avg_ms, 4), "launches": launches}
        if dominant.startswith("conv3x3_mfma"):  # HBM-bound by design (144-288 FLOP/B); the matrix-core rate it sustains meanwhile
            roof["mfma_tflops"] = round(model.fused.conv_stats[dominant][1] / timed_launches / (avg_ms * 1e-3) / 1e12, 1)
        roof["timed_launches"] = timed_launches
        # the deep levels' launches of the same kernel (128+ output channels): bound by the matrix cores
        if "conv3x3_mfma_deep" in prof and prof["conv3x3_mfma_deep"].get("timed_launches"):
            g = prof["conv3x3_mfma_deep"]
            st = model.fused.conv_stats["conv3x3_mfma_deep"]
            t_ms = g["ms_total"] / g["launches"]
            tf = st[1] / g["timed_launches"] / (t_ms * 1e-3) / 1e12
            roof_deep = {"bound": "mfma", "kernel": "conv3x3_mfma_deep", "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(tf / 2500.0, 4), "avg_launch_ms": round(t_ms, 4), "launches": g["launches"],
                         "timed_launches": g["timed_launches"],
                         "alg_gbps": round(st[0] / g["timed_launches"] / (t_ms * 1e-3) / 1e9, 1)}
        # every HBM-bound group's fraction of the 8 TB/s peak: algorithmic bytes per launch / average launch time
        for name, g in hip_groups.items():
            if name.startswith("conv") or name in ("first_conv", "style", "out_head", "fused_pointwise"):
                st = model.fused.conv_stats.get(name) if model.fused is not None else None
                if not st or not g.get("timed_launches"):
                    continue
                gb = st[0] / g["timed_launches"]
            else:
                gb = alg_bytes(name, B, C, Z, Y, X, table.n_obj, n_tiles_net)
            ms = g["ms_total"] / max(g["launches"], 1)
            if ms > 0:
                fracs[name] = round(gb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    net_ms = prof.get("unet_forward", {}).get("ms_total", 0.0) / max(steps, 1)
    hip_in_net_ms = sum(prof.get(k, {}).get("ms_total", 0.0) for k in ("fused_pointwise", "conv3x3_mfma", "conv3x3_mfma_deep", "conv3x3_mfma_head", "conv3x3_mfma_pair", "conv3x3_mfma_first_pair", "maxpool", "out_head",
                                                                       "first_conv", "conv1x1_mfma", "style")) / max(steps, 1)
    net_flops = model.net.flops_per_pixel() * n_tiles_net * BSIZE * BSIZE
    mfma = {"unet_ms_per_step": round(net_ms, 3), "of_which_hand_written_hip_ms": round(hip_in_net_ms, 3), "unet_tflops": round(net_flops / (net_ms * 1e-3) / 1e12, 2) if net_ms else None,
            "dtype": args.net_dtype, "peak_tflops_dense": 2500.0 if args.net_dtype != "float32" else 157.3,
            "path": "hand-written HIP (libaliby_hip.so)" if model.fused is not None else "torch module (MIOpen) - A/B reference, not the product path"}

    return roof, roof_deep, fracs, mfma


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks (one per GPU).  Under torch.distributed.run (RANK set) it must equal "
                    "WORLD_SIZE; from a plain environment with N > 1 this process starts the N ranks itself and relays rank 0's line")
    ap.add_argument("--rehearse", action="store_true", help="launcher + collectives only, no device work (CPU test of the N>1 path)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configuration (1-based)")
    ap.add_argument("--fovs", type=int, default=64, help="FOVs per step per GPU")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic FOVs generated per rank (replicated to --fovs)")
    ap.add_argument("--total-fovs", type=int, default=0, help="strong-scaling mode: this many FOVs in all, split over the ranks "
                    "(steps = ceil(total / (gpus * fovs)); --steps is then ignored)")
    ap.add_argument("--size", type=int, default=0, help="override the configuration's frame size")
    ap.add_argument("--net-dtype", default="bfloat16", choices=["bfloat16", "float32", "float16"],
                    help="bfloat16 (default, the product path): the hand-written MFMA network of libaliby_hip.so.  float32 / float16: the "
                    "plain torch module through MIOpen — an A/B reference for the network's numerics, NOT the product path; the line "
                    "then says so in mfma.path and carries no conv roofline")
    ap.add_argument("--net-batch", type=int, default=288, help="224x224 tiles per U-Net forward (the reference's batch_size knob)")
    ap.add_argument("--bsize", type=int, default=224, help="network tile size: 224 = cellpose 3's (the residual U-Net's own family, "
                    "the default and the headline); 256 = what cellpose 4's eval defaults to — 25 instead of 36 tiles per 1024^2 FOV")
    ap.add_argument("--time-every", type=int, default=7, help="bracket every n-th launch of the per-layer network kernels with HIP "
                    "events (they are launched ~600 times per step; 1 = every launch)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="informational: no per-kernel HIP events, so the network forward "
                    "runs as a replayed hipGraph (the product path); the line then carries no roofline objects")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the through-the-step-API leg (value_api)")
    ap.add_argument("--api-fovs", type=int, default=0, help="positions of the API leg (default: 16 batches of --fovs ~ a 384-well plate at 2-3 fields per well)")
    ap.add_argument("--overlap", action="store_true", help="experiment: dynamics + features of step k on a second stream while the "
                    "network of step k+1 runs (software pipelining across steps)")
    ap.add_argument("--inputs", default="", help="path prefix of an input cache (<prefix>.r<rank>of<world>.npz): loaded when present, "
                    "written when not; profiled runs load what an earlier, unprofiled command generated")
    ap.add_argument("--inputs-only", action="store_true", help="generate (and cache) the synthetic inputs, then exit: no GPU work")
    ap.add_argument("--true-3d", action="store_true", help="with --config 5: the stack as a VOLUME (an extension beyond what the reference "
                    "wires, SURVEY.md 8(d).5): every Z plane segmented, planes stitched along Z (IoU >= 0.01), 3-D intensity features; "
                    "a tile is one [C,Z,Y,X] stack.  Without it config 5 is the reference-faithful projected form")
    ap.add_argument("--host-procs", type=int, default=0, help="host processes for input generation / the CPU baseline (default: all)")
    args = ap.parse_args()
    global BSIZE
    BSIZE = args.bsize
    if "RANK" not in os.environ and args.gpus > 1:
        launch_ranks(args.gpus, sys.argv[1:])
    if int(os.environ.get("WORLD_SIZE", 1)) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', 1)}: start one rank per GPU "
                 f"(torch.distributed.run --nproc-per-node {args.gpus}), or run `python bench.py --gpus {args.gpus}` from a plain "
                 "environment and let it start the ranks")
    if args.rehearse:
        return rehearse(args)
    if args.config == 4:
        return main_timelapse(args)
    if args.true_3d:
        if args.config != 5:
            sys.exit("bench.py: --true-3d goes with --config 5")
        return main_volume(args)
    cfg = CONFIGS[args.config]
    size = args.size or cfg["size"]

    from aliby_amd import parallel

    rank, world, local_rank = parallel.rank_world()
    from aliby_amd import hostinfo

    procs = args.host_procs or hostinfo.usable_cores()  # this rank's share of the host, not os.cpu_count()
    # ---- everything that forks happens before the GPU is touched: synthetic inputs, then the CPU baseline ---------------
    distinct = max(1, min(args.distinct, args.fovs))
    base = make_inputs(args.config, cfg, [rank + world * i for i in range(distinct)], size, procs,
                       cache=(args.inputs and f"{args.inputs}.r{rank}of{world}.npz"))
    if args.inputs_only:
        return
    channels = list(range(cfg["C"])) if cfg["channels"] is None else list(cfg["channels"])
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.cpu_baseline import whole_tile  # (checker code: only this leg of the bench uses oracle/)
        from aliby_amd.pipe_builder import build_pipeline_steps

        kw = {} if cfg["features"] is None else {"features_to_extract": tuple(cfg["features"])}
        pipe = build_pipeline_steps(channels_to_segment={"nuclei": cfg["seg_channel"]}, channels_to_extract=channels, **kw)
        cpu = whole_tile(dict(pixels=base[0]["pixels"], nuclei=base[0]["nuclei"]), (base[0]["dP"], base[0]["prob"]),
                         pipe["steps"]["extract_nuclei"]["tree"], pipe["steps"].get("extractmulti_nuclei", {}).get("tree", {}),
                         workers=procs, net_tiles=int(np.ceil(size / BSIZE * 1.2)) ** 2)

    import torch

    from aliby_amd import _lib, synth  # noqa: F401

    # ALIBY_DIST_BACKEND=gloo: rehearsal of the N>1 path on a box with fewer GPUs than ranks (ranks share devices,
    # collectives go through host memory); the driver's runs use the default, RCCL ("nccl") with one GPU per rank
    backend = os.environ.get("ALIBY_DIST_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    parallel.init(backend if world > 1 else None)
    dist = torch.distributed if world > 1 else None

    from aliby_amd.extraction.batch import extract_batch
    from aliby_amd.extraction.engine import FeatureEngine, _ptr, _stream_ptr
    from aliby_amd.segment.cellpose_hip import CellposeModel

    C, Z, Y, X = cfg["C"], cfg["Z"], size, size
    B = args.fovs
    steps = args.steps
    scaling = "weak"
    if args.total_fovs:
        steps = max(1, -(-args.total_fovs // (world * B)))
        scaling = "strong"
    mono_tree, multi_tree = build_trees(channels, cfg["seg_channel"], cfg["features"])

    # ---- synthetic inputs, resident in HBM (positions rank, rank+world, ... of the FOV stream) ----------
    stacks = torch.empty((B, C, Z, Y, X), dtype=torch.uint16, device="cuda")
    dP_true = torch.empty((B, 2, Y, X), dtype=torch.float32, device="cuda")
    prob_true = torch.empty((B, Y, X), dtype=torch.float32, device="cuda")
    for b in range(B):
        k = b % distinct
        stacks[b] = torch.from_numpy(base[k]["pixels"]).cuda()
        dP_true[b] = torch.from_numpy(base[k]["dP"]).cuda()
        prob_true[b] = torch.from_numpy(base[k]["prob"]).cuda()
    n_obj_per_fov = float(np.mean([int(s["nuclei"].max()) for s in base]))
    tiles = torch.empty_like(stacks)
    rect = np.array([[0, 0, Y, X]], np.int32)
    flags = np.zeros(1, np.int32)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = CellposeModel(net_dtype=args.net_dtype, seed=0, flows_override=lambda x: (dP_true, prob_true),
                              run_network_with_override=True, batch_size=args.net_batch, bsize=args.bsize)
    eng = model.eng
    torch.cuda.synchronize()
    n_tiles_net = B * model._geometry(Y, X)["ny"] * model._geometry(Y, X)["nx"]

    def step():
        with eng.timed("stage_crop_pad"):
            _lib.check(eng.lib.aliby_crop_pad_u16(eng.ctx.handle, _ptr(stacks), B * C, Z, Y, X, _ptr(rect), 1, Y, X,
                                                  _ptr(tiles), _ptr(flags), _stream_ptr()))
        px = tiles.view(B, C, Z, Y, X)
        plane = model.select_and_project(px, cfg["seg_channel"])
        masks, _, _ = model.eval(plane, do_3D=False, stitch_threshold=0.0, normalize=True, z_axis=None)
        labels = masks if masks.ndim == 3 else masks[None]
        planes = (px, _lib.U16)
        # (the segmenter's counts are its frames' largest labels: the object table skips its own pass, as through the API)
        table = eng.object_table(labels, max_labels=model.last_counts)
        # the colocalisation tree (rank planes + one launch for all pairs) on a stream of its own beside the per-channel families
        m2 = None
        if multi_tree and step.side is not None:
            main = torch.cuda.current_stream()
            step.side.wait_stream(main)
            with torch.cuda.stream(step.side):
                m2, names2, _ = extract_batch(eng, labels, planes, multi_tree, multi=True, table=table)
        m1, names1, _ = extract_batch(eng, labels, planes, mono_tree, table=table)
        out = [m1]
        if multi_tree:
            if m2 is None:
                m2, names2, _ = extract_batch(eng, labels, planes, multi_tree, multi=True, table=table)
            else:
                main.wait_stream(step.side)
                m2.record_stream(main)
            out.append(m2)
        # rows -> pinned host memory on a side stream: the download of step k overlaps the start of step k+1
        pending = eng.to_host_async(tuple(out), slot=step.parity)
        step.parity ^= 1
        return pending, table, model.last_counts

    step.parity = 0
    step.side = torch.cuda.Stream() if os.environ.get("ALIBY_MULTI_STREAM", "1") != "0" else None

    # ---- experiment (--overlap): the step cut in two phases on two streams ------------------------------------------
    post_stream = torch.cuda.Stream() if args.overlap else None
    tiles2 = [tiles, torch.empty_like(tiles)] if args.overlap else None

    def phase_net(buf):
        tb = tiles2[buf]
        with eng.timed("stage_crop_pad"):
            _lib.check(eng.lib.aliby_crop_pad_u16(eng.ctx.handle, _ptr(stacks), B * C, Z, Y, X, _ptr(rect), 1, Y, X,
                                                  _ptr(tb), _ptr(flags), _stream_ptr()))
        px = tb.view(B, C, Z, Y, X)
        plane = model.select_and_project(px, cfg["seg_channel"])
        model.run_network(plane)  # output discarded, as in step(): the cost is what is measured
        ev = torch.cuda.Event()
        ev.record()
        return px, plane, ev

    def phase_post(state):
        from aliby_amd.segment import dynamics

        px, plane, ev = state
        with torch.cuda.stream(post_stream):
            post_stream.wait_event(ev)
            labels, counts = dynamics.masks_from_flows(eng, dP_true, prob_true, niter=200, cellprob_threshold=0.0, flow_threshold=0.4,
                                                       min_size=15, max_size_fraction=0.4)
            planes = (px, _lib.U16)
            m1, names1, table = extract_batch(eng, labels, planes, mono_tree, table=eng.object_table(labels, max_labels=counts))
            m2, names2, _ = extract_batch(eng, labels, planes, multi_tree, multi=True, table=table)
            pending = eng.to_host_async((m1, m2), slot=step.parity)
            step.parity ^= 1
        return pending, table, counts

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # everything allocated so far (modules, synthetic inputs) out of the cycle collector's sight: a full collection over it is
    # ~100 ms — one of those inside a 2 s timed region is 5 % (aliby_amd.runner does the same for run_positions callers)
    import gc

    gc.collect()
    gc.freeze()
    eng.profile = None
    for _ in range(args.warmup):
        step()[0].wait()
    eng.profile = None if args.no_kernel_timing else {}
    eng.profile_sample = {k: args.time_every for k in ("conv3x3_mfma", "conv3x3_mfma_deep", "fused_pointwise", "conv1x1_mfma")}  # (the head launch: every one)
    eng._sample_count = {}
    if model.fused is not None:
        model.fused.bytes_moved = 0
        model.fused.conv_stats = {}
    barrier()
    t0 = time.perf_counter()
    pending = None
    if args.overlap:
        state = phase_net(0)
        for k in range(steps):
            nxt_state = phase_net((k + 1) & 1) if k + 1 < steps else None  # queue the next network before this step's host read-backs
            nxt, table, counts = phase_post(state)
            if pending is not None:
                pending.wait()
            pending, state = nxt, nxt_state
    else:
        for _ in range(steps):
            nxt, table, counts = step()
            if pending is not None:
                pending.wait()  # the previous step's rows are on the host (its buffers may be reused two steps later)
            pending = nxt
    rows = tuple(torch.from_numpy(a) for a in pending.wait())  # the last step's rows land inside the timed region too
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if backend != "nccl":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = eng.collect_profile()
    eng.profile = None
    tiles_per_s = world * B * steps / dt
    n_cols = sum(r.shape[1] for r in rows)

    # ---- the one exchange step: gather the last step's rows on rank 0 (RCCL over xGMI) ------------------
    t0 = time.perf_counter()
    vals = torch.cat([r.cuda() for r in rows], dim=1)
    meta = torch.stack([torch.from_numpy(table.host["tile"].astype(np.int64)) * world + rank,
                        torch.from_numpy(table.host["label"].astype(np.int64)),
                        torch.zeros(table.n_obj, dtype=torch.int64), torch.zeros(table.n_obj, dtype=torch.int64)], 1).cuda()
    gv, gm = parallel.gather_rows(vals, meta)
    torch.cuda.synchronize()
    gather_ms = 1e3 * (time.perf_counter() - t0)

    roof, roof_deep, fracs, mfma = build_rooflines(prof, model, args, B, C, Z, Y, X, table.n_obj, n_tiles_net, steps)

    # ---- the same workload through the step API ------------------------------------------------------------------
    api = None
    if not args.no_api and not args.overlap:
        del stacks, tiles
        api = api_leg(args, cfg, base, channels, B, distinct, rank, world, dist, backend, eng)

    if rank == 0:
        feats = ("sizeshape + per channel [radial_zernikes, intensity, feret, texture, radial_distribution, zernike] + 10 pairs x "
                 "[pearson, costes, manders_fold, rwc]") if cfg["features"] is None else (
            "sizeshape + " + ", ".join(cfg["features"]) + f" on channels {channels}" + (" + colocalisation pairs" if multi_tree else ""))
        line = {
            "metric": "FOV tiles/sec (whole node)",
            "value": round(tiles_per_s, 3),
            "unit": "tiles/s",
            "n_gpus": world,
            "ranks": world,
            "backend": (dist.get_backend() if dist is not None else None),
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / steps, 3),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": f"u16 pixels, f64 feature accumulators, f32 dynamics, {args.net_dtype} U-Net",
            "data": "synthetic",
            "config": {
                "workload": f"{cfg['label']}: {B} FOV/step/GPU of {Y}x{X}x{C}ch (Z={Z}), ~{n_obj_per_fov:.0f} nuclei/FOV, Cellpose nuclei on "
                            f"ch{cfg['seg_channel']} + cp_measure ({distinct} distinct FOVs per rank"
                            + (f" replicated to {B}" if distinct < B else "") + ")"
                            + (f"; fixed job of {world * B * steps} FOVs split over {world} rank(s)" if args.total_fovs else ""),
                "features": feats,
                "segmentation": "U-Net forward with fixed-seed random weights (weights not obtainable offline; cost paid, output "
                                "discarded) + dynamics on analytic flows of the synthetic ground truth",
                "network_tiles": f"{n_tiles_net // B} tiles of {args.bsize} x {args.bsize} per FOV (tile_overlap 0.1), {args.net_batch} per forward",
                "objects_last_step": int(table.n_obj),
                "feature_vectors_per_s": round(tiles_per_s * float(table.n_obj) / B, 1),
                "columns": n_cols,
                "final_gather_ms": round(gather_ms, 2),
                "gathered_rows": int(gv.shape[0]) if gv is not None else None,
            },
            "roofline": roof,
            "roofline_mfma": roof_deep,
            "cpu_baseline": cpu,
            "mfma": mfma,
            "kernel_ms_per_step": {k: round(v["ms_total"] / steps, 3) for k, v in prof.items()},
            "kernel_hbm_frac": fracs,
        }
        if api is not None:
            line.update(api)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def api_leg(args, cfg, base, channels, B, distinct, rank, world, dist, backend, eng):
    """The same workload through the reference's API: one pipeline dict per position (build_pipeline_steps), inputs = pinned
    host arrays [1,C,Z,Y,X], aliby_amd.parallel.run_positions with B positions per device step, outputs = parquet + mask .npz on
    disk.  Returns {"value_api": tiles/s, "api": {...}, "api_split_ms_per_fov": {...}}."""
    import shutil
    import tempfile
    import warnings

    import torch

    from aliby_amd import runner
    from aliby_amd.pipe_builder import build_pipeline_steps

    n_pos = args.api_fovs or 16 * B  # (fill and drain of the pipeline, ~50 + ~65 ms, are part of the timed region)
    dev = torch.device("cuda", torch.cuda.current_device())
    dP_d = torch.stack([torch.from_numpy(b["dP"]) for b in base]).to(dev)
    prob_d = torch.stack([torch.from_numpy(b["prob"]) for b in base]).to(dev)
    pinned = [torch.from_numpy(b["pixels"][None]).pin_memory() for b in base]  # [1,C,Z,Y,X] each: the host side of the boundary
    seg = cfg["seg_channel"]
    # a plane is recognised by its first pixels (distinct FOVs differ in their noise) — on the device, so that the stand-in
    # for the network's output costs no host round trip
    keys_d = torch.from_numpy(np.stack([b["pixels"][seg].max(axis=0)[0, :8].astype(np.int32) for b in base])).to(dev)
    assert len({bytes(k) for k in keys_d.cpu().numpy()}) == len(base), "synthetic FOVs collide on their first 8 pixels"

    def override(x):
        sel = (x[:, 0, :8].to(torch.int32)[:, None, :] == keys_d[None]).all(-1).to(torch.int32).argmax(1)
        return dP_d.index_select(0, sel), prob_d.index_select(0, sel)

    kw = {} if cfg["features"] is None else {"features_to_extract": tuple(cfg["features"])}

    def pipelines(n):
        out = []
        for i in range(n):
            p = build_pipeline_steps(channels_to_segment={"nuclei": seg}, channels_to_extract=channels, **kw)
            p["steps"]["tile"]["image_kwargs"] = {"source": pinned[i % distinct].numpy()}
            p["steps"]["segment_nuclei"]["segmenter_kwargs"]["setup_params"] = SETUP
            out.append(p)
        return out

    WRITERS = None  # aliby_amd.runner sizes its writer threads / processes from the host share (ALIBY_WRITERS / ALIBY_WRITER_PROCS)
    SETUP = dict(flows_override=override, run_network_with_override=True, net_dtype=args.net_dtype, batch_size=args.net_batch, bsize=args.bsize)
    out_dir = Path(tempfile.mkdtemp(prefix=f"aliby_bench_r{rank}_"))
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            # warm-up: three batches — the runner keeps three page-locked arenas and three [B,C,Z,Y,X] device blocks in flight, and
            # each is allocated (hipHostMalloc / hipMalloc of ~0.5 GB, tens of ms on the launch thread) the first time it is used
            names = [f"w{rank}_{i:05d}" for i in range(3 * B)]
            runner.run_positions(pipelines(3 * B), names, out_dir / "warm", batch_size=B, shard=False, writers=WRITERS)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            names = [f"p{rank}_{i:05d}" for i in range(n_pos)]
            pipes = pipelines(n_pos)
            from aliby_amd import hostinfo

            cpu0 = hostinfo.cpu_stat()
            t0 = time.perf_counter()
            stats = {}
            prof = None
            if os.environ.get("ALIBY_PROFILE_API"):  # diagnostic: where the launch thread's time goes (cProfile, to stderr)
                import cProfile

                prof = cProfile.Profile()
                prof.enable()
            res = runner.run_positions(pipes, names, out_dir / "run", batch_size=B, shard=False, writers=WRITERS, stats=stats)
            if prof is not None:
                import pstats

                prof.disable()
                pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(35)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            cpu1 = hostinfo.cpu_stat()
            try:  # (stability of long runs: resident set of the process and device memory torch holds, after the leg)
                import resource

                stats["max_rss_mb"] = int(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024)
                stats["device_reserved_mb"] = int(torch.cuda.memory_reserved() >> 20)
            except Exception:
                pass
            stats["cgroup_cpu"] = {k: cpu1[k] - cpu0.get(k, 0) for k in cpu1 if k in ("usage_usec", "nr_periods", "nr_throttled", "throttled_usec")}
            rows = sum(r[0].num_rows for r in res if r[0] is not None)  # (None: ALIBY_ABLATE diagnostics)
            cols = len(res[0][0].column_names) if res[0][0] is not None else 0
            if dist is not None:
                t = torch.tensor([dt], dtype=torch.float64, device="cuda")
                if backend != "nccl":
                    t = t.cpu()
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            parquet_bytes = sum(f.stat().st_size for f in (out_dir / "run" / "profiles").glob("*.parquet"))
            # where the time goes: one more batch with every phase synchronised and timed (not part of value_api)
            split = runner.run_positions(pipelines(2 * B), [f"s{rank}_{i:05d}" for i in range(2 * B)], out_dir / "split", batch_size=B,
                                         shard=False, measure=True)  # (the second of two batches is the one reported)
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)
    return {
        "value_api": round(world * n_pos / dt, 3),
        "api": {"path": "pinned host arrays -> build_pipeline_steps() dicts -> aliby_amd.parallel.run_positions (Tiler -> segment -> "
                        "extract -> get_profiles_from_state -> parquet (zstd) + mask .npz on disk)",
                "positions_per_rank": n_pos, "positions_per_device_step": B, "seconds": round(dt, 3), "rows_written": rows,
                "columns": cols, "parquet_bytes_per_fov": int(parquet_bytes / max(n_pos, 1)), "main_thread": stats},
        "api_split_ms_per_fov": split,
    }


def _make_volume_input(job):
    """One synthetic stack for the true-3-D leg: config-5 pixels, ellipsoid ground truth per plane, analytic flows per plane."""
    from aliby_amd import synth

    fov, size, C, Z, n_target = job
    f = synth.make_fov(5, fov, shape=(size, size), n_channels=C, n_z=Z, n_target=n_target)
    gt = synth.ellipsoid_planes(f["nuclei"], Z, seed=fov)
    dP = np.zeros((Z, 2, size, size), np.float32)
    prob = np.empty((Z, size, size), np.float32)
    for z in range(Z):
        uniq = np.unique(gt[z])
        uniq = uniq[uniq != 0]
        fwd = np.zeros(int(gt[z].max()) + 1, gt.dtype)
        fwd[uniq] = np.arange(1, len(uniq) + 1)  # each plane labelled on its own, as a per-plane segmenter sees it
        dP[z], prob[z] = synth.analytic_flows(fwd[gt[z]])
    return dict(pixels=f["pixels"], gt=gt, dP=dP, prob=prob)


def main_volume(args):
    """Config 5 as a volume (BASELINE.json configs[4]: "3D Cellpose + 3D intensity features"; SURVEY.md 8(d).5's second number, an
    extension beyond reference behaviour, parity unpinned): per step B stacks [C=2, Z=32, 512, 512] resident in HBM go through
    stage -> every plane through normalise / tiles / U-Net / dynamics -> planes stitched along Z (the reference's stitch_threshold
    = 0.01) -> 3-D intensity statistics on both channels -> rows to the host.  A tile is one stack."""
    import multiprocessing as mp
    import warnings

    from aliby_amd import hostinfo, parallel

    cfg = CONFIGS[5]
    rank, world, local_rank = parallel.rank_world()
    size = args.size or cfg["size"]
    C, Z = cfg["C"], cfg["Z"]
    B = args.fovs if args.fovs != 64 else 8
    distinct = max(1, min(args.distinct, B))
    procs = args.host_procs or hostinfo.usable_cores()
    jobs = [(rank + world * i, size, C, Z, cfg["n_target"]) for i in range(distinct)]
    if procs > 1 and len(jobs) > 1 and not under_profiler():
        pool = mp.get_context("fork").Pool(min(procs, len(jobs)))
        try:
            base = pool.map(_make_volume_input, jobs, chunksize=1)
        finally:
            pool.close()
            pool.join()
    else:
        base = [_make_volume_input(j) for j in jobs]
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_volume(base[0])

    import torch

    from aliby_amd import _lib

    backend = os.environ.get("ALIBY_DIST_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    parallel.init(backend if world > 1 else None)
    dist = torch.distributed if world > 1 else None
    from aliby_amd.extraction.engine import _ptr, _stream_ptr
    from aliby_amd.segment.cellpose_hip import CellposeModel

    Y = X = size
    stacks = torch.empty((B, C, Z, Y, X), dtype=torch.uint16, device="cuda")
    dP_d = torch.empty((B * Z, 2, Y, X), dtype=torch.float32, device="cuda")
    prob_d = torch.empty((B * Z, Y, X), dtype=torch.float32, device="cuda")
    for b in range(B):
        k = b % distinct
        stacks[b] = torch.from_numpy(base[k]["pixels"]).cuda()
        dP_d[b * Z:(b + 1) * Z] = torch.from_numpy(base[k]["dP"]).cuda()
        prob_d[b * Z:(b + 1) * Z] = torch.from_numpy(base[k]["prob"]).cuda()
    tiles = torch.empty_like(stacks)
    rect = np.array([[0, 0, Y, X]], np.int32)
    flags = np.zeros(1, np.int32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = CellposeModel(net_dtype=args.net_dtype, seed=0, flows_override=lambda x: (dP_d, prob_d), run_network_with_override=True,
                              batch_size=args.net_batch, bsize=args.bsize)
    eng = model.eng
    n_tiles_net = B * Z * model._geometry(Y, X)["ny"] * model._geometry(Y, X)["nx"]
    state = {"parity": 0}

    def step():
        with eng.timed("stage_crop_pad"):
            _lib.check(eng.lib.aliby_crop_pad_u16(eng.ctx.handle, _ptr(stacks), B * C, Z, Y, X, _ptr(rect), 1, Y, X, _ptr(tiles), _ptr(flags),
                                                  _stream_ptr()))
        px = tiles.view(B, C, Z, Y, X)
        planes = px[:, cfg["seg_channel"]].reshape(B * Z, Y, X)
        masks, _, _ = model.eval(planes, do_3D=False, stitch_threshold=0.0, normalize=dict(norm3D=False), z_axis=None)
        with eng.timed("stitch_planes"):
            volume, counts = eng.stitch_planes(masks.view(B, Z, Y, X), threshold=0.01)
        rows = [eng.intensity3d(volume, px, c, counts) for c in range(C)]
        pending = eng.to_host_async(tuple(rows), slot=state["parity"])
        state["parity"] ^= 1
        return pending, counts

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    eng.profile = None
    for _ in range(max(args.warmup, 1)):
        step()[0].wait()
    eng.profile = None if args.no_kernel_timing else {}
    eng.profile_sample = {k: args.time_every for k in ("conv3x3_mfma", "conv3x3_mfma_deep", "fused_pointwise", "conv1x1_mfma")}
    eng._sample_count = {}
    if model.fused is not None:
        model.fused.bytes_moved = 0
        model.fused.conv_stats = {}
    steps = max(args.steps, 1)
    barrier()
    t0 = time.perf_counter()
    pending = None
    for _ in range(steps):
        nxt, counts = step()
        if pending is not None:
            pending.wait()
        pending = nxt
    rows = pending.wait()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if backend != "nccl":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = eng.collect_profile()
    eng.profile = None
    n_obj = int(np.sum(counts))
    roof, roof_deep, fracs, mfma = build_rooflines(prof, model, args, B * Z, 1, 1, Y, X, n_obj, n_tiles_net, steps)
    truth = [len(np.unique(b["gt"][b["gt"] > 0])) for b in base]
    if rank == 0:
        print(json.dumps({
            "metric": "FOV tiles/sec (whole node)", "value": round(world * B * steps / dt, 3), "unit": "tiles/s", "n_gpus": world, "ranks": world,
            "backend": (dist.get_backend() if dist is not None else None), "steps": steps, "warmup": max(args.warmup, 1),
            "ms_per_step": round(1e3 * dt / steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": f"u16 pixels, exact integer sums -> f64 features, f32 dynamics, {args.net_dtype} U-Net", "data": "synthetic",
            "config": {"workload": f"C5 as a VOLUME (extension beyond reference behaviour, parity unpinned; SURVEY.md 8(d).5): {B} stacks/step/GPU of "
                                   f"[{C} ch, Z={Z}, {Y}x{X}], every plane segmented ({B * Z} planes, {n_tiles_net} network tiles per step), planes "
                                   "stitched along Z by IoU >= 0.01, 12 3-D intensity columns per object and channel; a tile is one stack",
                       "segmentation": "U-Net forward with fixed-seed random weights (cost paid, output discarded) + dynamics on analytic flows of the "
                                       "per-plane ground truth (ellipsoids)",
                       "objects_last_step": n_obj, "objects_ground_truth": int(np.sum([truth[b % distinct] for b in range(B)])),
                       "planes_per_s": round(world * B * Z * steps / dt, 1), "columns": int(sum(r.shape[1] for r in rows))},
            "roofline": roof, "roofline_mfma": roof_deep, "cpu_baseline": cpu, "mfma": mfma,
            "kernel_ms_per_step": {k: round(v["ms_total"] / steps, 3) for k, v in prof.items()}, "kernel_hbm_frac": fracs,
        }))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline_volume(sample):
    """The true-3-D leg on one core: the CPU restatement's dynamics on a few planes (scaled to the stack), stitching and the 3-D
    intensity block on the whole stack.  The network forward is not part of this CPU figure."""
    from oracle import cellpose_restated as cr
    from oracle import volume_restated as vr
    from oracle.cpu_baseline import cpu_model

    Z = sample["gt"].shape[0]
    zs = [Z // 2 - 1, Z // 2, Z // 2 + 1]
    t0 = time.perf_counter()
    planes = {z: cr.finish_labels(cr.compute_masks(sample["dP"][z], sample["prob"][z])) for z in zs}
    t_dyn = (time.perf_counter() - t0) / len(zs) * Z
    per_plane = []
    for z in range(Z):  # (stitching and features are timed on per-plane labels derived from the ground truth)
        uniq = np.unique(sample["gt"][z])
        uniq = uniq[uniq != 0]
        fwd = np.zeros(int(sample["gt"][z].max()) + 1, np.int64)
        fwd[uniq] = np.arange(1, len(uniq) + 1)
        per_plane.append(fwd[sample["gt"][z]])
    t0 = time.perf_counter()
    vol, n = vr.stitch3d(np.stack(per_plane), 0.01)
    for c in range(sample["pixels"].shape[0]):
        vr.intensity3d(vol, sample["pixels"][c])
    t_rest = time.perf_counter() - t0
    del planes
    return {"value": round(1.0 / (t_dyn + t_rest), 4), "unit": "tiles/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
            "os_cpu_count": os.cpu_count(),
            "sample": f"oracle (CPU restatement) on one stack: NumPy dynamics on 3 of {Z} planes scaled to the stack ({t_dyn:.1f} s) + Z "
                      f"stitching + 3-D intensity on 2 channels ({t_rest:.1f} s), {n} objects; the network forward is not part of this CPU figure"}


def main_timelapse(args):
    """Config 4 (BASELINE.json configs[3]): yeast time-lapse positions, T x [1, 5, 512, 512], 117-px trap tiles — trap
    detection on the first frame, drift per timepoint, per-tile segmentation, IoU tracking, sizeshape — through the step API
    (aliby_amd.parallel.run_positions), B positions in lockstep.  A step = one timepoint of every position of the batch;
    a tile = one 117x117 trap window of one timepoint.  Inputs are device tensors (resident in HBM before the clock starts)."""
    import shutil
    import tempfile
    import warnings

    from aliby_amd import hostinfo, parallel, synth

    rank, world, local_rank = parallel.rank_world()
    B = args.fovs if args.fovs != 64 else 16
    K, W = max(args.steps, 2), max(args.warmup, 1)
    T = K + W
    tl = synth.make_timelapse(T=T, seed=11 + rank)
    half, tile = 117 // 2, 117
    tree = {"None": {"None": ["sizeshape"]}}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_timelapse(tl, tree, min(T, 6))

    import torch

    backend = os.environ.get("ALIBY_DIST_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    parallel.init(backend if world > 1 else None)
    dist = torch.distributed if world > 1 else None
    from aliby_amd import runner
    from aliby_amd.extraction.engine import FeatureEngine
    from aliby_amd.tile.tiles import TileLocations

    frames = torch.from_numpy(tl["pixels"]).cuda()  # [T,1,5,512,512]
    # analytic network-scale flows of the ground truth, per timepoint, cropped at the windows the tiler will use
    dP_full, prob_full = [], []
    for t in range(T):
        d, p = synth.analytic_flows(tl["labels"][t])
        dP_full.append(torch.from_numpy(d))
        prob_full.append(torch.from_numpy(p))
    dP_full, prob_full = torch.stack(dP_full).cuda(), torch.stack(prob_full).cuda()
    clock = {"calls": 0, "t0": 0, "n": B}
    state = {}

    def override(x):  # x [B*F,117,117]: every position is the same sample, so one set of windows serves the batch
        t = clock["t0"] + clock["calls"]
        clock["calls"] += 1
        tiler = state["tiler"]
        rects = tiler.rects(t)
        dP = torch.stack([dP_full[t, :, y : y + h, x0 : x0 + w] for y, x0, h, w in rects])
        pr = torch.stack([prob_full[t, y : y + h, x0 : x0 + w] for y, x0, h, w in rects])
        reps = x.shape[0] // dP.shape[0]
        return dP.repeat(reps, 1, 1, 1), pr.repeat(reps, 1, 1)

    setup = dict(flows_override=override, run_network_with_override=True, net_dtype=args.net_dtype, batch_size=args.net_batch, bsize=args.bsize)

    def pipelines(t0, ntps):
        out = []
        for i in range(B):
            out.append({
                "ntps": ntps,
                "steps": {
                    "tile": {"image_kwargs": {"source": frames[t0 : t0 + ntps]}, "tile_size": tile, "ref_channel": 0, "calculate_drift": True},
                    "segment_cells": {"segmenter_kwargs": {"kind": "cellpose", "per_tile": True, "setup_params": setup}, "channel_to_segment": 0},
                    "track": {"kind": "stitch", "stitch_threshold": 0.25},
                    "extract_cells": {"tree": tree},
                },
                "passed_data": {"track": [("masks", "segment_cells"), ("track_info", "track")],
                                "extract_cells": [("masks", "segment_cells"), ("pixels", "tile")]},
                "passed_methods": {"segment_cells": ("tile", "get_fczyx")},
                "save": ("segment_cells",), "save_interval": 1, "retain": {"tile": 1, "segment_cells": 2},
            })
        return out

    class _Spy:  # the override needs the windows of the batch's first tiler
        def __init__(self, fn):
            self.fn = fn

        def __call__(self, name, params, other=None):
            made = self.fn(name, params, other)
            if name == "tile" and "tiler" not in state:
                state["tiler"] = made
            return made

    from aliby_amd.pipe import init_step

    out_dir = Path(tempfile.mkdtemp(prefix=f"aliby_bench4_r{rank}_"))
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            clock.update(calls=0, t0=0)
            runner.run_positions(pipelines(0, W), [f"w{i}" for i in range(B)], out_dir / "warm", batch_size=B, shard=False,
                                 init_step_fn=_Spy(init_step))
            state.clear()
            FeatureEngine.shared_profile = None if args.no_kernel_timing else {}
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            clock.update(calls=0, t0=W)
            t0 = time.perf_counter()
            res = runner.run_positions(pipelines(W, K), [f"p{i}" for i in range(B)], out_dir / "run", batch_size=B, shard=False,
                                       init_step_fn=_Spy(init_step))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        n_tiles = len(state["tiler"].tile_locs)
        rows = sum(r[0].num_rows for r in res)
        prof = FeatureEngine(local_rank).collect_profile() if FeatureEngine.shared_profile is not None else {}
        FeatureEngine.shared_profile = None
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if backend != "nccl":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tiles_per_s = world * B * K * n_tiles / dt
    roof, fracs = None, {}
    if prof:
        F = B * n_tiles
        for name, g in prof.items():
            if name in ("unet_forward",) or name.startswith("conv") or name in ("first_conv", "style", "out_head"):
                continue
            gb = alg_bytes(name, F, 1, 5 if name in ("stage_crop_pad", "select_project", "reduce_z") else 1, tile, tile, rows // max(K, 1), 0)
            ms = g["ms_total"] / max(g["launches"], 1)
            if ms > 0:
                fracs[name] = (round(gb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), g["ms_total"], ms, gb, g["launches"])
        if fracs:
            dominant = max(fracs, key=lambda k: fracs[k][1])
            fr, _, ms, gb, launches = fracs[dominant]
            roof = {"bound": "hbm", "kernel": dominant, "achieved": round(gb / (ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": fr, "traffic": None, "alg_bytes_per_launch": gb, "avg_launch_ms": round(ms, 4), "launches": launches,
                    "note": "117-px tiles: a launch moves ~10 MB, so every kernel of this configuration is launch-latency bound"}
    if rank == 0:
        print(json.dumps({
            "metric": "FOV tiles/sec (whole node)", "value": round(tiles_per_s, 3), "unit": "tiles/s", "n_gpus": world, "ranks": world,
            "backend": (dist.get_backend() if dist is not None else None), "steps": K,
            "warmup": W, "ms_per_step": round(1e3 * dt / K, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": f"u16 pixels, f64 feature accumulators, f32 dynamics, {args.net_dtype} U-Net", "data": "synthetic",
            "config": {"workload": f"C4: {B} time-lapse positions/GPU in lockstep, T={K} timed timepoints of [1,5,512,512], {n_tiles} trap tiles "
                                   f"of 117x117 per position (detected on the first frame), drift per timepoint, per-tile Cellpose, IoU tracking, "
                                   "sizeshape; through the step API (run_positions), stacks resident in HBM",
                       "segmentation": "U-Net forward with fixed-seed random weights (cost paid, output discarded) + dynamics on analytic flows",
                       "rows_written": rows, "feature_vectors_per_s": round(world * rows / dt, 1)},
            "roofline": roof, "cpu_baseline": cpu,
            "kernel_ms_per_step": {k: round(v["ms_total"] / K, 3) for k, v in prof.items()},
            "kernel_hbm_frac": {k: v[0] for k, v in fracs.items()},
        }))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline_timelapse(tl, tree, n_tp):
    """Config 4 on one core: the CPU restatement of trap detection (once), then per timepoint drift, tile windows, NumPy
    dynamics per tile, IoU stitching and sizeshape through the reference-structured extraction — n_tp timepoints timed."""
    from oracle import aliby_extract as ox
    from oracle import cellpose_restated as cr
    from oracle import tiler_ref
    from oracle.cpu_baseline import cpu_model
    from oracle.drift_restated import phase_cross_correlation as pcc
    from oracle.track_restated import stitch_rois
    from oracle.traps_restated import segment_traps
    from aliby_amd import synth

    frames, half, tile = tl["pixels"], 117 // 2, 117
    t0 = time.perf_counter()
    centres = [c for c in segment_traps(frames[0, 0, 0], tile) if half < c[0] < 512 - half and half < c[1] < 512 - half]
    t_traps = time.perf_counter() - t0
    drifts, prev, info = [], None, None
    t0 = time.perf_counter()
    for t in range(n_tp):
        drifts.append(pcc(frames[max(0, t - 1), 0, 0], frames[t, 0, 0]).tolist())
        cum = np.sum(drifts, axis=0)
        ranges = []
        for cy, cx in centres:
            y, x = (np.array([cy, cx]) - cum).astype(int)
            ranges.append((slice(int(y) - half, int(y) - half + tile), slice(int(x) - half, int(x) - half + tile)))
        px = tiler_ref.get_fczyx(frames[t], ranges)
        masks = [cr.finish_labels(cr.compute_masks(*synth.analytic_flows(tiler_ref.relabel_sequential(tl["labels"][t][r])))) for r in ranges]
        if prev is not None:
            info = stitch_rois([[a, b] for a, b in zip(prev, masks)], info)
        prev = masks
        ox.process_tree_masks(tree, masks, px, ox.extract_tree)
    per_tp = (time.perf_counter() - t0) / n_tp
    n_tiles = len(centres)
    return {"value": round(n_tiles / per_tp, 4), "unit": "tiles/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
            "os_cpu_count": os.cpu_count(),
            "sample": f"oracle (CPU restatement) on one position: {n_tp} timepoints x {n_tiles} tiles, {per_tp:.2f} s per timepoint "
                      f"(drift + windows + NumPy dynamics per tile + IoU stitching + sizeshape on full-tile masks); trap detection once "
                      f"{t_traps:.1f} s (not in the rate); the network forward is not part of this CPU figure"}


if __name__ == "__main__":
    # stdout carries the ONE JSON line: whatever the steps print on the way (the reference's own progress messages, mirrored —
    # "Saving ...", "Tiler:TrapIdentification: Trying again.") goes to stderr
    import contextlib

    class _JsonOnly:
        def __init__(self, out, err):
            self.out, self.err = out, err

        def write(self, text):
            (self.out if text.lstrip().startswith('{"metric"') or (text == "\n" and self._json) else self.err).write(text)
            self._json = text.lstrip().startswith('{"metric"')
            return len(text)

        _json = False

        def flush(self):
            self.out.flush()
            self.err.flush()

    with contextlib.redirect_stdout(_JsonOnly(sys.stdout, sys.stderr)):
        main()
