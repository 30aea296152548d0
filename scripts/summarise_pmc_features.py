"""Per-kernel table of the SQ counters collected by scripts/pmc_features.sh: the LAST timed step's launches of every non-network
kernel, summed per kernel name.  Columns (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are
quad-cycles summed over waves, so their ratios are fractions of wave lifetime:
  valu_busy  = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   (a wave was issuing / executing a VALU instruction)
  wait_inst  = SQ_WAIT_INST_ANY   / SQ_WAVE_CYCLES    (issue stalls: dependency / pipe busy)
  wait_lds   = SQ_WAIT_INST_LDS   / SQ_WAVE_CYCLES
  wait_any   = SQ_WAIT_ANY        / SQ_WAVE_CYCLES    (parked at s_waitcnt / barrier)
  valu/obj   = SQ_INSTS_VALU per object row (wave-instructions), lds/obj likewise
  simd_valu  = SQ_ACTIVE_INST_VALU * 4 / (kernel duration x 1024 SIMDs x clock): share of the chip's VALU issue capacity in use."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
NET = ("k_conv", "k_first_conv", "k_fused", "k_out_head", "k_maxpool", "k_style", "k_nn", "k_pack", "k_tiles", "k_make_tiles", "k_average", "__amd")


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def load(which):
    files = glob.glob(f"gpurun_out/pmc_{tag}_{which}/**/*counter_collection.csv", recursive=True)
    if not files:
        return {}, {}
    rows = list(csv.DictReader(open(files[0])))
    # the last step: dispatches after the last k_normalize99 launch's predecessor crop ... simpler: keep the LAST occurrence
    # block, i.e. for each kernel name the launches whose Dispatch_Id is above the id of the second-to-last k_crop_copy
    crops = sorted({int(r["Dispatch_Id"]) for r in rows if "k_crop_copy" in r["Kernel_Name"]})
    lo = crops[-1] if crops else 0
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for r in rows:
        d = int(r["Dispatch_Id"])
        k = short(r["Kernel_Name"])
        if d < lo or k.startswith(NET):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(d)
    return agg, {k: len(v) for k, v in calls.items()}


def durations():
    files = glob.glob(f"gpurun_out/pmc_{tag}_feat1/**/*kernel_trace.csv", recursive=True)
    if not files:
        return {}
    rows = list(csv.DictReader(open(files[0])))
    crops = sorted(int(r["Dispatch_Id"]) for r in rows if "k_crop_copy" in r["Kernel_Name"])
    lo = crops[-1] if crops else 0
    out = collections.defaultdict(float)
    for r in rows:
        if int(r["Dispatch_Id"]) >= lo:
            out[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return out


a1, calls = load("feat1")
a2, _ = load("feat2")
dur = durations()
n_obj = None
try:
    n_obj = json.loads([ln for ln in open(f"gpurun_out/pmc_{tag}_feat1.log") if ln.startswith("{")][-1])["config"]["objects_last_step"]
except Exception:
    pass
out = open(f"gpurun_out/{tag}_pmc_per_object_kernels.csv", "w")
w = csv.writer(out)
w.writerow(["kernel", "launches", "us_under_pmc", "waves", "valu_busy", "wait_inst", "wait_lds", "wait_any", "active_any", "valu_insts_per_object",
            "lds_insts_per_object", "vmem_rd_per_object", "salu_per_object", "lds_bank_conflict_frac", "simd_valu_share"])
for k in sorted(a1, key=lambda k: -dur.get(k, 0)):
    c1, c2 = a1[k], a2.get(k, {})
    wc = c1.get("SQ_WAVE_CYCLES", 0) or 1
    wc2 = (c2.get("SQ_WAIT_ANY", 0) + c2.get("SQ_ACTIVE_INST_ANY", 0)) or 1
    per = (lambda v: round(v / n_obj, 1)) if n_obj else (lambda v: None)
    us = dur.get(k, 0)
    simd = c1.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (us * 1e-6 * 1024 * 2.1e9) if us else None
    w.writerow([k, calls.get(k), round(us, 1), int(c1.get("SQ_WAVES", 0)), round(c1.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3),
                round(c1.get("SQ_WAIT_INST_ANY", 0) / wc, 3), round(c1.get("SQ_WAIT_INST_LDS", 0) / wc, 3),
                round(c2.get("SQ_WAIT_ANY", 0) / wc, 3) if c2 else None, round(c2.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3) if c2 else None,
                per(c1.get("SQ_INSTS_VALU", 0)), per(c1.get("SQ_INSTS_LDS", 0)), per(c2.get("SQ_INSTS_VMEM_RD", 0)) if c2 else None,
                per(c2.get("SQ_INSTS_SALU", 0)) if c2 else None,
                round(c2.get("SQ_LDS_BANK_CONFLICT", 0) / max(c2.get("SQ_LDS_IDX_ACTIVE", 0), 1), 3) if c2 else None,
                round(simd, 3) if simd is not None else None])
out.close()
print(open(f"gpurun_out/{tag}_pmc_per_object_kernels.csv").read())
