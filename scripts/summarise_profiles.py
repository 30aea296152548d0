"""Turn the rocprofv3 outputs of scripts/profile_bench.sh into the small summaries committed under profiles/."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01c"


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name[:110]


# ---- kernel stats over the LAST traced step (steady state: after MIOpen's find phase)
files = glob.glob(f"gpurun_out/prof_{tag}/**/*kernel_trace.csv", recursive=True)
if files:
    rows = sorted(csv.DictReader(open(files[0])), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "k_normalize99" in r["Kernel_Name"]]
    seg = rows[marks[-2]:marks[-1]] if len(marks) >= 2 else rows
    agg = collections.OrderedDict()
    for r in seg:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(short(r["Kernel_Name"]), [0, 0, 10**12, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    with open(f"profiles/{tag}_kernel_stats_steady_step.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent"])
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, a[0], round(a[1] / 1e3, 1), round(a[1] / a[0] / 1e3, 2), round(a[2] / 1e3, 2), round(a[3] / 1e3, 2), round(100 * a[1] / tot, 2)])
    print("steady step: kernel time", tot / 1e6, "ms over", len(seg), "launches")
    # GPU idle time between consecutive kernels of the step (dependent launches on one stream): how launch-bound the step is
    end_so_far, idle, gaps = int(seg[0]["End_Timestamp"]), 0, []
    for r in seg[1:]:
        g = int(r["Start_Timestamp"]) - end_so_far
        if g > 0:
            idle += g
            gaps.append(g)
        end_so_far = max(end_so_far, int(r["End_Timestamp"]))
    span = end_so_far - int(seg[0]["Start_Timestamp"])
    gaps.sort()
    json.dump({"step_span_ms": round(span / 1e6, 3), "kernel_busy_ms": round((span - idle) / 1e6, 3), "idle_ms": round(idle / 1e6, 3),
               "launches": len(seg), "gaps": len(gaps), "median_gap_us": round(gaps[len(gaps) // 2] / 1e3, 2) if gaps else 0,
               "p90_gap_us": round(gaps[int(len(gaps) * 0.9)] / 1e3, 2) if gaps else 0,
               "idle_ms_in_gaps_over_20us": round(sum(g for g in gaps if g > 20000) / 1e6, 3),
               "note": "rocprofv3 --kernel-trace timestamps; tracing itself slows the host's launch rate"},
              open(f"profiles/{tag}_launch_gaps.json", "w"), indent=1)
    # the bench's timing groups, from the same trace: what `roofline.avg_launch_ms` / `roofline_mfma.avg_launch_ms` must agree with
    def grp(k):
        if k.startswith("k_conv_pair32"):
            return "conv3x3_mfma_pair"
        if k.startswith("k_conv_first_pair"):
            return "conv3x3_mfma_first_pair"
        if k.startswith("k_conv3x3_deep"):
            return "conv3x3_mfma_deep"
        if k.startswith("k_conv3x3"):
            params = k.split("(")[0].rstrip(">").split("<")[-1].split(", ")
            if k.startswith("k_conv3x3<") and len(params) == 8 and params[7] == "true":
                return "conv3x3_mfma_head"
            return "conv3x3_mfma_deep" if ", 128," in k else "conv3x3_mfma"
        if k.startswith("k_fused_act"):
            return "fused_pointwise"
        if k.startswith("k_out_head"):
            return "out_head"
        if k.startswith("k_conv1x1"):
            return "conv1x1_mfma"
        if k.startswith("k_first_conv"):
            return "first_conv"
        return None
    groups_t = {}
    for k, a in agg.items():
        g = grp(k)
        if g:
            t = groups_t.setdefault(g, [0, 0])
            t[0] += a[0]; t[1] += a[1]
    json.dump({g: {"launches_per_step": t[0], "total_ms_per_step": round(t[1] / 1e6, 3), "avg_launch_ms": round(t[1] / t[0] / 1e6, 4)}
               for g, t in groups_t.items()}, open(f"profiles/{tag}_group_avg_from_trace.json", "w"), indent=1)

# ---- HBM traffic of the hand-written kernels from the PMC passes
traffic = {}
for which, col in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    files = glob.glob(f"gpurun_out/pmc_{tag}_{which}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != col:
            continue
        k = short(r["Kernel_Name"])
        t = traffic.setdefault(k, {"FETCH_SIZE": [], "WRITE_SIZE": []})
        t[col].append(float(r["Counter_Value"]))
def conv_group(k):  # same split as fused_unet._launch_unit / _unit_head: 128-output-channel launches are the compute-bound group
    if k.startswith("k_conv_pair32"):
        return "conv3x3_mfma_pair"
    if k.startswith("k_conv_first_pair"):
        return "conv3x3_mfma_first_pair"
    if not k.startswith("k_conv3x3"):
        return None
    if k.startswith("k_conv3x3_deep"):
        return "conv3x3_mfma_deep"
    params = k.split("(")[0].rstrip(">").split("<")[-1].split(", ")
    if k.startswith("k_conv3x3<") and len(params) == 8 and params[7] == "true":
        return "conv3x3_mfma_head"
    return "conv3x3_mfma_deep" if ", 128," in k else "conv3x3_mfma"


groups = {"conv3x3_mfma": lambda k: conv_group(k) == "conv3x3_mfma", "conv3x3_mfma_deep": lambda k: conv_group(k) == "conv3x3_mfma_deep",
          "conv3x3_mfma_head": lambda k: conv_group(k) == "conv3x3_mfma_head", "conv3x3_mfma_pair": lambda k: conv_group(k) == "conv3x3_mfma_pair",
          "conv3x3_mfma_first_pair": lambda k: conv_group(k) == "conv3x3_mfma_first_pair",
          "fused_pointwise": lambda k: k.startswith("k_fused_act"), "out_head": lambda k: k.startswith("k_out_head"),
          "conv1x1_mfma": lambda k: k.startswith("k_conv1x1"), "first_conv": lambda k: k.startswith("k_first_conv")}
out = {}
with open(f"profiles/{tag}_pmc_hbm_traffic_summary.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "FETCH_SIZE_KB_avg (as reported)", "WRITE_SIZE_KB_avg", "hbm_bytes_per_launch (2*FETCH + WRITE, gfx950 correction)"])
    for k, t in sorted(traffic.items()):
        if not t["FETCH_SIZE"] or not t["WRITE_SIZE"] or not k.startswith("k_"):
            continue
        fa = sum(t["FETCH_SIZE"]) / len(t["FETCH_SIZE"]); wa = sum(t["WRITE_SIZE"]) / len(t["WRITE_SIZE"])
        w.writerow([k, len(t["FETCH_SIZE"]), round(fa, 1), round(wa, 1), round((2 * fa + wa) * 1024)])
for g, match in groups.items():
    fs = [v for k, t in traffic.items() if match(k) for v in t["FETCH_SIZE"]]
    ws = [v for k, t in traffic.items() if match(k) for v in t["WRITE_SIZE"]]
    if fs and ws:
        out[g] = {"kernel": ", ".join(sorted(k.split("(")[0] for k in traffic if match(k)))[:300], "launches": len(fs), "fetch_size_kb_per_launch": sum(fs) / len(fs),
                  "write_size_kb_per_launch": sum(ws) / len(ws),
                  "hbm_bytes_per_launch": (2 * sum(fs) / len(fs) + sum(ws) / len(ws)) * 1024,
                  "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 1`; "
                         "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); average over all launches "
                         "of the group in the run (same launch mix as the timed region: launches are per U-Net batch)",
                  "source": f"profiles/{tag}_pmc_hbm_traffic_summary.csv"}
if out:
    json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out.items()}), "MB per launch")
