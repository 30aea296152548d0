"""GPU busy / idle inside the through-the-API leg of bench.py, from a rocprofv3 --kernel-trace CSV.
usage: python scripts/api_gaps.py <dir with *kernel_trace.csv> <n batches of the timed API run>"""
import csv, glob, sys, json, collections
d, nb = sys.argv[1], int(sys.argv[2])
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_normalize99" in r["Kernel_Name"]]
# the last batch is the synchronised "split" pass; the nb before it are the timed run
lo, hi = marks[-1 - nb], marks[-1]
seg = rows[lo:hi]
busy_end, idle, gaps = int(seg[0]["End_Timestamp"]), 0, []
for a, r in zip(seg, seg[1:]):
    g = int(r["Start_Timestamp"]) - busy_end
    if g > 0:
        idle += g
        gaps.append((g, a["Kernel_Name"][:60], r["Kernel_Name"][:60]))
    busy_end = max(busy_end, int(r["End_Timestamp"]))
span = busy_end - int(seg[0]["Start_Timestamp"])
by = collections.Counter()
for g, a, b in gaps:
    if g > 50_000:
        by[(a.split("(")[0][:40], b.split("(")[0][:40])] += g
print(json.dumps({"span_ms": span / 1e6, "busy_ms": (span - idle) / 1e6, "idle_ms": idle / 1e6, "launches": len(seg),
                  "ms_per_fov_span": span / 1e6 / (nb * 64)}, indent=1))
for (a, b), g in by.most_common(14):
    print(f"{g / 1e6:8.2f} ms idle between  {a}  ->  {b}")
print("columns:", list(rows[0].keys()))
big = sorted(((int(r["Start_Timestamp"]) - int(a["End_Timestamp"]), i) for i, (a, r) in enumerate(zip(seg, seg[1:]))), reverse=True)[:12]
t0 = int(seg[0]["Start_Timestamp"])
for g, i in sorted(big, key=lambda x: x[1]):
    a, r = seg[i], seg[i + 1]
    print(f"t={(int(a['End_Timestamp']) - t0) / 1e6:9.2f} ms  gap {g / 1e6:7.2f} ms   {a['Kernel_Name'][:38]:38s} [q{a.get('Queue_Id')} s{a.get('Stream_Id')}] -> {r['Kernel_Name'][:38]:38s} [q{r.get('Queue_Id')} s{r.get('Stream_Id')}]")
