"""Per-kernel timing harness on the config-2 workload (not part of the product or the tests).
usage: python scripts/microbench.py [B]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from aliby_amd import synth, _lib
from aliby_amd.extraction.engine import FeatureEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
which = sys.argv[2] if len(sys.argv) > 2 else "nuclei"
eng = FeatureEngine(0)
base = [synth.make_fov(2, i) for i in range(2)]
px = torch.stack([torch.from_numpy(base[b % 2]["pixels"][:, 0]) for b in range(B)]).cuda()      # [B,5,Y,X]
lab = torch.stack([torch.from_numpy(base[b % 2][which]) for b in range(B)]).cuda()
tab = eng.object_table(lab)
print("objects", tab.n_obj, "max_area", tab.max_area, "max_h", tab.max_h, "max_w", tab.max_w)

def timeit(name, fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    print(f"{name:40s} {ms:8.3f} ms   {1e3*ms/max(tab.n_obj,1):8.3f} us/object")
    return ms

out = eng.new_output(tab.n_obj, 128)
U = _lib.U16
timeit("intensity(edge)", lambda: eng.intensity(lab, px, U, 0, tab, out, 0, True))
timeit("intensity(no edge)", lambda: eng.intensity(lab, px, U, 0, tab, out, 0, False))
timeit("sizeshape", lambda: eng.sizeshape(lab, tab, out, 0))
timeit("feret", lambda: eng.feret(lab, tab, out, 0))
timeit("texture", lambda: eng.texture(lab, px, U, 0, tab, out, 0))
for cols in (dict(pearson=0), dict(manders_fold=0), dict(rwc=0), dict(costes=0), dict(pearson=0, manders_fold=2, rwc=4, costes=6)):
    timeit("coloc " + "+".join(cols), lambda: eng.coloc(lab, px, U, 0, 1, tab, out, cols))
timeit("radial_zernikes", lambda: eng.zernike(lab, px, U, 0, tab, out, 0, True))
timeit("radial_distribution", lambda: eng.radial_distribution(lab, px, U, 0, tab, out, 0))
def geo():
    tab._binmaps = {}
    eng.radial_geometry(lab, tab, 4)
timeit("radial_geometry", geo)
timeit("cell_metrics", lambda: eng.cell_metrics(lab, px, U, 0, tab))
timeit("object_table", lambda: eng.object_table(lab))
