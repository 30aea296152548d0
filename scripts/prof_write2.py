import os, sys, time, numpy as np
import pyarrow as pa, pyarrow.parquet as pq
from concurrent.futures import ThreadPoolExecutor
n, ncol = 256, 1043
names = [f"0/max/texture/Feature_{j:04d}_xx" for j in range(ncol)]
tabs = [pa.table({nm: np.random.rand(n) for nm in names}) for _ in range(4)]
def w_mem(i):
    sink = pa.BufferOutputStream(); pq.write_table(tabs[i % 4], sink, compression="zstd")
N = 96
for cc in (256, 16, 1):
    pa.set_cpu_count(cc)
    for nt in (1, 4, 12):
        with ThreadPoolExecutor(nt) as ex:
            t = time.perf_counter(); list(ex.map(w_mem, range(N))); dt = time.perf_counter() - t
        print(f"arrow cpu_count {cc:3d} threads {nt:2d}: {dt / N * 1e3:6.2f} ms per file")
print("default pool", pa.default_memory_pool().backend_name)
for pool_name in ("system", "jemalloc", "mimalloc"):
    try:
        pool = getattr(pa, f"{pool_name}_memory_pool")()
    except Exception as e:
        print(pool_name, "unavailable", e); continue
    pa.set_memory_pool(pool)
    tabs2 = [pa.table({nm: np.random.rand(n) for nm in names}) for _ in range(4)]
    def w2(i):
        sink = pa.BufferOutputStream(); pq.write_table(tabs2[i % 4], sink, compression="zstd")
    for nt in (1, 12):
        with ThreadPoolExecutor(nt) as ex:
            t = time.perf_counter(); list(ex.map(w2, range(N))); dt = time.perf_counter() - t
        print(f"pool {pool_name:9s} threads {nt:2d}: {dt / N * 1e3:6.2f} ms per file")
