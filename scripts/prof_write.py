"""Where does concurrent parquet writing stop scaling on the GPU box?  threads vs processes, file vs memory sink."""
import os, sys, time, numpy as np
import pyarrow as pa, pyarrow.parquet as pq
from concurrent.futures import ThreadPoolExecutor, ProcessPoolExecutor
import multiprocessing as mp
n, ncol = 256, 1043
names = [f"0/max/texture/Feature_{j:04d}_xx" for j in range(ncol)]
def make():
    return pa.table({nm: np.random.rand(n) for nm in names})
tabs = [make() for _ in range(4)]
def w_file(i, d="/tmp"):
    pq.write_table(tabs[i % 4], f"{d}/w{i}.parquet", compression="zstd")
def w_shm(i):
    w_file(i, "/dev/shm")
def w_mem(i):
    sink = pa.BufferOutputStream(); pq.write_table(tabs[i % 4], sink, compression="zstd")
def w_mem_nodict(i):
    sink = pa.BufferOutputStream(); pq.write_table(tabs[i % 4], sink, compression="zstd", use_dictionary=False, write_statistics=False)
_T = None
def p_init():
    global _T
    _T = [make() for _ in range(4)]
def p_write(i):
    pq.write_table(_T[i % 4], f"/tmp/p{i}.parquet", compression="zstd")
def p_write_ipc(buf_i):
    buf, i = buf_i
    t = pa.ipc.open_stream(buf).read_all()
    pq.write_table(t, f"/tmp/q{i}.parquet", compression="zstd")
if __name__ == "__main__":
    N = 96
    for name, fn in (("file /tmp", w_file), ("file /dev/shm", w_shm), ("memory sink", w_mem), ("memory, no dict/stats", w_mem_nodict)):
        for nt in (1, 4, 12):
            with ThreadPoolExecutor(nt) as ex:
                t = time.perf_counter(); list(ex.map(fn, range(N))); dt = time.perf_counter() - t
            print(f"{name:24s} threads {nt:2d}: {dt / N * 1e3:6.2f} ms per file")
    print("arrow cpu_count", pa.cpu_count(), "io threads", pa.io_thread_count())
    for nproc in (4, 12):
        with ProcessPoolExecutor(nproc, mp_context=mp.get_context("spawn"), initializer=p_init) as ex:
            list(ex.map(p_write, range(nproc)))
            t = time.perf_counter(); list(ex.map(p_write, range(N))); dt = time.perf_counter() - t
            print(f"processes {nproc:2d} (own tables): {dt / N * 1e3:6.2f} ms per file")
            def ser(i):
                sink = pa.BufferOutputStream()
                with pa.ipc.new_stream(sink, tabs[i % 4].schema) as wr:
                    wr.write_table(tabs[i % 4])
                return sink.getvalue().to_pybytes(), i
            t = time.perf_counter(); payload = [ser(i) for i in range(N)]; ts = time.perf_counter() - t
            t = time.perf_counter(); list(ex.map(p_write_ipc, payload)); dt = time.perf_counter() - t
            print(f"processes {nproc:2d} (IPC payload): {dt / N * 1e3:6.2f} ms per file (+ {ts / N * 1e3:.2f} ms serialise in the parent)")
