"""How fast does this machine create and fill small files?  (The API leg writes two files per position: ~2 MB parquet + ~0.1 MB npz.)"""
import os, sys, time, tempfile, threading
from concurrent.futures import ThreadPoolExecutor

payload = os.urandom(2 << 20)


def run(root, threads, n=192):
    d = tempfile.mkdtemp(dir=root)

    def one(k):
        fd = os.open(f"{d}/{k}.bin", os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666)
        os.write(fd, payload)
        os.close(fd)

    with ThreadPoolExecutor(threads) as ex:
        t0 = time.perf_counter()
        list(ex.map(one, range(n)))
        dt = time.perf_counter() - t0
    for k in range(n):
        os.unlink(f"{d}/{k}.bin")
    os.rmdir(d)
    return 1e3 * dt / n


for root in sys.argv[1:] or [tempfile.gettempdir(), "/dev/shm"]:
    if os.path.isdir(root) and os.access(root, os.W_OK):
        print(root, {t: round(run(root, t), 3) for t in (1, 4, 12)}, "ms per 2 MB file (wall) at 1 / 4 / 12 threads")
