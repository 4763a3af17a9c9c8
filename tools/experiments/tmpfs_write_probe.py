"""How fast can this host write new files to tmpfs from N threads (one os.writev of 2.2 MB per file -- the size of a 3000-frame
motion pickle)?  The ceiling of the dataset path's writer (bench.py dataset_path)."""
import os, time, threading, tempfile, shutil, numpy as np
d = tempfile.mkdtemp(dir="/dev/shm")
buf = memoryview(np.random.default_rng(0).random(279000)).cast("B")
N = 1024
def work(lo, hi, tag):
    for i in range(lo, hi):
        fd = os.open(f"{d}/{tag}_{i}", os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666); os.writev(fd, [buf]); os.close(fd)
for w in (1, 1, 2, 4, 8, 16):
    t = time.perf_counter()
    th = [threading.Thread(target=work, args=(k * N // w, (k + 1) * N // w, w)) for k in range(w)]
    [x.start() for x in th]; [x.join() for x in th]
    dt = time.perf_counter() - t
    print(w, f"{N * buf.nbytes / dt / 1e9:.2f} GB/s = {N * 3000 / dt:.3e} frames/s of 744-byte frames")
    for i in range(N): os.unlink(f"{d}/{w}_{i}")
shutil.rmtree(d)
print("cpus", len(os.sched_getaffinity(0)))
