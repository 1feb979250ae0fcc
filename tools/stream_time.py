"""Out-of-core fit over a STREAMCSR file in row blocks (newStreamCSRDataset(f, cacheRows)): epoch wall time with the next
block loaded beside the current block's epoch (nfm_stream_prefetch_rows) and without (NIMFM_STREAM_PREFETCH=0).
usage: python tools/stream_time.py [n] [cacheRows]      (writes /tmp/nimfm_stream_x.bin, n x 64 entries, 16 B each)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
cache = int(sys.argv[2]) if len(sys.argv) > 2 else 250_000
d, m, k, B = 1_000_000, 64, 64, 8192
path = "/tmp/nimfm_stream_x.bin"
if len(sys.argv) > 3:  # the timed child
    import nimfm_amd as nf
    X, _ = nf.newStreamCSRDataset(path, cacheRows=cache)
    y = np.sign(np.random.default_rng(1).standard_normal(n))
    fm = nf.newFactorizationMachine("classification", nComponents=k, warmStart=True)
    fm.set_params(np.random.default_rng(2).standard_normal((1, k, d)) * 0.01, np.zeros(d), 0.0)
    opt = nf.newSGD(maxIter=1, loss="logistic", verbose=0, tol=0, mode="minibatch", batch=B, touchCap=16.0)
    opt.fit(X, y, fm)  # warm-up: page cache, plans of the blocks' shapes
    opt.maxIter = 3
    fm.warmStart = True
    t0 = time.perf_counter()
    opt.fit(X, y, fm)
    dt = (time.perf_counter() - t0) / 3
    print("prefetch %s: %.1f ms per epoch over %d blocks of %d rows = %.3g samples/s" % (
        os.environ.get("NIMFM_STREAM_PREFETCH", "1"), dt * 1e3, len(X.blocks()), cache, n / dt), flush=True)
    sys.exit(0)
rng = np.random.default_rng(3)
rec = np.dtype([("c", "<i8"), ("e", [("v", "<f8"), ("j", "<i8")], (m,))])
with open(path, "wb") as f:
    f.write(b"STREAMCSR")
    f.write(np.array([n, d, n * m], dtype="<i8").tobytes())
    f.write(np.array([1.0, -1.0], dtype="<f8").tobytes())
    step = 100_000
    for r0 in range(0, n, step):
        r = np.zeros(min(step, n - r0), dtype=rec)
        r["c"] = m
        # distinct ids per row: a random start + distinct strides inside disjoint ranges
        r["e"]["j"] = (rng.integers(0, d // m, size=(len(r), m)) + np.arange(m) * (d // m))
        r["e"]["v"] = rng.uniform(-1, 1, size=(len(r), m))
        f.write(r.tobytes())
print("wrote %s: %.2f GB" % (path, os.path.getsize(path) / 1e9), flush=True)
for pf in ("1", "0"):
    env = dict(os.environ, NIMFM_STREAM_PREFETCH=pf)
    subprocess.run([sys.executable, os.path.abspath(__file__), str(n), str(cache), "child"], env=env, check=True)
os.remove(path)
