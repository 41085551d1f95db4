"""Host LAPACK helper of the exact mode (ops.lapack_dlt): numpy's batched SVD is a serial, GIL-holding loop of one dgesdd per 8x9
system (~10 us each; Python threads give no speed-up, measured), and a lock-step round of 64 pairs flags ~6 000 rank-deficient
4-point samples: 70 ms of host time per round with the GPU idle.  ``null_vectors`` deals the systems to a few persistent WORKER
PROCESSES -- plain ``python -c`` children speaking a length-prefixed byte protocol over pipes (no multiprocessing: nothing
re-imports the caller's ``__main__``, nothing is forked from a process that holds a HIP context) -- each running the very same
``np.linalg.svd(A)[2][:, 8]`` (utils/outil.py:84-86) on its slice: per system the same LAPACK routine of the same numpy build on the
same data, hence the same bits (tests/test_oracle.py pins that), in 1/N of the time."""
import atexit
import os
import struct
import subprocess
import sys

import numpy as np

# utils/outil.py:68-87 up to the SVD: float32 products stored into a float64 8x9 system, then row 8 of Vh.  ONE source text, executed
# in this process (small batches) and in every worker: the same numpy expressions on the same float32 inputs -> the same bits.
_SOLVE_SRC = r"""
import numpy as np
def dlt_null_vectors(X, Y):
    k = X.shape[0]
    A = np.zeros((k, 8, 9))
    z, o = np.zeros(k), np.ones(k)
    for i in range(4):
        u, v, u_, v_ = Y[:, i, 0], Y[:, i, 1], X[:, i, 0], X[:, i, 1]
        A[:, 2 * i] = np.stack([z, z, z, -u, -v, -o, v_ * u, v_ * v, v_], axis=1)
        A[:, 2 * i + 1] = np.stack([u, v, o, z, z, z, -u_ * u, -u_ * v, -u_], axis=1)
    return np.linalg.svd(A)[2][:, 8]
"""
exec(_SOLVE_SRC)      # defines dlt_null_vectors here

_WORKER = _SOLVE_SRC + r"""
import sys, struct
rd, wr = sys.stdin.buffer, sys.stdout.buffer
while True:
    h = rd.read(8)
    if len(h) < 8:
        break
    k, = struct.unpack('<q', h)
    XY = np.frombuffer(rd.read(k * 96), dtype=np.float32).reshape(2, k, 4, 3)     # 96 bytes per system instead of a 576-byte matrix
    wr.write(np.ascontiguousarray(dlt_null_vectors(XY[0], XY[1])).tobytes())
    wr.flush()
"""

_workers = []


def _ncpu():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def start(n=None):
    """Start the worker processes (idempotent).  Called by pipelines built with degenerate="lapack" so that the first round
    does not pay the ~0.2 s interpreter start-up."""
    if _workers:
        return len(_workers)
    n = n or int(os.environ.get("RFX_LAPACK_WORKERS", "0")) or max(1, min(16, _ncpu() // 4))
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    for _ in range(n):
        _workers.append(subprocess.Popen([sys.executable, "-c", _WORKER], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env))
    atexit.register(stop)
    return n


def stop():
    while _workers:
        w = _workers.pop()
        try:
            w.stdin.close()
            w.wait(timeout=2)
        except Exception:  # noqa: BLE001 -- interpreter shutdown: best effort
            w.kill()


def null_vectors(X, Y, min_parallel=768):
    """X, Y (k,4,3) float32 source / target samples -> (k,9) float64: row 8 of Vh of numpy's full SVD of every DLT system, in order."""
    k = X.shape[0]
    X, Y = np.ascontiguousarray(X, dtype=np.float32), np.ascontiguousarray(Y, dtype=np.float32)
    if k < min_parallel or os.environ.get("RFX_LAPACK_WORKERS") == "1":
        return dlt_null_vectors(X, Y)                  # noqa: F821 -- defined by exec(_SOLVE_SRC)
    n = start()
    step = -(-k // n)
    jobs = []
    try:
        for w, i in zip(_workers, range(0, k, step)):
            m = min(step, k - i)
            w.stdin.write(struct.pack("<q", m))
            w.stdin.write(X[i:i + m].tobytes())
            w.stdin.write(Y[i:i + m].tobytes())
            w.stdin.flush()
            jobs.append((w, m))
        out = [np.frombuffer(w.stdout.read(m * 72), dtype=np.float64).reshape(-1, 9) for w, m in jobs]
        if any(o.shape[0] != m for o, (_, m) in zip(out, jobs)):
            raise RuntimeError("short read")
    except Exception:  # noqa: BLE001 -- a dead worker must not lose the round: drop the pool, solve in-process (same bits)
        stop()
        return dlt_null_vectors(X, Y)                  # noqa: F821
    return np.concatenate(out)
