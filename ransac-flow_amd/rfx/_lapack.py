"""Host LAPACK helper of the exact mode: binding of librfxhost.so (include/rfx_host_api.h).

The reference solves every 4-point DLT system with ``np.linalg.svd`` (utils/outil.py:84), i.e. with ``dgesdd`` of the LAPACK numpy
is linked to.  numpy's batched SVD is a serial, GIL-holding loop of one dgesdd per 8x9 system (~10 us each) and a lock-step round
of 64 pairs flags thousands of rank-deficient samples.  librfxhost.so (csrc/host_lapack.cpp) calls THE SAME routine -- resolved
through ``numpy.linalg._umath_linalg``'s own handle, so it is the symbol numpy's svd gufunc binds -- with numpy's argument set on
a pool of std::threads: per system the same LAPACK code on the same data, hence the same bits (tests/test_oracle.py pins that
against ``np.linalg.svd`` itself), without Python, pipes or the GIL.  Rounds 4-5 dealt the systems to ``python -c`` worker
processes instead (0.70x of the timed mode); those are gone.

``dlt_null_vectors`` below is the numpy restatement of utils/outil.py:68-87 the native solver is pinned on; the product path
never calls it.  There is no fallback: a missing librfxhost.so, or a numpy whose LAPACK cannot be resolved, raises."""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RFX_HOST_LIB") or os.path.normpath(os.path.join(_HERE, "..", "librfxhost.so"))

# name -> (restype, argtypes); mirrors include/rfx_host_api.h one to one
SIGNATURES = {
    "rfx_host_lapack_bind": (ctypes.c_int, [ctypes.c_char_p]),
    "rfx_host_lapack_symbol": (ctypes.c_char_p, []),
    "rfx_host_set_threads": (ctypes.c_int, [ctypes.c_int]),
    "rfx_host_dlt_null_vectors": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                 ctypes.c_void_p]),
}

_lib = None
_lock = threading.Lock()
_info = {}


def dlt_null_vectors(X, Y):
    """utils/outil.py:68-87 up to the float32 cast, in numpy: float32 products stored into a float64 8x9 system, full SVD, row 8 of
    Vh.  X, Y (k,4,3) float32 -> (k,9) float64.  The PIN of the native solver (tests), not a code path of the product."""
    k = X.shape[0]
    A = np.zeros((k, 8, 9))
    z, o = np.zeros(k), np.ones(k)
    for i in range(4):
        u, v, u_, v_ = Y[:, i, 0], Y[:, i, 1], X[:, i, 0], X[:, i, 1]
        A[:, 2 * i] = np.stack([z, z, z, -u, -v, -o, v_ * u, v_ * v, v_], axis=1)
        A[:, 2 * i + 1] = np.stack([u, v, o, z, z, z, -u_ * u, -u_ * v, -u_], axis=1)
    return np.linalg.svd(A)[2][:, 8]


def _ncpu():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def default_threads():
    """Solver threads of this process: RFX_LAPACK_THREADS, else the CPUs this process may run on divided by the ranks sharing
    the host (LOCAL_WORLD_SIZE / WORLD_SIZE under torch.distributed.run), at most 64."""
    env = int(os.environ.get("RFX_LAPACK_THREADS", "0"))
    if env > 0:
        return env
    world = int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or "1")
    return max(1, min(64, _ncpu() // max(1, world)))


def load():
    """Load librfxhost.so, attach prototypes, bind dgesdd through numpy's own linalg module.  Idempotent, thread-safe."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("librfxhost.so not found at %s -- build it with `make -C ransac-flow_amd/csrc` (g++; the exact RANSAC "
                               "mode has no other host solver)" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise RuntimeError("librfxhost.so is missing symbol %s (stale build?)" % name) from e
            fn.restype, fn.argtypes = res, args
        import numpy.linalg._umath_linalg as um          # the module whose svd gufunc the reference's np.linalg.svd runs
        bits = lib.rfx_host_lapack_bind(os.fsencode(um.__file__))
        if bits not in (32, 64):
            raise RuntimeError("no dgesdd symbol resolves through %s: the exact RANSAC mode needs the LAPACK numpy itself uses" % um.__file__)
        _info.update(symbol=lib.rfx_host_lapack_symbol().decode(), int_bits=bits, module=um.__file__,
                     threads=lib.rfx_host_set_threads(default_threads()))
        _lib = lib
    return _lib


def info():
    """dict(symbol, int_bits, module, threads) of the bound solver (loads it)."""
    load()
    return dict(_info)


def set_threads(n):
    _info["threads"] = load().rfx_host_set_threads(int(n))
    return _info["threads"]


def solve_rows(xy, out=None, dedupe=True, want_f64=False):
    """xy (k,16) float32 C-contiguous host array (rows as rfx_ransac_degenerate_gather writes them: 4 source points (x,y), 4 target
    points (x,y)) -> (k,9) float32 homographies = np.linalg.svd(A)[2][:, 8].astype(float32) of utils/outil.py:72-86; ``out``: a
    (>=k,9) float32 C-contiguous array to fill (pinned memory).  Releases the GIL for the duration (ctypes).
    Returns (H, n_solved[, hv float64])."""
    lib = load()
    if xy.dtype != np.float32 or not xy.flags.c_contiguous or xy.ndim != 2 or xy.shape[1] != 16:
        raise ValueError("xy must be a C-contiguous (k,16) float32 array")
    k = xy.shape[0]
    if out is None:
        out = np.empty((k, 9), dtype=np.float32)
    elif out.dtype != np.float32 or not out.flags.c_contiguous or out.shape[0] < k or out.shape[1:] != (9,):
        raise ValueError("out must be a C-contiguous (>=k,9) float32 array")
    hv = np.empty((k, 9), dtype=np.float64) if want_f64 else None
    ns = ctypes.c_int64(0)
    rc = lib.rfx_host_dlt_null_vectors(xy.ctypes.data, k, out.ctypes.data, hv.ctypes.data if hv is not None else None,
                                       1 if dedupe else 0, ctypes.byref(ns))
    if rc == -3:
        raise np.linalg.LinAlgError("SVD did not converge")          # what np.linalg.svd raises inside the reference
    if rc != 0:
        raise RuntimeError("rfx_host_dlt_null_vectors failed: %d" % rc)
    return (out[:k], ns.value, hv) if want_f64 else (out[:k], ns.value)


def null_vectors(X, Y):
    """X, Y (k,4,3) float32 source / target samples (outil.Homography's arguments) -> (k,9) float64 = row 8 of Vh of numpy's full
    SVD of every DLT system, in order -- through the native solver."""
    X, Y = np.asarray(X, dtype=np.float32), np.asarray(Y, dtype=np.float32)
    k = X.shape[0]
    xy = np.ascontiguousarray(np.concatenate([X[:, :, :2].reshape(k, 8), Y[:, :, :2].reshape(k, 8)], axis=1))
    return solve_rows(xy, want_f64=True)[2]


def start(n=None):
    """Load + bind the solver now (pipelines built in an exact mode call this so that the first round does not pay the dlopen)."""
    load()
    if n:
        set_threads(n)
    return _info["threads"]
