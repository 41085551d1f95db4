"""Drop-in for the reference's ``utils/outil.py`` (same names, argument meaning, return types and sentinels)
with every device op routed to librfx (HIP, gfx950).  Import it by bare name like the reference's scripts do
(``sys.path`` entry = this directory, or through ``run_reference_script.py``).

Scope: resizeImg, getWHTensor, getWHTensor_Int, mutualMatching, Homography, Prediction,
ScoreRANSAC, RANSAC.  ``Affine`` / ``Hough`` / ``Translation`` / ``SaliencyCoef`` are dead code in the reference
(RANSAC is hard-wired to 4 points, utils/outil.py:122-130) and raise NotImplementedError here.
"""
import numpy as np
import PIL.Image as Image
import torch

import os

from rfx import ops

# Rank-deficient 4-point samples (three matched points collinear in both images): the reference's value there is whatever the
# HOST's LAPACK returns (utils/outil.py:84), so the drop-in re-solves exactly those hypotheses with numpy's LAPACK
# (ops.lapack_dlt) -- every function below is host-synchronous in the reference anyway.  RFX_DEGENERATE=device keeps the
# device's own null vector (no extra sync).
_DEGENERATE = os.environ.get("RFX_DEGENERATE", "lapack")

# How mutualMatching's scores are summed (rfx_api.h, ABI 8: an argument of every call).  The drop-in's contract is "what a CPU run of
# the reference computes on THIS host", and the reference's score is the host's torch.mm (utils/outil.py:34), so the module resolves
# the host sgemm's K blocking ONCE, when it is imported (ops.resolve_score_chunk("host"): < 2 s of host arithmetic, never inside a
# call); RFX_SCORE_CHUNK=<products>|default overrides.  Both values are module attributes so that a script's log can state them.
SCORE_CHUNK, SCORE_CHUNK_SOURCE = ops.resolve_score_chunk(os.environ.get("RFX_SCORE_CHUNK") or "host")


def resizeImg(I, strideNet, minSize=400, mode=Image.LANCZOS):
    """utils/outil.py:6-19: scale so the smaller side is ``minSize``, round both sides to the net stride."""
    w, h = I.size
    ratio = min(w / minSize, h / minSize)
    w, h = w / ratio, h / ratio
    return I.resize((round(w / strideNet) * strideNet, round(h / strideNet) * strideNet), resample=mode)


def getWHTensor(feat):
    """utils/outil.py:21-24.  Returns (W, H): W = normalised *row* coordinate, H = *column* coordinate of
    every cell of ``feat`` (n,c,rows,cols), cell centres in [-1, 1]."""
    from rfx.pipeline import cell_coords_cached
    # the CPU reference's values bit for bit (true division; ATen's device kernel multiplies by a rounded reciprocal)
    return cell_coords_cached(feat.size(2), feat.size(3), feat.device)


def getWHTensor_Int(feat):
    """utils/outil.py:26-29: integer (row, col) of every cell."""
    rows, cols = feat.size(2), feat.size(3)
    W = torch.arange(rows, device=feat.device).view(-1, 1).expand(rows, cols).reshape(-1)
    Hh = torch.arange(cols, device=feat.device).view(1, -1).expand(rows, cols).reshape(-1)
    return W, Hh


def mutualMatching(featA, featB):
    """utils/outil.py:32-45 -> (index1, index2) int64 device tensors, ascending index1.
    One fused MFMA correlation + arg-max kernel chain; the nA x nB score matrix is never materialised."""
    return ops.mutual_nn(featA, featB, score_chunk=SCORE_CHUNK)


def Homography(X, Y):
    """utils/outil.py:68-87: X, Y (N,4,3) source / target samples -> H21 (N,3,3) float32 on X's device.
    Float64 Householder DLT with LAPACK dgesdd's sign on the device; the ~1 % of samples whose 8x9 system is rank deficient are
    re-solved by the host's LAPACK like the reference does for every sample (one sync; RFX_DEGENERATE=device: none)."""
    return ops.dlt4_homography(X, Y, degenerate=_DEGENERATE)


def Prediction(X, Y, H21):
    """utils/outil.py:97-100: X, Y (1,n,3) (or (n,3)), H21 (N,3,3) -> reprojection error (N,n)."""
    Xm = X[0] if X.dim() == 3 else X
    Ym = Y[0] if Y.dim() == 3 else Y
    return ops.prediction(Xm, Ym, H21)


def ScoreRANSAC(match1, match2, tolerance, samples, Transform):
    """utils/outil.py:102-113 -> (H21 (N,3,3), inlier counts (N,) int64 gated by det(H21) > 1e-6)."""
    _require_homography(Transform)
    return ops.score_hypotheses(match1, match2, samples, tolerance, degenerate=_DEGENERATE)


def RANSAC(nbIter, match1, match2, tolerance, nbPoint, Transform, samples=None):
    """utils/outil.py:117-164.  Returns (bestParams (3,3) float32 ndarray, bestInlier 0-d int64 ndarray,
    isInlier bool ndarray (n,), match2[isInlier] ndarray) or ``(None, 0, [], [])`` when a full chunk of 100
    hypotheses scores 0 (:145-146); raises TypeError like the reference when nothing ever beat 0 (:162).

    ``samples`` (optional, (nbIter,4) int64): the index draw.  Default: ``torch.randint(nbMatch, (nbIter, 4))``
    on the *CPU* generator -- the draw a CPU run of the reference makes (utils/outil.py:120), so a fixed
    ``torch.manual_seed`` reproduces the reference's hypotheses and, with them, its inlier indices bit for bit.
    """
    _require_homography(Transform)
    if nbPoint != 4:
        raise NotImplementedError("RANSAC is defined for nbPoint=4 only (the reference hard-codes 4, utils/outil.py:122-130)")
    nbMatch = len(match1)
    if samples is None:
        samples = torch.randint(nbMatch, (nbIter, nbPoint))
    bestH, inl, res = ops.ransac_h4(match1, match2, samples.to(match1.device), tolerance, degenerate=_DEGENERATE)
    status, cnt, _, _ = res.cpu().tolist()
    if status == 1:
        return None, 0, [], []
    if status == 2:
        raise TypeError("'NoneType' object is not subscriptable")
    inl_np = inl.cpu().numpy()
    return bestH.cpu().numpy(), np.asarray(cnt, dtype=np.int64), inl_np, match2[inl].cpu().numpy()


def _require_homography(Transform):
    if Transform is not Homography and getattr(Transform, "__name__", "") != "Homography":
        raise NotImplementedError("only the Homography transform is on the hot path")


def _out_of_scope(name):
    def f(*a, **k):
        raise NotImplementedError("outil.%s is dead code in the reference (unreachable from RANSAC) and is not provided" % name)
    f.__name__ = name
    return f


Affine = _out_of_scope("Affine")
Hough = _out_of_scope("Hough")
Translation = _out_of_scope("Translation")
SaliencyCoef = _out_of_scope("SaliencyCoef")
