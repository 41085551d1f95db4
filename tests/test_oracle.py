"""CPU tests: the oracle restatement (oracle/restate.py) against the committed golden vectors that were
produced by the real reference (tests/golden/make_golden.py), the host build of the DLT header against
numpy's LAPACK SVD, and the C-ABI surface of librfx.so.  No GPU needed."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import restate
from rfx import weights, synth, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


# ---------------------------------------------------------------- RANSAC


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_ransac_restatement_matches_reference_golden(seed):
    g = gold("ransac.npz")
    m1, m2 = torch.from_numpy(g["m1_%d" % seed]), torch.from_numpy(g["m2_%d" % seed])
    samples = torch.from_numpy(g["samples_%d" % seed])
    Hb, cnt, inl, m2in = restate.ransac(m1, m2, 0.05, samples)
    assert int(cnt) == int(g["count_%d" % seed])
    assert np.array_equal(inl, g["inlier_%d" % seed])            # inlier indices: bit exact
    assert np.abs(Hb - g["H_%d" % seed]).max() <= 1e-6
    assert Hb.dtype == np.float32 and inl.dtype == np.bool_
    assert np.array_equal(m2in, m2.numpy()[inl])


def test_score_ransac_restatement_matches_golden():
    g = gold("ransac.npz")
    for seed in range(4):
        m1, m2 = torch.from_numpy(g["m1_%d" % seed]), torch.from_numpy(g["m2_%d" % seed])
        uniq = restate.filter_samples(torch.from_numpy(g["samples_%d" % seed]))[:300]
        H21, cnt = restate.score_ransac(m1, m2, 0.05, uniq)
        assert np.abs(H21.numpy() - g["score_H_%d" % seed]).max() <= 1e-6
        assert np.array_equal(cnt.numpy(), g["score_counts_%d" % seed])


def test_ransac_abort_and_sentinels():
    g = gold("ransac.npz")
    m1, m2 = torch.from_numpy(g["abort_m1"]), torch.from_numpy(g["abort_m2"])
    res = restate.ransac(m1, m2, -1.0, torch.from_numpy(g["abort_samples"]))
    assert res[0] is None and res[1] == 0 and res[2] == [] and res[3] == []
    # fewer than 100 surviving hypotheses and nothing beats 0 -> the reference's TypeError (utils/outil.py:162)
    with pytest.raises(TypeError):
        restate.ransac(m1, m2, -1.0, torch.from_numpy(g["abort_samples"])[:50])


def test_duplicate_filter_keeps_order():
    s = torch.tensor([[0, 1, 2, 3], [1, 1, 2, 3], [4, 5, 6, 4], [7, 8, 9, 10], [3, 2, 2, 0]])
    assert restate.filter_samples(s).tolist() == [[0, 1, 2, 3], [7, 8, 9, 10]]


# ---------------------------------------------------------------- DLT null vector


def _host_dlt_lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("hostdlt")
    so = str(d / "libhostdlt.so")
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "host", "host_dlt.cpp")])
    return ctypes.CDLL(so)


def test_householder_dlt_is_lapack_sign_exact(tmp_path_factory):
    """The kernel's DLT header (compiled for the host) and the numpy restatement of SURVEY A.1 agree with
    numpy.linalg.svd's last row of Vh (the reference's utils/outil.py:84-86) incl. its sign."""
    lib = _host_dlt_lib(tmp_path_factory)
    g = gold("ransac.npz")
    m1, m2 = torch.from_numpy(g["m1_0"]), torch.from_numpy(g["m2_0"])
    torch.manual_seed(9)
    samples = restate.filter_samples(torch.randint(len(m1), (6000, 4)))
    X, Y = m1[samples].contiguous().numpy(), m2[samples].contiguous().numpy()
    N = len(X)
    h = np.zeros((N, 9))
    Hf = np.zeros((N, 9), dtype=np.float32)
    vp = ctypes.c_void_p
    lib.rfx_host_dlt4(X.ctypes.data_as(vp), Y.ctypes.data_as(vp), N, h.ctypes.data_as(vp), Hf.ctypes.data_as(vp))
    A = restate.dlt_matrix(X, Y)
    _, s, vh = np.linalg.svd(A)
    ref = vh[:, 8]
    good = s[:, 7] > 1e-9          # rank-8 systems (collinear samples are numerically rank deficient)
    assert good.mean() > 0.99
    assert ((h * ref).sum(1)[good] > 0).all()                       # sign: 100 %
    assert np.abs(Hf[good] - ref[good].astype(np.float32)).max() <= 1.2e-7
    py = restate.householder_nullvec(A[:64])
    wc = s[:64, 7] > 1e-3          # well conditioned: the two implementations agree to rounding
    assert np.abs(py - h[:64])[wc].max() < 1e-10
    # the float32 H of restate.homography_svd is what the reference returns
    Href = restate.homography_svd(torch.from_numpy(X[:64]), torch.from_numpy(Y[:64])).numpy().reshape(-1, 9)
    assert np.abs(Href - Hf[:64]).max() <= 1.2e-7


def test_rank_flag_of_the_dlt_separates_degenerate_samples_from_the_rest(tmp_path_factory):
    """dlt.h flags a 4-point sample whose 8x9 system is numerically rank deficient (one Sturm count on the bidiagonal; the flagged
    hypotheses are the ones ops.ransac_h4_batched(degenerate="lapack") hands to the host's LAPACK).  Against numpy's singular
    values on lattice matches under a near-translation -- where three matched points collinear in BOTH images are common: every
    system with sigma_8 / sigma_1 < 1e-10 is flagged, none with sigma_8 / sigma_1 > 1e-6 is, and on the unflagged ones the
    device restatement agrees with LAPACK to float32 rounding INCLUDING the sign (the part that cannot be pinned is exactly the
    flagged part)."""
    lib = _host_dlt_lib(tmp_path_factory)
    W, Hh = restate.get_wh(30, 40)
    cells = torch.stack((Hh, W, torch.ones_like(W)), 1)
    g = torch.Generator().manual_seed(3)
    pick = torch.randperm(len(cells), generator=g)[:400]
    m2 = cells[pick]
    m1 = m2.clone()
    m1[:, 0] += 2 / 40.0                                       # a shift by one lattice column: collinear stays collinear
    m1[::7] = cells[torch.randperm(len(cells), generator=g)[:len(m1[::7])]]
    samples = restate.filter_samples(torch.randint(len(m1), (20000, 4), generator=g))
    X, Y = m1[samples].contiguous().numpy(), m2[samples].contiguous().numpy()
    N = len(X)
    vp = ctypes.c_void_p
    flag = np.zeros(N, dtype=np.uint8)
    lib.rfx_host_dlt4_rank(X.ctypes.data_as(vp), Y.ctypes.data_as(vp), N, flag.ctypes.data_as(vp))
    _, s, vh = np.linalg.svd(restate.dlt_matrix(X, Y))
    ratio = s[:, 7] / s[:, 0]
    assert (ratio < 1e-10).sum() > 50                          # the case exists in quantity on a lattice
    assert flag[ratio < 1e-10].all() and not flag[ratio > 1e-6].any()
    assert 0.001 < flag.mean() < 0.2
    h = np.zeros((N, 9))
    Hf = np.zeros((N, 9), dtype=np.float32)
    lib.rfx_host_dlt4(X.ctypes.data_as(vp), Y.ctypes.data_as(vp), N, h.ctypes.data_as(vp), Hf.ctypes.data_as(vp))
    ok = flag == 0
    assert np.abs(Hf[ok] - vh[ok, 8].astype(np.float32)).max() <= 2.4e-7
    assert np.abs(Hf[~ok] - vh[~ok, 8].astype(np.float32)).max() > 1e-3      # ... and the flagged ones really do differ


def det_gate_cases(N, seed=0, spread=0.1):
    """N 4-point correspondences whose unit-norm DLT homography has |det| within +-``spread`` of the 1e-6 gate of
    utils/outil.py:108,113: H0 = U diag(1, 1, d) V^T (random orthogonal U, V) maps four spread-out target points onto a
    nearly collinear source quadruple.  Returns X (source), Y (target), each (N,4,3) float32; degenerate draws are dropped."""
    rng = np.random.RandomState(seed)
    base = np.array([[-0.8, -0.7], [0.75, -0.6], [0.7, 0.8], [-0.65, 0.72]])
    Xs, Ys = [], []
    while len(Xs) < N:
        U, _ = np.linalg.qr(rng.randn(3, 3))
        V, _ = np.linalg.qr(rng.randn(3, 3))
        d = 1e-6 * (1 + spread * (2 * rng.rand() - 1)) * 2 ** 1.5          # det(H0 / |H0|_F) ~ d / (2 + d^2)^1.5
        H0 = U @ np.diag([1, 1, d]) @ V.T
        Yh = np.concatenate([base + 0.1 * rng.randn(4, 2), np.ones((4, 1))], 1)
        Xh = Yh @ H0.T
        if np.abs(Xh[:, 2]).min() < 0.2:
            continue
        Xs.append(np.concatenate([Xh[:, :2] / Xh[:, 2:], np.ones((4, 1))], 1))
        Ys.append(Yh)
    return np.stack(Xs).astype(np.float32), np.stack(Ys).astype(np.float32)


def test_det3_lu_is_torch_det(tmp_path_factory):
    """The kernel's determinant (dlt.h: rfx_det3_lu_f32, compiled for the host) IS torch.det's float32 value -- the LU of H^T
    with MKL's operation order -- on random, near-singular and gate-straddling 3x3 matrices: bit for bit, so the
    ``det(H) > 1e-6`` gate of utils/outil.py:113 decides identically.  (A float32 cofactor expansion, the round-1/2 kernel,
    flips 1-2 % of the decisions within +-10 % of the gate: counted below.)"""
    lib = _host_dlt_lib(tmp_path_factory)
    rng = np.random.RandomState(3)
    mats = []
    for _ in range(4000):
        U, _ = np.linalg.qr(rng.randn(3, 3))
        V, _ = np.linalg.qr(rng.randn(3, 3))
        M = U @ np.diag([1, 1, 10 ** rng.uniform(-8, 0)]) @ V.T
        mats.append(M / np.linalg.norm(M) * rng.choice([-1, 1]))
    X, Y = det_gate_cases(3000, seed=1)
    Hg = restate.homography_svd(torch.from_numpy(X), torch.from_numpy(Y)).numpy()
    mats = np.concatenate([np.stack(mats).astype(np.float32), Hg, rng.randn(500, 3, 3).astype(np.float32)])
    out = np.zeros(len(mats), dtype=np.float32)
    vp = ctypes.c_void_p
    flat = np.ascontiguousarray(mats.reshape(-1, 9))
    lib.rfx_host_det3(flat.ctypes.data_as(vp), len(mats), out.ctypes.data_as(vp))
    ref = torch.det(torch.from_numpy(mats)).numpy()
    assert np.array_equal(out, ref), "bit-equal fraction %.4f" % np.mean(out == ref)
    # the gate-straddling family really straddles, and the cofactor expansion would have flipped some of its decisions
    d64 = np.linalg.det(Hg.astype(np.float64))
    band = np.abs(np.abs(d64) - 1e-6) < 1e-7
    assert band.sum() > 1000
    h = Hg.reshape(-1, 9)
    f = np.float32
    c0, c1, c2 = f(f(h[:, 4] * h[:, 8]) - f(h[:, 5] * h[:, 7])), f(f(h[:, 3] * h[:, 8]) - f(h[:, 5] * h[:, 6])), f(f(h[:, 3] * h[:, 7]) - f(h[:, 4] * h[:, 6]))
    cof = f(f(f(h[:, 0] * c0) - f(h[:, 1] * c1)) + f(h[:, 2] * c2))
    lo = len(mats) - 500 - len(Hg)
    flipped = int(((cof > 1e-6) != (ref[lo:lo + len(Hg)] > 1e-6)).sum())
    print("gate decisions a float32 cofactor expansion would flip: %d of %d within +-10 %% of the gate" % (flipped, int(band.sum())))


def test_householder_dlt_on_degenerate_samples(tmp_path_factory):
    """Fuzz of the 4-point DLT on the samples the coarse grid makes likely: three or four points of one grid
    row or column (collinear), near-collinear points, and two sources sent to one target.  Where the 8x9
    system keeps rank 8 the null vector must agree with numpy.linalg.svd (utils/outil.py:84-86) including
    its sign; where it does not (the null space has dimension >= 2, LAPACK's pick is not a function of
    the geometry) the header must still return a finite unit null vector, which is all the det(H)/inlier
    gates downstream (utils/outil.py:108-113) need to reject it."""
    lib = _host_dlt_lib(tmp_path_factory)
    rng = np.random.default_rng(17)
    rows, cols = 30, 40
    W, Hh = restate.get_wh(rows, cols)
    grid = torch.stack((Hh, W, torch.ones_like(W)), 1).numpy()          # (x, y, 1) as the match lists hold
    N = 4000
    idx = np.zeros((N, 4), dtype=np.int64)
    kind = rng.integers(0, 4, N)
    for k in range(N):
        r, c = rng.integers(rows), rng.integers(cols)
        if kind[k] == 0:        # four points of one grid row
            idx[k] = r * cols + rng.choice(cols, 4, replace=False)
        elif kind[k] == 1:      # three points of one grid column + one free point
            idx[k, :3] = rng.choice(rows, 3, replace=False) * cols + c
            idx[k, 3] = rng.integers(rows * cols)
        elif kind[k] == 2:      # four points of one diagonal
            t = rng.choice(min(rows, cols), 4, replace=False)
            idx[k] = t * cols + t
        else:                   # a generic sample
            idx[k] = rng.choice(rows * cols, 4, replace=False)
    X = grid[idx].astype(np.float32)
    # targets: a mild homography of the sources plus grid-sized jitter; kind 1 also collapses two targets
    Y = X.copy()
    Y[..., :2] = X[..., :2] * 0.9 + 0.05 + rng.normal(0, 0.02, (N, 4, 2)).astype(np.float32)
    Y[kind == 1, 1] = Y[kind == 1, 0]
    X, Y = np.ascontiguousarray(X), np.ascontiguousarray(Y)
    h = np.zeros((N, 9))
    Hf = np.zeros((N, 9), dtype=np.float32)
    vp = ctypes.c_void_p
    lib.rfx_host_dlt4(X.ctypes.data_as(vp), Y.ctypes.data_as(vp), N, h.ctypes.data_as(vp), Hf.ctypes.data_as(vp))
    A = restate.dlt_matrix(X, Y)
    _, s, vh = np.linalg.svd(A)
    ref = vh[:, 8]
    assert np.isfinite(h).all()
    assert np.abs(np.linalg.norm(h, axis=1) - 1).max() < 1e-12
    # always a null vector of A (8 equations, 9 unknowns: one exists whatever the rank)
    resid = np.abs(np.einsum("nij,nj->ni", A, h)).max(1)
    assert (resid <= 1e-12 * np.maximum(s[:, 0], 1)).all()
    rank8 = s[:, 7] > 1e-9 * s[:, 0]
    assert rank8[kind == 3].mean() > 0.99 and (~rank8).sum() > 500      # both regimes are exercised
    assert ((h * ref).sum(1)[rank8] > 0).all()
    # agreement degrades with the gap to the 8th singular value and not faster
    gap = s[:, 7] / s[:, 0]
    assert (np.abs(h - ref).max(1)[rank8] <= 1e-15 / gap[rank8] + 1e-12).all()


def test_identity_matches_give_identity_homography():
    """KAT: identical match lists -> H proportional to I, every match an inlier."""
    W, Hh = restate.get_wh(12, 16)
    m = torch.stack((Hh, W, torch.ones_like(W)), 1)
    torch.manual_seed(3)
    samples = torch.randint(len(m), (300, 4))
    Hb, cnt, inl, _ = restate.ransac(m, m, 0.05, samples)
    assert inl.all() and int(cnt) == len(m)
    Hn = Hb / Hb[2, 2]
    assert np.abs(Hn - np.eye(3)).max() < 1e-4


# ---------------------------------------------------------------- mutual matching


def test_mutual_matching_matches_golden():
    g = gold("mutual.npz")
    i1, i2 = restate.mutual_matching(torch.from_numpy(g["A"]), torch.from_numpy(g["B"]))
    assert np.array_equal(i1.numpy(), g["index1"]) and np.array_equal(i2.numpy(), g["index2"])
    assert (np.diff(g["index1"]) > 0).all()
    assert not np.isin(g["index2"], np.arange(5, 20)).any()   # masked (all-zero) columns never match


# ---------------------------------------------------------------- networks


def test_trunk_and_fine_nets_match_golden():
    g = gold("nets.npz")
    with torch.no_grad():
        t = restate.resnet50_trunk(weights.resnet50_trunk_sd(seed=31, randomize_bn=True), torch.from_numpy(g["trunk_in"]))
    scale = np.abs(g["trunk_out"]).max()
    assert np.abs(t.numpy() - g["trunk_out"]).max() <= 2e-5 * scale
    fe = weights.feature_extractor_sd(seed=32, randomize_bn=True)
    nf = weights.net_flow_coarse_sd(seed=33, randomize_bn=True)
    nm = weights.net_matchability_sd(seed=34, randomize_bn=True, last_std=0.02)
    with torch.no_grad():
        fa = F.normalize(restate.feature_extractor(fe, torch.from_numpy(g["fine_xa"])))
        fb = F.normalize(restate.feature_extractor(fe, torch.from_numpy(g["fine_xb"])))
        assert np.abs(fa.numpy() - g["fine_fa"]).max() < 2e-6
        c12 = restate.corr_neigh(fa, fb)
        assert np.abs(c12.numpy() - g["fine_corr"]).max() < 2e-6
        gc = torch.from_numpy(g["fine_corr"])
        assert np.abs(restate.net_flow_coarse(nf, gc, False).numpy() - g["fine_flow"]).max() < 1e-6
        assert np.abs(restate.net_flow_coarse(nf, gc, True).numpy() - g["fine_flow8"]).max() < 1e-6
        assert np.abs(restate.net_matchability(nm, gc, False).numpy() - g["fine_match"]).max() < 1e-6
        grid = restate.identity_grid(48, 64)
        fg, fc = restate.pred_flow_coarse(nf, gc, grid, True)
        assert np.abs(fg.numpy() - g["fine_flowGrad"]).max() < 1e-6
        assert np.abs(fc.numpy() - g["fine_flowCoarse"]).max() < 1e-6


def test_corr_neigh_centre_tap_is_one_for_normalised_features():
    f = F.normalize(torch.randn(1, 32, 5, 6), dim=1)
    c = restate.corr_neigh(f, f)
    assert torch.allclose(c[:, 24], torch.ones(1, 5, 6), atol=1e-6)
    assert c[0, 0, 0, 0] == 0  # top-left tap of the top-left pixel reads the zero padding


def test_flow_head_uniform_logits_give_zero_flow():
    sd = weights.net_flow_coarse_sd(seed=2)
    sd = {k: (torch.zeros_like(v) if k == "conv4.weight" else v) for k, v in sd.items()}
    flow = restate.net_flow_coarse(sd, torch.randn(1, 49, 6, 8), False)
    assert flow.abs().max() < 1e-7


def test_warp_and_sample_match_golden():
    g = gold("nets.npz")
    wg = restate.warp_grid(torch.from_numpy(g["warp_H"]), 48, 64)
    assert np.abs(wg.numpy() - g["warp_grid"]).max() < 1e-6
    out = restate.grid_sample(torch.from_numpy(g["fine_xa"]), wg)
    assert np.abs(out.numpy() - g["warp_sample"]).max() < 1e-6
    ident = restate.grid_sample(torch.from_numpy(g["fine_xa"]), _ac_false_identity(48, 64))
    assert np.abs(ident.numpy() - g["fine_xa"]).max() < 1e-6


def _ac_false_identity(h, w):
    xs = (torch.arange(w, dtype=torch.float32) + 0.5) / w * 2 - 1
    ys = (torch.arange(h, dtype=torch.float32) + 0.5) / h * 2 - 1
    return torch.stack((xs.view(1, w).expand(h, w), ys.view(h, 1).expand(h, w)), -1)[None]


# ---------------------------------------------------------------- config 1 end to end (CPU reference path)


def test_config1_coarse_to_fine_matches_reference_golden():
    g = gold("config1.npz")
    I1, I2 = synth.make_pair(240, 320, seed=0)
    ca = restate.CoarseAlignOracle(weights.resnet50_trunk_sd(seed=0), 7, 100, 0.05, 320, 1.2, variant="A")
    assert np.allclose(ca.scaleList, g["scaleList"])
    ca.setSource(I1)
    ca.setTarget(I2)
    assert ca.featsMultiScale.shape[1] == 2107 and ca.featt.shape[2:] == (15, 20)
    torch.manual_seed(123)
    r = ca.getCoarse(np.zeros((240, 320)))
    assert np.array_equal(r["index1"], g["index1"]) and np.array_equal(r["index2"], g["index2"])
    assert np.abs(r["H"] - g["H"]).max() <= 1e-6
    assert np.array_equal(r["inlierMask"], g["inlierMask"])
    nets = dict(feat=weights.feature_extractor_sd(seed=1), flow=weights.net_flow_coarse_sd(seed=2))
    with torch.no_grad():
        flowCoarse = restate.warp_grid(torch.from_numpy(r["H"])[None], 240, 320)
        st = restate.fine_step_quickstart(nets, ca.IsTensor, ca.ItTensor, flowCoarse)
    assert np.abs(st["flowDown"].numpy() - g["flowDown"]).max() < 1e-5
    assert np.abs(st["flow12"][:, ::8, ::8].numpy() - g["flow12_sub"]).max() < 1e-4


# ---------------------------------------------------------------- sky segmentation


def test_seg_restatement_matches_the_reference_golden():
    """oracle/restate.py::seg_scores / seg_get_sky (SegNet.getSky of segNet/segEval.py:23-43 over segNet/segModel.py, restated) against
    tests/golden/seg.npz -- the REFERENCE's own output (make_golden.py::gen_seg): class map, averaged class probabilities, both masks.
    Equal to the last bit on the authoring machine; another CPU's convolution kernels round differently, hence the small tolerance."""
    g = np.load(os.path.join(GOLD, "seg.npz"))
    enc, dec = weights.seg_encoder_sd(4, randomize_bn=True), weights.seg_decoder_sd(5, randomize_bn=True, logit_std=0.003)
    I1, _ = synth.make_pair(96, 128, seed=3)
    assert [list(s[::-1]) for s in restate.seg_test_sizes(128, 96)] == g["sizes"].tolist()
    sc = restate.seg_scores(enc, dec, I1)
    assert float((sc[0, :, ::8, ::8] - torch.from_numpy(g["scores_sub"])).abs().max()) < 1e-4
    pred = sc.max(dim=1)[1][0].numpy()
    assert (pred != g["pred"]).mean() < 1e-3
    seg_id = int(g["seg_id"])
    same = pred == g["pred"]
    assert np.array_equal(restate.seg_get_sky(enc, dec, I1, seg_id, True)[same], g["mask_fg"].astype(np.float32)[same])
    assert np.array_equal(restate.seg_get_sky(enc, dec, I1, seg_id, False)[same], g["mask_bg"].astype(np.float32)[same])


@pytest.mark.reference
def test_seg_restatement_matches_live_reference(tmp_path):
    """The same restatement against the live reference (segNet/segEval.py's SegNet, imported by oracle/ref_loader.load_seg under the
    collections.abc aliases its vendored Synchronized-BatchNorm package needs) on a fresh image and fresh weights: bit-equal scores."""
    import ref_loader
    S = ref_loader.load_seg()
    enc, dec = weights.seg_encoder_sd(14), weights.seg_decoder_sd(15, logit_std=0.002)
    img, _ = synth.make_pair(72, 104, seed=9)
    pe, pd_, pi = str(tmp_path / "e.pth"), str(tmp_path / "d.pth"), str(tmp_path / "i.png")
    torch.save(enc, pe)
    torch.save(dec, pd_)
    img.save(pi)
    sn = ref_loader.quiet(S["segEval"].SegNet, pe, pd_, 2, False)
    with torch.no_grad():
        IT = sn.dataset_test.getImg(pi)
        seg_size = (IT["img_ori"].shape[0], IT["img_ori"].shape[1])
        sc = torch.zeros(1, 150, *seg_size)
        for x in IT["img_data"]:
            sc = sc + sn.net(x, segSize=seg_size) / 5
    assert [tuple(x.shape[-2:])[::-1] for x in IT["img_data"]] == restate.seg_test_sizes(104, 72)
    assert torch.equal(restate.seg_scores(enc, dec, img), sc)
    for fg in (True, False):
        sn.segFg = fg
        assert np.array_equal(sn.getSky(pi), restate.seg_get_sky(enc, dec, img, 2, fg))


# ---------------------------------------------------------------- C ABI surface


def test_library_exports_every_declared_symbol():
    from rfx import _lapack
    hdr = open(os.path.join(ROOT, "include", "rfx_api.h")).read()
    declared = set(re.findall(r"\b(rfx_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    if not os.path.exists(_lib.LIB_PATH) or not os.path.exists(_lapack.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    # the host half of the exact mode: include/rfx_host_api.h <-> librfxhost.so <-> rfx/_lapack.py
    hh = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "rfx_host_api.h")).read(), flags=re.S)
    host_declared = set(re.findall(r"\b(rfx_host_[a-z0-9_]+)\s*\(", hh))
    assert host_declared == set(_lapack.SIGNATURES), (host_declared ^ set(_lapack.SIGNATURES))
    hl = _lapack.load()
    for name in host_declared:
        assert hasattr(hl, name), name
    assert lib.rfx_version().decode().startswith("rfx ")
    assert lib.rfx_mutual_nn_ws_bytes(8531, 1200) > 0 and lib.rfx_ransac_ws_bytes(900, 1000) > 0


def test_ops_refuse_cpu_tensors():
    from rfx import ops
    with pytest.raises(RuntimeError):
        ops.corr_neigh(torch.zeros(1, 8, 4, 4), torch.zeros(1, 8, 4, 4))
    with pytest.raises(RuntimeError):
        ops.ransac_h4(torch.zeros(8, 3), torch.zeros(8, 3), torch.zeros(4, 4, dtype=torch.int64), 0.05)


# ---------------------------------------------------------------- offline flow assembly

def _assembly_case():
    g = gold("assemble.npz")
    n, hd, wd = int(g["n"]), int(g["hd"]), int(g["wd"])
    flowDown, flowd2, param, md = synth.assembly_arrays(int(g["seed"]), n, hd, wd)
    t = torch.from_numpy
    return g, (t(flowDown), t(flowd2), t(param), t(md)), (hd, wd)


def test_assemble_restatement_matches_reference_golden():
    """The goldens are outputs of the reference's own getFlow_all / getFlow functions (make_golden.gen_assemble)."""
    g, (fd, fd2, pm, md), (hd, wd) = _assembly_case()
    for tag in "abc":
        th, multiH, oh, ow = g["hpatch_%s_cfg" % tag]
        fg, _, _ = restate.assemble_flow(fd, pm, md, (int(oh), int(ow)), float(th), bool(multiH), False)
        assert np.array_equal(fg.numpy(), g["hpatch_%s" % tag]), tag
        th, multiH = g["corr_%s_cfg" % tag]
        fg, mg, _ = restate.assemble_flow(fd, pm, md, (hd * 8, wd * 8), float(th), bool(multiH), True)
        assert np.array_equal(fg.numpy(), g["corr_%s_flow" % tag]), tag
        assert np.array_equal(mg.numpy()[..., None], g["corr_%s_match" % tag]), tag
    for tag in "abcd":
        th, cc, interp, multiH = g["kitti_%s_cfg" % tag]
        fg, _, _ = restate.assemble_flow_kitti(fd2, fd, pm, md, (hd * 8, wd * 8), float(th), bool(multiH), float(cc),
                                               bool(interp))
        assert np.array_equal(fg.numpy(), g["kitti_%s" % tag]), tag
    assert np.array_equal(restate.warp_grid(pm[:1], 72, 100).numpy(), g["hpatch_coarse"])
    # the multi-H settings really differ from the single-H ones (the merge is exercised)
    assert not np.array_equal(g["hpatch_a"], g["hpatch_b"]) and not np.array_equal(g["corr_a_flow"], g["corr_b_flow"])


def test_merge_multi_h_ownership_rule():
    flow = torch.stack([torch.full((2, 3, 2), float(i)) * 0.1 for i in range(3)])
    match = torch.tensor([[[0.9, 0.1, 0.1], [0.1, 0.1, 0.5]],
                          [[0.9, 0.9, 0.1], [0.1, 0.6, 0.5]],
                          [[0.9, 0.9, 0.9], [0.1, 0.9, 0.5]]])
    fg, mg, b = restate.merge_multi_h(flow, match, 0.5, True)
    assert torch.equal((fg[0, ..., 0] * 10).round().long(), torch.tensor([[0, 1, 2], [0, 1, 0]]))
    assert torch.equal(b[0], torch.tensor([[True, True, True], [False, True, True]]))
    assert torch.equal(mg[0], torch.tensor([[0.9, 0.9, 0.9], [0.1, 0.6, 0.5]]))
    fg1, _, b1 = restate.merge_multi_h(flow, match, 0.5, False)
    assert float(fg1.abs().max()) == 0.0 and torch.equal(b1[0], match[0] >= 0.5)


def test_assembly_file_lookup_follows_reference():
    from rfx import assemble
    names = ["mask_12_3H.npy", "flow_1_2H.npy", "flow_12_3H.npy", "junk"]
    assert assemble.find_nbH(12, names) == "3" and assemble.find_nbH(1, names) == "2"
    assert assemble.find_nbH(2, names) is None
    # a pair without saved arrays gives the reference's [] sentinels before any device work
    assert assemble.hpatch_getFlow_all(99, "/nonexistent", "/nonexistent", names, True, None, None, 0.5, 8, 8) == []
    assert assemble.corr_getFlow(99, "/nonexistent", names, "/nonexistent", "/nonexistent", True, 0.5) == ([], [])
    # inputs: the saved .npy arrays (any float dtype) or tensors (the views of a device record, no host round trip)
    a64 = np.arange(6, dtype=np.float64).reshape(2, 3)[:, ::2]
    t32 = assemble._to_dev(a64, "cpu")
    assert t32.dtype == torch.float32 and t32.is_contiguous() and torch.equal(t32, torch.tensor([[0., 2.], [3., 5.]]))
    tv = torch.arange(12, dtype=torch.float64).reshape(3, 4)[:, 1:3]
    t2 = assemble._to_dev(tv, "cpu")
    assert t2.dtype == torch.float32 and t2.is_contiguous() and torch.equal(t2, tv.float())


@pytest.mark.reference
@pytest.mark.parametrize("seed", [3, 4])
def test_assemble_restatement_matches_live_reference(seed, tmp_path):
    """Fresh seeds against the reference's getFlow functions executed from /root/reference (authoring container)."""
    import ref_loader
    kg = ref_loader.load()["kornia_geometry"]
    fh = ref_loader.script_functions("evaluation/evalHpatch/getResults.py", ["getFlow_all"])["getFlow_all"]
    fc = ref_loader.script_functions("evaluation/evalCorr/getResults.py", ["getFlow"])["getFlow"]
    n, hd, wd = 4, 6, 9
    flowDown, _, param, md = synth.assembly_arrays(seed, n, hd, wd)
    fine, coarse, maskp = [tmp_path / x for x in ("fine", "coarse", "mask")]
    for x in (fine, coarse, maskp):
        x.mkdir()
    np.save(fine / "flow_5_4H.npy", flowDown)
    np.save(fine / "mask_5_4H.npy", md)
    np.save(coarse / "flow_5_4H.npy", param)
    np.save(maskp / "maskBG_5_4H.npy", np.ones((hd * 8, wd * 8), bool))
    t = torch.from_numpy
    oh, ow = 40, 56
    ref = fh(5, str(fine), str(coarse), os.listdir(fine), True, kg.HomographyWarper(oh, ow), restate.identity_grid(oh, ow),
             0.4, ow, oh)
    fg, _, _ = restate.assemble_flow(t(flowDown), t(param), t(md), (oh, ow), 0.4, True, False)
    assert torch.equal(ref, fg)
    rf, rm = fc(5, str(fine), os.listdir(fine), str(coarse), str(maskp), True, 0.15)
    fg, mg, _ = restate.assemble_flow(t(flowDown), t(param), t(md), (hd * 8, wd * 8), 0.15, True, True)
    assert torch.equal(rf, fg) and torch.equal(rm[..., 0], mg)


def test_conv_dispatch_table_is_stable():
    """Host-side dispatch of rfx_conv2d_f32 (no GPU needed): which kernel instance a layer geometry gets.  Bits: 0-1 tile
    variant, 2 = 1x1 specialisation, 3 = wave-specialised form (off by default), 4 = 16-byte pixel loads, 5 = direct 3x3
    kernel with bits 6-7 = output patch shape (0: 8x16, 1: 16x8, 2: 32x4), 10 = k-major 1x1 kernel, 11 = 256-pixel patches of the
    direct kernel (one 64-channel tile, large launch), 12 = its instance with a ragged last K step (Cin % 8 != 0), 13 = direct
    3x3 / stride 2 kernel, 14 = chunked accumulation (direct 3x3 with K = 9 Cin >= 2048; stride-2 3x3 with K >= 1152; k-major 1x1 with K >= 512, 64-channel tiles)."""
    lib = _lib.load()
    kid = lambda N, Cin, Cout, k, s, p, Ho, Wo: lib.rfx_conv2d_kernel_id(N, Cin, Cout, k, k, s, p, Ho, Wo)
    # direct 3x3 / stride 1: big layers -> 128-channel tiles, 8x16 patches
    assert kid(128, 128, 128, 3, 1, 1, 120, 160) == 32
    assert kid(128, 64, 64, 3, 1, 1, 240, 320) == 33 | 2048                # Cout = 64 -> 64-channel tiles, 256-pixel patches
    assert kid(2, 64, 64, 3, 1, 1, 240, 320) == 33                         # ... only on launches of >= 1024 such patches
    assert kid(64, 256, 256, 3, 1, 1, 30, 40) == 32 | 64 | 16384            # 30x40 pads least with 16x8 patches; K = 2304: chunked sums
    assert kid(64, 256, 256, 3, 1, 1, 25, 33) == 32 | 128 | 16384           # 25x33 -> 32x4 patches
    assert kid(64, 224, 256, 3, 1, 1, 30, 40) == 32 | 64                    # K = 2016 < 2048: one chain
    assert kid(64, 49, 512, 3, 1, 1, 60, 80) == 32 | 4096                   # Cin % 8 != 0 -> the direct kernel's ragged instance
    assert kid(1, 49, 512, 3, 1, 1, 60, 80) == 32 | 1 | 4096                # ... a single image: 64-channel tiles fill the chip
    assert kid(64, 3, 64, 3, 1, 1, 480, 640) & 32 == 0                      # Cin < 8 (the stems' own kernels aside): implicit GEMM
    assert kid(64, 128, 128, 3, 2, 1, 60, 80) == 8192 | 16384               # 3x3 / stride 2 / pad 1, Cin % 8 == 0 -> direct stride-2 kernel <2>; K = 1152: chunked sums (round 5)
    assert kid(64, 64, 64, 3, 2, 1, 120, 160) == 8192 | 1                   # ... 64-channel tiles for Cout <= 64; K = 576: one chain
    assert kid(64, 128, 128, 3, 2, 1, 50, 66) == 8192 | 16384               # a chunked layer takes this kernel on EVERY map (50x66 pads to 56x80): its sums must not depend on the map
    assert kid(64, 64, 128, 3, 2, 1, 50, 66) & 8192 == 0                    # K = 576 (a chain in either kernel): a badly padding map keeps the implicit GEMM
    assert kid(64, 12, 128, 3, 2, 1, 60, 80) & 8192 == 0                    # Cin % 8 != 0 -> implicit GEMM
    assert kid(64, 128, 128, 3, 2, 0, 59, 79) & 8192 == 0                   # pad != 1 -> implicit GEMM
    # 1x1: 16-byte pixel loads only for stride 1 and H*W % 4 == 0
    # ... on the k-major kernel of conv1x1.hip (bit 10) when Cin % 32 == 0 and the tile is not the 64x64 one
    assert kid(64, 64, 256, 1, 1, 0, 120, 160) == 1024 | 4 | 16
    assert kid(64, 256, 1024, 1, 1, 0, 34, 45) == 1024 | 4                  # 1530 pixels per plane: scalar pixel loads
    assert kid(64, 256, 512, 1, 2, 0, 60, 80) == 4                          # strided 1x1 -> generic kernel
    assert kid(64, 256, 64, 1, 1, 0, 120, 160) == 1024 | 4 | 16 | 1         # Cout = 64 -> 64-wide channel tile
    assert kid(64, 1024, 256, 1, 1, 0, 30, 40) == 1024 | 4 | 16 | 1 | 16384  # K = 1024 (layer3 conv1): chunked sums, 64-channel tiles
    assert kid(1, 1024, 256, 1, 1, 0, 15, 20) == 1024 | 4 | 16 | 1 | 16384  # ... at every launch size (results do not depend on the batch)
    assert kid(64, 512, 128, 1, 1, 0, 60, 80) == 1024 | 4 | 16 | 1 | 16384  # K = 512 (layer2 conv1): two chunks of 256 since round 5
    assert kid(64, 256, 64, 1, 1, 0, 120, 160) & 16384 == 0                 # K = 256: one chain
    assert kid(64, 48, 256, 1, 1, 0, 120, 160) == 4 | 16                    # Cin % 32 != 0 -> generic kernel
    # tiny problems fall back to the 64x64 tile
    assert kid(1, 128, 49, 1, 1, 0, 12, 16) & 3 == 2
    assert lib.rfx_conv2d_tile_variant(64, 256, 120, 160) == 0 and lib.rfx_conv2d_tile_variant(1, 64, 12, 16) == 2


def test_fused_tail_patch_choice_and_packed_weight_layout():
    """Host logic added with the packed direct-3x3 kernel (no GPU needed): the fused Bottleneck tail avoids narrow patches
    (rfx_conv3x3_conv1x1_kernel_id), the plain kernel takes the shape with the least padded area over the STACKED batch, and
    ConvPlan packs wP exactly as include/rfx_api.h documents: wP[mt][s][h][m][kk] = w[mt*128 + m, k = s*72 + 2*kk + h]."""
    lib = _lib.load()
    fid = lib.rfx_conv3x3_conv1x1_kernel_id
    CH = 16384                                                   # round 5: the tails' 3x3 phase sums in chunks of 4 K steps (bit 14)
    assert fid(64, 120, 160, 64) == 512 | 1 | CH                 # 8x16 patch, 64-channel mid tile
    assert fid(64, 60, 80, 128) == 512 | CH                      # 128-channel mid tile
    assert fid(64, 100, 132, 64) == 512 | 1 | CH                 # W = 132: 8x16 (144) beats 16x8 (136 * 1.09) and 32x4 (132 * 1.4)
    assert fid(64, 112, 148, 64) == 512 | 1 | CH                 # W = 148: 8x16 (160) beats 16x8 (152 * 1.09)
    assert fid(64, 50, 66, 128) == 512 | 64 | CH                 # W = 66: 16x8 (72 * 1.09) beats 8x16 (80)
    # the stand-alone 3x3 of a tail asks for the same chunks (rfx_conv3x3_f32's k_chunk = 4; ops.ConvPlan.k_chunk, set by rfx/nets.py)
    k3 = lib.rfx_conv3x3_kernel_id
    assert k3(64, 64, 64, 120, 160, 0) == lib.rfx_conv2d_kernel_id(64, 64, 64, 3, 3, 1, 1, 120, 160) == 33 | 2048
    assert k3(64, 64, 64, 120, 160, 4) == 33 | 2048 | CH and k3(64, 128, 128, 60, 80, 4) == 32 | CH and k3(2, 128, 128, 60, 80, 4) == 33 | CH
    assert k3(64, 256, 256, 30, 40, 0) == k3(64, 256, 256, 30, 40, 4) == 32 | 64 | CH          # K >= 2048: chunked either way
    assert k3(64, 49, 512, 60, 80, 4) == 32 | 4096                                              # ragged Cin: never chunked
    kid = lambda N, Cin, Cout, H, W: lib.rfx_conv2d_kernel_id(N, Cin, Cout, 3, 3, 1, 1, H, W)
    assert kid(64, 256, 256, 26, 35) & 192 == 128                # plain kernel: least area wins (32x4 -> 36 columns)
    assert kid(64, 256, 256, 28, 37) & 192 == 64                 # 40 columns either way: the wider 16x8
    assert kid(1, 256, 256, 25, 33) & 192 == 128                 # one image: 26 stacked rows -> one 32x4 patch row
    from rfx import ops
    import torch as _t
    g = _t.Generator().manual_seed(1)
    w = _t.randn(200, 16, 3, 3, generator=g)
    plan = ops.ConvPlan(w, None, 1, 1, ops.ACT_NONE, device="cpu")
    assert plan.wP.shape == (2, 2, 2, 128, 36)
    w2 = w.reshape(200, 144)
    for (mt, s_, h, m, kk) in ((0, 0, 0, 0, 0), (0, 1, 1, 5, 35), (1, 0, 1, 71, 7), (1, 1, 0, 3, 20)):
        assert plan.wP[mt, s_, h, m, kk] == w2[mt * 128 + m, s_ * 72 + 2 * kk + h]
    assert float(plan.wP[1, :, :, 72:, :].abs().max()) == 0.0   # channels past Cout are zero rows
    # area threshold of the small-component filter: the largest a with a / size <= cc_th in float64
    for size, th in ((1200, 0.01), (466992, 0.01), (997, 0.05), (12, 1.0), (7, 0.0001), (1000, 0.3)):
        a = ops.cc_max_area(size, th)
        assert a / float(size) <= th and (a + 1) / float(size) > th



def test_host_sgemm_probe_and_score_chunk_resolution():
    """Host logic of the score chunk (no GPU needed): ops.host_sgemm_k_block() finds the K blocking with which a float32
    emulation reproduces THIS host's torch.mm (the reference's score product, utils/outil.py:34) bit for bit -- 384 on the authoring
    container's Xeon, 192 on the GPU box's EPYC (profiles/r04_mm_blocking_probe.json) -- and ops.resolve_score_chunk() turns a
    pipeline's ``score_chunk`` argument into (products per chunk, origin): the value every rfx_mutual_nn*_f32 call of that pipeline
    carries explicitly since ABI 8 (no process-wide setter, no probe inside a launch)."""
    from rfx import ops
    lib = _lib.load()
    assert not hasattr(lib, "rfx_mutual_nn_set_chunk")                     # ABI 7's process-wide setter is gone
    kc = ops.host_sgemm_k_block()
    assert kc in (None, 128, 192, 256, 384, 512, 1024)
    assert ops.host_sgemm_k_block() == kc                                  # cached
    env = os.environ.pop("RFX_SCORE_CHUNK", None)
    try:
        assert ops.resolve_score_chunk(None) == (256, "default (fixed 256 products)")          # the FALLBACK is pinned: 256
        assert ops.resolve_score_chunk("default")[0] == 256
        assert ops.resolve_score_chunk(192) == (192, "argument") and ops.resolve_score_chunk("384")[0] == 384
        assert ops.resolve_score_chunk(0)[0] == 256 and ops.resolve_score_chunk(-5)[0] == -1    # 0 = the library default in EVERY layer (rfx_api.h, ops.mutual_nn); < 0 = one chain
        with pytest.raises(ValueError):
            ops.resolve_score_chunk(100)
        v, src = ops.resolve_score_chunk("host")
        assert v == (kc if kc else 256) and (("host probe" in src) if kc else ("fallback" in src))
        os.environ["RFX_SCORE_CHUNK"] = "192"
        assert ops.resolve_score_chunk(None) == (192, "RFX_SCORE_CHUNK")
        os.environ["RFX_SCORE_CHUNK"] = "host"
        assert ops.resolve_score_chunk(None)[0] == v
    finally:
        os.environ.pop("RFX_SCORE_CHUNK", None)
        if env is not None:
            os.environ["RFX_SCORE_CHUNK"] = env
    if kc:                                                                 # the emulation the probe relies on, on a second, independent problem
        import numpy as np
        g = torch.Generator().manual_seed(3)
        A = torch.relu(torch.randn(1024, 400, generator=g)); B = torch.relu(torch.randn(1024, 300, generator=g))
        A, B = A / A.norm(dim=0, keepdim=True), B / B.norm(dim=0, keepdim=True)
        mm = (A.t() @ B).numpy()
        a, b = A.numpy().astype(np.float64), B.numpy().astype(np.float64)
        tot = np.zeros(mm.shape, np.float32)
        for k0 in range(0, 1024, kc):
            acc = np.zeros(mm.shape, np.float32)
            for k in range(k0, min(1024, k0 + kc)):
                acc = (acc.astype(np.float64) + np.outer(a[k], b[k])).astype(np.float32)
            tot = (tot.astype(np.float64) + acc).astype(np.float32)
        assert float((tot != mm).mean()) <= 1e-4


def _degenerate_lattice_samples(k, seed=0, n_rows=30, n_cols=40):
    """4-point samples on the feature-cell lattice (getWHTensor coordinates): half general, half with three points collinear in BOTH
    images (the rank-deficient systems the exact mode exists for), some repeated (late rounds draw few distinct samples)."""
    rng = np.random.RandomState(seed)

    def pts(m):
        r = ((rng.randint(0, n_rows, (m, 4)).astype(np.float32) + np.float32(0.5)) / np.float32(n_rows) - np.float32(0.5)) * 2
        c = ((rng.randint(0, n_cols, (m, 4)).astype(np.float32) + np.float32(0.5)) / np.float32(n_cols) - np.float32(0.5)) * 2
        return np.stack([c, r, np.ones_like(c)], axis=2).astype(np.float32)
    X, Y = pts(k), pts(k)
    h = k // 2
    X[:h, 2, :2] = 0.5 * (X[:h, 0, :2] + X[:h, 1, :2])
    Y[:h, 2, :2] = 0.5 * (Y[:h, 0, :2] + Y[:h, 1, :2])
    X[h:h + k // 8], Y[h:h + k // 8] = X[0], Y[0]
    return X, Y


def test_native_lapack_helper_equals_numpy_svd_bit_for_bit():
    """The exact mode's host stage (ops.ransac_h4_batched_finish -> rfx/_lapack.py -> librfxhost.so, include/rfx_host_api.h): the
    flagged 4-point samples are re-solved by the dgesdd numpy's own svd gufunc binds (resolved through numpy.linalg._umath_linalg's
    handle), called with numpy's argument set on a pool of std::threads.  Pinned here against ``np.linalg.svd(A)[2][:, 8]``
    (utils/outil.py:84-86) ITSELF, bit for bit in float64 and after the float32 cast: on general systems AND on rank-deficient
    ones (the 2-D null space whose LAPACK pick is rounding noise), with and without the duplicate cache, at 1 / 3 / 8 solver
    threads, for ragged sizes around the pool's block sizes, and for the row layout the device gather writes."""
    from rfx import ops, _lapack
    inf = _lapack.info()
    assert inf["int_bits"] in (32, 64) and "gesdd" in inf["symbol"], inf
    X, Y = _degenerate_lattice_samples(4999)
    ref = _lapack.dlt_null_vectors(X, Y)                                            # numpy, in process: the reference's expression
    A = np.zeros((3, 8, 9))
    for i in range(4):                                                              # and the restatement against the formula, once
        u, v, u_, v_ = Y[:3, i, 0], Y[:3, i, 1], X[:3, i, 0], X[:3, i, 1]
        A[:, 2 * i, 3:] = np.stack([-u, -v, -np.ones(3), v_ * u, v_ * v, v_], axis=1)
        A[:, 2 * i + 1, :3] = np.stack([u, v, np.ones(3)], axis=1)
        A[:, 2 * i + 1, 6:] = np.stack([-u_ * u, -u_ * v, -u_], axis=1)
    assert np.array_equal(np.linalg.svd(A)[2][:, 8], ref[:3])
    xy = np.ascontiguousarray(np.concatenate([X[:, :, :2].reshape(-1, 8), Y[:, :, :2].reshape(-1, 8)], axis=1))
    try:
        for threads in (1, 3, 8):
            assert _lapack.set_threads(threads) == threads
            for dedupe in (False, True):
                for k in (1, 63, 64, 1025, 4999):
                    H, n_solved, hv = _lapack.solve_rows(xy[:k], dedupe=dedupe, want_f64=True)
                    assert np.array_equal(hv.view(np.int64), ref[:k].view(np.int64)), (threads, dedupe, k)
                    assert H.dtype == np.float32 and np.array_equal(H, ref[:k].astype(np.float32))
                    assert n_solved <= k and (dedupe and k == 4999) == (n_solved < k)
        out = np.zeros((6000, 9), np.float32)                                       # a caller-owned (pinned) result buffer
        assert _lapack.solve_rows(xy, out=out)[0].base is out and np.array_equal(out[:4999], ref.astype(np.float32))
    finally:
        _lapack.set_threads(_lapack.default_threads())
    assert np.array_equal(ops.lapack_dlt(X, Y), ref.reshape(-1, 3, 3).astype(np.float32))
    assert _lapack.solve_rows(np.zeros((0, 16), np.float32))[0].shape == (0, 9)
    with pytest.raises(ValueError):
        _lapack.solve_rows(xy[:, :8])


def test_l2norm_order_is_atens():
    """F.normalize(x, dim=1) (quick_start/coarseAlignFeatMatch.py:106,124) on the CPU sums the squares of a cell's C channels as ONE
    fused-multiply-add chain in channel order (ATen's binary_kernel_reduce with NormTwoOps), takes a correctly rounded sqrt, clamps at
    1e-12 and divides: a float32 emulation of exactly that reproduces torch bit for bit on every shape and thread count -- the order
    csrc/pool.hip::l2norm_nchw*_kernel follow since round 5.  The four-interleaved-chain sum of rounds 1-4 is MORE accurate against
    float64 but differs from the reference's norm in most cells, by up to ~1e-6 relative: a coherent factor on every score of the
    cell, an order of magnitude above what the convolutions' round-off does to a score."""
    def fma(a, b, c):
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
    worst4 = 0.0
    for shape in ((1, 1024, 15, 20), (2, 256, 30, 40), (3, 1024, 5, 7), (1, 50, 9, 11)):
        g = torch.Generator().manual_seed(shape[2])
        x = torch.relu(torch.randn(*shape, generator=g)) * torch.rand(1, shape[1], 1, 1, generator=g)
        N, C = shape[0], shape[1]
        X = x.reshape(N, C, -1).numpy()
        s = np.zeros((N, X.shape[2]), np.float32)
        s4 = [np.zeros((N, X.shape[2]), np.float32) for _ in range(4)]
        for c in range(C):
            s = fma(X[:, c], X[:, c], s)
            s4[c % 4] = fma(X[:, c], X[:, c], s4[c % 4])
        d = np.maximum(np.sqrt(s).astype(np.float32), np.float32(1e-12))
        want = (X / d[:, None, :]).astype(np.float32)
        keep = torch.get_num_threads()
        try:
            for threads in (1, 4):
                torch.set_num_threads(threads)
                got = F.normalize(x).reshape(N, C, -1).numpy()
                assert np.array_equal(got, want), (shape, threads)
        finally:
            torch.set_num_threads(keep)                 # later tests pin exact CPU results: leave the pool as it was
        n4 = np.sqrt((s4[0] + s4[1]) + (s4[2] + s4[3])).astype(np.float32)
        worst4 = max(worst4, float(np.abs(n4 / np.sqrt(s).astype(np.float32) - 1).max()))
    assert worst4 > 2e-7            # the old order is NOT the reference's: up to ~1e-6 relative on the norm of a cell


def test_timed_mode_parity_checker_replays_a_device_draw_dump(tmp_path, monkeypatch):
    """The checker behind ``parsed.parity_timed_mode`` (oracle/parity_sweep.py::compare_loop on a dump that carries the device's own
    Philox samples), exercised WITHOUT a GPU: the "device" here is the port oracle/restate.py running the multi-homography loop of
    evaluation/evalHpatch/evaluation.py:211-243 with the draws rfx_draw_samples_i64 would make -- Philox4x32-10 keyed by (seed,
    pair id, round), tests/philox_ref.py -- and leaving the per-round state in dump_gpu_loop's format.  The checker (the reference
    itself where it is present) must regenerate every round's samples from (draw seed, pair id, round) and find them equal to the
    dumped ones, replay every round from the dumped state exactly -- count, status, bit-equal inliers, |dH| = 0, flows, accept
    decision, next mask -- and conclude that its own free-running loop is implied.  A dump whose samples were tampered with is caught."""
    import json
    import philox_ref
    import parity_sweep
    cfg = "t_small"
    monkeypatch.setitem(parity_sweep.CONFIGS, cfg, dict(variant="B", nbScale=3, scaleR=1.2, size="min", nbIter=400, loop="hpatch", H=240, W=320,
                                                        maxCoarse=2, th=0.01, amp=0.05))
    monkeypatch.setattr(parity_sweep, "_W", {})
    c, seed, draw_seed, sub = parity_sweep.CONFIGS[cfg], 7, 1000, 4
    sds = parity_sweep.state_dicts(parity_sweep.MULTIH_MATCH_STD)
    nets = dict(feat=sds["feat"], flow=sds["flow"], match=sds["match"])
    rnd = [0]

    def device_draw(n, it):                                             # what the lock-step driver draws in round rnd[0] for pair id `seed`
        return torch.from_numpy(philox_ref.draw_samples([n], it, draw_seed, rnd[0], pair_ids=[seed])[0])
    ca = restate.CoarseAlignOracle(sds["trunk"], c["nbScale"], c["nbIter"], 0.05, 240, c["scaleR"], variant="B", sample_fn=device_draw)
    Is, It = parity_sweep.make_pair(cfg, seed, 240, 320)
    ca.setPair(Is, It)
    h, w = ca.It.size[1], ca.It.size[0]
    with torch.no_grad():
        featt = F.normalize(restate.feature_extractor(nets["feat"], ca.ItTensor))
    grid = restate.identity_grid(h, w)
    Mask = np.zeros((h, w), np.float32)
    pack = lambda m: np.packbits(m > 0.5, axis=-1)
    rows, nb = [], 0
    while nb <= c["maxCoarse"]:
        fg = Mask.copy()
        r = ca.getCoarse(fg)
        assert r is not None
        with torch.no_grad():
            flow12, match, fd8, md8 = restate.pred_flow_mask(nets, ca.IsTensor, featt, restate.warp_grid(torch.from_numpy(r["H"])[None], h, w), grid)
        gain = float((match * (1 - fg)).mean())
        acc = gain > c["th"] or nb == 0
        new = ((Mask + match * (1 - fg)) >= 1.0).astype(np.float32) if acc else Mask
        inl = np.zeros(len(ca.index1), bool)                            # the dump keeps (cap,) inlier flags per round, rows beyond n unused
        inl[:len(r["inlier"])] = r["inlier"]
        rows.append(dict(mask_before=pack(fg), mask_after=pack(new), n=len(r["match1"]), status=0, winner=0, inlier=np.packbits(inl),
                         H=r["H"], flowDown8=fd8[0], matchDown8=md8[0], flow12_sub=flow12[0, ::sub, ::sub].numpy(), accept=int(acc), gain=gain,
                         samples=r["samples"].astype(np.int32), round_id=rnd[0]))
        rnd[0] += 1
        if not acc:
            break
        Mask, nb = new, nb + 1
    assert len(rows) >= 2 and nb >= 1
    d = {k: np.stack([np.asarray(q[k]) for q in rows]) for k in rows[0]}
    d.update(seed=seed, index1=ca.index1.numpy(), index2=ca.index2.numpy(), nbH=nb, sub=sub, final_mask=pack(Mask))
    json.dump(dict(score_chunk_products=192, score_chunk_source="test", draw="device", degenerate="device", draw_seed=draw_seed, draw_epoch=0),
              open(str(tmp_path / "meta.json"), "w"))
    path = str(tmp_path / ("pair_%d.npz" % seed))
    np.savez_compressed(path, **d)
    rec = parity_sweep.compare_loop(cfg, seed, path)
    assert rec["draw"] == "device" and rec["identical_list"] and rec["rounds"] == len(rows)
    for q in rec["round_records"]:
        assert q["count_equal"] and q["status_equal"] and q["inlier_bit_exact"] and q["H_delta"] == 0.0, q
        assert q["flow12_delta"] < 1e-6 and q["flowDown8_delta"] < 1e-6 and q["accept_equal"] and q.get("mask_diff_frac", 0.0) == 0.0, q
    assert rec["free_run"]["same_nbH"] and rec["free_run"].get("implied_by_exact_rounds")
    s = parity_sweep.summarise_loop(cfg, [rec], 1, 0.0)
    assert s["rounds_exact_given_state"] == "%d/%d" % (len(rows), len(rows)) and s["rounds_degenerate_winner"] == 0
    # a dump whose samples are not the Philox draws of (seed, pair id, round) is refused
    d["samples"] = d["samples"].copy()
    d["samples"][0, 0, 0] = (d["samples"][0, 0, 0] + 1) % d["n"][0]
    np.savez_compressed(path, **d)
    with pytest.raises(AssertionError, match="dumped device samples"):
        parity_sweep.compare_loop(cfg, seed, path)


def test_ctypes_prototypes_mirror_the_header_argument_by_argument():
    """The binding's SIGNATURES table (rfx/_lib.py) against include/rfx_api.h, parsed: for every entry point the same number of
    parameters and, per parameter, the same kind -- any pointer -> c_void_p, int / int32_t -> c_int, long long -> c_longlong,
    uint64_t -> c_uint64, float -> c_float, double -> c_double, size_t -> c_size_t -- and the same return type.  A C-ABI change that
    the mirror misses (ABI 8 moved three signatures) would otherwise only show as garbage arguments on a GPU box."""
    import ctypes as C
    hdr = open(os.path.join(ROOT, "include", "rfx_api.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)                       # comments carry example prototypes
    protos = re.findall(r"\b(const\s+char\s*\*|int|size_t)\s+(rfx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S)
    assert len(protos) == len(_lib.SIGNATURES), (len(protos), len(_lib.SIGNATURES))

    def kind(decl):
        d = " ".join(decl.split())
        if "*" in d:
            return C.c_void_p
        base = d.rsplit(" ", 1)[0] if " " in d else d                       # drop the parameter name
        base = base.replace("const ", "").strip()
        return {"int": C.c_int, "int32_t": C.c_int, "long long": C.c_longlong, "uint64_t": C.c_uint64, "float": C.c_float,
                "double": C.c_double, "size_t": C.c_size_t}[base]
    same = lambda a, b: a is b or {a, b} <= {C.c_int, C.c_int32}
    for ret, name, args in protos:
        res, argtypes = _lib.SIGNATURES[name]
        want_res = {"int": C.c_int, "size_t": C.c_size_t}.get(ret, C.c_char_p)
        assert same(res, want_res), (name, res, ret)
        params = [] if args.strip() in ("", "void") else [p for p in args.split(",")]
        assert len(params) == len(argtypes), (name, len(params), len(argtypes))
        for i, (p, t) in enumerate(zip(params, argtypes)):
            assert same(kind(p), t), (name, i, p.strip(), t)


def test_kernel_resources_are_what_design_says():
    """Static check of the built library (scripts/kernel_resources.py reads registers / LDS / scratch / spills of every gfx950 kernel
    out of librfx.so's code objects and locates each scratch instruction relative to the MFMA loops): the occupancy each hot kernel
    was tiled for is the occupancy the register and LDS budgets grant, the DLT kernel keeps its fp64 matrices in registers, and no
    kernel pays a spill per K step beyond one scratch instruction per 36 MFMAs (the two known cases: the chunked stride-2 3x3 kernel
    and the 128-channel fused Bottleneck tail, both at the 256-VGPR cap).  profiles/r06_kernel_resources.tsv is this table."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "scripts", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    rows = kr.table(_lib.LIB_PATH)
    by = {r["kernel"]: r for r in rows}
    assert len(rows) > 150 and len(by) == len(rows)
    for r in rows:
        assert r["vgpr"] + r["agpr"] <= 512 and r["lds"] <= 160 * 1024, r
        assert r["k_loop_scratch"] <= r["k_loop_mfma"] // 36, r                       # never more than 1 scratch op per 36 MFMAs
        if r["lds"] > 32 * 1024:
            assert r["wg_per_cu"] * ((r["wg"] + 63) // 64) >= 8, r                    # >= 2 wavefronts per SIMD on every tiled kernel
    hot = ["conv3x3_direct_kernel<2, 16, false, 2, false, 4>", "conv3x3_direct_kernel<2, 16, false, 2, false, 0>",
           "conv3x3_direct_kernel<1, 16, false, 2, false, 4>", "conv1x1_kmajor_kernel<1, true, 8>", "conv1x1_kmajor_kernel<1, true, 0>",
           "conv2d_mfma_kernel<2, 2, true, false, false>", "mnn_tile_kmajor_kernel<true, 1>", "mnn_tile_kmajor_kernel<false, 1>",
           "stem7_conv_maxpool_kernel", "stem_conv_maxblur_kernel", "ransac_dlt_kernel<false>", "ransac_dlt_kernel<true>",
           "l2norm_nchw_kernel", "l2norm_nchw_q4_kernel"]
    for k in hot:
        assert by[k]["scratch"] == 0 and by[k]["vgpr_spills"] == 0, by[k]
    # the two-workgroups-per-CU kernels: 256 threads, <= 256 registers, <= 80 KB of LDS
    for k in hot[:8]:
        assert by[k]["wg"] == 256 and by[k]["wg_per_cu"] >= 2 and by[k]["k_loop_mfma"] == 0, by[k]   # (no scratch: loops not analysed)
    # the 128x128 1x1 tile and the fused tails sit AT the cap and spill a handful of registers, all outside the K loops
    for k in ("conv1x1_kmajor_kernel<2, true, 0>", "conv1x1_kmajor_kernel<2, false, 0>"):
        assert by[k]["vgpr"] == 256 and by[k]["vgpr_spills"] <= 8 and by[k]["k_loop_scratch"] == 0 and by[k]["k_loop_mfma"] == 64, by[k]
    # one K step of the dominant kernel: 144 MFMAs (8 channels x 9 taps = 36 k-pairs on 4 sub-tiles), two barriers, no scratch; the
    # committed mix table (profiles/r05_kloop_mix.tsv) is this build's
    mix = {r["kernel"]: r for r in kr.mix_table(r"^conv3x3_direct_kernel<2, 16, false, 2, false, 4>$|^conv1x1_kmajor_kernel<2, true, 0>$",
                                                _lib.LIB_PATH)}
    dom = mix["conv3x3_direct_kernel<2, 16, false, 2, false, 4>"]
    assert (dom["mfma"], dom["barrier"], dom["scratch"]) == (144, 2, 0) and dom["valu"] < dom["mfma"], dom
    lines = [l.rstrip("\n").split("\t") for l in open(os.path.join(ROOT, "profiles", "r05_kloop_mix.tsv")) if not l.startswith("#")]
    for l in lines[1:]:
        row = dict(zip(lines[0], l))
        if row["kernel"] in mix:
            assert all(str(mix[row["kernel"]][c]) == row[c] for c in ("mfma", "ds_read", "ds_write", "vmem_load", "barrier")), row
    # the committed table is the table of this build
    path = os.path.join(ROOT, "profiles", "r06_kernel_resources.tsv")
    lines = [l.rstrip("\n").split("\t") for l in open(path) if not l.startswith("#")]
    assert lines[0] == kr.COLS
    committed = {l[1]: dict(zip(kr.COLS, l)) for l in lines[1:]}
    assert set(committed) == set(by)
    for k, r in by.items():
        assert all(str(r[c]) == committed[k][c] for c in ("wg", "vgpr", "lds", "scratch", "wg_per_cu", "k_loop_scratch")), (k, r, committed[k])


def test_split_weights_are_exact_bf16_pieces_in_fragment_order():
    """ops.split_weights (the host half of rfx_conv1x1_split_f32 / rfx_conv3x3_split_f32): hi + mid + lo == w exactly for every
    float32 weight, each piece on the bf16 grid, packed [k / 16][piece][(k % 16) / 8][m (Mpad)][k % 8] with zero rows past Cout."""
    from rfx import ops
    g = torch.Generator().manual_seed(3)
    w = torch.randn(70, 48, generator=g) * torch.logspace(-6, 3, 48)[None, :]
    ws = ops.split_weights(w)
    assert ws.dtype == torch.int16 and tuple(ws.shape) == (3, 3, 2, 128, 8)
    pc = ws.view(torch.bfloat16).float()
    for kb in range(3):
        for h in range(2):
            blk = pc[kb, :, h, :70, :]                                        # (piece, m, 8)
            assert torch.equal(blk.double().sum(0).float(), w[:, 16 * kb + 8 * h:16 * kb + 8 * h + 8])
            assert torch.equal(blk[0], w[:, 16 * kb + 8 * h:16 * kb + 8 * h + 8].bfloat16().float())
    assert float(pc[:, :, :, 70:, :].abs().max()) == 0.0
