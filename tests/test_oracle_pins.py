"""Pins of the evaluation-side restatements of oracle/restate.py against the REFERENCE ITSELF:

  * ``pred_flow_mask`` / ``pred_flow_mask_kitti``   <- PredFlowMask of evaluation/evalHpatch/evaluation.py:23-55 and
                                                       evaluation/evalKITTI/evaluation.py:49-81
  * ``CoarseAlignOracle(variant="B")``                <- evaluation/evalHpatch/coarseAlignFeatMatch.py:102-179
  * ``multi_h_loop``                                   <- the ``while nbCoarse <= args.maxCoarse`` loop, evalHpatch/evaluation.py:211-243
  * ``multi_h_loop_kitti`` (+ ``remove_small_cc_eval``, ``resize_img``)
                                                     <- the ``while True`` loop, evalKITTI/evaluation.py:270-336

Golden fixtures (tests/golden/{predflowmask,coarse_b,multi_h,kitti_loop}.npz) were produced by running the reference's
own functions / classes / loop statements, compiled from where they lie under /root/reference by oracle/ref_loader.py
(tests/golden/make_golden.py); they are checked on every machine.  The ``-m reference`` tests repeat the comparison
live on fresh seeds in the authoring container.  CPU only."""
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import restate
from rfx import weights, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
MULTIH_MATCH_STD = 3.0


def gold(name):
    return np.load(os.path.join(GOLD, name))


def _tt(im):
    return torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)[None]


def _nets(match_std, randomize_bn=False, seeds=(1, 2, 3)):
    return dict(feat=weights.feature_extractor_sd(seed=seeds[0], randomize_bn=randomize_bn),
                flow=weights.net_flow_coarse_sd(seed=seeds[1], randomize_bn=randomize_bn),
                match=weights.net_matchability_sd(seed=seeds[2], randomize_bn=randomize_bn, last_std=match_std))


# ---------------------------------------------------------------- PredFlowMask (a19)


def test_pred_flow_mask_matches_reference_golden():
    g = gold("predflowmask.npz")
    nets = _nets(float(g["match_std"]), True, tuple(int(x) for x in g["seeds"]))
    I1, I2 = synth.make_pair(96, 128, seed=5, homography=True)
    Is, It = _tt(I1), _tt(I2)
    Hm = torch.from_numpy(g["hp_H"])
    with torch.no_grad():
        fc = restate.warp_grid(Hm, 96, 128)
        featt = F.normalize(restate.feature_extractor(nets["feat"], It))
        f12, match, fd8, md8 = restate.pred_flow_mask(nets, Is, featt, fc, restate.identity_grid(96, 128))
        assert np.abs(f12.numpy() - g["hp_flow12"]).max() < 1e-6
        assert np.abs(match - g["hp_match"]).max() < 1e-6
        assert np.abs(fd8 - g["hp_flowDown8"]).max() < 1e-6
        assert np.abs(md8 - g["hp_matchDown8"]).max() < 1e-6
        # KITTI variant: pre-sampled images, cycle-checked matchability, output grid larger than the inputs
        gh, gw = (int(x) for x in g["ki_grid_hw"])
        f12, match, fd8, md8 = restate.pred_flow_mask_kitti(nets, restate.grid_sample(Is, fc), It, fc,
                                                            restate.identity_grid(gh, gw))
        assert f12.shape == (1, gh, gw, 2)
        assert np.abs(f12.numpy() - g["ki_flow12"]).max() < 1e-6
        assert np.abs(match - g["ki_match"]).max() < 1e-6
        assert np.abs(fd8.numpy() - g["ki_flowDown8"]).max() < 1e-6
        assert np.abs(md8.numpy() - g["ki_matchDown8"]).max() < 1e-6


def test_remove_small_cc_eval_matches_reference_golden():
    g = gold("predflowmask.npz")
    th, cc = (float(x) for x in g["cc_cfg"])
    out = restate.remove_small_cc_eval(g["cc_in"].copy(), th, cc)
    assert np.array_equal(out, g["cc_out"])
    assert (out != g["cc_in"]).any(), "fixture must exercise the removal branch"
    assert restate.remove_small_cc_eval(g["cc_in"].copy(), th, 0) is not None


# ---------------------------------------------------------------- CoarseAlign variant B (a7)


def _oracle_b(nbScale, nbIter, minSize, scaleR):
    return restate.CoarseAlignOracle(weights.resnet50_trunk_sd(0), nbScale, nbIter, 0.05, minSize, scaleR, variant="B")


def test_coarse_align_b_matches_reference_golden():
    g = gold("coarse_b.npz")
    nbScale, nbIter, minSize, scaleR = g["cfg"]
    ca = _oracle_b(int(nbScale), int(nbIter), int(minSize), float(scaleR))
    I1, I2 = synth.make_pair(240, 320, seed=6, homography=True)
    ca.setPair(I1, I2)
    assert (ca.It.size[1], ca.It.size[0]) == tuple(int(x) for x in g["hw"])
    # the cached match list of setPair (evalHpatch/coarseAlignFeatMatch.py:139-147): float and integer coordinates
    for name, ref in (("W1M", "W1"), ("H1M", "H1"), ("W2M", "W2"), ("H2M", "H2"), ("W2MI", "W2I"), ("H2MI", "H2I")):
        assert np.array_equal(getattr(ca, name).numpy(), g[ref]), name
    for k in range(4):
        torch.manual_seed(300 + k)
        r = ca.getCoarse(g["mask_%d" % k])
        if g["H_%d" % k].size == 0:
            assert r is None                                    # everything masked: < 4 matches -> None (:171-172)
        else:
            assert r is not None and r["H"].dtype == np.float32
            assert np.abs(r["H"] - g["H_%d" % k]).max() <= 1e-6, k
    # masks 1 and 2 really filter matches
    torch.manual_seed(301)
    assert len(ca.getCoarse(g["mask_1"])["index1"]) < len(g["W1"])


# ---------------------------------------------------------------- multi-H loop (a20 / f1)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_multi_h_loop_matches_reference_golden(tag):
    g = gold("multi_h.npz")
    seed, maxCoarse, th, draw_seed = g["%s_cfg" % tag]
    nets = _nets(float(g["match_std"]))
    ca = _oracle_b(3, 300, 240, 1.2)
    I1, I2 = synth.make_pair(240, 320, seed=int(seed), homography=True)
    ca.setPair(I1, I2)
    torch.manual_seed(int(draw_seed))
    o = restate.multi_h_loop(ca, nets, max_coarse=int(maxCoarse), mask_region_th=float(th))
    assert len(o["H"]) == int(g["%s_nb" % tag]) >= 2
    assert np.abs(np.stack(o["H"]) - g["%s_H" % tag]).max() <= 1e-6
    assert np.abs(np.concatenate(o["flowDown8"]) - g["%s_flowDown8" % tag]).max() < 1e-6
    assert np.abs(np.concatenate(o["matchDown8"]) - g["%s_matchDown8" % tag]).max() < 1e-6
    assert np.array_equal(o["masks"][-1], g["%s_mask" % tag])
    assert 0.05 < float(o["masks"][0].mean()) < 1.0            # the mask grows: later homographies see filtered matches


@pytest.mark.parametrize("tag", ["a", "b"])
def test_kitti_two_resolution_loop_matches_reference_golden(tag):
    g = gold("kitti_loop.npz")
    seed, fine, cc_th, th, draw_seed = g["%s_cfg" % tag]
    nets = _nets(float(g["match_std"]))
    Is, It = synth.make_pair(96, 312, seed=int(seed), homography=True, amp=0.03)
    ca = _oracle_b(3, 300, 160, 1.2)
    ca.setPair(Is, It)
    sizes = [int(x) for x in g["%s_sizes" % tag]]
    r, d2 = restate.resize_img(It, 8, int(fine)), restate.resize_img(It, 8, int(fine) // 2)
    assert [It.size[1], It.size[0], r.size[1], r.size[0], d2.size[1], d2.size[0]] == sizes
    torch.manual_seed(int(draw_seed))
    o = restate.multi_h_loop_kitti(ca, nets, Is, It, int(fine), mask_region_th=float(th), cc_th=float(cc_th))
    assert len(o["H"]) == int(g["%s_nb" % tag]) >= 2
    assert np.abs(np.stack(o["H"]) - g["%s_H" % tag]).max() <= 1e-6
    assert np.abs(np.concatenate(o["flowD2"]) - g["%s_flowD2" % tag]).max() < 1e-6
    assert np.abs(np.concatenate(o["flowDown8"]) - g["%s_flowDown8" % tag]).max() < 1e-6
    assert np.abs(np.concatenate(o["matchDown8"]) - g["%s_matchDown8" % tag]).max() < 1e-6
    assert np.array_equal(o["masks"][-1], g["%s_mask" % tag])


# ---------------------------------------------------------------- live reference, fresh seeds (authoring container)


def _live():
    import sys
    sys.path.insert(0, os.path.join(GOLD))
    import make_golden
    return make_golden


@pytest.mark.reference
def test_eval_side_restatements_match_live_reference():
    """Fresh seeds, the reference executed here: PredFlowMask (both variants), CoarseAlign-B under a random mask, the
    Hpatch multi-H loop and the KITTI two-resolution loop."""
    import ref_loader
    mg = _live()
    R = ref_loader.load()
    # --- PredFlowMask, both variants
    fh = ref_loader.script_functions("evaluation/evalHpatch/evaluation.py", ["PredFlowMask"])["PredFlowMask"]
    fk = ref_loader.script_functions("evaluation/evalKITTI/evaluation.py", ["PredFlowMask"])["PredFlowMask"]
    net = mg._ref_networks(match_std=0.5, randomize_bn=True, seeds=(51, 52, 53))
    nets = _nets(0.5, True, (51, 52, 53))
    I1, I2 = synth.make_pair(80, 112, seed=21, homography=True)
    Is, It = _tt(I1), _tt(I2)
    Hm = torch.tensor([[[0.97, -0.03, 0.02], [0.02, 1.04, -0.03], [-0.01, 0.02, 1.0]]])
    with torch.no_grad():
        fc = R["kornia_geometry"].HomographyWarper(80, 112).warp_grid(Hm)
        grid = restate.identity_grid(80, 112)
        ref = fh(Is, F.normalize(net["netFeatCoarse"](It)), fc, grid, net)
        got = restate.pred_flow_mask(nets, Is, F.normalize(restate.feature_extractor(nets["feat"], It)),
                                     restate.warp_grid(Hm, 80, 112), grid)
        # identical features and grids; the only arithmetic that differs is the summation order of the 7x7 correlation
        # (49 narrow-mul-sum ops in model/model.py:138-149 vs one unfold in the restatement): float32 round-off
        for a, b in zip(ref, got):
            assert np.abs(np.asarray(a) - np.asarray(b)).max() < 5e-5
        IsS = F.grid_sample(Is, fc)
        ref = fk(IsS, It, fc, restate.identity_grid(100, 140), net)
        got = restate.pred_flow_mask_kitti(nets, IsS, It, fc, restate.identity_grid(100, 140))
        for a, b in zip(ref, got):
            assert np.abs(np.asarray(a) - np.asarray(b)).max() < 5e-5
    # --- CoarseAlign B + Hpatch loop
    net = mg._ref_networks(match_std=MULTIH_MATCH_STD)
    nets = _nets(MULTIH_MATCH_STD)
    loop = ref_loader.script_loop("evaluation/evalHpatch/evaluation.py", "nbCoarse <= args.maxCoarse")
    I1, I2 = synth.make_pair(160, 208, seed=31, homography=True)
    ca_ref = mg._ref_coarse_b(nbScale=3, nbIter=200, minSize=160, scaleR=1.3)
    ca_ref.setPair(I1, I2)
    ca = _oracle_b(3, 200, 160, 1.3)
    ca.setPair(I1, I2)
    assert np.array_equal(ca.W1M.numpy(), ca_ref.W1MutualMatch.numpy()) and np.array_equal(ca.H2MI.numpy(), ca_ref.H2MutualMatchInt.numpy())
    Itw, Ith = ca_ref.It.size
    Mt = (np.random.RandomState(4).rand(Ith, Itw) > 0.6).astype(np.float32)
    torch.manual_seed(77)
    Href = ca_ref.getCoarse(Mt)
    torch.manual_seed(77)
    assert np.abs(ca.getCoarse(Mt)["H"] - Href).max() <= 1e-6
    with torch.no_grad():
        ns = dict(args=types.SimpleNamespace(maxCoarse=3, maskRegionTh=0.01), coarseModel=ca_ref, network=net,
                  featt=F.normalize(net["netFeatCoarse"](ca_ref.ItTensor)), grid=restate.identity_grid(Ith, Itw),
                  warper=R["kornia_geometry"].HomographyWarper(Ith, Itw), It_bg=np.ones((Ith, Itw), np.float32),
                  Mask=np.zeros((Ith, Itw), np.float32), Coarse_Flow_Tensor=[], Fine_Flow_Tensor=[], Fine_Mask_Tensor=[],
                  nbCoarse=0, PredFlowMask=ref_loader.script_functions("evaluation/evalHpatch/evaluation.py", ["PredFlowMask"])["PredFlowMask"])
        torch.manual_seed(78)
        loop(ns)
    torch.manual_seed(78)
    o = restate.multi_h_loop(ca, nets, max_coarse=3, mask_region_th=0.01)
    assert len(o["H"]) == ns["nbCoarse"]
    assert np.abs(np.stack(o["H"]) - np.concatenate(ns["Coarse_Flow_Tensor"])).max() <= 1e-6
    assert np.abs(np.concatenate(o["flowDown8"]) - np.concatenate(ns["Fine_Flow_Tensor"])).max() < 1e-5
    assert float((o["masks"][-1] != ns["Mask"]).mean()) < 1e-3       # thresholded at exactly 1.0: round-off may flip a pixel


@pytest.mark.reference
def test_kitti_loop_matches_live_reference():
    """Fresh seed: the reference's `while True` KITTI loop (evaluation/evalKITTI/evaluation.py:270-336) executed here."""
    import ref_loader
    mg = _live()
    R = ref_loader.load()
    fk = ref_loader.script_functions("evaluation/evalKITTI/evaluation.py", ["PredFlowMask", "remove_small_cc", "get_info"])
    import torchvision.transforms as tvt
    fk["get_info"].__globals__["transforms"] = tvt
    loop = ref_loader.script_loop("evaluation/evalKITTI/evaluation.py", "True")
    net, nets = mg._ref_networks(match_std=MULTIH_MATCH_STD), _nets(MULTIH_MATCH_STD)
    Is, It = synth.make_pair(80, 264, seed=17, homography=True, amp=0.03)
    ca_ref = mg._ref_coarse_b(nbScale=3, nbIter=200, minSize=128, scaleR=1.2)
    fine, cc_th, th = 96, 0.001, 0.01
    It_resize, It_d2 = R["outil"].resizeImg(It, 8, fine), R["outil"].resizeImg(It, 8, fine // 2)
    with torch.no_grad():
        w_org, h_org, _, grid_org, _ = fk["get_info"](It)
        _, _, tensor_s, _, _ = fk["get_info"](Is)
        _, _, tensor_resize, grid_resize, warper_resize = fk["get_info"](It_resize)
        _, _, tensor_d2, grid_d2, warper_d2 = fk["get_info"](It_d2)
        ca_ref.setPair(Is, It)
    ns = dict(args=types.SimpleNamespace(cc_th=cc_th, maskRegionTh=th), coarseModel=ca_ref, network=net,
              It_bg=np.ones((h_org, w_org), np.float32), Mask=np.zeros((h_org, w_org), np.float32), warper_d2=warper_d2,
              warper_resize=warper_resize, tensor_s=tensor_s, tensor_d2=tensor_d2, tensor_resize=tensor_resize,
              grid_d2=grid_d2, grid_resize=grid_resize, grid_org=grid_org, Homography=[], Org_D2=[], Finetune_D2=[],
              Org_Mask=[], Finetune_Mask=[], Org=[], Finetune=[], nbCoarse=0, PredFlowMask=fk["PredFlowMask"],
              remove_small_cc=fk["remove_small_cc"])
    torch.manual_seed(91)
    loop(ns)
    ca = _oracle_b(3, 200, 128, 1.2)
    ca.setPair(Is, It)
    torch.manual_seed(91)
    o = restate.multi_h_loop_kitti(ca, nets, Is, It, fine, mask_region_th=th, cc_th=cc_th)
    assert len(o["H"]) == ns["nbCoarse"] >= 1
    assert np.abs(np.stack(o["H"]) - torch.cat(ns["Homography"]).numpy()).max() <= 1e-6
    assert np.abs(np.concatenate(o["flowDown8"]) - torch.cat(ns["Finetune"]).numpy()).max() < 1e-5
    assert float((o["masks"][-1] != ns["Mask"]).mean()) < 1e-3


# ------------------------------------------------------------------------------------------------ the staged reference (oracle/_ref)

_STAGED_PROBE = r'''
import json, sys, os, hashlib
import numpy as np, torch
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "ransac-flow_amd")]
torch.set_num_threads(4)
import ref_loader, ref_oracle, parity_sweep
from rfx import synth, weights
out = dict(kind=ref_loader.kind(), staged=ref_loader.staged())
h = hashlib.sha256()
# (1) quick_start: the reference's CoarseAlign (variant A) + outil.RANSAC + nn.Modules, end to end on its own homography
r, ca = parity_sweep.oracle_pair("qs", 3, 128, 160)
for k in ("index1", "index2", "H", "inlier", "flow12"):
    h.update(np.ascontiguousarray(r[k]).tobytes())
# (2) the multi-homography ``while`` statement of evaluation/evalHpatch/evaluation.py + PredFlowMask compiled out of the script
sds = parity_sweep.state_dicts(parity_sweep.MULTIH_MATCH_STD)
cb = ref_oracle.CoarseAlignOracle(sds["trunk"], 3, 300, 0.05, 128, 1.2, variant="B", sample_fn=lambda n, it: parity_sweep.draw(9, n, it))
cb.setPair(*synth.make_pair(128, 160, seed=9, homography=True))
o = ref_oracle.multi_h_loop(cb, dict(feat=sds["feat"], flow=sds["flow"], match=sds["match"]), max_coarse=3, mask_region_th=0.01)
out["nbH"] = len(o["H"])
for a in o["H"] + o["flowDown8"] + o["matchDown8"] + o["masks"]:
    h.update(np.ascontiguousarray(a).tobytes())
# (3) the KITTI ``while True`` statement, get_info, remove_small_cc
ck = ref_oracle.CoarseAlignOracle(sds["trunk"], 3, 300, 0.05, 160, 1.2, variant="B", sample_fn=lambda n, it: parity_sweep.draw(12, n, it))
Is, It = synth.make_pair(96, 312, seed=12, homography=True, amp=0.03)
ck.setPair(Is, It)
k = ref_oracle.multi_h_loop_kitti(ck, dict(feat=sds["feat"], flow=sds["flow"], match=sds["match"]), Is, It, 128, mask_region_th=0.01, cc_th=0.002)
out["nbH_kitti"] = len(k["H"])
for a in k["H"] + k["flowD2"] + k["flowDown8"] + k["matchDown8"] + k["masks"]:
    h.update(np.ascontiguousarray(a).tobytes())
# (4) the offline assembly functions of the three getResults.py scripts resolve
for rel, names in (("evaluation/evalHpatch/getResults.py", ["getFlow_all", "getFlow_onlyCoarse"]), ("evaluation/evalCorr/getResults.py", ["getFlow", "getFlow_Coarse"])):
    assert all(callable(f) for f in ref_loader.script_functions(rel, names).values())
out["sha"] = h.hexdigest()
print("PROBE" + json.dumps(out))
'''


@pytest.mark.reference
def test_staged_reference_equals_the_source_tree(tmp_path):
    """oracle/make_ref.py byte-compiles the reference from where it lies; the compiled tree (what the GPU box gets) must BE the
    reference: the same probe -- variant-A alignment end to end, the Hpatch multi-homography loop statement, the KITTI loop
    statement, each on fresh seeds -- run once on /root/reference and once on a freshly staged tree gives bit-identical outputs."""
    import json
    import subprocess
    if not os.path.isdir("/root/reference/utils"):
        pytest.skip("needs the reference SOURCE tree (authoring container) to stage from")
    import make_ref
    staged = make_ref.build(out=str(tmp_path / "_ref"), verbose=False)
    assert not any(f.endswith(".py") for _, _, fs in os.walk(staged) for f in fs)          # compiled form only: no source travels
    res = {}
    for name, root in (("source", "/root/reference"), ("staged", staged)):
        r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + _STAGED_PROBE], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, RFX_REFERENCE_ROOT=root))
        assert r.returncode == 0, r.stderr[-2000:]
        res[name] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("PROBE")][0][5:])
    assert res["source"]["staged"] is False and res["staged"]["staged"] is True
    assert res["source"]["nbH"] >= 1 and res["source"]["nbH_kitti"] >= 1
    assert {k: v for k, v in res["source"].items() if k not in ("kind", "staged")} == \
           {k: v for k, v in res["staged"].items() if k not in ("kind", "staged")}


def test_product_path_never_touches_the_oracle_or_the_staged_reference():
    """Nothing under ransac-flow_amd/ (the product) may import, open or mention oracle/, oracle/_ref, ref_loader, ref_oracle,
    restate or parity_sweep; bench.py may do so only in its CPU legs (functions that run in child processes / after the timed
    region)."""
    import re
    bad = re.compile(r"\b(oracle|_ref\b|ref_loader|ref_oracle|restate|parity_sweep|make_ref|/root/reference)")
    hits = []
    pkg = os.path.join(ROOT, "ransac-flow_amd")
    for d, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                for i, ln in enumerate(open(os.path.join(d, f), errors="replace"), 1):
                    code = ln.split("#", 1)[0] if f.endswith(".py") else ln
                    if bad.search(code) and "import" in code or re.search(r"(open|load|join)\(.*(oracle|_ref)", code):
                        hits.append("%s:%d: %s" % (os.path.relpath(os.path.join(d, f), ROOT), i, ln.strip()))
    assert not hits, "\n".join(hits)
    # and the only python modules that import the oracle side are tests/, oracle/ itself, bench.py and __graft_entry__.py
    importers = set()
    for d, _, fs in os.walk(ROOT):
        if any(p in d for p in ("gpurun_out", ".git", "__pycache__")):
            continue
        for f in fs:
            if f.endswith(".py") and re.search(r"^\s*(import|from)\s+(ref_loader|ref_oracle|restate|parity_sweep|make_ref)\b",
                                               open(os.path.join(d, f), errors="replace").read(), re.M):
                importers.add(os.path.relpath(os.path.join(d, f), ROOT).split(os.sep)[0])
    assert importers <= {"tests", "oracle", "bench.py", "__graft_entry__.py", "scripts"}, importers
