"""Generates tests/golden/*.npz by running the REAL reference (/root/reference, imported under the stubs
of oracle/ref_loader.py) on seeded synthetic inputs.  Authoring-container only; the fixtures are committed
so that the oracle restatement and the HIP path can be checked against reference outputs on any machine
(the reference ships no golden vectors of its own for this path).

    python tests/golden/make_golden.py

Weights are not stored: they are regenerated from rfx.weights with the seeds recorded in each fixture.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))

import ref_loader  # noqa: E402
from rfx import weights, synth  # noqa: E402


def lattice_matches(seed, rows=30, cols=40, outlier=0.6):
    """SURVEY 8d stage-level RANSAC input: target lattice, planted H, 60% outliers."""
    R = ref_loader.load()
    g = torch.Generator().manual_seed(seed)
    feat = torch.zeros(1, 1, rows, cols)
    W, Hh = R["outil"].getWHTensor(feat)
    n = rows * cols
    m2 = torch.stack((Hh, W, torch.ones(n)), 1)
    Ht = torch.tensor([[1.05, .02, .03], [-.01, .97, -.02], [.01, .02, 1.0]])
    m1 = m2 @ Ht.t()
    m1 = m1 / m1[:, 2:]
    out = torch.rand(n, generator=g) < outlier
    m1[out] = m2[torch.randint(n, (int(out.sum()),), generator=g)]
    return m1.contiguous(), m2.contiguous()


def gen_ransac():
    R = ref_loader.load()
    outil = R["outil"]
    cases = {}
    for seed in range(4):
        m1, m2 = lattice_matches(seed)
        nb_iter = [1000, 250, 1000, 2050][seed]
        torch.manual_seed(100 + seed)
        samples = torch.randint(len(m1), (nb_iter, 4))
        # run the reference's RANSAC with its own randint producing exactly `samples`
        torch.manual_seed(100 + seed)
        Hb, cnt, inl, m2in = outil.RANSAC(nb_iter, m1, m2, 0.05, 4, outil.Homography)
        uniq = samples[~((samples[:, 0] == samples[:, 1]) | (samples[:, 0] == samples[:, 2]) | (samples[:, 0] == samples[:, 3]) |
                         (samples[:, 1] == samples[:, 2]) | (samples[:, 1] == samples[:, 3]) | (samples[:, 2] == samples[:, 3]))]
        H21, counts = outil.ScoreRANSAC(m1, m2, 0.05, uniq[:300], outil.Homography)
        cases["m1_%d" % seed] = m1.numpy()
        cases["m2_%d" % seed] = m2.numpy()
        cases["samples_%d" % seed] = samples.numpy()
        cases["H_%d" % seed] = Hb
        cases["count_%d" % seed] = np.asarray(cnt)
        cases["inlier_%d" % seed] = inl
        cases["score_H_%d" % seed] = H21.numpy()
        cases["score_counts_%d" % seed] = counts.numpy()
    # abort case: a negative tolerance makes every count 0 -> the first full chunk aborts (utils/outil.py:145-146)
    g = torch.Generator().manual_seed(7)
    m1 = torch.cat((torch.rand(50, 2, generator=g) * 2 - 1, torch.ones(50, 1)), 1)
    m2 = torch.cat((torch.rand(50, 2, generator=g) * 2 - 1, torch.ones(50, 1)), 1)
    torch.manual_seed(5)
    samples = torch.randint(50, (400, 4))
    torch.manual_seed(5)
    res = outil.RANSAC(400, m1, m2, -1.0, 4, outil.Homography)
    assert res[0] is None
    cases["abort_m1"], cases["abort_m2"], cases["abort_samples"] = m1.numpy(), m2.numpy(), samples.numpy()
    np.savez_compressed(os.path.join(HERE, "ransac.npz"), **cases)
    print("ransac.npz", {k: v.shape for k, v in cases.items() if k.startswith("H_") or k.startswith("count")})


def gen_mutual():
    R = ref_loader.load()
    g = torch.Generator().manual_seed(11)
    C, nA, nB = 64, 333, 97
    A = F.normalize(torch.relu(torch.randn(C, nA, generator=g)), dim=0)
    B = F.normalize(torch.relu(torch.randn(C, nB, generator=g)), dim=0)
    B[:, 5:20] = 0  # masked columns
    i1, i2 = R["outil"].mutualMatching(A, B)
    np.savez_compressed(os.path.join(HERE, "mutual.npz"), A=A.numpy(), B=B.numpy(), index1=i1.numpy(), index2=i2.numpy())
    print("mutual.npz", len(i1))


def gen_nets():
    """Small-shape stage outputs of the reference modules with rfx.weights (randomize_bn=True)."""
    R = ref_loader.load()
    model = R["model"]
    out = {}
    g = torch.Generator().manual_seed(21)
    # trunk on 64x96
    r50 = ref_loader.quiet(R["resnet50"].resnet50)
    sd = weights.resnet50_trunk_sd(seed=31, randomize_bn=True)
    missing = r50.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    trunk = torch.nn.Sequential(r50.conv1, r50.bn1, r50.relu, r50.maxpool, r50.layer1, r50.layer2, r50.layer3).eval()
    x = torch.randn(1, 3, 64, 96, generator=g)
    with torch.no_grad():
        out["trunk_in"] = x.numpy()
        out["trunk_out"] = trunk(x).numpy()
    # fine nets on 48x64
    fe = ref_loader.quiet(model.FeatureExtractor)
    fe.load_state_dict(weights.feature_extractor_sd(seed=32, randomize_bn=True))
    fe.eval()
    nf = ref_loader.quiet(model.NetFlowCoarse, 7)
    nf.load_state_dict(weights.net_flow_coarse_sd(seed=33, randomize_bn=True))
    nf.eval()
    nm = ref_loader.quiet(model.NetMatchability, 7)
    nm.load_state_dict(weights.net_matchability_sd(seed=34, randomize_bn=True, last_std=0.02))
    nm.eval()
    corr = model.CorrNeigh(7).eval()
    xa = torch.rand(1, 3, 48, 64, generator=g)
    xb = torch.rand(1, 3, 48, 64, generator=g)
    with torch.no_grad():
        fa, fb = F.normalize(fe(xa)), F.normalize(fe(xb))
        c12 = corr(fa, fb)
        flow = nf(c12, False)
        flow8 = nf(c12, True)
        mt = nm(c12, False)
        grid = torch.cat((torch.linspace(-1, 1, 64).view(1, 1, -1, 1).expand(1, 48, 64, 1),
                          torch.linspace(-1, 1, 48).view(1, -1, 1, 1).expand(1, 48, 64, 1)), dim=3)
        fg, fc = model.predFlowCoarse(c12, nf, grid, True)
    out.update(fine_xa=xa.numpy(), fine_xb=xb.numpy(), fine_fa=fa.numpy(), fine_fb=fb.numpy(), fine_corr=c12.numpy(),
               fine_flow=flow.numpy(), fine_flow8=flow8.numpy(), fine_match=mt.numpy(), fine_flowGrad=fg.numpy(),
               fine_flowCoarse=fc.numpy())
    # warp / sample
    Hm = torch.tensor([[[1.02, 0.03, 0.01], [-0.02, 0.98, 0.03], [0.01, -0.02, 1.0]]])
    warper = R["kornia_geometry"].HomographyWarper(48, 64)
    wg = warper.warp_grid(Hm)
    out["warp_H"] = Hm.numpy()
    out["warp_grid"] = wg.numpy()
    out["warp_sample"] = F.grid_sample(xa, wg).numpy()
    np.savez_compressed(os.path.join(HERE, "nets.npz"), **out)
    print("nets.npz", {k: v.shape for k, v in out.items()})


def gen_config1():
    """BASELINE config 1: quick_start path on one 240x320 synthetic pair, CPU reference, nbIter=100."""
    R = ref_loader.load()
    model = R["model"]
    I1, I2 = synth.make_pair(240, 320, seed=0)
    ca = ref_loader.quiet(R["CoarseAlignA"], 7, 100, 0.05, "Homography", 320, scaleR=1.2)
    trunk_sd = weights.resnet50_trunk_sd(seed=0)
    # the coarse model's net is Sequential(conv1,bn1,relu,maxpool,layer1,layer2,layer3): keys are indices
    names = ["conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3"]
    remap = {}
    for k, v in trunk_sd.items():
        top, rest = k.split(".", 1)
        remap["%d.%s" % (names.index(top), rest)] = v
    ca.net.load_state_dict(remap)
    ca.net.eval()
    ca.setSource(I1)
    ca.setTarget(I2)
    torch.manual_seed(123)
    bestPrm, inlierMask = ca.getCoarse(np.zeros((ca.It.size[1], ca.It.size[0])))
    # re-derive the matches the reference used (same call it makes inside getCoarse with an all-ones mask)
    i1, i2 = R["outil"].mutualMatching(ca.featsMultiScale, ca.featt.contiguous().view(1024, -1))
    fe = ref_loader.quiet(model.FeatureExtractor)
    fe.load_state_dict(weights.feature_extractor_sd(seed=1))
    fe.eval()
    nf = ref_loader.quiet(model.NetFlowCoarse, 7)
    nf.load_state_dict(weights.net_flow_coarse_sd(seed=2))
    nf.eval()
    corr = model.CorrNeigh(7).eval()
    h, w = ca.It.size[1], ca.It.size[0]
    warper = R["kornia_geometry"].HomographyWarper(h, w)
    with torch.no_grad():
        Hm = torch.from_numpy(bestPrm).unsqueeze(0)
        flowCoarse = warper.warp_grid(Hm)
        img1_coarse = F.grid_sample(ca.IsTensor, flowCoarse)
        feat1 = F.normalize(fe(img1_coarse))
        feat2 = F.normalize(fe(ca.ItTensor))
        corr12 = corr(feat1, feat2)
        flowDown = nf(corr12, False)
        gridX = torch.linspace(-1, 1, steps=w).view(1, 1, -1, 1).expand(1, h, w, 1)
        gridY = torch.linspace(-1, 1, steps=h).view(1, -1, 1, 1).expand(1, h, w, 1)
        grid = torch.cat((gridX, gridY), dim=3)
        flowUp = F.interpolate(flowDown, size=(h, w), mode="bilinear").permute(0, 2, 3, 1) + grid
        flow12 = F.grid_sample(flowCoarse.permute(0, 3, 1, 2), flowUp).permute(0, 2, 3, 1).contiguous()
    np.savez_compressed(os.path.join(HERE, "config1.npz"), H=bestPrm, inlierMask=inlierMask, index1=i1.numpy(),
                        index2=i2.numpy(), flowDown=flowDown.numpy(), flow12_sub=flow12[:, ::8, ::8].numpy(),
                        feat_t_sub=ca.featt[0, ::64].numpy(), scaleList=np.asarray(ca.scaleList))
    print("config1.npz: matches", len(i1), "inliers", int(inlierMask.sum()), "H", bestPrm.ravel()[:3])


def gen_assemble():
    """Offline flow assembly: the reference's own getFlow* functions, compiled out of the three
    getResults.py scripts, run on seeded .npy files written in the on-disk format they expect."""
    import tempfile
    kg = ref_loader.load()["kornia_geometry"]
    fh = ref_loader.script_functions("evaluation/evalHpatch/getResults.py", ["getFlow_all", "getFlow_onlyCoarse"])
    fc = ref_loader.script_functions("evaluation/evalCorr/getResults.py", ["getFlow", "getFlow_Coarse"])
    fk = ref_loader.script_functions("evaluation/evalKITTI/getResults.py",
                                     ["getFlow_all", "remove_small_cc", "interpolate_flow_match"])
    n, hd, wd = 3, 10, 14
    flowDown, flowd2, param, md = synth.assembly_arrays(11, n, hd, wd)
    d = tempfile.mkdtemp()
    fine, coarse, maskp, kit = [os.path.join(d, x) for x in ("fine", "coarse", "mask", "kitti")]
    for x in (fine, coarse, maskp, kit):
        os.makedirs(x)
    H8, W8 = hd * 8, wd * 8
    np.save(os.path.join(fine, "flow_7_3H.npy"), flowDown)
    np.save(os.path.join(coarse, "flow_7_3H.npy"), param)
    np.save(os.path.join(fine, "mask_7_3H.npy"), md)
    np.save(os.path.join(maskp, "maskBG_7_3H.npy"), np.ones((H8, W8), bool))
    np.save(os.path.join(kit, "Homograpy_7_3.npy"), param)
    np.save(os.path.join(kit, "Finetune_D2_7_3.npy"), flowd2)
    np.save(os.path.join(kit, "Finetune_7_3.npy"), flowDown)
    np.save(os.path.join(kit, "Finetune_Mask_7_3.npy"), md)
    np.save(os.path.join(kit, "BG_7_3H.npy"), np.ones((H8, W8), bool))
    out = dict(seed=11, n=n, hd=hd, wd=wd)
    # Hpatch: arbitrary output size (the script uses minSize x minSize), no cycle check
    oh, ow = 72, 100
    grid = torch.cat((torch.linspace(-1, 1, ow).view(1, 1, -1, 1).expand(1, oh, ow, 1),
                      torch.linspace(-1, 1, oh).view(1, -1, 1, 1).expand(1, oh, ow, 1)), dim=3)
    for tag, th, multiH in (("a", 0.45, True), ("b", 0.45, False), ("c", 0.7, True)):
        r = fh["getFlow_all"](7, fine, coarse, os.listdir(fine), multiH, kg.HomographyWarper(oh, ow), grid, th, ow, oh)
        out["hpatch_%s" % tag] = r.numpy()
        out["hpatch_%s_cfg" % tag] = np.asarray([th, float(multiH), oh, ow])
    out["hpatch_missing"] = np.asarray(len(fh["getFlow_all"](8, fine, coarse, os.listdir(fine), True, None, grid, 0.5, ow, oh)))
    out["hpatch_coarse"] = fh["getFlow_onlyCoarse"](7, fine, coarse, os.listdir(fine), True,
                                                   kg.HomographyWarper(oh, ow), grid, 0.5, ow, oh).numpy()
    for tag, th, multiH in (("a", 0.2, True), ("b", 0.2, False), ("c", 0.35, True)):
        rf, rm = fc["getFlow"](7, fine, os.listdir(fine), coarse, maskp, multiH, th)
        out["corr_%s_flow" % tag] = rf.numpy()
        out["corr_%s_match" % tag] = rm.numpy()
        out["corr_%s_cfg" % tag] = np.asarray([th, float(multiH)])
    grid8 = torch.cat((torch.linspace(-1, 1, W8).view(1, 1, -1, 1).expand(1, H8, W8, 1),
                       torch.linspace(-1, 1, H8).view(1, -1, 1, 1).expand(1, H8, W8, 1)), dim=3)
    for tag, th, cc, interp, multiH in (("a", 0.2, 0.0, False, True), ("b", 0.25, 0.01, False, True),
                                        ("c", 0.25, 0.02, True, True), ("d", 0.2, 0.01, True, False)):
        r = fk["getFlow_all"](7, kit, 3, "Finetune", kg.HomographyWarper(H8, W8), multiH, grid8, th, cc, interp)
        out["kitti_%s" % tag] = r.numpy()
        out["kitti_%s_cfg" % tag] = np.asarray([th, cc, float(interp), float(multiH)])
    np.savez_compressed(os.path.join(HERE, "assemble.npz"), **out)
    print("assemble.npz:", sorted(k for k in out if not k.endswith("_cfg"))[:6], "...")


def _ref_networks(match_std=0.02, randomize_bn=False, seeds=(1, 2, 3)):
    """The reference's own four modules (model/model.py) carrying rfx.weights state dicts."""
    R = ref_loader.load()
    model = R["model"]
    fe = ref_loader.quiet(model.FeatureExtractor)
    fe.load_state_dict(weights.feature_extractor_sd(seed=seeds[0], randomize_bn=randomize_bn))
    nf = ref_loader.quiet(model.NetFlowCoarse, 7)
    nf.load_state_dict(weights.net_flow_coarse_sd(seed=seeds[1], randomize_bn=randomize_bn))
    nm = ref_loader.quiet(model.NetMatchability, 7)
    nm.load_state_dict(weights.net_matchability_sd(seed=seeds[2], randomize_bn=randomize_bn, last_std=match_std))
    net = {"netFeatCoarse": fe, "netCorr": model.CorrNeigh(7), "netFlowCoarse": nf, "netMatch": nm}
    for m in net.values():
        m.eval()
    return net


def _ref_coarse_b(nbScale, nbIter, minSize, scaleR, trunk_seed=0):
    """The reference's evaluation-side CoarseAlign (evaluation/evalHpatch/coarseAlignFeatMatch.py:35-179), trunk
    weights from rfx.weights (argument order: nbScale, nbIter, tolerance, transform, minSize, segId, segFg, scaleR,
    imageNet, segNet)."""
    R = ref_loader.load()
    ca = ref_loader.quiet(R["CoarseAlignB"], nbScale, nbIter, 0.05, "Homography", minSize, 2, False, scaleR, True, False)
    names = ["conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3"]
    remap = {}
    for k, v in weights.resnet50_trunk_sd(seed=trunk_seed).items():
        top, rest = k.split(".", 1)
        remap["%d.%s" % (names.index(top), rest)] = v
    ca.net.load_state_dict(remap)
    ca.net.eval()
    return ca


def _grid(h, w):
    return torch.cat((torch.linspace(-1, 1, w).view(1, 1, -1, 1).expand(1, h, w, 1),
                      torch.linspace(-1, 1, h).view(1, -1, 1, 1).expand(1, h, w, 1)), dim=3)


def gen_predflowmask():
    """The reference's own PredFlowMask functions -- evaluation/evalHpatch/evaluation.py:23-55 and the KITTI variant
    evaluation/evalKITTI/evaluation.py:49-81 -- compiled out of the two scripts and run on seeded inputs with the
    reference's model.* modules.  Pins restate.pred_flow_mask / pred_flow_mask_kitti."""
    R = ref_loader.load()
    fh = ref_loader.script_functions("evaluation/evalHpatch/evaluation.py", ["PredFlowMask"])["PredFlowMask"]
    fk = ref_loader.script_functions("evaluation/evalKITTI/evaluation.py", ["PredFlowMask", "remove_small_cc"])
    net = _ref_networks(match_std=0.02, randomize_bn=True, seeds=(41, 42, 43))
    out = dict(seeds=np.asarray([41, 42, 43]), match_std=np.asarray(0.02))
    I1, I2 = synth.make_pair(96, 128, seed=5, homography=True)
    tt = lambda im: torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)[None]
    Is, It = tt(I1), tt(I2)
    Hm = torch.tensor([[[1.03, 0.02, 0.04], [-0.02, 0.97, 0.05], [0.015, -0.01, 1.0]]])
    h, w = 96, 128
    grid = _grid(h, w)
    with torch.no_grad():
        flowCoarse = R["kornia_geometry"].HomographyWarper(h, w).warp_grid(Hm)
        featt = F.normalize(net["netFeatCoarse"](It))
        f12, match, fd8, md8 = fh(Is, featt, flowCoarse, grid, net)
        out.update(hp_H=Hm.numpy(), hp_flow12=f12.numpy(), hp_match=match, hp_flowDown8=fd8, hp_matchDown8=md8)
        # KITTI variant: both images already sampled; output grid LARGER than the input images (the second call of
        # evalKITTI/evaluation.py:302 passes grid_org with resize-resolution tensors)
        IsSample = F.grid_sample(Is, flowCoarse)
        gh, gw = 120, 168
        f12, match, fd8, md8 = fk["PredFlowMask"](IsSample, It, flowCoarse, _grid(gh, gw), net)
        out.update(ki_flow12=f12.numpy(), ki_match=match, ki_flowDown8=fd8.numpy(), ki_matchDown8=md8.numpy(),
                   ki_grid_hw=np.asarray([gh, gw]))
    # remove_small_cc (evalKITTI/evaluation.py:85-100) on a seeded blob map
    g = torch.Generator().manual_seed(9)
    blob = F.avg_pool2d(torch.rand(1, 1, 66, 90, generator=g), 3, 1)[0, 0].numpy() * 1.9
    blob = np.clip(blob, 0, 1).astype(np.float32)
    out["cc_in"] = blob.copy()
    out["cc_out"] = fk["remove_small_cc"](blob.copy(), 0.99, 0.002)
    out["cc_cfg"] = np.asarray([0.99, 0.002])
    np.savez_compressed(os.path.join(HERE, "predflowmask.npz"), **out)
    print("predflowmask.npz", {k: np.asarray(v).shape for k, v in out.items()})


def gen_coarse_b():
    """The reference's variant-B CoarseAlign.setPair / getCoarse (evaluation/evalHpatch/coarseAlignFeatMatch.py:102-179)
    on a config-1-sized pair with three different masks.  Pins restate.CoarseAlignOracle(variant="B")."""
    I1, I2 = synth.make_pair(240, 320, seed=6, homography=True)
    ca = _ref_coarse_b(nbScale=5, nbIter=400, minSize=240, scaleR=1.5)
    ca.setPair(I1, I2)
    h, w = ca.It.size[1], ca.It.size[0]
    out = dict(cfg=np.asarray([5, 400, 240, 1.5]), hw=np.asarray([h, w]),
               W1=ca.W1MutualMatch.numpy(), H1=ca.H1MutualMatch.numpy(), W2=ca.W2MutualMatch.numpy(),
               H2=ca.H2MutualMatch.numpy(), W2I=ca.W2MutualMatchInt.numpy(), H2I=ca.H2MutualMatchInt.numpy())
    masks = [np.zeros((h, w), np.float32), np.zeros((h, w), np.float32), np.zeros((h, w), np.float32),
             np.ones((h, w), np.float32)]
    masks[1][:, : w // 2] = 1                      # left half already explained
    yy, xx = np.mgrid[0:h, 0:w]
    masks[2][((yy - h / 2) ** 2 + (xx - w / 2) ** 2) < (h / 3) ** 2] = 1   # a disc
    for k, Mt in enumerate(masks):
        torch.manual_seed(300 + k)
        Hb = ca.getCoarse(Mt)
        out["mask_%d" % k] = Mt
        out["H_%d" % k] = np.zeros((0,), np.float32) if Hb is None else Hb      # mask 3 excludes everything -> None
    np.savez_compressed(os.path.join(HERE, "coarse_b.npz"), **out)
    print("coarse_b.npz: matches", len(out["W1"]), "H0", out["H_0"].ravel()[:3], "none for full mask:", out["H_3"].size == 0)


MULTIH_MATCH_STD = 3.0      # saturating matchability head: the explained-region mask of the multi-H loop grows


class _RoundRecorder:
    """Per-round state of a reference driver loop, observed from outside (nothing in the loop is changed): the mask every
    ``coarseModel.getCoarse(fgMask)`` call receives, the number of matches every ``outil.RANSAC`` call sees and the H it
    returns, and the full-resolution matchability the accept test reads (the ``matchFine`` of PredFlowMask for Hpatch; the
    output of ``remove_small_cc`` for KITTI).  Stored as <tag>_round_fg (rounds,h,w) packed bits, <tag>_round_n,
    <tag>_round_H, <tag>_round_match: what a teacher-forced test needs to replay round k from the reference's own state."""

    def __init__(self, R, ca):
        self.fg, self.n, self.H, self.match = [], [], [], []
        self.outil, self.ca = R["outil"], ca
        self._ransac, self._get = self.outil.RANSAC, ca.getCoarse

        def ransac(nbIter, match1, match2, *a, **k):
            self.n.append(len(match1))
            r = self._ransac(nbIter, match1, match2, *a, **k)
            self.H.append(np.zeros((3, 3), np.float32) if r[0] is None else np.asarray(r[0], dtype=np.float32))
            return r

        def get(Mt):
            self.fg.append(np.asarray(Mt, dtype=np.float32).copy())
            k = len(self.n)
            r = self._get(Mt)
            if len(self.n) == k:              # fewer than 4 matches: RANSAC was not called
                self.n.append(-1)
                self.H.append(np.zeros((3, 3), np.float32))
            return r
        self.outil.RANSAC, ca.getCoarse = ransac, get

    def wrap_match(self, fn, out_index):
        """Records element ``out_index`` of fn's result (None: the result itself) as the round's matchability map."""
        def rec(*a, **k):
            r = fn(*a, **k)
            m = r if out_index is None else r[out_index]
            self.match.append(np.asarray(m, dtype=np.float32).copy())
            return r
        return rec

    def close(self):
        self.outil.RANSAC = self._ransac
        self.ca.getCoarse = self._get

    def save(self, out, tag, every=1):
        out["%s_round_fg" % tag] = np.packbits(np.stack(self.fg) > 0.5, axis=-1)
        out["%s_round_n" % tag] = np.asarray(self.n)
        out["%s_round_H" % tag] = np.stack(self.H)
        out["%s_round_match" % tag] = np.stack(self.match[every - 1::every])


def gen_multi_h():
    """The reference's multi-homography driver loop itself -- the ``while nbCoarse <= args.maxCoarse`` statement of
    evaluation/evalHpatch/evaluation.py:211-243, compiled out of the script and executed on the variables its module
    level sets up (:172-208) -- with the reference's CoarseAlign (variant B), PredFlowMask and model.* modules.
    Pins restate.multi_h_loop."""
    import types
    R = ref_loader.load()
    pfm = ref_loader.script_functions("evaluation/evalHpatch/evaluation.py", ["PredFlowMask"])["PredFlowMask"]
    loop = ref_loader.script_loop("evaluation/evalHpatch/evaluation.py", "nbCoarse <= args.maxCoarse")
    net = _ref_networks(match_std=MULTIH_MATCH_STD)
    out = dict(match_std=np.asarray(MULTIH_MATCH_STD))
    for tag, seed, maxCoarse, th in (("a", 7, 2, 0.01), ("b", 8, 4, 0.02)):
        I1, I2 = synth.make_pair(240, 320, seed=seed, homography=True)
        ca = _ref_coarse_b(nbScale=3, nbIter=300, minSize=240, scaleR=1.2)
        ca.setPair(I1, I2)
        Itw, Ith = ca.It.size
        rounds = _RoundRecorder(R, ca)
        with torch.no_grad():
            featt = F.normalize(net["netFeatCoarse"](ca.ItTensor))
            ns = dict(args=types.SimpleNamespace(maxCoarse=maxCoarse, maskRegionTh=th), coarseModel=ca, network=net,
                      featt=featt, grid=_grid(Ith, Itw), warper=R["kornia_geometry"].HomographyWarper(Ith, Itw),
                      It_bg=np.ones((Ith, Itw), dtype=np.float32), Mask=np.zeros((Ith, Itw), dtype=np.float32),
                      Coarse_Flow_Tensor=[], Fine_Flow_Tensor=[], Fine_Mask_Tensor=[], nbCoarse=0,
                      PredFlowMask=rounds.wrap_match(pfm, 1))
            torch.manual_seed(500 + seed)
            loop(ns)
        rounds.close()
        rounds.save(out, tag)
        n = ns["nbCoarse"]
        out["%s_cfg" % tag] = np.asarray([seed, maxCoarse, th, 500 + seed])
        out["%s_nb" % tag] = np.asarray(n)
        out["%s_H" % tag] = np.concatenate(ns["Coarse_Flow_Tensor"], axis=0)
        out["%s_flowDown8" % tag] = np.concatenate(ns["Fine_Flow_Tensor"], axis=0)
        out["%s_matchDown8" % tag] = np.concatenate(ns["Fine_Mask_Tensor"], axis=0)
        out["%s_mask" % tag] = ns["Mask"]
        print("multi_h %s: %d homographies, explained %.3f" % (tag, n, float(ns["Mask"].mean())))
    np.savez_compressed(os.path.join(HERE, "multi_h.npz"), **out)


def gen_kitti_loop():
    """The reference's two-resolution KITTI driver itself -- the ``while True`` statement of
    evaluation/evalKITTI/evaluation.py:270-336 compiled out of the script, with the script's own get_info / PredFlowMask
    / remove_small_cc functions (:29-36, :49-81, :85-100), outil.resizeImg (utils/outil.py:6-19) and the reference's
    CoarseAlign (variant B) -- on a KITTI-shaped (3.3:1) synthetic pair.  Pins restate.multi_h_loop_kitti (SURVEY 8f1,
    BASELINE config 5)."""
    import types
    R = ref_loader.load()
    fk = ref_loader.script_functions("evaluation/evalKITTI/evaluation.py", ["PredFlowMask", "remove_small_cc", "get_info"])
    import torchvision.transforms as tvt                      # the stub of ref_loader
    fk["get_info"].__globals__["transforms"] = tvt
    loop = ref_loader.script_loop("evaluation/evalKITTI/evaluation.py", "True")
    net = _ref_networks(match_std=MULTIH_MATCH_STD)
    out = dict(match_std=np.asarray(MULTIH_MATCH_STD))
    for tag, seed, fine, cc_th, th in (("a", 12, 128, 0.0, 0.01), ("b", 13, 112, 0.002, 0.02)):
        Is, It = synth.make_pair(96, 312, seed=seed, homography=True, amp=0.03)
        ca = _ref_coarse_b(nbScale=3, nbIter=300, minSize=160, scaleR=1.2)
        get_info = fk["get_info"]
        It_resize = R["outil"].resizeImg(It, 8, fine)
        It_d2 = R["outil"].resizeImg(It, 8, fine // 2)
        with torch.no_grad():
            w_org, h_org, tensor_org, grid_org, warper_org = get_info(It)
            _, _, tensor_s, _, _ = get_info(Is)
            w_resize, h_resize, tensor_resize, grid_resize, warper_resize = get_info(It_resize)
            w_d2, h_d2, tensor_d2, grid_d2, warper_d2 = get_info(It_d2)
            ca.setPair(Is, It)
        rounds = _RoundRecorder(R, ca)
        ns = dict(args=types.SimpleNamespace(cc_th=cc_th, maskRegionTh=th), coarseModel=ca, network=net,
                  It_bg=np.ones((h_org, w_org), dtype=np.float32), Mask=np.zeros((h_org, w_org), dtype=np.float32),
                  warper_d2=warper_d2, warper_resize=warper_resize, tensor_s=tensor_s, tensor_d2=tensor_d2,
                  tensor_resize=tensor_resize, grid_d2=grid_d2, grid_resize=grid_resize, grid_org=grid_org,
                  Homography=[], Org_D2=[], Finetune_D2=[], Org_Mask=[], Finetune_Mask=[], Org=[], Finetune=[], nbCoarse=0,
                  PredFlowMask=fk["PredFlowMask"], remove_small_cc=rounds.wrap_match(fk["remove_small_cc"], None))
        torch.manual_seed(700 + seed)
        loop(ns)
        rounds.close()
        rounds.save(out, tag)
        n = ns["nbCoarse"]
        cat = lambda lst: torch.cat(lst, dim=0).numpy().astype(np.float32)
        out["%s_cfg" % tag] = np.asarray([seed, fine, cc_th, th, 700 + seed])
        out["%s_sizes" % tag] = np.asarray([h_org, w_org, h_resize, w_resize, h_d2, w_d2])
        out["%s_nb" % tag] = np.asarray(n)
        out["%s_H" % tag] = cat(ns["Homography"])
        out["%s_flowD2" % tag] = cat(ns["Finetune_D2"])
        out["%s_flowDown8" % tag] = cat(ns["Finetune"])
        out["%s_matchDown8" % tag] = cat(ns["Finetune_Mask"])
        out["%s_mask" % tag] = ns["Mask"]
        print("kitti loop %s: %d homographies, explained %.3f, sizes" % (tag, n, float(ns["Mask"].mean())), out["%s_sizes" % tag])
    np.savez_compressed(os.path.join(HERE, "kitti_loop.npz"), **out)


def gen_seg():
    """the REFERENCE's SegNet.getSky (segNet/segEval.py:23-43 on segNet/segModel.py, imported by ref_loader.load_seg)
    on one synthetic 96x128 image with seeded random-init weights (rfx/weights.py seg_encoder_sd(4) / seg_decoder_sd(5, logit_std=0.003: class scores that are not saturated), BN
    statistics perturbed): the class map, the averaged class probabilities on a pixel lattice, and the two masks the scripts
    use (segId of the most frequent class, segFg True / False)."""
    import tempfile
    S = ref_loader.load_seg()
    enc, dec = weights.seg_encoder_sd(4, randomize_bn=True), weights.seg_decoder_sd(5, randomize_bn=True, logit_std=0.003)
    I1, _ = synth.make_pair(96, 128, seed=3)
    d = tempfile.mkdtemp()
    pe, pd_, pi = os.path.join(d, "enc.pth"), os.path.join(d, "dec.pth"), os.path.join(d, "img.png")
    torch.save(enc, pe)
    torch.save(dec, pd_)
    I1.save(pi)
    sn = ref_loader.quiet(S["segEval"].SegNet, pe, pd_, 1, True)
    with torch.no_grad():
        IT = sn.dataset_test.getImg(pi)
        seg_size = (IT["img_ori"].shape[0], IT["img_ori"].shape[1])
        scores = torch.zeros(1, 150, *seg_size)
        for img in IT["img_data"]:
            scores = scores + sn.net(img, segSize=seg_size) / 5                   # segEval.py:30-35
    pred = scores.max(dim=1)[1][0].numpy()
    seg_id = int(np.bincount(pred.reshape(-1)).argmax())
    sn.segId = seg_id
    fg = sn.getSky(pi)
    sn.segFg = False
    bg = sn.getSky(pi)
    assert np.array_equal(fg, 1 - (pred == seg_id)) and np.array_equal(bg, (pred == seg_id).astype(np.float32))
    print("seg: %d classes predicted, segId %d covers %.2f" % (np.unique(pred).size, seg_id, bg.mean()))
    np.savez_compressed(os.path.join(HERE, "seg.npz"), pred=pred.astype(np.uint8), scores_sub=scores[0, :, ::8, ::8].numpy(),
                        seg_id=np.asarray(seg_id), mask_fg=fg.astype(np.uint8), mask_bg=bg.astype(np.uint8),
                        sizes=np.asarray([x.shape[-2:] for x in IT["img_data"]]))


if __name__ == "__main__":
    torch.set_num_threads(8)
    only = sys.argv[1:]
    if only:                                   # python make_golden.py gen_multi_h gen_coarse_b ...
        for name in only:
            globals()[name]()
        sys.exit(0)
    gen_ransac()
    gen_mutual()
    gen_nets()
    gen_config1()
    gen_assemble()
    gen_predflowmask()
    gen_coarse_b()
    gen_multi_h()
    gen_kitti_loop()
    gen_seg()
