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


if __name__ == "__main__":
    torch.set_num_threads(8)
    gen_ransac()
    gen_mutual()
    gen_nets()
    gen_config1()
    gen_assemble()
