"""CPU tests of the drop-in module surface (ransac-flow_amd/dropin): names, signatures, state-dict keys,
sentinel behaviour that does not need a device.  Tests marked ``reference`` compare against the real
reference modules (authoring container only)."""
import inspect
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "ransac-flow_amd", "dropin")


@pytest.fixture(scope="module")
def dropin():
    """Import the drop-ins under private names so they do not shadow the oracle's reference modules."""
    import importlib.util
    sys.path.insert(0, DROPIN)
    mods = {}
    saved = {k: sys.modules.get(k) for k in ("outil", "model", "coarseAlignFeatMatch")}
    try:
        for name in ("outil", "model", "coarseAlignFeatMatch"):
            spec = importlib.util.spec_from_file_location(name, os.path.join(DROPIN, name + ".py"))
            m = importlib.util.module_from_spec(spec)
            sys.modules[name] = m
            spec.loader.exec_module(m)
            mods[name] = m
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        sys.path.remove(DROPIN)
    return mods


def test_model_state_dict_keys_and_checkpoint_format(dropin, tmp_path):
    from rfx import weights
    model = dropin["model"]
    nets = {"netFeatCoarse": model.FeatureExtractor(), "netCorr": model.CorrNeigh(7),
            "netFlowCoarse": model.NetFlowCoarse(7), "netMatch": model.NetMatchability(7)}
    sds = {"netFeatCoarse": weights.feature_extractor_sd(1), "netCorr": {}, "netFlowCoarse": weights.net_flow_coarse_sd(2),
           "netMatch": weights.net_matchability_sd(3)}
    ck = str(tmp_path / "ckpt.pth")
    torch.save(sds, ck)                                   # the dict-of-4-state-dicts format of train/train.py:293-297
    param = torch.load(ck)
    for key in param:                                     # the loop of quick_start/align2images.py:47-50
        nets[key].load_state_dict(param[key])
        nets[key].eval()
    assert set(nets["netFeatCoarse"].state_dict()) == set(sds["netFeatCoarse"])
    assert "gridX" not in nets["netFlowCoarse"].state_dict()          # plain attributes, as in the reference
    assert nets["netFlowCoarse"].gridX.view(7, 7)[0].tolist() == [-3, -2, -1, 0, 1, 2, 3]
    assert nets["netFlowCoarse"].gridY.view(7, 7)[:, 0].tolist() == [-3, -2, -1, 0, 1, 2, 3]
    with pytest.raises(RuntimeError):                     # parameters on the CPU: fails loudly, no fallback
        nets["netFeatCoarse"](torch.zeros(1, 3, 16, 16))
    nets["netFlowCoarse"].train()
    with pytest.raises(NotImplementedError):
        nets["netFlowCoarse"](torch.zeros(1, 49, 4, 4))


def test_outil_host_helpers(dropin):
    import restate
    outil = dropin["outil"]
    feat = torch.zeros(1, 4, 5, 7)
    W, H = outil.getWHTensor(feat)
    Wr, Hr = restate.get_wh(5, 7)
    assert torch.equal(W, Wr) and torch.equal(H, Hr)
    Wi, Hi = outil.getWHTensor_Int(feat)
    assert Wi.tolist()[:8] == [0] * 7 + [1] and Hi.tolist()[:8] == list(range(7)) + [0]
    import PIL.Image as Image
    assert outil.resizeImg(Image.new("RGB", (300, 200)), 16, 400).size == (608, 400)
    with pytest.raises(NotImplementedError):
        outil.Affine(None, None)
    with pytest.raises(RuntimeError):                     # CPU tensors are refused by the HIP ops
        outil.mutualMatching(torch.zeros(8, 4), torch.zeros(8, 4))


@pytest.mark.reference
def test_signatures_match_reference(dropin):
    import ref_loader
    R = ref_loader.load()
    ro, rm = R["outil"], R["model"]
    for name in ("resizeImg", "getWHTensor", "getWHTensor_Int", "mutualMatching", "Homography", "Prediction", "ScoreRANSAC"):
        assert str(inspect.signature(getattr(dropin["outil"], name))) == str(inspect.signature(getattr(ro, name))), name
    ours = list(inspect.signature(dropin["outil"].RANSAC).parameters)
    assert ours[:6] == list(inspect.signature(ro.RANSAC).parameters)           # + optional samples=
    for name in ("predFlowCoarse", "predFlowCoarseNoGrad", "predMatchability"):
        assert str(inspect.signature(getattr(dropin["model"], name))) == str(inspect.signature(getattr(rm, name))), name
    for cls in ("FeatureExtractor", "CorrNeigh", "NetFlowCoarse", "NetMatchability"):
        a = ref_loader.quiet(getattr(rm, cls), *([] if cls == "FeatureExtractor" else [7]))
        b = getattr(dropin["model"], cls)(*([] if cls == "FeatureExtractor" else [7]))
        assert list(a.state_dict()) == list(b.state_dict()), cls
        assert [tuple(v.shape) for v in a.state_dict().values()] == [tuple(v.shape) for v in b.state_dict().values()]
        assert str(inspect.signature(a.forward)) == str(inspect.signature(b.forward)), cls
    ca = dropin["coarseAlignFeatMatch"]
    pa = list(inspect.signature(R["CoarseAlignA"].__init__).parameters)
    assert list(inspect.signature(ca.CoarseAlignA.__init__).parameters)[:len(pa)] == pa
    pb = list(inspect.signature(R["CoarseAlignB"].__init__).parameters)
    assert list(inspect.signature(ca.CoarseAlignB.__init__).parameters)[:len(pb)] == pb
    for meth in ("setSource", "setTarget", "getCoarse", "skyFromSeg"):
        assert hasattr(ca.CoarseAlignA, meth)
    assert hasattr(ca.CoarseAlignB, "setPair")
    # segNet/segEval.py: SegNet(encoderPth, decoderPth, segId=1, segFg=True) / getSky(imgPath) -- same leading parameters,
    # same defaults (the drop-in adds device= at the end); a CoarseAlign built with segNet=False has no sky mask, like the reference's
    import importlib.util
    spec = importlib.util.spec_from_file_location("_rfx_dropin_segEval_probe", os.path.join(DROPIN, "segEval.py"))
    se = importlib.util.module_from_spec(spec)
    sys.path.insert(0, DROPIN)
    try:
        spec.loader.exec_module(se)
    finally:
        sys.path.remove(DROPIN)
    rs = ref_loader.load_seg()["segEval"].SegNet
    ref_params = inspect.signature(rs.__init__).parameters
    our_params = inspect.signature(se.SegNet.__init__).parameters
    assert list(our_params)[:len(ref_params)] == list(ref_params)
    assert all(our_params[k].default == ref_params[k].default for k in ref_params)
    assert str(inspect.signature(se.SegNet.getSky)) == str(inspect.signature(rs.getSky))


@pytest.mark.reference
def test_restatement_matches_live_reference():
    """Pins oracle/restate.py against the real reference on fresh seeds (the goldens pin fixed ones)."""
    import ref_loader
    import restate
    R = ref_loader.load()
    outil = R["outil"]
    torch.manual_seed(77)
    g = torch.Generator().manual_seed(5)
    A = torch.nn.functional.normalize(torch.relu(torch.randn(32, 210, generator=g)), dim=0)
    B = torch.nn.functional.normalize(torch.relu(torch.randn(32, 77, generator=g)), dim=0)
    r1, r2 = outil.mutualMatching(A, B)
    o1, o2 = restate.mutual_matching(A, B)
    assert torch.equal(r1, o1) and torch.equal(r2, o2)
    W, Hh = restate.get_wh(20, 25)
    m2 = torch.stack((Hh, W, torch.ones_like(W)), 1)
    m1 = m2 @ torch.tensor([[0.95, 0.03, 0.02], [0.02, 1.04, -0.03], [0.0, 0.01, 1.0]]).t()
    m1 = m1 / m1[:, 2:]
    m1[::3] = m2[torch.randperm(len(m2), generator=g)[:len(m1[::3])]]
    for seed in (1, 2, 3):
        torch.manual_seed(seed)
        samples = torch.randint(len(m1), (450, 4))
        torch.manual_seed(seed)
        Hb, cnt, inl, _ = outil.RANSAC(450, m1, m2, 0.05, 4, outil.Homography)
        Ho, co, io, _ = restate.ransac(m1, m2, 0.05, samples)
        assert np.array_equal(Hb, Ho) and int(cnt) == int(co) and np.array_equal(inl, io)


@pytest.mark.reference
def test_unchanged_reference_script_runs_up_to_the_first_device_call(tmp_path):
    """The REAL quick_start/align2images.py, unmodified, under dropin/run_reference_script.py in this GPU-less container:
    every import of the script (coarseAlignFeatMatch, outil, model, kornia.geometry, torchvision, pandas, matplotlib ...)
    must resolve to the drop-ins / stand-ins, its argument parser must run, the four drop-in modules must be constructed
    (align2images.py:37-41) and the run must stop at the script's first ``.cuda()`` (:44) with the HIP-device error --
    not with an ImportError, AttributeError or a silent CPU fallback."""
    import subprocess
    import ref_loader
    from rfx import weights
    script = os.path.join(ref_loader.REF_ROOT, "quick_start", "align2images.py")      # staged tree: the launcher takes the .pyc
    ck = tmp_path / "ck.pth"
    torch.save({"netFeatCoarse": weights.feature_extractor_sd(1), "netCorr": {}, "netFlowCoarse": weights.net_flow_coarse_sd(2),
                "netMatch": weights.net_matchability_sd(3)}, str(ck))
    launcher = os.path.join(ROOT, "ransac-flow_amd", "dropin", "run_reference_script.py")
    for how in ("argv", "env"):
        env = dict(os.environ, MPLBACKEND="Agg")
        cmd = [sys.executable, launcher]
        if how == "argv":
            cmd.append(script)
        else:
            env.update(RFX_REFERENCE_ROOT=ref_loader.REF_ROOT, RFX_REFERENCE_SCRIPT="quick_start/align2images.py")
        cmd += ["--resumePth", str(ck), "--outdir", str(tmp_path / "out")]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        err = out.stderr
        assert out.returncode != 0
        assert "ImportError" not in err and "ModuleNotFoundError" not in err and "AttributeError" not in err, err[-1500:]
        assert "align2images.py" in err, err[-1500:]                                 # died inside the script ...
        assert ref_loader.staged() or ".cuda()" in err, err[-1500:]                  # ... at a .cuda() call (bytecode has no source lines)
        assert any(k in err for k in ("No HIP GPUs are available", "Found no NVIDIA driver", "not compiled with CUDA", "HIP")), err[-800:]


@pytest.mark.parametrize("rel,args,tail", [
    ("evaluation/evalCorr/evaluation.py", ["--imageNet", "--nbScale", "3", "--minSize", "240"], ["MegaDepth", "--endIndex", "1"]),
    ("evaluation/evalYFCC/evaluation.py", ["--imageNet", "--nbScale", "3", "--minSize", "240"], ["YFCC", "--endIndex", "1"]),
])
def test_unchanged_evalcorr_and_evalyfcc_scripts_run_up_to_the_first_device_call(tmp_path, rel, args, tail):
    """Round 5: the two remaining evaluation drivers, unmodified, under dropin/run_reference_script.py in this GPU-less container --
    every import of the scripts (coarseAlignFeatMatch, outil, model, pandas, kornia.geometry, torchvision and ``from scipy.misc import
    imresize``, which the scipy of the reference's own requirements.txt no longer has: the launcher's restatement) must resolve, the
    argument parsers with their sub-commands must run, the four network modules must be constructed, and the run must stop at the
    script's first ``.cuda()`` (evalCorr/evaluation.py:126, evalYFCC/evaluation.py:116) with the HIP-device error."""
    import subprocess
    import ref_loader
    from rfx import weights
    script = os.path.join(ref_loader.REF_ROOT, rel)
    ck = tmp_path / "ck.pth"
    torch.save({"netFeatCoarse": weights.feature_extractor_sd(1), "netCorr": {}, "netFlowCoarse": weights.net_flow_coarse_sd(2),
                "netMatch": weights.net_matchability_sd(3)}, str(ck))
    launcher = os.path.join(ROOT, "ransac-flow_amd", "dropin", "run_reference_script.py")
    cmd = [sys.executable, launcher, script] + args + ["--resumePth", str(ck), "--outDir", str(tmp_path / "out")] + tail
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, MPLBACKEND="Agg"))
    err = out.stderr
    assert out.returncode != 0
    assert "ImportError" not in err and "ModuleNotFoundError" not in err and "AttributeError" not in err, err[-1500:]
    assert os.path.basename(os.path.dirname(rel)) in err, err[-1500:]              # died inside the script ...
    assert any(k in err for k in ("No HIP GPUs are available", "Found no NVIDIA driver", "not compiled with CUDA", "HIP")), err[-800:]
