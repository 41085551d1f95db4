"""TEST INFRASTRUCTURE ONLY -- the oracle interface of the parity sweeps / cpu_baseline served by THE REFERENCE ITSELF.

``oracle/restate.py`` is a port; this module offers the same call surface (the subset ``oracle/parity_sweep.py``,
``bench.py``'s CPU legs, ``__graft_entry__.smoke()`` and the tests use) but every piece of arithmetic is executed by the
reference's own code objects, loaded by ``oracle/ref_loader.py`` from ``/root/reference`` (authoring container) or from the
byte-compiled ``oracle/_ref`` (GPU box; recipe ``oracle/make_ref.py``):

  CoarseAlignOracle      the reference's ``CoarseAlign`` class, variant A (quick_start/coarseAlignFeatMatch.py:26-173) or B
                         (evaluation/evalHpatch/coarseAlignFeatMatch.py:35-179), UNMODIFIED; what the sweeps need from inside
                         it (feature matrices, cell coordinates, the match list, RANSAC's inlier mask) is observed by wrapping
                         ``outil.getWHTensor`` / ``outil.mutualMatching`` / ``outil.RANSAC`` from outside while its methods run
  ransac / score_ransac  ``outil.RANSAC`` / ``outil.ScoreRANSAC`` (utils/outil.py:102-164)
  pred_flow_mask(_kitti) the ``PredFlowMask`` functions of evaluation/evalHpatch/evaluation.py:23-55 and
                         evaluation/evalKITTI/evaluation.py:49-81, compiled out of the scripts
  multi_h_loop(_kitti)   the driver ``while`` STATEMENTS of evalHpatch/evaluation.py:211-243 / evalKITTI/evaluation.py:270-336
  kitti_fine_round       ONE pass of the KITTI ``while True`` statement with the homography handed in (a stand-in coarse model
                         returns it once, then None)
  fine_step_quickstart   the reference's nn.Modules called in the order of quick_start/align2images.py:66,87-97 (those lines
                         sit inside ``align2images(args)`` between plotting calls and cannot be executed on their own: the
                         call sequence is the only restated part)

The RANSAC index draw (``torch.randint`` at utils/outil.py:120) is the one thing injected: ``outil``'s module global ``torch``
is replaced by a proxy whose ``randint`` returns the caller's explicit sample tensor when one is armed and is the real
``torch.randint`` otherwise; every other attribute is torch's own.  The reference's code is not touched.

Third-party pieces that are NOT in the reference tree stay what ``ref_loader`` stubs them with: kornia's ``warp_grid``
(restated: parity-unpinned), torchvision's ToTensor / Normalize, skimage's ``measure.label`` (scipy's labelling).
"""
import contextlib
import io
import types

import numpy as np
import torch
import torch.nn.functional as F

import ref_loader
import restate

KIND = None          # set on first load: ref_loader.kind()

# helpers that carry no reference arithmetic (grids, integer cell indices, diagnostics of the sweeps)
identity_grid = restate.identity_grid
get_wh_int = restate.get_wh_int
dlt_matrix = restate.dlt_matrix
filter_samples = restate.filter_samples

_S = {}


class _TorchWithDraw:
    """``torch`` as seen by utils/outil.py: identical except that ``randint`` hands out an armed sample tensor once."""

    def __init__(self, real):
        self.__dict__["_real"] = real
        self.__dict__["armed"] = None          # callable (nbMatch, nbIter) -> LongTensor (nbIter, 4), or None

    def __getattr__(self, name):
        return getattr(self._real, name)

    def randint(self, *a, **k):
        fn = self.__dict__["armed"]
        if fn is None:
            return self._real.randint(*a, **k)
        high, size = a[0], a[1]
        s = fn(int(high), int(size[0]))
        s = torch.as_tensor(s, dtype=torch.long)
        assert tuple(s.shape) == tuple(size) and (len(s) == 0 or int(s.max()) < high), "injected draw does not fit (%s, %s)" % (high, size)
        return s


def _ref():
    global KIND
    if not _S:
        R = ref_loader.load()
        _S["R"] = R
        _S["outil"] = R["outil"]
        _S["proxy"] = _TorchWithDraw(torch)
        R["outil"].torch = _S["proxy"]
        _S["pfm_h"] = ref_loader.script_functions("evaluation/evalHpatch/evaluation.py", ["PredFlowMask"])["PredFlowMask"]
        fk = ref_loader.script_functions("evaluation/evalKITTI/evaluation.py", ["PredFlowMask", "remove_small_cc", "get_info"])
        import torchvision.transforms as tvt                      # ref_loader's stand-in
        fk["get_info"].__globals__["transforms"] = tvt
        _S["kitti"] = fk
        _S["loop_h"] = ref_loader.script_loop("evaluation/evalHpatch/evaluation.py", "nbCoarse <= args.maxCoarse")
        _S["loop_k"] = ref_loader.script_loop("evaluation/evalKITTI/evaluation.py", "True")
        _S["nets"], _S["ca"] = {}, {}
        KIND = ref_loader.kind()
    return _S


@contextlib.contextmanager
def _armed(fn):
    S = _ref()
    old = S["proxy"].__dict__["armed"]
    S["proxy"].__dict__["armed"] = fn
    try:
        yield
    finally:
        S["proxy"].__dict__["armed"] = old


def kind():
    _ref()
    return KIND


# ------------------------------------------------------------------------------------------------ networks


def _module(what, sd):
    """The reference's nn.Module for a state dict (cached per dict object)."""
    S = _ref()
    key = (what, id(sd))
    if key not in S["nets"]:
        model = S["R"]["model"]
        with contextlib.redirect_stdout(io.StringIO()):
            m = {"feat": model.FeatureExtractor, "flow": lambda: model.NetFlowCoarse(7), "match": lambda: model.NetMatchability(7)}[what]()
        m.load_state_dict(sd)
        m.eval()
        S["nets"][key] = (m, sd)                # keep sd alive: the key is its id
    return S["nets"][key][0]


def network(nets):
    """nets = dict(feat=sd, flow=sd[, match=sd]) -> the reference's ``network`` dict (quick_start/align2images.py:37-41)."""
    S = _ref()
    out = {"netFeatCoarse": _module("feat", nets["feat"]), "netCorr": S["R"]["model"].CorrNeigh(7),
           "netFlowCoarse": _module("flow", nets["flow"])}
    if nets.get("match") is not None:
        out["netMatch"] = _module("match", nets["match"])
    return out


def feature_extractor(sd, x):
    with torch.no_grad():
        return _module("feat", sd)(x)


def warp_grid(Hm, h, w):
    return _ref()["R"]["kornia_geometry"].HomographyWarper(h, w).warp_grid(Hm)


def mutual_matching(featA, featB):
    return _ref()["outil"].mutualMatching(featA, featB)


def score_ransac(match1, match2, tol, samples):
    o = _ref()["outil"]
    return o.ScoreRANSAC(match1, match2, tol, samples, o.Homography)


def ransac(match1, match2, tol, samples):
    """outil.RANSAC (utils/outil.py:117-164) on an explicit index draw."""
    o = _ref()["outil"]
    with _armed(lambda n, it: samples):
        return o.RANSAC(len(samples), match1, match2, tol, 4, o.Homography)


def fine_step_quickstart(nets, IsTensor, ItTensor, flowCoarse):
    net = network(nets)
    h, w = ItTensor.shape[2], ItTensor.shape[3]
    with torch.no_grad():
        img1_coarse = F.grid_sample(IsTensor, flowCoarse)                                    # align2images.py:66
        feat1 = F.normalize(net["netFeatCoarse"](img1_coarse))                               # :87
        feat2 = F.normalize(net["netFeatCoarse"](ItTensor))                                  # :88
        corr12 = net["netCorr"](feat1, feat2)                                                # :89
        flowDown = net["netFlowCoarse"](corr12, False)                                       # :90
        flowUp = F.interpolate(flowDown, size=(h, w), mode="bilinear")                       # :92
        flowUp = flowUp.permute(0, 2, 3, 1) + identity_grid(h, w)                            # :93-94
        flow12 = F.grid_sample(flowCoarse.permute(0, 3, 1, 2), flowUp).permute(0, 2, 3, 1).contiguous()   # :95
        img1_fine = F.grid_sample(IsTensor, flow12)                                          # :97
    return dict(img1_coarse=img1_coarse, feat1=feat1, feat2=feat2, corr12=corr12, flowDown=flowDown, flow12=flow12,
                img1_fine=img1_fine)


def pred_flow_mask(nets, IsTensor, featt, flowCoarse, grid):
    with torch.no_grad():
        return _ref()["pfm_h"](IsTensor, featt, flowCoarse, grid, network(nets))


def pred_flow_mask_kitti(nets, IsSample, ItSample, flowCoarse, grid):
    with torch.no_grad():
        return _ref()["kitti"]["PredFlowMask"](IsSample, ItSample, flowCoarse, grid, network(nets))


# ------------------------------------------------------------------------------------------------ coarse aligner


def _load_trunk(ca, trunk_sd):
    names = ["conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3"]
    remap = {}
    for k, v in trunk_sd.items():
        top, rest = k.split(".", 1)
        if top in names:
            remap["%d.%s" % (names.index(top), rest)] = v
    ca.net.load_state_dict(remap)
    ca.net.eval()
    return ca


class CoarseAlignOracle:
    """Same constructor and result dicts as ``restate.CoarseAlignOracle``; the work is done by the reference's class."""

    def __init__(self, trunk_sd, nbScale, nbIter, tolerance, minSize, scaleR, variant="A", sample_fn=None):
        S = _ref()
        key = (variant, nbScale, nbIter, tolerance, minSize, scaleR, id(trunk_sd))
        if key not in S["ca"]:
            R = S["R"]
            if variant == "A":
                ca = ref_loader.quiet(R["CoarseAlignA"], nbScale, nbIter, tolerance, "Homography", minSize, scaleR=scaleR)
            else:
                ca = ref_loader.quiet(R["CoarseAlignB"], nbScale, nbIter, tolerance, "Homography", minSize, 2, False, scaleR, True, False)
            S["ca"][key] = (_load_trunk(ca, trunk_sd), trunk_sd)
        self.ref = S["ca"][key][0]
        self.variant, self.nbIter, self.tol = variant, nbIter, tolerance
        self.scaleList = self.ref.scaleList
        self.sample_fn = sample_fn
        self.last = None

    # ---- observation of the reference's internals from outside
    @contextlib.contextmanager
    def _watch(self):
        o = _ref()["outil"]
        rec = dict(wh=[], mm=[], ransac=[])
        g0, m0, r0 = o.getWHTensor, o.mutualMatching, o.RANSAC

        def getwh(feat):
            r = g0(feat)
            rec["wh"].append(r)
            return r

        def mm(a, b):
            r = m0(a, b)
            rec["mm"].append((a, b, r[0], r[1]))
            return r

        def rs(nbIter, match1, match2, *a, **k):
            r = r0(nbIter, match1, match2, *a, **k)
            rec["ransac"].append((match1, match2, r))
            return r
        o.getWHTensor, o.mutualMatching, o.RANSAC = getwh, mm, rs
        try:
            yield rec
        finally:
            o.getWHTensor, o.mutualMatching, o.RANSAC = g0, m0, r0

    def _mirror(self):
        r = self.ref
        self.It, self.ItTensor = r.It, r.ItTensor
        if hasattr(r, "Is"):
            self.Is, self.IsTensor = r.Is, r.IsTensor

    def setSource(self, Is_org):
        assert self.variant == "A"
        self.ref.setSource(Is_org)
        self.Is, self.IsTensor = self.ref.Is, self.ref.IsTensor
        self.featsMultiScale, self.WMultiScale, self.HMultiScale = self.ref.featsMultiScale, self.ref.WMultiScale, self.ref.HMultiScale

    def setTarget(self, It_org):
        assert self.variant == "A"
        self.ref.setTarget(It_org)
        self.It, self.ItTensor, self.featt, self.Wt, self.Ht = self.ref.It, self.ref.ItTensor, self.ref.featt, self.ref.Wt, self.ref.Ht

    def setPair(self, Is_org, It_org):
        assert self.variant == "B"
        with self._watch() as rec:
            self.ref.setPair(Is_org, It_org)
        self._mirror()
        ns = len(self.scaleList)
        assert len(rec["wh"]) == ns + 1 and len(rec["mm"]) == 1
        self.WMultiScale = torch.cat([w for w, _ in rec["wh"][:ns]])
        self.HMultiScale = torch.cat([h for _, h in rec["wh"][:ns]])
        self.Wt, self.Ht = rec["wh"][ns]
        A, B, i1, i2 = rec["mm"][0]
        self.featsMultiScale = A
        self.featt = B.view(1, 1024, self.ref.W2, self.ref.H2)
        self.index1, self.index2 = i1, i2
        self._own = None

    def _mask_to_feat(self, Mt):
        # the two mask lines of getCoarse (evalHpatch/coarseAlignFeatMatch.py:158-160), used by the sweeps only to REPORT which
        # cached matches a round kept; the decision itself is made by the reference's getCoarse below
        ext = torch.from_numpy((1 - Mt).astype(np.float32))[None, None]
        m = F.interpolate(ext, size=(self.featt.shape[2], self.featt.shape[3]), mode="bilinear", align_corners=False)
        return m > 0.5

    def set_matches(self, i1, i2):
        """Variant B: make the reference's getCoarse run on a GIVEN match list (the device's) instead of its own cached one."""
        r = self.ref
        if self._own is None:
            self._own = (r.W1MutualMatch, r.H1MutualMatch, r.W2MutualMatch, r.H2MutualMatch, r.W2MutualMatchInt, r.H2MutualMatchInt)
        if i1 is None:
            (r.W1MutualMatch, r.H1MutualMatch, r.W2MutualMatch, r.H2MutualMatch, r.W2MutualMatchInt, r.H2MutualMatchInt) = self._own
            return
        i1, i2 = torch.as_tensor(np.asarray(i1), dtype=torch.long), torch.as_tensor(np.asarray(i2), dtype=torch.long)
        WI, HI = get_wh_int(self.featt.shape[2], self.featt.shape[3])
        r.W1MutualMatch, r.H1MutualMatch = self.WMultiScale[i1], self.HMultiScale[i1]
        r.W2MutualMatch, r.H2MutualMatch = self.Wt[i2], self.Ht[i2]
        r.W2MutualMatchInt, r.H2MutualMatchInt = WI[i2], HI[i2]

    def getCoarse(self, Mt, sample_fn=None):
        """The reference's getCoarse(Mt).  Returns dict(H, inlier, index1, index2, count, n) or None on its sentinel paths;
        ``self.last`` keeps what was observed (n = -1 when fewer than 4 matches survived: RANSAC was not called)."""
        fn = sample_fn or self.sample_fn
        with self._watch() as rec, (_armed(fn) if fn is not None else contextlib.nullcontext()):
            out = self.ref.getCoarse(np.asarray(Mt))
        Hb = out[0] if self.variant == "A" else out
        if self.variant == "A":
            _, _, i1, i2 = rec["mm"][-1]
        else:
            i1, i2 = self.index1, self.index2
            if rec["ransac"] and len(rec["ransac"][-1][0]) != len(i1):
                valid = self._mask_to_feat(np.asarray(Mt))[0, 0][self.ref.W2MutualMatchInt, self.ref.H2MutualMatchInt]
                if len(self.ref.W2MutualMatchInt) == len(i1):
                    i1, i2 = i1[valid], i2[valid]
        self.last = dict(n=(len(rec["ransac"][-1][0]) if rec["ransac"] else -1), index1=i1.numpy(), index2=i2.numpy())
        if Hb is None:
            return None
        m1, m2, (Hr, cnt, inl, _) = rec["ransac"][-1]
        res = dict(H=np.asarray(Hb, dtype=np.float32), inlier=np.asarray(inl), index1=self.last["index1"], index2=self.last["index2"],
                   match1=m1.numpy(), match2=m2.numpy(), count=int(cnt), n=len(m1))
        if self.variant == "A":
            res["inlierMask"] = out[1]
        return res


# ------------------------------------------------------------------------------------------------ driver loops


def _ns_hpatch(ca, nets, max_coarse, th, It_bg, pfm=None):
    S = _ref()
    Itw, Ith = ca.It.size
    with torch.no_grad():
        featt = F.normalize(network(nets)["netFeatCoarse"](ca.ItTensor))
    return dict(args=types.SimpleNamespace(maxCoarse=max_coarse, maskRegionTh=th), coarseModel=ca.ref, network=network(nets), featt=featt,
                grid=identity_grid(Ith, Itw), warper=S["R"]["kornia_geometry"].HomographyWarper(Ith, Itw),
                It_bg=np.ones((Ith, Itw), dtype=np.float32) if It_bg is None else It_bg, Mask=np.zeros((Ith, Itw), dtype=np.float32),
                Coarse_Flow_Tensor=[], Fine_Flow_Tensor=[], Fine_Mask_Tensor=[], nbCoarse=0, PredFlowMask=pfm or S["pfm_h"])


def multi_h_loop(ca, nets, max_coarse=10, mask_region_th=0.01, It_bg=None):
    """evaluation/evalHpatch/evaluation.py:211-243 -- the reference's own ``while`` statement on the variables its module level
    sets up (:172-208).  ``ca``: a variant-B CoarseAlignOracle after setPair; its sample_fn (if any) supplies each round's draw."""
    S = _ref()
    ns = _ns_hpatch(ca, nets, max_coarse, mask_region_th, It_bg)
    with torch.no_grad(), (_armed(ca.sample_fn) if ca.sample_fn is not None else contextlib.nullcontext()):
        S["loop_h"](ns)
    return dict(H=[np.asarray(h[0], dtype=np.float32) for h in ns["Coarse_Flow_Tensor"]], flowDown8=list(ns["Fine_Flow_Tensor"]),
                matchDown8=list(ns["Fine_Mask_Tensor"]), masks=[ns["Mask"]], nbH=ns["nbCoarse"])


def kitti_setup(Is, It, fine_size):
    """evaluation/evalKITTI/evaluation.py:222-256: the script's get_info + outil.resizeImg on the pair."""
    S = _ref()
    gi, o = S["kitti"]["get_info"], S["outil"]
    It_resize, It_d2 = o.resizeImg(It, 8, fine_size), o.resizeImg(It, 8, fine_size // 2)
    with torch.no_grad():
        w_org, h_org, _, grid_org, _ = gi(It)
        _, _, tensor_s, _, _ = gi(Is)
        w_r, h_r, tensor_resize, grid_resize, warper_resize = gi(It_resize)
        w_d2, h_d2, tensor_d2, grid_d2, warper_d2 = gi(It_d2)
    return dict(tensor_s=tensor_s, tensor_resize=tensor_resize, tensor_d2=tensor_d2, org=(h_org, w_org), resize=(h_r, w_r), d2=(h_d2, w_d2),
                grid_org=grid_org, grid_resize=grid_resize, grid_d2=grid_d2, warper_resize=warper_resize, warper_d2=warper_d2)


def _ns_kitti(coarse_model, nets, T, th, cc_th, It_bg, Mask=None, nb=0):
    S = _ref()
    h_org, w_org = T["org"]
    return dict(args=types.SimpleNamespace(cc_th=cc_th, maskRegionTh=th), coarseModel=coarse_model, network=network(nets),
                It_bg=np.ones((h_org, w_org), dtype=np.float32) if It_bg is None else It_bg,
                Mask=np.zeros((h_org, w_org), dtype=np.float32) if Mask is None else Mask,
                warper_d2=T["warper_d2"], warper_resize=T["warper_resize"], tensor_s=T["tensor_s"], tensor_d2=T["tensor_d2"],
                tensor_resize=T["tensor_resize"], grid_d2=T["grid_d2"], grid_resize=T["grid_resize"], grid_org=T["grid_org"],
                Homography=[], Org_D2=[], Finetune_D2=[], Org_Mask=[], Finetune_Mask=[], Org=[], Finetune=[], nbCoarse=nb,
                PredFlowMask=S["kitti"]["PredFlowMask"], remove_small_cc=S["kitti"]["remove_small_cc"])


def multi_h_loop_kitti(ca, nets, Is, It, fine_size, mask_region_th=0.01, cc_th=0.0, It_bg=None):
    """evaluation/evalKITTI/evaluation.py:270-336 -- the reference's ``while True`` statement."""
    S = _ref()
    T = kitti_setup(Is, It, fine_size)
    ns = _ns_kitti(ca.ref, nets, T, mask_region_th, cc_th, It_bg)
    with (_armed(ca.sample_fn) if ca.sample_fn is not None else contextlib.nullcontext()):
        S["loop_k"](ns)
    cat = lambda lst: [t.numpy() for t in lst]
    return dict(H=[np.asarray(h[0].numpy(), dtype=np.float32) for h in ns["Homography"]], flowD2=cat(ns["Finetune_D2"]),
                flowDown8=cat(ns["Finetune"]), matchDown8=cat(ns["Finetune_Mask"]), masks=[ns["Mask"]], nbH=ns["nbCoarse"])


class _OneShotCoarse:
    """Stand-in for ``coarseModel`` in a teacher-forced pass of a driver loop: hands the given homography to the loop once."""

    def __init__(self, Hm):
        self.H, self.calls = np.asarray(Hm, dtype=np.float32).reshape(3, 3), 0

    def getCoarse(self, Mt):
        self.calls += 1
        return self.H.copy() if self.calls == 1 else None


def kitti_fine_round(nets, T, Hm, cc_th):
    """ONE pass of the KITTI ``while True`` statement for a given homography (evalKITTI/evaluation.py:279-321): returns
    (match after the small-component filter, flow_d2, flowDown8, matchDown8, flow12) -- the variables the statement leaves."""
    S = _ref()
    # nbCoarse = 1 and maskRegionTh = inf: the pass computes everything up to the accept test (:322), then takes ``else: break``
    ns = _ns_kitti(_OneShotCoarse(torch.as_tensor(Hm).numpy()), nets, T, float("inf"), cc_th, None, nb=1)
    S["loop_k"](ns)
    return ns["matchFine_finetune"], ns["flowFine_d2"], ns["flowFineDown8_org"], ns["matchFineDown8_org"], ns["flowFine_org"]
