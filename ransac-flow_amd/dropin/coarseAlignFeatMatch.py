"""Drop-in for the reference's ``coarseAlignFeatMatch.py`` -- the ``CoarseAlign`` object in its three call
signatures:

  CoarseAlignA  quick_start/coarseAlignFeatMatch.py:26-173   setSource / setTarget / getCoarse -> (H, InlierMask)
  CoarseAlignB  evaluation/eval{Hpatch,KITTI,Corr}/coarseAlignFeatMatch.py:35-179   setPair / getCoarse -> H
  CoarseAlignC  evaluation/evalYFCC/coarseAlignFeatMatch.py:35-196   like A with ``use_cuda``

``CoarseAlign`` is bound to the variant named by ``RFX_COARSE_VARIANT`` (A, default).  Public attributes the
reference's scripts read are kept: ``It``, ``Is`` (PIL), ``IsTensor``, ``ItTensor`` (1x3xhxw on the device),
``featt``, ``scaleList``, ``nbIter``, ``tolerance``.

All device work (ResNet-50 conv4 trunk, L2 norm, all-pairs correlation + mutual NN, RANSAC, mask resize)
runs in librfx HIP kernels; PIL resizing and ToTensor/Normalize stay on the host like in the reference.
``imageNet=True`` wants torchvision's ImageNet weights: they are used when torchvision can provide them,
otherwise the constructor raises (as the reference would) unless the caller opts in explicitly:
``trunk_state_dict=``, ``RFX_TRUNK_WEIGHTS=<.pth>`` or ``RFX_ALLOW_RANDOM_TRUNK=1`` (seeded random-init trunk).
"""
import os
import sys

import numpy as np
import PIL.Image as Image
import torch

import outil
from rfx import ops, weights
from rfx.nets import ResNet50Trunk
from rfx.pipeline import scale_list, resize_dims, pil_to_tensor, IMAGENET_MEAN, IMAGENET_STD

_TRUNK_KEYS = ("conv1.", "bn1.", "layer1.", "layer2.", "layer3.")


def _trunk_state_dict(imageNet, explicit=None):
    """Weights of conv1..layer3.  Like the reference (quick_start/coarseAlignFeatMatch.py:36), ``imageNet=True`` fails
    loudly when torchvision's pretrained ResNet-50 cannot be had; the only ways around it are explicit:
    ``trunk_state_dict=``, ``RFX_TRUNK_WEIGHTS=<state-dict .pth>`` or ``RFX_ALLOW_RANDOM_TRUNK=1`` (benchmarks and
    smoke runs on this offline image; alignment with such a trunk is meaningless on real photographs)."""
    if explicit is not None:
        return explicit
    pth = os.environ.get("RFX_TRUNK_WEIGHTS")
    if pth:
        sd = torch.load(pth, map_location="cpu")
        sd = sd.get("model", sd) if isinstance(sd, dict) else sd
        sd = {k.replace("module.", ""): v for k, v in sd.items()}
        return {k: v for k, v in sd.items() if k.startswith(_TRUNK_KEYS)}
    if imageNet:
        try:
            import torchvision.models as models
            sd = models.resnet50(pretrained=True).state_dict()
            return {k: v for k, v in sd.items() if k.startswith(_TRUNK_KEYS)}
        except (ImportError, OSError) as e:     # no torchvision / no network for the weight download
            if os.environ.get("RFX_ALLOW_RANDOM_TRUNK", "0") != "1":
                raise RuntimeError(
                    "CoarseAlign(imageNet=True): torchvision's ImageNet ResNet-50 weights are unavailable (%s: %s). Pass "
                    "trunk_state_dict=, point RFX_TRUNK_WEIGHTS at a ResNet-50 state dict, or set RFX_ALLOW_RANDOM_TRUNK=1 "
                    "to run with a seeded random-init trunk (benchmark / smoke use only)." % (type(e).__name__, e)) from e
            print("[rfx] RFX_ALLOW_RANDOM_TRUNK=1: seeded random-init ResNet-50 trunk (ImageNet weights unavailable: %s)"
                  % type(e).__name__)
            return weights.resnet50_trunk_sd(seed=0)
    featPth = "../../model/pretrained/resnet50_moco.pth"
    param = torch.load(featPth, map_location="cpu")
    print("Loading pretrained model from {}".format(featPth))
    sd = {k.replace("module.", ""): v for k, v in param["model"].items()}
    return {k: v for k, v in sd.items() if k.startswith(_TRUNK_KEYS)}


class _CoarseBase:
    _resize_mode = "max"

    def _init_common(self, nbScale, nbIter, tolerance, transform, minSize, imageNet, scaleR, trunk_state_dict, device):
        self.nbIter = nbIter
        self.tolerance = tolerance
        self.device = torch.device(device)
        self.net = ResNet50Trunk(_trunk_state_dict(imageNet, trunk_state_dict), self.device)
        if transform == "Affine":
            raise NotImplementedError("the Affine transform is unreachable in the reference (outil.RANSAC samples 4 points)")
        self.Transform = outil.Homography
        self.nbPoint = 4
        self.strideNet = 16
        self.minSize = minSize
        self.scaleList = scale_list(nbScale, scaleR)
        self._mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
        self._std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
        print(self.scaleList)

    # host-side pre-processing ------------------------------------------------------------------
    def toTensor(self, pil):
        return pil_to_tensor(pil)

    def preproc(self, pil):
        return (pil_to_tensor(pil) - self._mean) / self._std

    def _resize(self, I, size):
        w, h = I.size
        nw, nh = resize_dims(w, h, size, self._resize_mode, self.strideNet)
        return I.resize((nw, nh), resample=Image.LANCZOS)

    def ResizeMaxSize(self, I, minSize):
        self_mode, self._resize_mode = self._resize_mode, "max"
        try:
            return self._resize(I, minSize)
        finally:
            self._resize_mode = self_mode

    def ResizeMinSize(self, I, minSize):
        self_mode, self._resize_mode = self._resize_mode, "min"
        try:
            return self._resize(I, minSize)
        finally:
            self._resize_mode = self_mode

    # device work ----------------------------------------------------------------------------------
    def _feat(self, pil):
        x = self.preproc(pil).unsqueeze(0).to(self.device)
        return ops.l2norm(self.net(x))

    def setSource(self, Is_org):
        with torch.no_grad():
            IsList = [self._resize(Is_org, int(self.minSize * s)) for s in self.scaleList]
            self.Is = IsList[len(self.scaleList) // 2]
            self.IsTensor = self.toTensor(self.Is).unsqueeze(0).to(self.device)
            feats, Ws, Hs = [], [], []
            for I in IsList:
                feat = self._feat(I)
                W, Hh = outil.getWHTensor(feat)
                feats.append(feat.view(1024, -1))
                Ws.append(W)
                Hs.append(Hh)
            self.featsMultiScale = torch.cat(feats, dim=1)
            self.WMultiScale = torch.cat(Ws)
            self.HMultiScale = torch.cat(Hs)

    def setTarget(self, It_org):
        with torch.no_grad():
            self.It = self._resize(It_org, self.minSize)
            self.ItTensor = self.toTensor(self.It).unsqueeze(0).to(self.device)
            self.featt = self._feat(self.It)
            self.Wt, self.Ht = outil.getWHTensor(self.featt)

    # evaluation/evalHpatch/coarseAlignFeatMatch.py:63-64: the reference builds its SegNet in the constructor from two fixed
    # checkpoint paths (relative to the script's directory) whenever ``segNet`` is true -- which is also the signature's default.
    # Here the same two files are read when they exist; otherwise the network is built at the first skyFromSeg call (and a missing
    # checkpoint surfaces there as the FileNotFoundError the reference raises at construction): callers that never ask for a sky mask
    # (quick_start, the --segNet-less evaluation runs) need no segmentation weights.  RFX_SEG_ENCODER / RFX_SEG_DECODER override the paths.
    SEG_ENCODER_PTH = '../../model/pretrained/ade20k_resnet50dilated_encoder.pth'
    SEG_DECODER_PTH = '../../model/pretrained/ade20k_resnet50dilated_decoder.pth'

    def _init_seg(self, segNet, segId, segFg, eager=False):
        self._seg_args, self.segNet = (segId, segFg) if segNet else None, None
        enc = os.environ.get("RFX_SEG_ENCODER", self.SEG_ENCODER_PTH)
        if segNet and (eager or os.path.isfile(enc)):
            self._build_seg()

    def _build_seg(self):
        import segEval
        self.segNet = segEval.SegNet(os.environ.get("RFX_SEG_ENCODER", self.SEG_ENCODER_PTH),
                                     os.environ.get("RFX_SEG_DECODER", self.SEG_DECODER_PTH), self._seg_args[0], self._seg_args[1])

    def skyFromSeg(self, path):
        """:152-153 -> SegNet.getSky (segNet/segEval.py:23-43), forward pass on the device (rfx/segnet.py)."""
        if getattr(self, "_seg_args", None) is None:
            raise AttributeError("'CoarseAlign' object has no attribute 'segNet'")      # what the reference says for segNet=False
        if self.segNet is None:
            self._build_seg()
        return self.segNet.getSky(path)

    def _mask_to_feature_res(self, Mt):
        """1 - Mt -> bilinear resize to the feature grid (align_corners=False) -> > 0.5."""
        ext = torch.from_numpy((1 - Mt).astype(np.float32)).to(self.device)[None, None]
        m = ops.resize_bilinear(ext, (self.featt.size(2), self.featt.size(3)), align_corners=False)
        return m > 0.5

    def _ransac(self, match1, match2):
        return outil.RANSAC(self.nbIter, match1, match2, self.tolerance, self.nbPoint, self.Transform)


class CoarseAlignA(_CoarseBase):
    """quick_start/coarseAlignFeatMatch.py:26-173."""
    _resize_mode = "max"

    def __init__(self, nbScale, nbIter, tolerance, transform, minSize, segId=1, segFg=True, imageNet=True, scaleR=2,
                 trunk_state_dict=None, device="cuda"):
        self._init_common(nbScale, nbIter, tolerance, transform, minSize, imageNet, scaleR, trunk_state_dict, device)

    def getCoarse(self, Mt):
        with torch.no_grad():
            keep = self._mask_to_feature_res(Mt)
            # zeroed target features (reference :143) == a 0/1 column mask inside the correlation kernel
            index1, index2 = ops.mutual_nn(self.featsMultiScale, self.featt.view(1024, -1), keep.float().view(-1),
                                           score_chunk=outil.SCORE_CHUNK)
            W1, H1 = self.WMultiScale[index1], self.HMultiScale[index1]
            W2, H2 = self.Wt[index2], self.Ht[index2]
            ones = torch.ones_like(W1)
            match1 = torch.stack((H1, W1, ones), dim=1)
            match2 = torch.stack((H2, W2, ones), dim=1)
            if len(match1) < self.nbPoint:
                return None, []
            bestParam, _, indexInlier, _ = self._ransac(match1, match2)
            if bestParam is None:
                return None, []
            index2Inlier = index2.cpu().numpy()[indexInlier]
            nr, nc = self.featt.size(2), self.featt.size(3)
            Wt, Ht = self.Wt.cpu(), self.Ht.cpu()
            InlierMask = np.zeros((nr, nc), dtype=np.float32)
            InlierMask[((Wt[index2Inlier] / 2 + 0.5) * nr).numpy().astype(np.int64),
                       ((Ht[index2Inlier] / 2 + 0.5) * nc).numpy().astype(np.int64)] = 1
            return bestParam.astype(np.float32), InlierMask


class CoarseAlignC(CoarseAlignA):
    """evaluation/evalYFCC/coarseAlignFeatMatch.py:35-196 (ResizeMinSize, ``use_cuda`` flag)."""
    _resize_mode = "min"

    def __init__(self, nbScale, nbIter, tolerance, transform, minSize, segId=1, segFg=True, use_cuda=True, imageNet=True,
                 segNet=True, scaleR=2, trunk_state_dict=None, device="cuda"):
        if not use_cuda:
            raise RuntimeError("use_cuda=False: the MI355X path has no CPU fallback")
        self._init_common(nbScale, nbIter, tolerance, transform, minSize, imageNet, scaleR, trunk_state_dict, device)
        self._init_seg(segNet, segId, segFg)


class CoarseAlignB(_CoarseBase):
    """evaluation/evalHpatch/coarseAlignFeatMatch.py:35-179: matches computed once in setPair, filtered by the
    mask at integer cell coordinates in getCoarse, which returns H only."""
    _resize_mode = "min"

    def __init__(self, nbScale, nbIter, tolerance, transform, minSize, segId, segFg, scaleR=2, imageNet=True, segNet=True,
                 trunk_state_dict=None, device="cuda"):
        self._init_common(nbScale, nbIter, tolerance, transform, minSize, imageNet, scaleR, trunk_state_dict, device)
        self._init_seg(segNet, segId, segFg)

    def setPair(self, Is_org, It_org):
        with torch.no_grad():
            self.setSource(Is_org)
            self.setTarget(It_org)
            WtInt, HtInt = outil.getWHTensor_Int(self.featt)
            self.W2, self.H2 = self.featt.size(2), self.featt.size(3)
            index1, index2 = outil.mutualMatching(self.featsMultiScale, self.featt.view(1024, -1))
            self.W1MutualMatch, self.H1MutualMatch = self.WMultiScale[index1], self.HMultiScale[index1]
            self.W2MutualMatch, self.H2MutualMatch = self.Wt[index2], self.Ht[index2]
            self.W2MutualMatchInt, self.H2MutualMatchInt = WtInt[index2], HtInt[index2]

    def getCoarse(self, Mt):
        with torch.no_grad():
            keep = self._mask_to_feature_res(Mt)[0, 0]
            valid = keep[self.W2MutualMatchInt, self.H2MutualMatchInt]
            ones = torch.ones(int(valid.sum().item()), dtype=torch.float32, device=self.device)
            match1 = torch.stack((self.H1MutualMatch[valid], self.W1MutualMatch[valid], ones), dim=1)
            match2 = torch.stack((self.H2MutualMatch[valid], self.W2MutualMatch[valid], ones), dim=1)
            if len(match1) < self.nbPoint:
                return None
            bestParam, _, indexInlier, _ = self._ransac(match1, match2)
            if bestParam is None:
                return None
            return bestParam.astype(np.float32)


CoarseAlign = {"A": CoarseAlignA, "B": CoarseAlignB, "C": CoarseAlignC}[os.environ.get("RFX_COARSE_VARIANT", "A").upper()]
