"""Drop-in for the reference's ``segNet/segEval.py``: ``SegNet(encoderPth, decoderPth, segId, segFg).getSky(imgPath)`` with the
forward pass on the MI355X (rfx/segnet.py: ResNet-50-dilated encoder + PPM decoder over five scales, csrc/seg.hip + the
convolution family incl. the dilated 3x3 entry point).  Same constructor arguments, same checkpoints (the two ``.pth`` state
dicts ``ModelBuilder.build_encoder / build_decoder`` load, segNet/segModel.py:268-292), same return value: a float32 (H, W)
numpy array, ``1 - (pred == segId)`` for ``segFg`` else ``(pred == segId)`` (segNet/segEval.py:38-43)."""
import torch

from rfx.segnet import SegNetDevice


class SegNet:
    def __init__(self, encoderPth, decoderPth, segId=1, segFg=True, device="cuda"):
        load = lambda p: p if isinstance(p, dict) else torch.load(p, map_location="cpu")      # a state dict or its file
        print('Loading weights for net_encoder')
        enc = load(encoderPth)
        print('Loading weights for net_decoder')
        dec = load(decoderPth)
        self.net = SegNetDevice(enc, dec, segId=segId, segFg=segFg, device=device)
        self.segId, self.segFg = segId, segFg

    def getSky(self, imgPath):
        return self.net.getSky(imgPath)
