#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY -- runs one of the reference's entry scripts ITSELF, unchanged, on the CPU.

    python oracle/run_ref_script.py quick_start/align2images.py [--rfx-ransac-seed S] [script args ...]

The counterpart of ``ransac-flow_amd/dropin/run_reference_script.py`` (which runs the same script on the MI355X drop-ins): here
the script's imports resolve to the REFERENCE's own modules (``coarseAlignFeatMatch``, ``outil``, ``model`` from the tree
``oracle/ref_loader.py`` found: /root/reference, or the byte-compiled oracle/_ref) under ref_loader's three stubs (``.cuda()``
-> identity, stand-in torchvision / kornia / scipy.misc.imresize; for evalYFCC the CoarseAlign instance is switched to its own
``use_cuda=False`` branch after construction).  Used by the GPU tests to compare an unchanged script's outputs on the device
with the outputs of the reference's own CPU run of the same command on the same box.

``--rfx-ransac-seed S``: reseed the CPU generator with S + k before the k-th ``outil.RANSAC`` call (utils/outil.py:120 draws
from it), so that the device run -- whose drop-in RANSAC draws from the same generator -- sees the same hypotheses.
``RFX_SEG_ENCODER / RFX_SEG_DECODER=<.pth>``: the sky-segmentation checkpoints of a ``--segNet`` run (the reference reads them from two
fixed paths inside its own tree).
``RFX_TRUNK_WEIGHTS=<.pth>``: the ResNet-50 trunk weights behind ``models.resnet50(pretrained=True)`` (no ImageNet weights
offline), shared with the drop-in.
"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def main():
    import ref_loader
    import torch
    argv = sys.argv[1:]
    if not argv:
        sys.exit(__doc__)
    rel, rest = argv[0], argv[1:]
    seed = None
    if rest[:1] == ["--rfx-ransac-seed"]:
        seed, rest = int(rest[1]), rest[2:]
    script = ref_loader.ref_path(rel)
    R = ref_loader.load()                   # stubs + the reference's outil / model in sys.modules
    sys.modules["model"] = R["model"]       # the scripts ``import model`` (ref_loader registers it as ref_model)
    if seed is not None:
        outil, n = R["outil"], [0]
        real = outil.RANSAC

        def ransac(*a, **k):
            torch.manual_seed(seed + n[0])
            n[0] += 1
            return real(*a, **k)
        outil.RANSAC = ransac
    if "--segNet" in rest:
        # evaluation/evalHpatch/evaluation.py --segNet: ``import segEval`` must resolve to the reference's own module, and its two
        # hard-wired checkpoint paths (evaluation/evalHpatch/coarseAlignFeatMatch.py:63: ../../model/pretrained/*.pth, inside the
        # read-only reference tree) to the files RFX_SEG_ENCODER / RFX_SEG_DECODER name -- the same override the drop-in honours
        S = ref_loader.load_seg()
        real_seg = S["segEval"].SegNet

        class SegNet(real_seg):
            def __init__(self, encoderPth, decoderPth, *a, **k):
                super().__init__(os.environ.get("RFX_SEG_ENCODER", encoderPth), os.environ.get("RFX_SEG_DECODER", decoderPth), *a, **k)
        S["segEval"].SegNet = SegNet
    if os.path.basename(os.path.dirname(script)) == "evalYFCC":
        # evaluation/evalYFCC/evaluation.py:143 hard-codes ``use_cuda=True`` and its CoarseAlign then builds small tensors with
        # ``torch.device("cuda")`` (evalYFCC/coarseAlignFeatMatch.py:154-155,161,176) -- next to ``.cuda()`` (identity here) the one
        # other place a CPU run of that script names the device.  The class carries its own CPU branch (``use_cuda=False``:
        # :113-114,122-123,141-143,157): the CPU oracle switches the instance to it right after construction, nothing else changes
        import importlib.util
        spec = importlib.util.spec_from_file_location("coarseAlignFeatMatch", ref_loader.ref_path("evaluation/evalYFCC/coarseAlignFeatMatch.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)

        class CoarseAlign(mod.CoarseAlign):
            def __init__(self, *a, **k):
                super().__init__(*a, **k)
                self.use_cuda = False
        mod.CoarseAlign = CoarseAlign
        sys.modules["coarseAlignFeatMatch"] = mod
    torch.set_num_threads(int(os.environ.get("RFX_CPU_THREADS", "8")))
    sys.argv = [script] + rest
    os.chdir(os.path.dirname(script))
    sys.path.insert(0, os.path.dirname(script))
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
