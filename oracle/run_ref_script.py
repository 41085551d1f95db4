#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY -- runs one of the reference's entry scripts ITSELF, unchanged, on the CPU.

    python oracle/run_ref_script.py quick_start/align2images.py [--rfx-ransac-seed S] [script args ...]

The counterpart of ``ransac-flow_amd/dropin/run_reference_script.py`` (which runs the same script on the MI355X drop-ins): here
the script's imports resolve to the REFERENCE's own modules (``coarseAlignFeatMatch``, ``outil``, ``model`` from the tree
``oracle/ref_loader.py`` found: /root/reference, or the byte-compiled oracle/_ref) under ref_loader's three stubs (``.cuda()``
-> identity, stand-in torchvision / kornia).  Used by the GPU tests to compare an unchanged script's outputs on the device
with the outputs of the reference's own CPU run of the same command on the same box.

``--rfx-ransac-seed S``: reseed the CPU generator with S + k before the k-th ``outil.RANSAC`` call (utils/outil.py:120 draws
from it), so that the device run -- whose drop-in RANSAC draws from the same generator -- sees the same hypotheses.
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
    torch.set_num_threads(int(os.environ.get("RFX_CPU_THREADS", "8")))
    sys.argv = [script] + rest
    os.chdir(os.path.dirname(script))
    sys.path.insert(0, os.path.dirname(script))
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
