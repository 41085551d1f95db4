#!/usr/bin/env python
"""Run one of the reference's entry scripts UNCHANGED on the MI355X path.

    python run_reference_script.py /path/to/RANSAC-Flow/quick_start/align2images.py [script args ...]
    RFX_REFERENCE_ROOT=/mnt/RANSAC-Flow RFX_REFERENCE_SCRIPT=quick_start/align2images.py python run_reference_script.py [args]

The reference's scripts import the hot path by bare module name after ``sys.path.append`` --
``from coarseAlignFeatMatch import CoarseAlign`` (quick_start/align2images.py:2), ``import outil`` (:5),
``import model`` (:7), ``import kornia.geometry as tgm`` (:19).  This launcher pre-loads the drop-in modules
of this directory into ``sys.modules`` under exactly those names (so the script's own imports resolve to
them), picks the CoarseAlign variant from the script's directory, installs small stand-ins for packages the
script imports but this image lacks (kornia's HomographyWarper -> librfx warp_grid kernel; torchvision
transforms; scipy.misc.imresize), seats the ``segEval`` drop-in, optionally rebinds ``torch.nn.functional.grid_sample`` /
``interpolate`` / ``normalize`` to the librfx kernels for float32 HIP tensors (``RFX_PATCH_FUNCTIONAL=1``,
default on), and then runs the script with ``runpy`` from its own directory.  ``RFX_RANSAC_SEED=S`` makes the RANSAC draws
reproducible (see ``seed_ransac_calls``).  A byte-compiled deployment of the reference tree (``align2images.pyc``) runs as is.
"""
import importlib
import os
import runpy
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)


def variant_for(script_path):
    d = os.path.basename(os.path.dirname(os.path.abspath(script_path)))
    return {"quick_start": "A", "evalYFCC": "C"}.get(d, "B" if d.startswith("eval") else "A")


def install_kornia_shim():
    try:
        import kornia.geometry  # noqa: F401
        return
    except Exception:
        pass
    import torch
    from rfx import ops

    class HomographyWarper:
        """kornia 0.1.4 HomographyWarper(h, w).warp_grid(H) on the rfx_warp_grid_f32 kernel."""

        def __init__(self, height, width, *a, **k):
            self.height, self.width = height, width

        def warp_grid(self, H):
            return ops.warp_grid(H.float().cuda(), self.height, self.width)

    ko, kg = types.ModuleType("kornia"), types.ModuleType("kornia.geometry")
    kg.HomographyWarper = HomographyWarper
    ko.geometry = kg
    sys.modules["kornia"], sys.modules["kornia.geometry"] = ko, kg


def install_torchvision_shim():
    try:
        import torchvision  # noqa: F401
        return
    except Exception:
        pass
    import numpy as np
    import torch
    import PIL.Image as Image
    tv, tvm, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.models"), types.ModuleType("torchvision.transforms")

    class ToTensor:
        def __call__(self, pic):
            a = np.asarray(pic, dtype=np.uint8)
            a = a[:, :, None] if a.ndim == 2 else a
            return torch.from_numpy(a.copy()).permute(2, 0, 1).contiguous().float().div(255)

    class Normalize:
        def __init__(self, mean, std):
            self.m, self.s = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

        def __call__(self, t):
            return (t - self.m) / self.s

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToPILImage:
        def __call__(self, t):
            return Image.fromarray((t.detach().cpu().clamp(0, 1) * 255).byte().permute(1, 2, 0).numpy())

    def resnet50(pretrained=False, **k):
        raise ImportError("torchvision is not installed (stand-in module): no pretrained ResNet-50 available")

    tvt.ToTensor, tvt.Normalize, tvt.Compose, tvt.ToPILImage = ToTensor, Normalize, Compose, ToPILImage
    tvm.resnet50 = resnet50
    tv.models, tv.transforms = tvm, tvt
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.transforms": tvt})


def install_misc_shims():
    import scipy
    if not hasattr(scipy, "misc") or not hasattr(scipy.misc, "imresize"):
        sm = types.ModuleType("scipy.misc")

        def imresize(arr, size, interp="bilinear", mode=None):
            """scipy.misc.imresize (removed in scipy 1.3; evaluation/evalYFCC/evaluation.py:23,201,212, evalCorr/evaluation.py:24,186)
            for the 2-D arrays those scripts pass, restated from scipy 1.2.3 scipy/misc/pilutil.py: bytescale of non-uint8 data
            ([min, max] -> [0, 255]; a constant array -> zeros), 8-bit image, PIL resize to (cols, rows), back to an array."""
            import numpy as np
            import PIL.Image as Image
            data = np.asarray(arr)
            if data.ndim != 2:
                raise NotImplementedError("imresize stand-in: 2-D arrays only")
            if data.dtype != np.uint8:
                cmin, cmax = data.min(), data.max()
                cscale = (cmax - cmin) or 1
                data = (((data - cmin) * (255.0 / cscale)).clip(0, 255) + 0.5).astype(np.uint8)
            im = Image.frombytes("L", (data.shape[1], data.shape[0]), data.tobytes())
            resample = {"nearest": 0, "lanczos": 1, "bilinear": 2, "bicubic": 3, "cubic": 3}[interp]
            return np.asarray(im.resize((int(size[1]), int(size[0])), resample=resample))
        sm.imresize = imresize
        sys.modules["scipy.misc"] = sm
        scipy.misc = sm
    # ``import segEval`` after sys.path.append('../../segNet') (evaluation/evalHpatch/coarseAlignFeatMatch.py:22-23): the drop-in
    # of this directory (SegNet.getSky on the device, rfx/segnet.py)
    sys.modules["segEval"] = importlib.import_module("segEval")
    try:
        import skimage.measure  # noqa: F401
    except ImportError:
        # evaluation/evalKITTI/evaluation.py:24,85-100: measure.label(binary, background=0) = 8-connected components in 2-D
        import numpy as np
        from scipy import ndimage
        sk, skm = types.ModuleType("skimage"), types.ModuleType("skimage.measure")
        skm.label = lambda m, background=0: ndimage.label(m, structure=np.ones((3, 3), dtype=np.int32))[0]
        sk.measure = skm
        sys.modules["skimage"], sys.modules["skimage.measure"] = sk, skm


def patch_functional():
    """F.grid_sample / F.interpolate(bilinear) / F.normalize(dim=1) -> librfx for float32 HIP 4-D tensors."""
    import torch
    import torch.nn.functional as F
    from rfx import ops
    _gs, _ip, _nm = F.grid_sample, F.interpolate, F.normalize

    def grid_sample(input, grid, mode="bilinear", padding_mode="zeros", align_corners=None):
        if input.is_cuda and input.dtype == torch.float32 and input.dim() == 4 and mode == "bilinear" and padding_mode == "zeros":
            return ops.grid_sample(input, grid, bool(align_corners))
        return _gs(input, grid, mode=mode, padding_mode=padding_mode, align_corners=align_corners)

    def interpolate(input, size=None, scale_factor=None, mode="nearest", align_corners=None, **kw):
        if input.is_cuda and input.dtype == torch.float32 and input.dim() == 4 and mode == "bilinear" and not kw:
            sf, out_size = None, size                    # the caller's own arguments stay untouched for the fallback below
            if out_size is None:
                sf = scale_factor if isinstance(scale_factor, (tuple, list)) else (scale_factor, scale_factor)
                out_size = (int(input.shape[2] * sf[0]), int(input.shape[3] * sf[1]))
            # torch maps coordinates with 1/scale_factor when a scale factor is given; the kernel uses in/out sizes:
            # the two agree only when out = in * factor exactly (integer factors, the x8 of the reference)
            if sf is None or all(float(f).is_integer() for f in sf):
                out_size = (out_size, out_size) if isinstance(out_size, int) else out_size
                return ops.resize_bilinear(input, out_size, bool(align_corners))
        return _ip(input, size=size, scale_factor=scale_factor, mode=mode, align_corners=align_corners, **kw)

    def normalize(input, p=2.0, dim=1, eps=1e-12, out=None):
        if input.is_cuda and input.dtype == torch.float32 and input.dim() == 4 and p == 2 and dim == 1 and eps == 1e-12 and out is None:
            return ops.l2norm(input)
        return _nm(input, p=p, dim=dim, eps=eps, out=out)

    F.grid_sample, F.interpolate, F.normalize = grid_sample, interpolate, normalize


def setup(script_path):
    for p in (PKG, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.setdefault("RFX_COARSE_VARIANT", variant_for(script_path))
    install_torchvision_shim()
    install_kornia_shim()
    install_misc_shims()
    for name in ("outil", "model", "coarseAlignFeatMatch"):
        sys.modules[name] = importlib.import_module(name)
    if os.environ.get("RFX_PATCH_FUNCTIONAL", "1") == "1":
        patch_functional()
    if os.environ.get("RFX_RANSAC_SEED"):
        seed_ransac_calls(int(os.environ["RFX_RANSAC_SEED"]))


def seed_ransac_calls(seed):
    """Reproducible runs (``RFX_RANSAC_SEED=S``): reseed the CPU generator with S + k before the k-th ``outil.RANSAC`` call.  The
    drop-in draws its hypotheses from that generator exactly like a CPU run of the reference does (utils/outil.py:120), so a
    CPU run of the reference seeded the same way scores the same 4-point samples (evaluation/evalKITTI/evaluation.py:182 seeds
    once per process; per call makes the draw independent of how much of the stream earlier stages consumed)."""
    import torch
    outil, n = sys.modules["outil"], [0]
    real = outil.RANSAC

    def RANSAC(*a, **k):
        torch.manual_seed(seed + n[0])
        n[0] += 1
        return real(*a, **k)
    outil.RANSAC = RANSAC


def main():
    # the script path comes from argv or, for harnesses that cannot pass arguments, from RFX_REFERENCE_SCRIPT (absolute, or
    # relative to RFX_REFERENCE_ROOT -- e.g. a reference tree mounted on the GPU box); its own arguments follow
    argv = sys.argv[1:]
    env_script = os.environ.get("RFX_REFERENCE_SCRIPT")
    if env_script and (not argv or argv[0].startswith("-")):
        root = os.environ.get("RFX_REFERENCE_ROOT", "")
        argv = [env_script if os.path.isabs(env_script) else os.path.join(root, env_script)] + argv
    if not argv:
        print(__doc__)
        sys.exit(2)
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script) and script.endswith(".py") and os.path.isfile(script + "c"):
        script += "c"                      # a byte-compiled deployment of the reference tree (align2images.pyc): runpy runs it as is
    if not os.path.isfile(script):
        sys.exit("run_reference_script: no such script: %s" % script)
    setup(script)
    sys.argv = [script] + argv[1:]
    os.chdir(os.path.dirname(script))
    # ``python script.py`` puts the script's directory first on sys.path (evaluation/evalHpatch/evaluation.py:9 imports its
    # sibling ``utils``); the drop-ins stay in front of same-named siblings because they are already in sys.modules
    sys.path.insert(0, os.path.dirname(script))
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
