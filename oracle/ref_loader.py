"""TEST INFRASTRUCTURE ONLY -- loads the *real* reference (XiSHEN0220/RANSAC-Flow) on CPU.

This module imports the reference's own Python modules from ``/root/reference`` under three
stubs so that they run on a GPU-less host:

1. ``Tensor.cuda`` / ``Module.cuda`` -> identity, ``torch.cuda.FloatTensor = torch.FloatTensor``
   (the reference hard-codes ``.cuda()``: utils/outil.py:86, quick_start/coarseAlignFeatMatch.py:51,100,106).
2. a fake ``torchvision`` (``models.resnet50`` -> the reference's own model/resnet50.py:185,
   ``transforms.{ToTensor,Normalize,Compose,ToPILImage}``).
3. a fake ``kornia.geometry.HomographyWarper`` whose ``warp_grid`` restates kornia 0.1.4's
   documented behaviour (requirements.txt:59; kornia itself is not in /root/reference).

Where it finds the reference (``REF_ROOT``): ``$RFX_REFERENCE_ROOT`` if set; else ``/root/reference`` (the source tree:
authoring container only); else ``oracle/_ref`` -- the reference BYTE-COMPILED from where it lies by the committed recipe
``oracle/make_ref.py`` (run by ``__graft_entry__.build()``), git-ignored, which travels to the GPU box like a built ``.so``.
Both forms execute the same code objects: ``tests/test_oracle_pins.py::test_staged_reference_equals_the_source_tree``.

It is used by ``tests/golden/make_golden.py`` to generate the committed golden vectors, by the tests that pin
``oracle/restate.py`` against the real reference, and (through ``oracle/ref_oracle.py``) by the parity sweeps and
``bench.py``'s ``cpu_baseline`` leg.  Nothing in the product path (``ransac-flow_amd/``) may import it.
"""
import os
import sys
import types
import contextlib
import io

import numpy as np
import torch

STAGED_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def _default_root():
    env = os.environ.get("RFX_REFERENCE_ROOT")
    if env:
        return env
    if os.path.isdir("/root/reference/utils"):
        return "/root/reference"
    return STAGED_ROOT


REF_ROOT = _default_root()


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "utils"))


def staged() -> bool:
    """True when REF_ROOT holds the byte-compiled form (oracle/make_ref.py), not the source tree."""
    return available() and not os.path.isfile(os.path.join(REF_ROOT, "utils", "outil.py"))


def kind() -> str:
    return "reference (byte-compiled by oracle/make_ref.py)" if staged() else "reference (source tree)"


def ref_path(rel):
    """Path of a reference file: the source where the tree is mounted, else its compiled form under oracle/_ref."""
    p = os.path.join(REF_ROOT, rel)
    if os.path.isfile(p) or not rel.endswith(".py"):
        return p
    pc = p[:-3] + ".pyc"
    if os.path.isfile(pc):
        _check_magic()
        return pc
    raise FileNotFoundError("reference file %s not found under %s (run oracle/make_ref.py where /root/reference exists)" % (rel, REF_ROOT))


_magic_ok = []


def _check_magic():
    if _magic_ok:
        return
    import importlib.util
    import json
    m = json.load(open(os.path.join(REF_ROOT, "MANIFEST.json")))
    if m["magic"] != importlib.util.MAGIC_NUMBER.hex():
        raise RuntimeError("oracle/_ref was compiled by python %s (bytecode magic %s), this interpreter has %s: rebuild it"
                           % (m["python"], m["magic"], importlib.util.MAGIC_NUMBER.hex()))
    _magic_ok.append(True)


def _extract(rel_path):
    """Code objects oracle/make_ref.py compiled out of a reference script (staged form only)."""
    import marshal
    _check_magic()
    with open(os.path.join(REF_ROOT, "_extract", rel_path[:-3] + ".marshal"), "rb") as f:
        return marshal.load(f)


_loaded = {}


def _imresize(arr, size, interp="bilinear", mode=None):
    """``scipy.misc.imresize`` as evaluation/evalYFCC/evaluation.py:23,201,212 and evaluation/evalCorr/evaluation.py:24,186 call it
    (a 2-D array, ``size`` = (rows, cols)).  Third-party code that is NOT in /root/reference and not in the scipy the reference's
    own requirements.txt:110 pins (1.10.1 -- the function was removed in scipy 1.3.0, so the two scripts cannot even be imported
    under their pinned environment): restated from its last release, scipy 1.2.3 ``scipy/misc/pilutil.py`` -- ``toimage`` with
    ``bytescale`` (non-uint8 data are stretched from [min, max] to [0, 255], a constant array becomes all zeros; + 0.5, truncate),
    an 8-bit 'L' image, ``Image.resize((cols, rows), resample)``, ``fromimage``.  PARITY-UNPINNED against the original binary
    (absent); pinned on its documented behaviour.  The drop-in launcher installs the same restatement
    (ransac-flow_amd/dropin/run_reference_script.py), so both runs of a script see the same background map."""
    import PIL.Image as Image
    data = np.asarray(arr)
    if data.ndim != 2:
        raise NotImplementedError("imresize stand-in: 2-D arrays only (what the evaluation scripts pass)")
    if data.dtype != np.uint8:
        cmin, cmax = data.min(), data.max()
        cscale = (cmax - cmin) or 1
        data = (((data - cmin) * (255.0 / cscale)).clip(0, 255) + 0.5).astype(np.uint8)
    im = Image.frombytes("L", (data.shape[1], data.shape[0]), data.tobytes())
    resample = {"nearest": 0, "lanczos": 1, "bilinear": 2, "bicubic": 3, "cubic": 3}[interp]
    return np.asarray(im.resize((int(size[1]), int(size[0])), resample=resample))


def _install_stubs():
    # 1. .cuda() -> identity
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor

    # 2. torchvision shim
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvt = types.ModuleType("torchvision.transforms")

    def _resnet50(pretrained=False, **kw):
        if "resnet50" not in sys.modules:
            import importlib.util
            spec = importlib.util.spec_from_file_location("resnet50", ref_path("model/resnet50.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            sys.modules["resnet50"] = mod
        net = sys.modules["resnet50"].resnet50()
        pth = os.environ.get("RFX_TRUNK_WEIGHTS")          # shared with the drop-in (dropin/coarseAlignFeatMatch.py): no ImageNet weights offline
        if pth:
            sd = torch.load(pth, map_location="cpu")
            missing = net.load_state_dict(sd, strict=False)
            assert not missing.unexpected_keys, missing.unexpected_keys
        return net

    tvm.resnet50 = _resnet50

    class ToTensor:
        def __call__(self, pic):
            arr = np.asarray(pic, dtype=np.uint8)
            if arr.ndim == 2:
                arr = arr[:, :, None]
            t = torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous()
            return t.float().div(255)

    class Normalize:
        def __init__(self, mean, std):
            self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
            self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

        def __call__(self, t):
            return (t - self.mean) / self.std

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToPILImage:
        def __call__(self, t):
            import PIL.Image as Image
            arr = (t.detach().cpu().clamp(0, 1) * 255).byte().permute(1, 2, 0).numpy()
            return Image.fromarray(arr)

    tvt.ToTensor, tvt.Normalize, tvt.Compose, tvt.ToPILImage = ToTensor, Normalize, Compose, ToPILImage
    tvt.transforms = tvt                 # segNet/segData.py:2 ``from torchvision.transforms import transforms``
    tv.models, tv.transforms = tvm, tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm
    sys.modules["torchvision.transforms"] = tvt

    # 3. kornia shim (kornia==0.1.4.post2 HomographyWarper)
    ko = types.ModuleType("kornia")
    kg = types.ModuleType("kornia.geometry")

    class HomographyWarper:
        def __init__(self, height, width):
            self.height, self.width = height, width
            xs = torch.linspace(-1, 1, width)
            ys = torch.linspace(-1, 1, height)
            gx = xs.view(1, 1, width).expand(1, height, width)
            gy = ys.view(1, height, 1).expand(1, height, width)
            self.grid = torch.stack([gx, gy, torch.ones_like(gx)], dim=-1)  # 1,h,w,3

        def warp_grid(self, H):
            # (x', y', z') = H (x, y, 1); return (x'/z', y'/z')
            B = H.shape[0]
            pts = self.grid.expand(B, -1, -1, -1).reshape(B, -1, 3)
            out = torch.bmm(pts, H.transpose(1, 2))
            out = out[..., :2] / out[..., 2:]
            return out.view(B, self.height, self.width, 2)

    kg.HomographyWarper = HomographyWarper
    ko.geometry = kg
    sys.modules["kornia"] = ko
    sys.modules["kornia.geometry"] = kg

    # eval variants import scipy.misc.imresize (removed from scipy) and segEval
    import scipy
    sm = types.ModuleType("scipy.misc")
    sm.imresize = _imresize
    sys.modules["scipy.misc"] = sm
    scipy.misc = sm
    sys.modules.setdefault("segEval", types.ModuleType("segEval"))
    # evaluation/evalKITTI/evaluation.py:24 imports skimage.measure (label: 8-connectivity in 2-D) -- not installed: scipy's labelling
    if "skimage" not in sys.modules:
        try:
            import skimage  # noqa: F401
        except ImportError:
            from scipy import ndimage
            sk, skm = types.ModuleType("skimage"), types.ModuleType("skimage.measure")
            skm.label = lambda m, background=0: ndimage.label(m, structure=np.ones((3, 3), dtype=np.int32))[0]
            sk.measure = skm
            sys.modules["skimage"], sys.modules["skimage.measure"] = sk, skm
    if not hasattr(torch.cuda, "_rfx_long_stub"):
        torch.cuda.LongTensor = torch.LongTensor          # evaluation/evalKITTI/evaluation.py:217 (indexRoll, unused afterwards)
        torch.cuda._rfx_long_stub = True


def load():
    """Return dict with the reference modules: outil, model, resnet50, CoarseAlignA (quick_start), CoarseAlignB (evalHpatch)."""
    if _loaded:
        return _loaded
    if not available():
        raise RuntimeError("reference not available at %s" % REF_ROOT)
    _install_stubs()
    import importlib.util

    def _imp(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    sys.path.insert(0, os.path.join(REF_ROOT, "model"))
    sys.path.insert(0, os.path.join(REF_ROOT, "utils"))
    with contextlib.redirect_stdout(io.StringIO()):
        outil = _imp("outil", ref_path("utils/outil.py"))
        sys.modules["outil"] = outil
        downsample = _imp("downsample", ref_path("model/downsample.py"))
        sys.modules["downsample"] = downsample
        resnet50 = _imp("resnet50", ref_path("model/resnet50.py"))
        sys.modules["resnet50"] = resnet50
        model = _imp("ref_model", ref_path("model/model.py"))
        cwd = os.getcwd()
        try:
            os.chdir(os.path.join(REF_ROOT, "quick_start"))
            caA = _imp("ref_coarseAlignA", ref_path("quick_start/coarseAlignFeatMatch.py"))
            os.chdir(os.path.join(REF_ROOT, "evaluation", "evalHpatch"))
            # variant B refers to a module-level name `resnet50` only when imageNet=False
            caB = _imp("ref_coarseAlignB", ref_path("evaluation/evalHpatch/coarseAlignFeatMatch.py"))
        finally:
            os.chdir(cwd)
    _loaded.update(dict(outil=outil, model=model, resnet50=resnet50, downsample=downsample,
                        CoarseAlignA=caA.CoarseAlign, CoarseAlignB=caB.CoarseAlign,
                        kornia_geometry=sys.modules["kornia.geometry"]))
    return _loaded


_seg = {}


def load_seg():
    """The reference's sky-segmentation modules on the CPU: segNet/segModel.py, segData.py, segEval.py and the
    vendored Synchronized-BatchNorm package they import (segNet/lib/nn) -> dict(segModel, segData, segEval).  One more stub besides
    load()'s three: the vendored package reads ``collections.Mapping / Sequence`` (segNet/lib/nn/parallel/data_parallel.py), names
    Python 3.10 only keeps in ``collections.abc`` -- aliased here; with them the reference's own modules import unchanged.
    ``sys.modules["segEval"]`` becomes the REFERENCE's module (load()'s stubs seat an empty placeholder for scripts that never
    build a SegNet), so a CPU run of evaluation/evalHpatch/evaluation.py --segNet resolves ``import segEval`` to it."""
    if _seg:
        return _seg
    if not os.path.isfile(ref_path("segNet/segModel.py")):
        raise RuntimeError("reference segNet not available under %s" % REF_ROOT)
    _install_stubs()
    import collections
    import collections.abc
    import importlib.util
    for n in ("Mapping", "Sequence", "Iterable", "Callable", "MutableMapping"):
        if not hasattr(collections, n):
            setattr(collections, n, getattr(collections.abc, n))
    seg_dir = os.path.join(REF_ROOT, "segNet")
    if seg_dir not in sys.path:
        sys.path.insert(0, seg_dir)          # ``from lib.nn import SynchronizedBatchNorm2d`` (segNet/segModel.py:5)

    def _imp(name, rel):
        spec = importlib.util.spec_from_file_location(name, ref_path(rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod
    with contextlib.redirect_stdout(io.StringIO()):
        segModel = _imp("segModel", "segNet/segModel.py")
        segData = _imp("segData", "segNet/segData.py")
        segEval = _imp("segEval", "segNet/segEval.py")
    _seg.update(segModel=segModel, segData=segData, segEval=segEval)
    return _seg


def script_functions(rel_path, names):
    """Compile ONLY the named top-level functions of a reference *script* (e.g. evaluation/evalHpatch/getResults.py,
    whose module level parses argv and walks a dataset directory) and return them, bound to a namespace holding the
    modules those functions use (np, torch, F, os, and the kornia stub as ``tgm``).  The reference source is read
    and executed from where it lies; nothing is copied."""
    import ast
    _install_stubs()
    if staged():
        have = _extract(rel_path)["functions"]
        missing = set(names) - set(have)
        codes = [have[n] for n in have if n in names]            # dicts keep the script's definition order
    else:
        path = os.path.join(REF_ROOT, rel_path)
        tree = ast.parse(open(path).read(), filename=path)
        keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
        missing = set(names) - {n.name for n in keep}
        codes = [compile(ast.Module(body=keep, type_ignores=[]), path, "exec")]
    if missing:
        raise KeyError("not in %s: %s" % (rel_path, sorted(missing)))
    from scipy import ndimage
    # skimage is not installed: measure.label(mask, background=0) (8-connectivity in 2-D) -> scipy's labelling
    measure = types.SimpleNamespace(
        label=lambda m, background=0: ndimage.label(m, structure=np.ones((3, 3), dtype=np.int32))[0])
    ns = {"np": np, "torch": torch, "F": torch.nn.functional, "os": os, "tgm": sys.modules["kornia.geometry"],
          "nd": ndimage, "measure": measure}
    for code in codes:
        exec(code, ns)
    return {n: ns[n] for n in names}


def script_loop(rel_path, test_src, extra_ns=None):
    """Compile ONE loop statement of a reference *script* -- the first ``while`` whose condition reads ``test_src``
    (e.g. "nbCoarse <= args.maxCoarse" = the multi-homography driver of evaluation/evalHpatch/evaluation.py:211-243,
    or "True" = evaluation/evalKITTI/evaluation.py:270-336) -- and return ``run(ns)``, which executes that statement,
    from where it lies in the reference tree, in the caller's namespace ``ns`` (the variables the script's module level
    would have set up before the loop).  Nothing is copied: the source is parsed and executed in place."""
    import ast
    _install_stubs()
    want = ast.dump(ast.parse(test_src, mode="eval").body)
    if staged():
        code = _extract(rel_path)["loops"].get(want)
    else:
        path = os.path.join(REF_ROOT, rel_path)
        tree = ast.parse(open(path).read(), filename=path)
        code = None
        for node in ast.walk(tree):
            if isinstance(node, ast.While) and ast.dump(node.test) == want:
                code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
                break
    if code is None:
        raise KeyError("no `while %s` in %s" % (test_src, rel_path))
    from scipy import ndimage
    measure = types.SimpleNamespace(
        label=lambda m, background=0: ndimage.label(m, structure=np.ones((3, 3), dtype=np.int32))[0])
    base = {"np": np, "torch": torch, "F": torch.nn.functional, "os": os, "tgm": sys.modules["kornia.geometry"],
            "measure": measure}
    base.update(extra_ns or {})

    def run(ns):
        for k, v in base.items():
            ns.setdefault(k, v)
        exec(code, ns)
        return ns
    return run


def quiet(fn, *a, **k):
    """Run fn with stdout silenced (the reference prints scaleList / 'Not initializing')."""
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)
