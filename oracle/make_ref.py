#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY -- recipe that builds ``oracle/_ref/``: the reference's hot path in COMPILED form.

``/root/reference`` does not exist on the GPU box, and reference SOURCES are never copied into this repository.  What this
recipe produces is the Python analogue of compiling a C reference into ``oracle/_ref/*.so``: every hot-path file of the
reference is byte-compiled FROM WHERE IT LIES (``compile(source, path, "exec")`` -> a ``.pyc`` written by
``importlib._bootstrap_external._code_to_timestamp_pyc``) into the git-ignored directory ``oracle/_ref/`` under the
reference's own relative path.  The directory travels to the GPU box with the snapshot exactly like ``librfx.so`` does and is
importable there through Python's sourceless-module machinery (same interpreter, same bytecode magic -- checked on load).

Besides whole modules, the statements that ``oracle/ref_loader.py`` compiles OUT of the reference's scripts -- the named
top-level functions (``script_functions``) and the driver ``while`` loops (``script_loop``) -- are compiled here with the same
AST selection and stored as marshalled code objects under ``oracle/_ref/_extract/``.

Also staged (data, not code): the two sample images quick_start/align2images.py:107-108 defaults to.

Run by ``__graft_entry__.build()`` whenever ``/root/reference`` is present; ``python oracle/make_ref.py`` by hand.
Consumers: ``oracle/ref_loader.py`` (the checker), ``tests/``, ``bench.py``'s cpu_baseline / parity children.  Nothing under
``ransac-flow_amd/`` may read ``oracle/_ref`` (tests/test_dropin_cpu.py greps for it).
"""
import ast
import hashlib
import importlib.util
import json
import marshal
import os
import shutil
import sys
from importlib._bootstrap_external import _code_to_timestamp_pyc

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.environ.get("RFX_REFERENCE_SOURCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")

# the files on (or either side of) the hot path
MODULES = [
    "utils/outil.py",
    "model/model.py", "model/downsample.py", "model/resnet50.py",
    "quick_start/coarseAlignFeatMatch.py", "quick_start/align2images.py",
    "evaluation/evalHpatch/coarseAlignFeatMatch.py", "evaluation/evalHpatch/evaluation.py", "evaluation/evalHpatch/utils.py",
    "evaluation/evalHpatch/getResults.py",
    "evaluation/evalKITTI/coarseAlignFeatMatch.py", "evaluation/evalKITTI/evaluation.py", "evaluation/evalKITTI/getResults.py",
    "evaluation/evalCorr/getResults.py", "evaluation/evalCorr/coarseAlignFeatMatch.py", "evaluation/evalCorr/evaluation.py",
    "evaluation/evalYFCC/coarseAlignFeatMatch.py", "evaluation/evalYFCC/evaluation.py",
    # SURVEY.md 8f4 (round 6): the sky-segmentation forward pass and the vendored Synchronized-BatchNorm package its model imports
    "segNet/segModel.py", "segNet/segData.py", "segNet/segEval.py",
    "segNet/lib/nn/__init__.py", "segNet/lib/nn/modules/__init__.py", "segNet/lib/nn/modules/batchnorm.py",
    "segNet/lib/nn/modules/comm.py", "segNet/lib/nn/modules/replicate.py",
    "segNet/lib/nn/parallel/__init__.py", "segNet/lib/nn/parallel/data_parallel.py",
]
# scripts whose functions / loops ref_loader extracts (the module level of these parses argv and walks a dataset)
EXTRACT = [
    "evaluation/evalHpatch/evaluation.py", "evaluation/evalHpatch/getResults.py",
    "evaluation/evalKITTI/evaluation.py", "evaluation/evalKITTI/getResults.py",
    "evaluation/evalCorr/getResults.py", "quick_start/align2images.py",
]
DATA = ["img/ArtMiner_Detail_Res13_10.png", "img/ArtMiner_Detail_Res13_11.png"]


def extract_codes(source, path):
    """{'functions': {name: code}, 'loops': {ast.dump(test): code}} -- the same selection ref_loader makes from source."""
    tree = ast.parse(source, filename=path)
    fns, loops = {}, {}
    for n in tree.body:
        if isinstance(n, ast.FunctionDef):
            fns[n.name] = compile(ast.Module(body=[n], type_ignores=[]), path, "exec")
    for n in ast.walk(tree):
        if isinstance(n, ast.While):
            key = ast.dump(n.test)
            if key not in loops:
                loops[key] = compile(ast.Module(body=[n], type_ignores=[]), path, "exec")
    return {"functions": fns, "loops": loops}


def build(src=SRC, out=OUT, verbose=True):
    if not os.path.isdir(os.path.join(src, "utils")):
        raise RuntimeError("no reference tree at %s" % src)
    if os.path.isdir(out):
        shutil.rmtree(out)
    manifest = dict(python=sys.version.split()[0], magic=importlib.util.MAGIC_NUMBER.hex(), source_root=src, files={})
    for rel in MODULES:
        path = os.path.join(src, rel)
        raw = open(path, "rb").read()
        code = compile(raw, path, "exec", dont_inherit=True)
        dst = os.path.join(out, rel[:-3] + ".pyc")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, "wb") as f:
            f.write(_code_to_timestamp_pyc(code, 0, len(raw)))
        manifest["files"][rel] = dict(sha256=hashlib.sha256(raw).hexdigest(), lines=raw.count(b"\n"))
    for rel in EXTRACT:
        path = os.path.join(src, rel)
        dst = os.path.join(out, "_extract", rel[:-3] + ".marshal")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, "wb") as f:
            marshal.dump(extract_codes(open(path).read(), path), f)
    for rel in DATA:
        dst = os.path.join(out, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(src, rel), dst)
    json.dump(manifest, open(os.path.join(out, "MANIFEST.json"), "w"), indent=1)
    if verbose:
        print("oracle/_ref: %d compiled modules, %d extracts, %d data files (python %s)"
              % (len(MODULES), len(EXTRACT), len(DATA), manifest["python"]))
    return out


if __name__ == "__main__":
    build()
