"""Drop-in for the flow-assembly functions of the reference's three ``getResults.py`` scripts (which are scripts,
not modules: their bodies parse argv and walk dataset folders that are out of scope).  Same names, argument lists
and return conventions (CPU tensors, ``[]`` sentinels); the arithmetic runs in librfx on the HIP device:

    from getResults import hpatch, corr, kitti
    flow = hpatch.getFlow_all(pairID, finePath, coarsePath, flowList, multiH, warper, grid, th, outW, outH)
    flow, match = corr.getFlow(pairID, finePath, flowList, coarsePath, maskPath, multiH, th)
    flow = kitti.getFlow_all(pairID, predDir, nbH, res_name, warper_org, multiH, grid_org, th, cc_th, interpolate)

Reference: evaluation/evalHpatch/getResults.py:16-80, evaluation/evalCorr/getResults.py:78-164,
evaluation/evalKITTI/getResults.py:95-152.
"""
import os
import sys
import types

_PKG = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from rfx import assemble as _a  # noqa: E402


def _cpu(x):
    if isinstance(x, tuple):
        return tuple(_cpu(v) for v in x)
    return x.cpu() if hasattr(x, "cpu") else x


def _wrap(fn):
    def call(*args, **kw):
        return _cpu(fn(*args, **kw))
    call.__doc__ = fn.__doc__
    call.__name__ = fn.__name__
    return call


hpatch = types.SimpleNamespace(getFlow_all=_wrap(_a.hpatch_getFlow_all),
                               getFlow_onlyCoarse=_wrap(_a.hpatch_getFlow_onlyCoarse))
corr = types.SimpleNamespace(getFlow=_wrap(_a.corr_getFlow), getFlow_Coarse=_wrap(_a.corr_getFlow_Coarse))
kitti = types.SimpleNamespace(getFlow_all=_wrap(_a.kitti_getFlow_all),
                              getFlow_onlyCoarse=_wrap(_a.kitti_getFlow_onlyCoarse))
