"""Offline flow assembly: what the reference's ``getResults.py`` scripts do with the arrays the
evaluation scripts saved -- per pair ``n`` homographies, their stride-8 fine flows and matchability maps -- to obtain
ONE dense flow (and confidence) at the evaluation resolution:

    coarse_i = warp_grid(H_i);  flowUp_i = clamp(up(flowDown_i) + identity, -1, 1);  flow_i = grid_sample(coarse_i, flowUp_i)
    score_i  = up(match12_i) [* grid_sample(up(match21_i), flowUp_i)] * in_bounds(flow_i)
    owner(p) = 0 if score_0(p) >= th else first i >= 1 with score_i(p) >= th (multiH) else 0

Reference: evaluation/evalHpatch/getResults.py:16-63, evaluation/evalCorr/getResults.py:78-136,
evaluation/evalKITTI/getResults.py:95-141.  Device work = librfx kernels (rfx_warp_grid_f32, rfx_compose_flow_f32,
rfx_resize_bilinear_f32, rfx_grid_sample_f32, rfx_match_score_f32, rfx_merge_multi_h_f32); reading the ``.npy`` files
KITTI's connected-component filter runs on the device (rfx_remove_small_cc_f32); its nearest-neighbour fill stays on the host like in the reference (scipy).

On-disk formats (written by evaluation/evalHpatch/evaluation.py:254-260 and friends):
    <fine>/flow_{id}_{n}H.npy   (n,2,h/8,w/8) float   fine flow, stride 8
    <fine>/mask_{id}_{n}H.npy   (n,2,h/8,w/8) float   matchability 1->2 | 2->1
    <coarse>/flow_{id}_{n}H.npy (n,3,3)        float   homographies
    <mask>/maskBG_{id}_{n}H.npy (h,w) bool              background mask (loaded by evalCorr, unused in the assembly)
    KITTI: Homograpy_{id}_{n}.npy, {res}_D2_{id}_{n}.npy, {res}_{id}_{n}.npy, {res}_Mask_{id}_{n}.npy, BG_{id}_{n}H.npy
"""
import os

import numpy as np
import torch

from . import ops


def find_nbH(pairID, flowList):
    """The reference's lookup (evalHpatch/getResults.py:17-22): first listed name whose second '_' field is the pair
    id; returns the text between the second '_' and 'H', or None."""
    for name in flowList:
        parts = name.split("_")
        if len(parts) > 2 and parts[1] == str(pairID):
            return parts[2].split("H")[0]
    return None


def _to_dev(a, device):
    """numpy array (the saved .npy files) or tensor (the views of an ops.MultiHRecords: no host round trip) -> float32 on ``device``."""
    if torch.is_tensor(a):
        return a.to(device=device, dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a).astype(np.float32))).to(device)


def assemble(flowDown, param, matchDown, out_hw, th, multiH, cycle, device="cuda"):
    """-> (flowGlobal (1,H,W,2), matchGlobal (1,H,W), binary (1,H,W) bool) device tensors."""
    H, W = int(out_hw[0]), int(out_hw[1])
    fd, pm, md = _to_dev(flowDown, device), _to_dev(param, device), _to_dev(matchDown, device)
    coarse = ops.warp_grid(pm, H, W)
    flow, inb, fup = ops.compose_flow(fd, coarse, clamp=True, want_inb=True, want_flow_up=cycle)
    m = ops.resize_bilinear(md, (H, W), align_corners=False)
    cyc = ops.grid_sample(m[:, 1:2].contiguous(), fup)[:, 0] if cycle else None
    return ops.merge_multi_h(flow, m[:, 0], th, multiH, cyc=cyc, inb=inb)


def remove_small_cc(score, cc_th, match_th=0.99):
    """evalKITTI/getResults.py:66-84 (skimage.measure.label on the host there) on the device: rfx_remove_small_cc_f32."""
    return ops.remove_small_cc(score, cc_th, match_th)


def interpolate_flow_match(flowGlobal, binary):
    """evalKITTI/getResults.py:87-93 on the host."""
    from scipy import ndimage
    idx = ndimage.distance_transform_edt((~binary[0]).cpu().numpy(), return_distances=False, return_indices=True)
    f = flowGlobal[0].cpu().numpy()[tuple(idx)]
    return torch.from_numpy(f).unsqueeze(0).to(flowGlobal.device)


def assemble_kitti(flowd2Down, flowDown, param, matchDown, out_hw, th, multiH, cc_th=0.0, interpolate=False,
                   device="cuda"):
    H, W = int(out_hw[0]), int(out_hw[1])
    d2d, fd = _to_dev(flowd2Down, device), _to_dev(flowDown, device)
    pm, md = _to_dev(param, device), _to_dev(matchDown, device)
    hom = ops.warp_grid(pm, H, W)
    d2, _, _ = ops.compose_flow(d2d, hom, clamp=True)
    flow, inb, fup = ops.compose_flow(fd, d2, clamp=True, want_inb=True, want_flow_up=True)
    m = ops.resize_bilinear(md, (H, W), align_corners=False)
    cyc = ops.grid_sample(m[:, 1:2].contiguous(), fup)[:, 0]
    if cc_th == 0:
        fg, mg, binary = ops.merge_multi_h(flow, m[:, 0], th, multiH, cyc=cyc, inb=inb)
    else:
        score = remove_small_cc(ops.match_score(m[:, 0], cyc=cyc, inb=inb), cc_th)
        fg, mg, binary = ops.merge_multi_h(flow, score, th, multiH)
    if interpolate:
        fg = interpolate_flow_match(fg, binary)
    return fg, mg, binary


# ---- file-level entry points with the reference's argument lists -------------------------------------------------

def _load(path):
    return np.load(path).astype(np.float32)


def hpatch_getFlow_all(pairID, finePath, coarsePath, flowList, multiH, warper, grid, th, outW, outH, device="cuda"):
    """evalHpatch/getResults.py:16-63.  ``warper`` / ``grid`` are accepted for signature parity and ignored (the grid
    and the homography warp are generated on the device).  Returns [] when the pair has no saved flow."""
    nbH = find_nbH(pairID, flowList)
    if nbH is None:
        return []
    flow = _load(os.path.join(finePath, "flow_{:d}_{}H.npy".format(pairID, nbH)))
    param = _load(os.path.join(coarsePath, "flow_{:d}_{}H.npy".format(pairID, nbH)))
    match = _load(os.path.join(finePath, "mask_{:d}_{}H.npy".format(pairID, nbH)))
    return assemble(flow, param, match, (outH, outW), th, multiH, False, device)[0]


def hpatch_getFlow_onlyCoarse(pairID, finePath, coarsePath, flowList, multiH, warper, grid, th, outW, outH,
                              device="cuda"):
    """evalHpatch/getResults.py:65-80: the first homography's grid."""
    nbH = find_nbH(pairID, flowList)
    if nbH is None:
        return []
    param = _load(os.path.join(coarsePath, "flow_{:d}_{}H.npy".format(pairID, nbH)))
    return ops.warp_grid(_to_dev(param[:1], device), int(outH), int(outW))


def corr_getFlow(pairID, finePath, flowList, coarsePath, maskPath, multiH, th, device="cuda"):
    """evalCorr/getResults.py:78-136 -> (flowGlobal (1,8h,8w,2), matchGlobal (1,8h,8w,1)) or ([], [])."""
    nbH = find_nbH(pairID, flowList)
    if nbH is None:
        return [], []
    flow = _load(os.path.join(finePath, "flow_{:d}_{}H.npy".format(pairID, nbH)))
    param = _load(os.path.join(coarsePath, "flow_{:d}_{}H.npy".format(pairID, nbH)))
    match = _load(os.path.join(finePath, "mask_{:d}_{}H.npy".format(pairID, nbH)))
    h, w = flow.shape[2], flow.shape[3]
    fg, mg, _ = assemble(flow, param, match, (h * 8, w * 8), th, multiH, True, device)
    return fg, mg.unsqueeze(3)


def corr_getFlow_Coarse(pairID, flowList, finePath, coarsePath, device="cuda"):
    """evalCorr/getResults.py:138-164 -> (grid of homography 0 at 8x, all-ones confidence)."""
    nbH = find_nbH(pairID, flowList)
    if nbH is None:
        return [], []
    flow = np.load(os.path.join(finePath, "flow_{:d}_{}H.npy".format(pairID, nbH)))
    param = _load(os.path.join(coarsePath, "flow_{:d}_{}H.npy".format(pairID, nbH)))
    h, w = flow.shape[2], flow.shape[3]
    return (ops.warp_grid(_to_dev(param[:1], device), h * 8, w * 8),
            torch.ones(1, h * 8, w * 8, 1, device=device))


def kitti_getFlow_all(pairID, predDir, nbH, res_name, warper_org, multiH, grid_org, th, cc_th, interpolate,
                      device="cuda"):
    """evalKITTI/getResults.py:95-141; the output size is that of ``grid_org`` (1,H,W,2), as in the reference."""
    param = _load(os.path.join(predDir, "Homograpy_{}_{}.npy".format(pairID, nbH)))
    flowd2 = _load(os.path.join(predDir, "{}_D2_{}_{}.npy".format(res_name, pairID, nbH)))
    flow = _load(os.path.join(predDir, "{}_{}_{}.npy".format(res_name, pairID, nbH)))
    match = _load(os.path.join(predDir, "{}_Mask_{}_{}.npy".format(res_name, pairID, nbH)))
    H, W = int(grid_org.shape[1]), int(grid_org.shape[2])
    return assemble_kitti(flowd2, flow, param, match, (H, W), th, multiH, cc_th, interpolate, device)[0]


def kitti_getFlow_onlyCoarse(pairID, predDir, nbH, res_name, warper_org, multiH, grid_org, th, cc_th, interpolate,
                             device="cuda"):
    """evalKITTI/getResults.py:144-152."""
    param = _load(os.path.join(predDir, "Homograpy_{}_{}.npy".format(pairID, nbH)))
    return ops.warp_grid(_to_dev(param[:1], device), int(grid_org.shape[1]), int(grid_org.shape[2]))
