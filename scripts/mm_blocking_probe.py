#!/usr/bin/env python
"""EVIDENCE SCRIPT -- how does the host's torch.mm (the reference's score product, utils/outil.py:34) sum over k, and does the
device's mutual-NN kernel reproduce it BIT FOR BIT?

CPU part: a float32 emulation of "fma chain inside blocks of KC products, block sums added to a running total" for a range of
KC, compared element by element with torch.mm on post-ReLU, L2-normalised columns (C = 1024).  On MKL 2024.2 (the BLAS of this
torch build) KC = 384 reproduces torch.mm exactly, on the authoring container's Xeon and -- this script answers it -- on the GPU
box's host.  GPU part (when a device is visible): the per-row score maxima that mnn_tile_*_kernel leaves in its workspace, for
score_chunk = one chain / 192 / 256 / 384 products, compared with the row maxima of torch.mm: the share of bit-equal rows.

    python scripts/mm_blocking_probe.py [--out profiles/r04_mm_blocking_probe.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))


def blocked(a, b, kc):
    K, nA, nB = a.shape[0], a.shape[1], b.shape[1]
    tot = np.zeros((nA, nB), np.float32)
    for k0 in range(0, K, kc):
        acc = np.zeros((nA, nB), np.float32)
        for k in range(k0, min(K, k0 + kc)):
            acc = (acc.astype(np.float64) + np.outer(a[k], b[k])).astype(np.float32)     # fma: exact product, one rounding
        tot = (tot.astype(np.float64) + acc).astype(np.float32)
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    a_ = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    C, nA, nB = 1024, 384, 256
    A = F.normalize(torch.relu(torch.randn(C, nA, generator=g)), dim=0)
    B = F.normalize(torch.relu(A[:, torch.randint(nA, (nB,), generator=g)] + 0.5 * torch.randn(C, nB, generator=g)), dim=0)
    out = dict(blas=[l.strip() for l in torch.__config__.show().split("\n") if "Math Kernel" in l][:1],
               cpu=os.popen("lscpu | grep 'Model name' | head -1").read().strip(), C=C, cpu_emulation={})
    a64, b64 = A.numpy().astype(np.float64), B.numpy().astype(np.float64)
    for th in (1, 8):
        torch.set_num_threads(th)
        mm = (A.t() @ B).numpy()
        row = {}
        for kc in (128, 192, 256, 320, 336, 384, 448, 512, 1024):
            x = blocked(a64, b64, kc)
            row[str(kc)] = dict(share_of_elements_differing=float((x != mm).mean()), max_abs=float(np.abs(x.astype(np.float64) - mm).max()))
        out["cpu_emulation"]["threads_%d" % th] = row
        print(th, {k: round(v["share_of_elements_differing"], 4) for k, v in row.items()})
    if torch.cuda.is_available():
        from rfx import ops, _lib
        lib = _lib.load()
        out["host_sgemm_k_block"] = ops.host_sgemm_k_block()
        print("ops.host_sgemm_k_block():", out["host_sgemm_k_block"])
        dev = torch.device("cuda:0")
        nA2, nB2 = 3000, 1100
        A2 = F.normalize(torch.relu(torch.randn(C, nA2, generator=g)), dim=0)
        B2 = F.normalize(torch.relu(A2[:, torch.randint(nA2, (nB2,), generator=g)] + 0.5 * torch.randn(C, nB2, generator=g)), dim=0)
        torch.set_num_threads(8)
        S = A2.t() @ B2
        cpu_rowmax, cpu_arg = S.max(dim=1)
        al = lambda x: (x + 255) & ~255
        tA, tB = (nA2 + 127) // 128, (nB2 + 127) // 128
        o_rowval = 2 * al(tB * nA2 * 4) + 2 * al(tA * nB2 * 4)
        Ad, Bd = A2.to(dev), B2.to(dev)
        out["device_vs_torch_mm_row_maxima"] = {}
        for chunk in ("0", "6", "8", "12"):
            ws = torch.zeros(lib.rfx_mutual_nn_ws_bytes(nA2, nB2), dtype=torch.uint8, device=dev)
            i1 = torch.empty(nB2, dtype=torch.int64, device=dev); i2 = torch.empty_like(i1)
            cnt = torch.zeros(1, dtype=torch.int32, device=dev)
            ops._call("rfx_mutual_nn_f32", dev, ops._p(Ad), nA2, nA2, ops._p(Bd), nB2, nB2, C, ops._p(None), ops._p(i1), ops._p(i2), ops._p(cnt),
                      ops._p(ws), int(chunk) * 32 if chunk != "0" else -1)       # score_chunk argument (ABI 8): products per chunk; < 0 = one chain
            rowval = ws[o_rowval:o_rowval + 4 * nA2].view(torch.float32).cpu()
            r = dict(share_of_rows_bit_equal=float((rowval == cpu_rowmax).float().mean()), max_abs=float((rowval - cpu_rowmax).abs().max()))
            out["device_vs_torch_mm_row_maxima"]["score_chunk=%s" % (int(chunk) * 32 if chunk != "0" else "chain")] = r
            print("device chunk", chunk, r)
    if a_.out:
        os.makedirs(os.path.dirname(os.path.abspath(a_.out)), exist_ok=True)
        json.dump(out, open(a_.out, "w"), indent=1)


if __name__ == "__main__":
    main()
