#!/usr/bin/env python
"""Micro-benchmark of the mutual-NN tile kernel forms (RFX_MNN_FORM: 0 = k-major images, 1 = transposed images) on the bench
shapes: config 3 (64 pairs, nA 13 065 x nB 1 200), quick_start (64 x 8 531 x 1 200), config 5 (8 pairs, 25 747 x 8 250).
Times the whole rfx_mutual_nn_batched_f32 chain (tile + reduce + compact) with HIP events; checks the two forms' lists equal.

    python scripts/ubench/mnn_bench.py [--iters 10]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))
import torch  # noqa: E402
from rfx import ops, _lib  # noqa: E402


def run(B, nA, nB, C, iters, dev):
    lib = _lib.load()
    ld = (nA + 3) // 4 * 4
    g = torch.Generator(device=dev).manual_seed(nA)
    A = torch.nn.functional.normalize(torch.relu(torch.randn(B, C, ld, device=dev, generator=g)), dim=1)
    Bm = torch.nn.functional.normalize(torch.relu(torch.randn(B, C, nB, device=dev, generator=g)), dim=1)
    ws = torch.empty(lib.rfx_mutual_nn_ws_bytes(nA, nB) * B, dtype=torch.uint8, device=dev)
    cap = min(nA, nB)
    idx1 = torch.empty((B, cap), dtype=torch.int64, device=dev)
    idx2 = torch.empty((B, cap), dtype=torch.int64, device=dev)
    cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    out = {}
    for form in ("0", "1"):
        os.environ["RFX_MNN_FORM"] = form
        call = lambda: ops._call("rfx_mutual_nn_batched_f32", dev, p(A), ld, nA, C * ld, p(Bm), nB, nB, C * nB, C, ctypes.c_void_p(0),
                                 p(idx1), p(idx2), p(cnt), p(ws), B, 0)
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        n = cnt.cpu()
        out[form] = dict(ms=round(ms, 3), tflops=round(2.0 * B * nA * nB * C / ms / 1e9, 1), matches=int(n.sum()),
                         lists=(idx1.clone(), idx2.clone(), n))
    same = all(torch.equal(out["0"]["lists"][0][b, :int(out["0"]["lists"][2][b])], out["1"]["lists"][0][b, :int(out["1"]["lists"][2][b])]) and
               torch.equal(out["0"]["lists"][1][b, :int(out["0"]["lists"][2][b])], out["1"]["lists"][1][b, :int(out["1"]["lists"][2][b])]) for b in range(B))
    row = dict(B=B, nA=nA, nB=nB, C=C, kmajor=dict(ms=out["0"]["ms"], tflops=out["0"]["tflops"]),
               transposed=dict(ms=out["1"]["ms"], tflops=out["1"]["tflops"]), identical_lists=bool(same), matches=out["0"]["matches"])
    print(json.dumps(row), flush=True)
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = [run(64, 13065, 1200, 1024, a.iters, dev), run(64, 8531, 1200, 1024, a.iters, dev), run(8, 25747, 8250, 1024, max(2, a.iters // 3), dev)]
    if a.out:
        json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
