#!/usr/bin/env python
"""Micro-benchmark of the 7x7 correlation kernel's tile variants (rfx_corr_neigh_variant_f32) on the bench shape
(N x 256 x 60 x 80, N = 64 / 128) with ROTATING input sets (3 x 630 MB at N = 64 > the 256 MB Infinity Cache, as in
the pipeline where x / y were just written by the L2-norm kernel), HIP-event timed on the launch stream.
Prints one line per (N, variant): avg us, algorithmic GB/s, fraction of the 8 TB/s HBM peak; checks every variant
bit-for-bit against variant 3 and variant 3 against a float64 einsum on a slice.

    python scripts/ubench/corr_bench.py [--n 64 128] [--variants 1 2 3 4 5 6 7 8 9] [--iters 30]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))
import torch  # noqa: E402
from rfx import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[64, 128])
    ap.add_argument("--variants", type=int, nargs="+", default=[3, 1, 2, 4, 5, 6, 21, 22])
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--shape", type=int, nargs=3, default=[256, 60, 80])
    ap.add_argument("--sets", type=int, default=3)
    ap.add_argument("--out", type=str, default=None)
    ap.add_argument("--bidir", type=int, default=0, help="1: images 0..N/2-1 = (a,b), N/2.. = (b,a) [both directions, far "
                    "apart]; 2: (a0,b0),(b0,a0),(a1,b1),.. interleaved [both directions of a pair on neighbouring workgroups]")
    ap.add_argument("--pairs", action="store_true", help="also time both directions of N pairs: one BIDIR launch vs two launches")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    C, H, W = a.shape
    rows = []
    for N in a.n:
        g = torch.Generator(device=dev).manual_seed(N)
        sets = [(torch.nn.functional.normalize(torch.randn(N, C, H, W, device=dev, generator=g), dim=1),
                 torch.nn.functional.normalize(torch.randn(N, C, H, W, device=dev, generator=g), dim=1)) for _ in range(a.sets)]
        if a.bidir:
            ns = []
            for (xa, xb) in sets:
                h = N // 2
                A, B = xa[:h], xb[:h]
                if a.bidir == 1:
                    ns.append((torch.cat((A, B)), torch.cat((B, A))))
                else:
                    ns.append((torch.stack((A, B), 1).reshape(N, C, H, W), torch.stack((B, A), 1).reshape(N, C, H, W)))
            sets = ns
        ref = ops.corr_neigh(*sets[0], variant=3)
        for k in range(60):                       # clock ramp: the first kernels after start-up run at idle clocks
            ops.corr_neigh(*sets[k % a.sets], variant=3)
        # float64 check of the reference variant on one image
        x0, y0 = sets[0][0][:1].double(), sets[0][1][:1].double()
        yp = torch.nn.functional.pad(y0, (3, 3, 3, 3))
        chk = torch.stack([(x0 * yp[:, :, i:i + H, j:j + W]).sum(1) for i in range(7) for j in range(7)], dim=1)
        err = float((ref[:1].double() - chk).abs().max())
        assert err < 1e-5, err
        nbytes = (2 * C + 49) * 4.0 * N * H * W
        for v in a.variants:
            try:
                out = ops.corr_neigh(*sets[0], variant=v)
            except Exception as e:  # noqa: BLE001
                print("N=%d variant %d: %s" % (N, v, e))
                continue
            same = bool(torch.equal(out, ref))
            for k in range(3):
                ops.corr_neigh(*sets[k % a.sets], variant=v)
            evs = []
            for k in range(a.iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.corr_neigh(*sets[k % a.sets], variant=v)
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
            avg, med, best = sum(ts) / len(ts), ts[len(ts) // 2], ts[0]
            row = dict(N=N, bidir=a.bidir, variant=v, avg_us=round(avg, 1), med_us=round(med, 1), min_us=round(best, 1),
                       gbs=round(nbytes / avg / 1e3, 1), frac=round(nbytes / avg / 1e3 / 8000.0, 4), bit_identical=same,
                       f64_err_ref=err)
            rows.append(row)
            print(json.dumps(row), flush=True)
        # both directions of a pair from one pass (ops.corr_neigh_bidir) against two one-direction launches over the same pairs
        if a.pairs:
            out2 = torch.empty((2 * N, 49, H, W), device=dev)
            r12, r21 = ops.corr_neigh(*sets[0]), ops.corr_neigh(sets[0][1], sets[0][0])
            b12, b21 = ops.corr_neigh_bidir(*sets[0], out=out2)
            same = bool(torch.equal(b12, r12) and torch.equal(b21, r21))
            for mode in ("bidir", "two_launches"):
                evs = []
                for k in range(a.iters + 3):
                    x, y = sets[k % a.sets]
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    if mode == "bidir":
                        ops.corr_neigh_bidir(x, y, out=out2)
                    else:
                        ops.corr_neigh(x, y)
                        ops.corr_neigh(y, x)
                    e1.record()
                    evs.append((e0, e1))
                torch.cuda.synchronize()
                ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs[3:])
                avg = sum(ts) / len(ts)
                minb, perdir = (2 * C + 98) * 4.0 * N * H * W, 2 * nbytes
                row = dict(N=N, pairs=N, mode=mode, avg_us=round(avg, 1), min_us=round(ts[0], 1), bit_identical=same,
                           frac_min_traffic=round(minb / avg / 1e3 / 8000.0, 4), frac_per_direction=round(perdir / avg / 1e3 / 8000.0, 4))
                rows.append(row)
                print(json.dumps(row), flush=True)
            del out2
        del sets, ref
        torch.cuda.empty_cache()
    if a.out:
        json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
