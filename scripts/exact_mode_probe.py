#!/usr/bin/env python
"""BASELINE config 3 step (64 pairs 480x640, multi-homography loop) timed in the four driver modes:
   {device null vector, host LAPACK (exact)} x {one lock-step group, RFX_MULTIH_SPLIT groups on streams}
and the per-round cost of the exact mode's host stage (flagged samples, dgesdd calls after the duplicate cache, host ms).
Checks on the way: the records of the split and unsplit drivers are bit-identical (per mode).
    python scripts/exact_mode_probe.py [--batch 64] [--steps 5] [--split 2] [--out profiles/r06_exact_mode_probe.json]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--split", type=int, default=2)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch
    from rfx import weights, synth, ops, _lapack
    from rfx.pipeline import AlignPipeline
    dev = torch.device("cuda:0")
    B = a.batch
    sds = dict(trunk=weights.resnet50_trunk_sd(0), feat=weights.feature_extractor_sd(1), flow=weights.net_flow_coarse_sd(2),
               match=weights.net_matchability_sd(3, last_std=3.0))
    seeds = list(range(B))
    out = {"lapack": _lapack.info(), "batch": B, "steps": a.steps, "modes": {}}
    recs = {}
    for degen in ("device", "lapack"):
        pipe = AlignPipeline(sds, nbScale=7, nbIter=10000, tolerance=0.05, minSize=480, scaleR=2.0, variant="B", device=dev,
                             draw="device", seed=1000, degenerate=degen, score_chunk="host")
        raw = pipe.upload_raw([synth.make_pair(480, 640, seed=s, homography=True) for s in seeds])
        for split in (1, a.split):
            def step():
                prep = pipe.prepare_device(*raw)
                R = ops.MultiHRecords(B, 60, 80, dev, max_h=11)
                pipe.multi_h_batched(prep, maxCoarse=10, maskRegionTh=0.01, records=R, want_lists=False, pair_ids=seeds, split=split)
                return R.rec
            pipe.exact_log = None
            step(); step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                rec = step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            pipe.exact_log = [] if degen == "lapack" else None
            rec = step()
            torch.cuda.synchronize()
            key = "%s_split%d" % (degen, split)
            recs[key] = rec.clone()
            m = {"ms_per_step": round(ms, 2), "pairs_per_s": round(B / ms * 1e3, 2), "homographies": int(rec[:, 0].sum())}
            if pipe.exact_log:
                L = pipe.exact_log
                m["rounds"] = [dict(round=r["round"], group=r["lo"], active=r["active"], flagged=int(sum(r["n_degenerate"])),
                                    solved=int(r["n_solved"]), host_ms=round(r["host_ms"], 3)) for r in L]
                m["host_ms_per_step"] = round(sum(r["host_ms"] for r in L), 2)
                m["flagged_per_step"] = int(sum(sum(r["n_degenerate"]) for r in L))
                m["solved_per_step"] = int(sum(r["n_solved"] for r in L))
            out["modes"][key] = m
            print(key, {k: v for k, v in m.items() if k != "rounds"}, flush=True)
        del pipe
        torch.cuda.empty_cache()
    for degen in ("device", "lapack"):
        k1, k2 = "%s_split1" % degen, "%s_split%d" % (degen, a.split)
        out["%s_split_records_bit_identical" % degen] = bool(torch.equal(recs[k1], recs[k2]))
    d, l = recs["device_split1"], recs["lapack_split1"]
    out["pairs_whose_records_differ_device_vs_lapack"] = int((d != l).any(dim=1).sum())
    base = out["modes"]["device_split1"]["pairs_per_s"]
    for k, m in out["modes"].items():
        m["vs_device_split1"] = round(m["pairs_per_s"] / base, 4)
    print(json.dumps({k: v for k, v in out.items() if k != "modes"}))
    print(json.dumps({k: {kk: vv for kk, vv in m.items() if kk != "rounds"} for k, m in out["modes"].items()}))
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
