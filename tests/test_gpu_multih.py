"""The device kernels that took the host out of the multi-homography rounds (csrc/multih.hip), the two-direction
correlation (csrc/corr.hip, BIDIR) and the HIP-graph trunk path, each against an independent formulation: the ATen glue
they replaced (evaluation/evalHpatch/coarseAlignFeatMatch.py:156-170, evaluation/evalHpatch/evaluation.py:211-243 as the
per-pair drivers still spell it), the numpy Philox restatement (tests/philox_ref.py, pinned on the published known-answer
vectors by tests/test_philox_cpu.py), and two one-direction correlation launches.  ``-m gpu``."""
import numpy as np
import pytest
import torch

import philox_ref
from rfx import ops, weights, synth
from rfx.pipeline import AlignPipeline, cell_coords

pytestmark = pytest.mark.gpu


def test_cell_coordinates_are_the_cpu_reference_values_bit_for_bit(dev):
    """outil.getWHTensor's cell centres as the CPU reference computes them (true division): the RANSAC inputs must be the
    oracle's exactly -- ATen's device kernel for tensor / scalar multiplies by a rounded reciprocal and lands one ulp away
    on 40 % of the cells, which a 4-point DLT amplifies by its conditioning."""
    import restate
    for (r, c) in [(30, 40), (60, 80), (25, 33), (50, 165), (45, 60), (7, 23)]:
        Wd, Hd = cell_coords(r, c, dev)
        Wo, Ho = restate.get_wh(r, c)
        assert torch.equal(Wd.cpu(), Wo) and torch.equal(Hd.cpu(), Ho)
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ransac-flow_amd", "dropin", "outil.py")
    spec = importlib.util.spec_from_file_location("_rfx_dropin_outil_probe", path)      # private name: no sys.modules["outil"]
    outil = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(outil)
    W, H = outil.getWHTensor(torch.empty((1, 4, 25, 33), device=dev))
    Wo, Ho = restate.get_wh(25, 33)
    assert torch.equal(W.cpu(), Wo) and torch.equal(H.cpu(), Ho)


def test_device_index_draw_is_philox_modulo_the_device_counts(dev):
    n = torch.tensor([1200, 4, 0, 7, 8531, 3], dtype=torch.int32, device=dev)
    for seed, stream in ((0, 1), (0x1234_5678_9ABC_DEF0, 0xFFFF_FFFF_0000_0003)):
        got = ops.draw_samples(n, 1000, seed, stream).cpu().numpy()
        ref = philox_ref.draw_samples(n.cpu().numpy(), 1000, seed, stream)
        assert np.array_equal(got, ref)
    # uniform over [0, n): every residue of a small range appears, mean near (n-1)/2
    big = ops.draw_samples(torch.tensor([1200], dtype=torch.int32, device=dev), 50000, 7, 9).cpu().numpy()
    assert big.min() == 0 and big.max() == 1199 and abs(big.mean() - 599.5) < 5
    # a different stream id / pair index gives different draws
    a = ops.draw_samples(n[:1].repeat(2), 100, 1, 1).cpu()
    assert not torch.equal(a[0], a[1])
    assert not torch.equal(a[0], ops.draw_samples(n[:1], 100, 1, 2).cpu()[0])


def _aten_filter(idx1, idx2, cnt, b, Mask, bg, rt, ct, HA, WA, Ht, Wt):
    """The glue the kernel replaced, for one pair (pipeline.multi_h spells the same)."""
    n = int(cnt[b])
    i1, i2 = idx1[b, :n], idx2[b, :n]
    bgb = torch.ones_like(Mask[b]) if bg is None else bg[b]
    fg = ((Mask[b] + (1 - bgb)) > 0.5).float()
    keep = ops.resize_bilinear((1 - fg)[None, None], (rt, ct), align_corners=False)[0, 0] > 0.5
    valid = keep[i2 // ct, i2 % ct]
    ones = torch.ones(int(valid.sum()), device=Mask.device)
    m1 = torch.stack((HA[i1][valid], WA[i1][valid], ones), dim=1)
    m2 = torch.stack((Ht[i2][valid], Wt[i2][valid], ones), dim=1)
    return m1, m2, torch.nonzero(valid)[:, 0]


@pytest.mark.parametrize("shape", [(240, 320, 15, 20), (96, 312, 10, 33), (376, 1242, 50, 165)])
def test_filter_matches_equals_the_aten_glue(dev, shape):
    h, w, rt, ct = shape
    g = torch.Generator().manual_seed(h + w)
    B, nA = 4, 3 * rt * ct
    cap = rt * ct
    Wt, Ht = cell_coords(rt, ct, dev)
    WA, HA = torch.rand(nA, generator=g).to(dev), torch.rand(nA, generator=g).to(dev)
    cnt = torch.tensor([cap, cap // 2, 3, 0], dtype=torch.int32, device=dev)
    idx1 = torch.stack([torch.sort(torch.randperm(nA, generator=g)[:cap]).values for _ in range(B)]).to(dev)
    idx2 = torch.stack([torch.randperm(cap, generator=g) for _ in range(B)]).to(dev)
    # blobby 0/1 masks (non-integer resize ratios in two of the shapes) and a background map on half of the cases
    blob = torch.nn.functional.avg_pool2d(torch.rand(B, 1, h + 16, w + 16, generator=g), 17, 1)[:, 0]
    Mask = (blob > blob.median()).float().to(dev).contiguous()
    for bg in (None, (torch.rand(B, h, w, generator=g) > 0.2).float().to(dev)):
        for active in (None, torch.tensor([2, 0, 3, 1], dtype=torch.int32, device=dev), torch.tensor([1], dtype=torch.int32, device=dev)):
            m1, m2, n, kept = ops.filter_matches(idx1, idx2, cnt, active, Mask, bg, rt, ct, HA, WA, Ht, Wt, want_kept=True)
            order = list(range(B)) if active is None else active.cpu().tolist()
            for k, b in enumerate(order):
                r1, r2, rk = _aten_filter(idx1, idx2, cnt, b, Mask, bg, rt, ct, HA, WA, Ht, Wt)
                nk = int(n[k])
                assert nk == r1.shape[0]
                assert torch.equal(m1[k, :nk], r1) and torch.equal(m2[k, :nk], r2)
                assert torch.equal(kept[k, :nk].long(), rk)
                assert float(m1[k, nk:].abs().max() if nk < cap else 0) == 0 and (nk == cap or int(kept[k, nk:].max()) == -1)


@pytest.mark.parametrize("mode", [0, 1])
def test_multih_accept_equals_the_reference_rule(dev, mode):
    g = torch.Generator().manual_seed(40 + mode)
    B, h, w, h8, w8 = 5, 96, 136, 12, 17
    act = torch.tensor([4, 0, 2, 3], dtype=torch.int32, device=dev)
    a = act.shape[0]
    # saturating "matchability": many exact ones, values straddling 0.9999
    logits = torch.randn(a, h, w, generator=g) * 12
    match = torch.sigmoid(logits).to(dev).contiguous()
    Mask0 = (torch.rand(B, h, w, generator=g) > 0.7).float().to(dev).contiguous()
    bg = (torch.rand(B, h, w, generator=g) > 0.1).float().to(dev)
    res = torch.tensor([[0, 50, 3, 280], [0, 9, 1, 290], [1, 0, -1, 290], [0, 12, 8, 300]], dtype=torch.int32, device=dev)
    n_match = torch.tensor([300, 3, 200, 40], dtype=torch.int32, device=dev)
    nbH0 = torch.tensor([1, 0, 2, 0, 3], dtype=torch.int32, device=dev)        # pair 3: first homography -> accepted regardless of gain
    bestH = torch.randn(a, 3, 3, generator=g).to(dev)
    f8 = torch.randn(a, 2, h8, w8, generator=g).to(dev)
    m12, m21 = torch.rand(a, 1, h8, w8, generator=g).to(dev), torch.rand(a, 1, h8, w8, generator=g).to(dev)
    fd2 = torch.randn(a, 2, 6, 9, generator=g).to(dev) if mode else None
    for use_bg in (True, False):
        bgx = bg if use_bg else None
        bgv = bg if use_bg else torch.ones_like(bg)
        fg = ((Mask0[act.long()] + (1 - bgv[act.long()])) > 0.5).float()
        stat = (match > 0.9999).float() * (1 - fg) if mode else match * (1 - fg)
        gain_ref = stat.double().mean(dim=(1, 2)).float()
        th = float(gain_ref.sort().values[1] + gain_ref.sort().values[2]) / 2          # some above, some below
        ok = (n_match >= 4) & (res[:, 0] == 0) & ((gain_ref > th) | (nbH0[act.long()] == 0))
        upd = Mask0[act.long()] + match * (1 - fg)
        new_ref = ((upd > 0.9999) if mode else (upd >= 1.0)).float()
        Mask, nbH = Mask0.clone(), nbH0.clone()
        R = ops.MultiHRecords(B, h8, w8, dev, max_h=4, hd2=6 if mode else 0, wd2=9 if mode else 0)
        acc, gain = ops.multih_accept(match, Mask, bgx, act, res, n_match, nbH, th, mode, bestH=bestH, flowDown8=f8, match12Down8=m12,
                                      match21Down8=m21, flowD2=fd2, records=R)
        assert torch.equal(acc.bool(), ok), (acc, ok, gain, gain_ref)
        assert (gain - gain_ref).abs().max() < 1e-7
        nbv, status, RH, Rf, Rm, Rd2 = R.views()
        for k, b in enumerate(act.tolist()):
            if ok[k]:
                assert torch.equal(Mask[b], new_ref[k])
                s = int(nbH0[b])
                assert int(nbH[b]) == s + 1 and float(nbv[b]) == s + 1 and float(status[b]) == 0
                if s < 4:
                    assert torch.equal(RH[b, s], bestH[k]) and torch.equal(Rf[b, s], f8[k])
                    assert torch.equal(Rm[b, s, 0], m12[k, 0]) and torch.equal(Rm[b, s, 1], m21[k, 0])
                    if mode:
                        assert torch.equal(Rd2[b, s], fd2[k])
            else:
                assert torch.equal(Mask[b], Mask0[b]) and int(nbH[b]) == int(nbH0[b]) and float(status[b]) == 1
        assert torch.equal(Mask[1], Mask0[1])                                            # a pair outside the active list is untouched
        assert ok.any() and not ok.all()


@pytest.mark.parametrize("shape", [(32, 16, 60, 80), (16, 8, 120, 64), (16, 8, 120, 48), (12, 8, 90, 120), (6, 8, 81, 268),   # tuned 80/64/48-column tiles
                                   (512, 2, 64, 32), (300, 2, 64, 32), (2, 64, 17, 24),                                        # 64- / 32- / 16-row x 16-column tiles
                                   (3, 32, 9, 26), (2, 64, 41, 134)])                                                          # widths that are not a multiple of 4
def test_two_direction_correlation_is_bit_identical_to_two_launches(dev, shape):
    """corr(y, x) is corr(x, y) at mirrored taps / shifted pixels (same channel-ordered products): the BIDIR epilogue must
    reproduce the second launch bit for bit, borders included, on every tile shape the automatic choice can take."""
    N, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.nn.functional.normalize(torch.randn(N, C, H, W, generator=g), dim=1).to(dev)
    y = torch.nn.functional.normalize(torch.randn(N, C, H, W, generator=g), dim=1).to(dev)
    c12, c21 = ops.corr_neigh(x, y), ops.corr_neigh(y, x)
    both = torch.full((2 * N, 49, H, W), float("nan"), device=dev)
    b12, b21 = ops.corr_neigh_bidir(x, y, out=both)
    assert torch.equal(b12, c12)
    assert torch.equal(b21, c21)
    assert torch.equal(both[:N], c12) and torch.equal(both[N:], c21)          # every element written (no NaN left)
    # the index identity itself, on the one-direction results
    i, j, r, c = 1, 5, H // 2, W // 2
    assert float(c21[0, (6 - i) * 7 + (6 - j), r + i - 3, c + j - 3]) == float(c12[0, i * 7 + j, r, c])


def test_graphed_trunk_pass_equals_the_eager_pass(dev, monkeypatch):
    """B <= 4: the trunk pass is captured into a HIP graph at the SECOND sighting of a shape and replayed afterwards.  Two
    different inputs of one shape must give their own, eager-identical features through the replay (a capture taken on the
    wrong device / stream would replay stale outputs), and the capture cache is bounded."""
    pipe = AlignPipeline(dict(trunk=weights.resnet50_trunk_sd(0)), nbScale=3, nbIter=10, tolerance=0.05, minSize=160, scaleR=1.2,
                         device=dev)
    preps = [pipe.prepare([synth.make_pair(128, 160, seed=s)]) for s in (1, 2, 3)]
    monkeypatch.setenv("RFX_GRAPH", "0")
    monkeypatch.setenv("RFX_GROUPED", "0")
    eager = [pipe.features(p) for p in preps]              # one launch per layer AND level, 8 streams: the reference form
    monkeypatch.setenv("RFX_GROUPED", "1")
    grouped = [pipe.features(p) for p in preps]            # one grouped launch per kernel instance and layer (blockIdx.y = image)
    for got, ref in zip(grouped, eager):
        assert torch.equal(got["featA"], ref["featA"]) and torch.equal(got["featB"], ref["featB"])
    monkeypatch.setenv("RFX_GRAPH", "1")
    first = pipe.features(preps[0])                        # first sighting: eager
    assert not getattr(pipe, "_graphs", {})
    second = pipe.features(preps[1])                       # second sighting: capture + replay
    assert len(pipe._graphs) == 1
    third = pipe.features(preps[2])                        # replay
    fourth = pipe.features(preps[0])
    for got, ref in ((first, eager[0]), (second, eager[1]), (third, eager[2]), (fourth, eager[0])):
        assert torch.equal(got["featA"], ref["featA"]) and torch.equal(got["featB"], ref["featB"])
    assert not torch.equal(second["featA"], third["featA"])
    # bounded cache: more shapes than MAX_GRAPHS, each seen twice
    for k in range(pipe.MAX_GRAPHS + 2):
        p = pipe.prepare([synth.make_pair(96 + 16 * k, 160, seed=k)])
        pipe.features(p); pipe.features(p)
    assert len(pipe._graphs) <= pipe.MAX_GRAPHS


def test_graph_from_raw_images_equals_the_eager_calls(dev, monkeypatch):
    """prepare_and_features: device pyramid + trunk pass as ONE graph from the raw uint8 images (B <= 4).  Three different pairs of
    one shape through first sighting (eager), capture and replay must equal the eager prepare_device + features, bit for bit,
    for the features AND the tensors the fine stage reads afterwards."""
    pipe = AlignPipeline(dict(trunk=weights.resnet50_trunk_sd(0)), nbScale=3, nbIter=10, tolerance=0.05, minSize=160, scaleR=1.2,
                         device=dev)
    raws = [pipe.upload_raw([synth.make_pair(128, 160, seed=s)]) for s in (4, 5, 6)]
    monkeypatch.setenv("RFX_GRAPH", "0")
    refs = []
    for r in raws:
        p = pipe.prepare_device(*r)
        f = pipe.features(p)
        refs.append((p["IsTensor"].clone(), p["ItTensor"].clone(), f["featA"].clone(), f["featB"].clone()))
    monkeypatch.setenv("RFX_GRAPH", "1")
    for k in (0, 1, 2, 0):
        p, f = pipe.prepare_and_features(*raws[k])
        got = (p["IsTensor"], p["ItTensor"], f["featA"], f["featB"])
        for g, r in zip(got, refs[k]):
            assert torch.equal(g, r), k
    assert any(key[0] == "raw" for key in pipe._graphs)
    res = pipe.align_prepared(p, fine=False, feats=f)
    assert len(res) == 1


def test_lock_step_driver_device_draw_records_and_determinism(dev):
    """The throughput form as bench.py runs it: device-side draw (no explicit samples), result records filled on the device.
    Same seed + same call sequence -> the same homographies; the records hold exactly what the per-pair lists hold; the
    RANSAC of every round is the oracle's on the device's own matches and draws (checked through the recorded H)."""
    import restate
    sds = dict(trunk=weights.resnet50_trunk_sd(0), feat=weights.feature_extractor_sd(1), flow=weights.net_flow_coarse_sd(2),
               match=weights.net_matchability_sd(3, last_std=3.0))
    pipe = AlignPipeline(sds, nbScale=3, nbIter=300, tolerance=0.05, minSize=240, scaleR=1.2, variant="B", device=dev, seed=5)
    assert pipe.draw == "device"
    pairs = [synth.make_pair(240, 320, seed=s, homography=True) for s in (7, 8, 9, 10)]
    prep = pipe.prepare_device(*pipe.upload_raw(pairs))
    feats = pipe.features(prep)
    runs = []
    for _ in range(2):
        pipe.reseed(5)
        R = ops.MultiHRecords(4, 30, 40, dev)
        outs = pipe.multi_h_batched(prep, maxCoarse=3, maskRegionTh=0.01, feats=feats, records=R)
        runs.append((outs, R))
    (o1, R1), (o2, R2) = runs
    assert torch.equal(R1.rec, R2.rec)
    nbv, status, RH, Rf, Rm, _ = R1.views()
    assert max(o["nbH"] for o in o1) >= 2
    for b, o in enumerate(o1):
        assert int(nbv[b]) == o["nbH"] == len(o["H"]) and float(status[b]) == (0.0 if o["H"] else 1.0)
        for k in range(o["nbH"]):
            assert torch.equal(RH[b, k], o["H"][k]) and torch.equal(Rf[b, k], o["flowDown8"][k][0])
            assert torch.equal(Rm[b, k], o["matchDown8"][k][0])
        assert float(RH[b, o["nbH"]:].abs().max()) == 0
    # first round by hand: the device draw of driver call 1, round 0 is Philox(seed 5, stream (1 << 32) | 0, ids = batch
    # positions); RANSAC on the device's matches with that draw must be the oracle's
    idx1, idx2, cnt = pipe._mutual_batched(feats, 4)
    Mask = torch.zeros((4, 240, 320), device=dev)
    M1, M2, n = ops.filter_matches(idx1, idx2, cnt, None, Mask, None, feats["rt"], feats["ct"], feats["HA"], feats["WA"], feats["Ht"], feats["Wt"])
    assert torch.equal(n, cnt)
    smp = philox_ref.draw_samples(n.cpu().numpy(), 300, 5, 1 << 32)
    for b in range(4):
        nb_ = int(n[b])
        Hb, c, inl, _ = restate.ransac(M1[b, :nb_].cpu(), M2[b, :nb_].cpu(), 0.05, torch.from_numpy(smp[b]))
        assert np.abs(Hb - o1[b]["H"][0].cpu().numpy()).max() <= 1.2e-7


def test_lock_step_driver_with_hopeless_pairs(dev):
    """A batch holding an alignable pair, a pair whose target is ALL background (It_bg = 0: every cached match is masked out,
    fewer than 4 survive -> the reference returns its None sentinel before RANSAC,
    evaluation/evalHpatch/coarseAlignFeatMatch.py:165-166) and a noise pair: the device-resident rounds must neither crash nor
    let one pair disturb another -- the good pair equals the same pair run alone (same draws), the record of a pair without
    homography carries status 1 / nbH 0 / zero payload."""
    import PIL.Image as Image
    rng = np.random.RandomState(3)
    good = synth.make_pair(240, 320, seed=7, homography=True)
    blank = (good[0], Image.fromarray(np.full((240, 320, 3), 127, dtype=np.uint8)))
    noise = (Image.fromarray((rng.rand(240, 320, 3) * 255).astype(np.uint8)), Image.fromarray((rng.rand(240, 320, 3) * 255).astype(np.uint8)))
    sds = dict(trunk=weights.resnet50_trunk_sd(0), feat=weights.feature_extractor_sd(1), flow=weights.net_flow_coarse_sd(2),
               match=weights.net_matchability_sd(3, last_std=3.0))
    pipe = AlignPipeline(sds, nbScale=3, nbIter=300, tolerance=0.05, minSize=240, scaleR=1.2, variant="B", device=dev)

    def draws_for_pair0():
        calls = [0]

        def fn(b, n, it):
            if b != 0:
                return torch.randint(n, (it, 4))
            calls[0] += 1
            return torch.randint(n, (it, 4), generator=torch.Generator().manual_seed(31 * calls[0]))
        return fn
    alone = pipe.multi_h_batched(pipe.prepare_device(*pipe.upload_raw([good])), maxCoarse=3, sample_fn=draws_for_pair0())
    prep = pipe.prepare_device(*pipe.upload_raw([good, blank, noise]))
    R = ops.MultiHRecords(3, 30, 40, dev)
    bg = torch.ones((3, 240, 320), device=dev)
    bg[1] = 0
    outs = pipe.multi_h_batched(prep, maxCoarse=3, sample_fn=draws_for_pair0(), records=R, It_bg=bg)
    assert outs[0]["nbH"] == alone[0]["nbH"] >= 1
    for k in range(outs[0]["nbH"]):
        assert torch.equal(outs[0]["H"][k], alone[0]["H"][k]) and torch.equal(outs[0]["flowDown8"][k], alone[0]["flowDown8"][k])
    assert torch.equal(outs[0]["mask"], alone[0]["mask"])
    nbv, status, RH, Rf, Rm, _ = R.views()
    assert torch.isfinite(R.rec).all()
    for b, o in enumerate(outs):
        assert int(nbv[b]) == o["nbH"] and float(status[b]) == (0.0 if o["nbH"] else 1.0)
        if o["nbH"] == 0:
            assert float(RH[b].abs().max()) == 0 and float(Rf[b].abs().max()) == 0
    assert outs[1]["nbH"] == 0 and float(outs[1]["mask"].sum()) == 0       # all-background target: no match survives, no homography
    # and the device-draw default on the same batch: runs, deterministic under reseed
    pipe.reseed(11); a = pipe.multi_h_batched(prep, maxCoarse=2, It_bg=bg)
    pipe.reseed(11); b = pipe.multi_h_batched(prep, maxCoarse=2, It_bg=bg)
    assert [o["nbH"] for o in a] == [o["nbH"] for o in b] and all(torch.equal(x["mask"], y["mask"]) for x, y in zip(a, b))


def test_grouped_launches_are_bit_identical_to_single_launches(dev):
    """ops.launch_group on every convolution family the trunk uses (stem, k-major 1x1, generic 1x1 / strided, direct 3x3, fused
    Bottleneck tail) with 5 inputs of different sizes incl. a batch of 2: the recorded launches come back as ONE launch per
    kernel instance and must reproduce the single launches bit for bit; more than 8 problems split into several launches; an
    exception inside the block drops the recording."""
    from rfx import nets
    trunk = nets.ResNet50Trunk(weights.resnet50_trunk_sd(0), dev)
    g = torch.Generator().manual_seed(3)
    sizes = [(1, 96, 128), (1, 80, 112), (2, 64, 96), (1, 48, 80), (1, 112, 144)]
    xs = [torch.randn(n, 3, h, w, generator=g).to(dev) for n, h, w in sizes]
    ref = [trunk(x) for x in xs]
    got = trunk.forward_group(xs)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    # 11 problems of one kernel instance: 8 + 3
    blk = trunk.blocks[0]
    ys = [ops.stem_conv7_maxpool(x, trunk.conv1) for x in xs] * 2 + [ops.stem_conv7_maxpool(xs[0], trunk.conv1)]
    single = [blk["c1"](y) for y in ys]
    with ops.launch_group(dev):
        many = [blk["c1"](y) for y in ys]
    for a, b in zip(many, single):
        assert torch.equal(a, b)
    # an error inside the block aborts the recording; the next launch is immediate again
    with pytest.raises(ValueError):
        with ops.launch_group(dev):
            blk["c1"](ys[0])
            raise ValueError("boom")
    assert torch.equal(blk["c1"](ys[0]), single[0])


def test_device_draw_is_keyed_by_pair_id_and_round_not_by_batch_composition(dev):
    """with absolute ``pair_ids`` a pair's hypotheses -- hence its homographies -- are a function of (seed, id,
    round): the same pair gives bit-identical records alone, inside a batch of four, and in a batch where it sits at another
    position next to pairs that stop at other rounds (which is what sharding a stream over N ranks changes)."""
    sds = dict(trunk=weights.resnet50_trunk_sd(0), feat=weights.feature_extractor_sd(1), flow=weights.net_flow_coarse_sd(2),
               match=weights.net_matchability_sd(3, last_std=3.0))
    pipe = AlignPipeline(sds, nbScale=3, nbIter=300, tolerance=0.05, minSize=240, scaleR=1.2, variant="B", device=dev, seed=5)
    seeds = [7, 8, 9, 10]
    pairs = {s: synth.make_pair(240, 320, seed=s, homography=True) for s in seeds}

    def run(order):
        prep = pipe.prepare_device(*pipe.upload_raw([pairs[s] for s in order]))
        R = ops.MultiHRecords(len(order), 30, 40, dev)
        pipe.multi_h_batched(prep, maxCoarse=3, maskRegionTh=0.01, records=R, want_lists=False, pair_ids=order)
        return {s: R.rec[k].clone() for k, s in enumerate(order)}
    full = run(seeds)
    assert len({int(r[0]) for r in full.values()}) >= 2          # pairs stop at different rounds: the active list shrinks unevenly
    alone = run([9])
    shuffled = run([10, 9, 7])
    assert torch.equal(full[9], alone[9]) and torch.equal(full[9], shuffled[9])
    assert torch.equal(full[10], shuffled[10]) and torch.equal(full[7], shuffled[7])
    # the draw itself: ids enter the Philox counter where the batch position used to
    n = torch.tensor([900, 900], dtype=torch.int32, device=dev)
    a = ops.draw_samples(n, 64, 3, 2, torch.tensor([41, 5], dtype=torch.int32, device=dev)).cpu().numpy()
    assert np.array_equal(a, philox_ref.draw_samples([900, 900], 64, 3, 2, pair_ids=[41, 5]))
    assert np.array_equal(a[1], ops.draw_samples(n[:1], 64, 3, 2, torch.tensor([5], dtype=torch.int32, device=dev)).cpu().numpy()[0])


def test_record_overflow_is_flagged_not_silent(dev):
    """a pair that accepts more homographies than its record has slots gets status 3 and nbH clamped to max_h (the
    device counter keeps the true number); contiguity of the in-place operands is checked before any copy."""
    B, h, w, h8, w8 = 1, 16, 16, 2, 2
    R = ops.MultiHRecords(B, h8, w8, dev, max_h=2)
    mask = torch.zeros((B, h, w), device=dev)
    nbH = torch.zeros(B, dtype=torch.int32, device=dev)
    res = torch.zeros((1, 4), dtype=torch.int32, device=dev)
    n = torch.tensor([10], dtype=torch.int32, device=dev)
    for k in range(3):
        match = torch.zeros((1, h, w), device=dev)
        match[0, k * 4:(k + 1) * 4] = 1.0                              # every round explains a new band: always accepted
        acc, _ = ops.multih_accept(match, mask, None, None, res, n, nbH, 0.01, 0, bestH=torch.eye(3, device=dev)[None] * (k + 1),
                                   flowDown8=torch.full((1, 2, h8, w8), float(k), device=dev), match12Down8=torch.zeros((1, 1, h8, w8), device=dev),
                                   match21Down8=torch.zeros((1, 1, h8, w8), device=dev), records=R)
        assert int(acc[0]) == 1
        nbv, status, RH, _, _, _ = R.views()
        assert int(nbH[0]) == k + 1 and int(nbv[0]) == min(k + 1, 2) and float(status[0]) == (3.0 if k + 1 > 2 else 0.0)
    assert float(RH[0, 1, 0, 0]) == 2.0                                  # the slots hold homographies 1 and 2; the third was not stored
    with pytest.raises(ValueError):
        ops.multih_accept(match, torch.zeros((B, h, 2 * w), device=dev)[:, :, ::2], None, None, res, n, nbH, 0.01, 0)


def test_lock_step_groups_on_streams_give_the_one_group_records_bit_for_bit(dev):
    """multi_h_batched(split=k): the rounds of the batch as k lock-step groups on k HIP streams, driven as coroutines from one
    host thread (the exact mode's LAPACK stage and the accept readback of one group hide under the other groups' kernels).
    Every kernel computes a pair independently and the device draws are keyed by absolute pair id / position, so the records
    -- homography counts, homographies, /8 flows and matchability maps -- are the one-group driver's bit for bit: in the
    exact mode ("lapack", the default) and in the device-null-vector mode, with and without ``pair_ids``, for group counts
    that do and do not divide the batch."""
    sds = dict(trunk=weights.resnet50_trunk_sd(0), feat=weights.feature_extractor_sd(1), flow=weights.net_flow_coarse_sd(2),
               match=weights.net_matchability_sd(3, last_std=3.0))
    seeds = [3, 4, 5, 6, 7]
    pairs = [synth.make_pair(240, 320, seed=s, homography=True) for s in seeds]
    for degen in ("lapack", "device"):
        pipe = AlignPipeline(sds, nbScale=3, nbIter=2000, tolerance=0.05, minSize=240, scaleR=1.2, variant="B", device=dev, seed=5,
                             degenerate=degen)
        raw = pipe.upload_raw(pairs)

        def run(split, ids):
            pipe.reseed(5)
            pipe.exact_log = []
            R = ops.MultiHRecords(len(seeds), 30, 40, dev)
            outs = pipe.multi_h_batched(pipe.prepare_device(*raw), maxCoarse=4, maskRegionTh=0.01, records=R, pair_ids=ids, split=split)
            torch.cuda.synchronize()
            return R.rec.clone(), outs, list(pipe.exact_log)
        for ids in (seeds, None):
            one, outs1, log1 = run(1, ids)
            assert int(one[:, 0].sum()) >= len(seeds) + 2 and len({int(x) for x in one[:, 0]}) >= 2     # rounds happen, groups shrink unevenly
            for k in (2, 3, 5):
                rec, outs, logk = run(k, ids)
                assert torch.equal(rec, one), (degen, ids is None, k)
                for a, b in zip(outs, outs1):
                    assert a["nbH"] == b["nbH"] and torch.equal(a["mask"], b["mask"])
                    assert all(torch.equal(x, y) for x, y in zip(a["H"], b["H"]))
                if degen == "lapack":
                    assert {r["lo"] for r in logk} == {len(seeds) * g // k for g in range(k)}          # every group ran its own host stage
                    assert sum(sum(r["n_degenerate"]) for r in logk) == sum(sum(r["n_degenerate"]) for r in log1)
            assert (degen == "lapack") == bool(log1)
        pipe.exact_log = None
