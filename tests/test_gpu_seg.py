"""SURVEY.md 8f4 -- the sky-segmentation forward pass on the device (rfx/segnet.py, csrc/seg.hip, rfx_conv2d_dilated_f32) against
torch's CPU kernels op by op, against the committed golden output of the REFERENCE's own SegNet.getSky
(tests/golden/seg.npz, made by tests/golden/make_golden.py::gen_seg from segNet/segEval.py + segNet/segModel.py), and end to end
against the reference / its restatement on this box's CPU with a float64 proof obligation for every pixel whose class differs.
``-m gpu``."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import restate
from rfx import ops, weights, synth
from rfx.ops import ConvPlan, ACT_RELU
from rfx.segnet import SegNetDevice, input_sizes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bn(c, g):
    return dict(weight=1.0 + 0.2 * (torch.rand(c, generator=g) - 0.5), bias=0.1 * torch.randn(c, generator=g),
                running_mean=0.1 * torch.randn(c, generator=g), running_var=1.0 + 0.4 * (torch.rand(c, generator=g) - 0.5))


@pytest.mark.parametrize("dil,cin,cout,hw", [(2, 64, 96, (41, 52)), (4, 32, 160, (33, 47)), (2, 256, 256, (25, 32))])
def test_dilated_conv_is_the_zero_stuffed_convolution_bit_for_bit(dev, dil, cin, cout, hw):
    """rfx_conv2d_dilated_f32 (segNet/segModel.py:196-205: 3x3, dilation = padding = 2 / 4) against (a) torch's CPU convolution with
    ``dilation=`` (+ folded BN + residual + ReLU) and (b) the UNdilated entry point on the zero-stuffed (2d+1)x(2d+1) kernel: the
    stuffed zeros contribute fma(0, x, acc) = acc exactly and the non-zero taps keep their (channel, kh, kw) order, so the two device
    results must be equal bit for bit."""
    g = torch.Generator().manual_seed(dil * 100 + cin)
    x = torch.randn(2, cin, *hw, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cout)) ** 0.5
    bn = _bn(cout, g)
    res = torch.randn(2, cout, *hw, generator=g)
    plan = ConvPlan(w, bn, 1, dil, ACT_RELU, dev, dilation=dil)
    got = plan(x.to(dev), residual=res.to(dev))
    ref = F.relu(F.batch_norm(F.conv2d(x, w, padding=dil, dilation=dil), bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"],
                              False, 0.0, 1e-5) + res)
    assert got.shape == ref.shape and float((got.cpu() - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max()))
    k = 2 * dil + 1
    ws = torch.zeros(cout, cin, k, k)
    ws[:, :, ::dil, ::dil] = w
    stuffed = ConvPlan(ws, bn, 1, dil, ACT_RELU, dev)(x.to(dev), residual=res.to(dev))
    assert torch.equal(got, stuffed)


def test_adaptive_avgpool_softmax_argmax_match_torch(dev):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 96, 47, 63, generator=g)
    for s in (1, 2, 3, 6, (5, 7)):
        got = ops.adaptive_avgpool2d(x.to(dev), s).cpu()
        ref = F.adaptive_avg_pool2d(x, s)
        assert got.shape == ref.shape and float((got - ref).abs().max()) <= 1e-6
    lg = torch.randn(2, 150, 37, 51, generator=g) * 3
    sc = ops.softmax_accum(lg.to(dev), None, div=5.0)
    ref = F.softmax(lg, dim=1) / 5
    assert float((sc.cpu() - ref).abs().max()) < 2e-7
    lg2 = torch.randn(2, 150, 37, 51, generator=g) * 3
    assert ops.softmax_accum(lg2.to(dev), sc, div=5.0) is sc
    ref = ref + F.softmax(lg2, dim=1) / 5
    assert float((sc.cpu() - ref).abs().max()) < 3e-7
    pred_ref = ref.max(dim=1)[1]
    for cid in (int(pred_ref.flatten().mode()[0]), 3):
        m, pred = ops.argmax_mask(sc, cid, complement=False, want_pred=True)
        same = pred.cpu().long() == pred_ref
        assert float(same.float().mean()) > 0.9999                                   # (a float32 tie of two sums would be a flip)
        assert torch.equal(m.cpu()[same], (pred_ref == cid).float()[same])
        mc = ops.argmax_mask(sc, cid, complement=True)
        assert torch.equal(mc, 1 - m)
    with pytest.raises(ValueError):
        ops.softmax_accum(lg.to(dev), torch.zeros(2, 150, 37, 52, device=dev))


def _near_tie_report(pred_dev, scores64):
    """Pixels whose device class differs from the float64 arg-max must be float64 near-ties: the gap between the float64 score of
    the float64 winner and that of the device's class, relative to the winner."""
    s = scores64[0]
    best, arg = s.max(dim=0)
    diff = (pred_dev.long() != arg)
    if not bool(diff.any()):
        return 0, 0.0
    gap = (best - s.gather(0, pred_dev.long()[None])[0])[diff] / best[diff]
    return int(diff.sum()), float(gap.max())


def test_device_segnet_equals_the_reference_golden_and_the_cpu_run(dev):
    """SegNetDevice.get_sky (five scales: ResNet-50-dilated encoder, PPM decoder, softmax average, arg-max) against
    (1) tests/golden/seg.npz -- the reference's own SegNet.getSky on a 96x128 image: class map and both masks;
    (2) a fresh 200x264 image against the CPU checker of this box (the reference itself through oracle/ref_loader.load_seg where a
        reference tree is present, else the restatement oracle/restate.py::seg_get_sky, pinned on it): equal masks except at pixels
        whose class is a float64 near-tie (scores of the same expression evaluated in double)."""
    g = np.load(os.path.join(GOLD, "seg.npz"))
    enc, dec = weights.seg_encoder_sd(4, randomize_bn=True), weights.seg_decoder_sd(5, randomize_bn=True, logit_std=0.003)
    I1, _ = synth.make_pair(96, 128, seed=3)
    assert [list(s[::-1]) for s in input_sizes(128, 96)] == g["sizes"].tolist()
    seg_id = int(g["seg_id"])
    net = SegNetDevice(enc, dec, segId=seg_id, segFg=True, device=dev)
    sc = net.scores(I1)
    assert float((sc[0, :, ::8, ::8].cpu() - torch.from_numpy(g["scores_sub"])).abs().max()) < 2e-4
    mask, pred = net.get_sky_device(I1, want_pred=True)
    frac = float((pred.cpu().numpy() != g["pred"]).mean())
    print("golden: %.5f of the pixels differ in class" % frac)
    assert frac < 2e-3
    same = pred.cpu().numpy() == g["pred"]
    assert np.array_equal(mask.cpu().numpy()[same], g["mask_fg"].astype(np.float32)[same])
    net.segFg = False
    assert np.array_equal(net.getSky(I1)[same], g["mask_bg"].astype(np.float32)[same])
    # (2) fresh image, CPU checker of this box, float64 evidence
    I2, _ = synth.make_pair(200, 264, seed=11)
    net2 = SegNetDevice(enc, dec, segId=seg_id, segFg=False, device=dev)
    mask_d, pred_d = net2.get_sky_device(I2, want_pred=True)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    import ref_loader
    def _has_seg():
        try:
            return ref_loader.available() and os.path.isfile(ref_loader.ref_path("segNet/segModel.py"))
        except FileNotFoundError:
            return False
    if _has_seg():
        import tempfile
        d = tempfile.mkdtemp()
        pe, pd_, pi = (os.path.join(d, n) for n in ("e.pth", "d.pth", "i.png"))
        torch.save(enc, pe)
        torch.save(dec, pd_)
        I2.save(pi)
        S = ref_loader.load_seg()
        cpu_mask = ref_loader.quiet(S["segEval"].SegNet, pe, pd_, seg_id, False).getSky(pi)
        checker = ref_loader.kind()
    else:
        cpu_mask = restate.seg_get_sky(enc, dec, I2, seg_id, False)
        checker = "port (oracle/restate.py)"
    differ = mask_d.cpu().numpy() != cpu_mask
    print("checker: %s; mask pixels differing: %d of %d" % (checker, int(differ.sum()), differ.size))
    assert differ.mean() < 2e-3
    s64 = restate.seg_scores(enc, dec, I2, dtype=torch.float64)
    n_diff, worst = _near_tie_report(pred_d.cpu(), s64)
    print("device class != float64 arg-max at %d pixels, worst relative gap %.2e" % (n_diff, worst))
    assert worst < 2e-5
    arg64 = s64[0].max(dim=0)[1]
    cpu_is64 = torch.from_numpy(cpu_mask) == (arg64 == seg_id).float()
    assert not bool((torch.from_numpy(differ) & (pred_d.cpu().long() == arg64) & cpu_is64).any())   # every mask difference sits on a near-tie
