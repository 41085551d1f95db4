"""Stage-wise parity: every librfx kernel (through the C ABI) against the CPU oracle on the same seeded
inputs, plus the reference golden vectors.  Needs a real MI355X: run with ``-m gpu``.

Tolerances: integer / index outputs bit exact; float32 outputs within round-off of the value range
(stated per test).  Where a float near-tie could flip an arg-max the test verifies that every
disagreement IS such a near-tie (margin check in float64) and bounds their rate."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import restate
from rfx import ops, nets, weights

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def test_flow_grad_clamp_matches_torch(dev):
    """rfx_flow_grad_clamp_f32 == model.predFlowCoarse's tail in ATen (model/model.py:333-340): clamp(flow^T + grid) bit for bit,
    flowGrad within one rounding of torch.norm; broadcast and per-sample grids, the NoGrad variant, odd sizes."""
    g = torch.Generator().manual_seed(2)
    for (B, H, W, gb) in ((1, 37, 53, 1), (3, 16, 24, 3), (2, 2, 2, 1), (4, 60, 80, 1)):
        f = torch.randn(B, 2, H, W, generator=g) * 0.7
        grid = torch.rand(gb, H, W, 2, generator=g) * 2.4 - 1.2
        fg, flow = ops.flow_grad_clamp(f.to(dev), grid.to(dev))
        ref_flow = torch.clamp(f.permute(0, 2, 3, 1) + grid, min=-1, max=1)
        ref_fg = torch.norm(f[:, :, 1:, 1:] - f[:, :, :-1, :-1], dim=1, keepdim=True)
        assert torch.equal(flow.cpu(), ref_flow)
        assert fg.shape == ref_fg.shape and (fg.cpu() - ref_fg).abs().max() <= 2.4e-7 * float(ref_fg.abs().max())
        fg2, flow2 = ops.flow_grad_clamp(f.to(dev), grid.to(dev), want_grad=False)
        assert fg2 is None and torch.equal(flow2, flow)


# ------------------------------------------------------------------ small-component filter
def test_remove_small_cc_equals_host_labelling(dev):
    """rfx_remove_small_cc_f32 (lock-free union-find on the device) == the reference's host filter
    (evaluation/evalKITTI/evaluation.py:85-100, restated in oracle/restate.py and pinned on the reference's own function):
    bit for bit on blob maps, diagonal-only (8-connected) chains, all-foreground / all-background maps, one-pixel images,
    components exactly at the area threshold, and a batch of maps in one launch."""
    g = torch.Generator().manual_seed(5)
    maps = []
    for (H, W, sigma) in ((61, 97, 3.0), (120, 333, 6.0), (47, 50, 1.0)):
        n = torch.randn(1, 1, H, W, generator=g)
        k = int(4 * sigma) | 1
        ax = torch.arange(k) - k // 2
        ker = torch.exp(-ax.float() ** 2 / (2 * sigma ** 2))
        ker = (ker[:, None] * ker[None, :]) / ker.sum() ** 2
        sm = F.conv2d(n, ker[None, None], padding=k // 2)[0, 0]
        maps.append((torch.sigmoid(40 * sm / sm.std()) * 0.999999).contiguous())
    diag = torch.zeros(40, 40)
    for i in range(30):
        diag[i, i] = 1.0                      # a chain that only 8-connectivity joins
        diag[39 - i, i] = 0.995 if i < 12 else 0.0
    maps += [diag, torch.ones(9, 13), torch.zeros(9, 13), torch.ones(1, 1), torch.full((3, 5), 0.99)]
    for m in maps:
        for cc_th in (0.01, 0.05, 0.3, 1.0):
            ref = restate.remove_small_cc_eval(m.numpy().copy(), 0.99, cc_th)
            out = ops.remove_small_cc(m.to(dev), cc_th, 0.99).cpu().numpy()
            assert np.array_equal(out, ref), (tuple(m.shape), cc_th)
    # a component of exactly the threshold area: 12 of 1200 pixels = 0.01 -> removed at cc_th 0.01, kept just below
    t = torch.zeros(30, 40)
    t[3, 5:17] = 1.0
    t[10:20, 10:30] = 1.0
    assert ops.cc_max_area(1200, 0.01) == 12
    assert np.array_equal(ops.remove_small_cc(t.to(dev), 0.01).cpu().numpy(), restate.remove_small_cc_eval(t.numpy().copy(), 0.99, 0.01))
    assert ops.remove_small_cc(t.to(dev), 0.01)[3, 5] == 0 and ops.remove_small_cc(t.to(dev), 0.0099)[3, 5] == 1
    # batch: the maps of one launch do not see each other
    b = torch.stack([maps[0][:40, :40], diag, maps[2][:40, :40]]).contiguous()
    outb = ops.remove_small_cc(b.to(dev), 0.05).cpu().numpy()
    for j in range(3):
        assert np.array_equal(outb[j], restate.remove_small_cc_eval(b[j].numpy().copy(), 0.99, 0.05))


# ------------------------------------------------------------------ conv family

CONV_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad, bn, res, act
    (1, 3, 64, 96, 64, 7, 2, 3, True, False, 1),      # ResNet conv1
    (2, 64, 16, 24, 64, 1, 1, 0, True, False, 1),     # bottleneck 1x1
    (2, 64, 16, 24, 256, 1, 1, 0, True, True, 1),     # 1x1 + residual
    (1, 128, 17, 23, 128, 3, 2, 1, True, False, 1),   # strided 3x3, odd sizes
    (1, 256, 8, 12, 512, 1, 2, 0, True, False, 0),    # strided 1x1 downsample
    (3, 49, 12, 16, 512, 3, 1, 1, True, False, 1),    # head conv1
    (1, 128, 12, 16, 49, 3, 1, 1, False, False, 0),   # head conv4 (Cout=49)
    (1, 128, 12, 16, 1, 3, 1, 1, False, False, 2),    # matchability conv4 + sigmoid
    (1, 3, 40, 56, 64, 3, 1, 1, True, False, 1),      # FeatureExtractor conv1 (K=27)
    (4, 256, 30, 40, 256, 3, 1, 1, True, True, 1),    # layer3-like 3x3 (<2,2> / <1,2> tiles)
    (16, 256, 30, 40, 1024, 1, 1, 0, True, True, 1),  # big enough for the 128x128 tile
    (2, 64, 19, 21, 64, 3, 1, 1, True, True, 1),      # direct 3x3 kernel <1>, ragged 8x16 patches + residual
    (8, 128, 60, 80, 256, 3, 1, 1, True, False, 1),   # direct 3x3 kernel <2>
    (1, 8, 5, 3, 130, 3, 1, 1, False, False, 0),      # direct 3x3: one K step, image smaller than a patch
    (2, 64, 25, 33, 64, 3, 1, 1, True, True, 1),      # direct 3x3 <1, 4>: 32x4 patch
    (64, 16, 25, 33, 128, 3, 1, 1, True, False, 1),   # direct 3x3 <2, 4>
    (48, 16, 30, 40, 256, 3, 1, 1, True, True, 1),    # direct 3x3 <2, 8>: 16x8 patch
    (3, 24, 28, 37, 40, 3, 1, 1, False, False, 0),    # direct 3x3 <1, 8>, Cout not a multiple of the tile
    (40, 64, 34, 45, 256, 1, 1, 0, True, True, 1),    # k-major 1x1 <2, false>: odd plane, scalar pixel loads
    (8, 96, 34, 45, 72, 1, 1, 0, True, False, 1),     # k-major 1x1 <1, false>: three K steps, ragged channel tile
    (9, 32, 30, 40, 64, 1, 1, 0, False, True, 0),     # k-major 1x1 <1, true>: ONE K step, ragged pixel tile
    (66, 128, 16, 20, 130, 1, 1, 0, True, True, 1),   # k-major 1x1 <2, true>: ragged channel + pixel tiles
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_matches_cpu(dev, case):
    N, Cin, H, W, Cout, k, stride, pad, bn, res, act = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bnd = None
    if bn:
        bnd = dict(weight=1 + 0.3 * torch.randn(Cout, generator=g), bias=0.2 * torch.randn(Cout, generator=g),
                   running_mean=0.2 * torch.randn(Cout, generator=g), running_var=0.5 + torch.rand(Cout, generator=g))
    ref = F.conv2d(x, w, stride=stride, padding=pad)
    if bn:
        ref = F.batch_norm(ref, bnd["running_mean"], bnd["running_var"], bnd["weight"], bnd["bias"], False, 0.0, 1e-5)
    r = torch.randn(ref.shape, generator=g) if res else None
    if res:
        ref = ref + r
    if act == 1:
        ref = F.relu(ref)
    elif act == 2:
        ref = torch.sigmoid(ref)
    plan = ops.ConvPlan(w, bnd, stride, pad, act, dev)
    out = plan(x.to(dev), residual=r.to(dev) if res else None)
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    assert relerr(out, ref) < 2e-5, relerr(out, ref)


@pytest.mark.parametrize("shape", [(2, 64, 19, 21, 64), (8, 128, 60, 80, 256), (1, 8, 5, 3, 130), (64, 16, 25, 33, 128),
                                   (48, 16, 30, 40, 256), (3, 24, 28, 37, 40), (5, 72, 33, 47, 192),
                                   # the images of a batch are tiled as one stack (conv3x3.hip): patches that span several tiny
                                   # images, one-row / one-pixel images, a stack that ends inside a patch
                                   (7, 16, 5, 3, 64), (9, 8, 1, 1, 8), (5, 8, 2, 70, 32), (33, 8, 3, 17, 136), (2, 8, 31, 5, 64),
                                   # Cin not a multiple of the 8-channel K step (the heads' 49-channel correlation volume): zero
                                   # weight rows + zero-filled patch slots for the missing channels of the last step
                                   (4, 49, 60, 80, 512), (2, 9, 7, 11, 64), (3, 15, 12, 9, 40), (1, 49, 5, 3, 49), (2, 17, 20, 33, 130),
                                   # one 64-channel tile + a launch large enough: 256-pixel (16x16) patches, TN = 4
                                   (24, 64, 120, 160, 64), (70, 8, 50, 70, 24), (170, 16, 33, 47, 64), (64, 49, 60, 80, 49)])
def test_direct_3x3_equals_implicit_gemm_bit_for_bit(dev, shape):
    """rfx_conv3x3_f32 (weights packed in the kernel's LDS order, include/rfx_api.h) == rfx_conv2d_f32 (generic wT / ktab
    packing) on the same layer, bit for bit: same k order in both kernels."""
    N, Cin, H, W, Cout = shape
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(N, Cin, H, W, generator=g).to(dev)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bnd = dict(weight=1 + 0.3 * torch.randn(Cout, generator=g), bias=0.2 * torch.randn(Cout, generator=g),
               running_mean=0.2 * torch.randn(Cout, generator=g), running_var=0.5 + torch.rand(Cout, generator=g))
    r = torch.randn(N, Cout, H, W, generator=g).to(dev)
    plan = ops.ConvPlan(w, bnd, 1, 1, ops.ACT_RELU, dev)
    assert plan.wP is not None
    direct = plan(x, residual=r)
    generic = torch.empty_like(direct)
    ops._call("rfx_conv2d_f32", x.device, ops._p(x), ops._p(plan.wT), ops._p(plan.ktab), ops._p(plan.scale), ops._p(plan.shift),
              ops._p(r), ops._p(generic), N, Cin, H, W, Cout, 3, 3, 1, 1, ops.ACT_RELU)
    torch.cuda.synchronize()
    assert torch.equal(direct, generic)


def lib_kid(N, Cin, Cout, Ho, Wo):
    from rfx import _lib
    return _lib.load().rfx_conv2d_kernel_id(N, Cin, Cout, 3, 3, 2, 1, Ho, Wo)


@pytest.mark.parametrize("shape", [(64, 64, 60, 80, 128), (16, 128, 120, 160, 256), (3, 8, 5, 3, 130), (2, 16, 1, 1, 8), (5, 24, 33, 47, 40),
                                   (4, 64, 34, 66, 64), (9, 8, 2, 70, 32), (2, 256, 31, 53, 256), (70, 16, 17, 18, 192), (1, 8, 16, 32, 64)])
def test_direct_3x3_stride2_equals_implicit_gemm_bit_for_bit(dev, shape):
    """Round 4: rfx_conv3x3_s2_f32 (direct 3x3 / stride 2 / pad 1 kernel, input patch de-interleaved by column parity in LDS)
    == rfx_conv2d_f32 on the same layer, bit for bit -- odd and even map sizes, maps smaller than a patch, ragged channel tiles,
    both tile heights -- and the ConvPlan routes stride-2 layers to it."""
    N, Cin, H, W, Cout = shape
    g = torch.Generator().manual_seed(Cin * 5 + Cout + H)
    x = torch.randn(N, Cin, H, W, generator=g).to(dev)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bnd = dict(weight=1 + 0.3 * torch.randn(Cout, generator=g), bias=0.2 * torch.randn(Cout, generator=g),
               running_mean=0.2 * torch.randn(Cout, generator=g), running_var=0.5 + torch.rand(Cout, generator=g))
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    r = torch.randn(N, Cout, Ho, Wo, generator=g).to(dev)
    plan = ops.ConvPlan(w, bnd, 2, 1, ops.ACT_RELU, dev)
    assert plan.wP is not None
    # the kernel itself, whatever the dispatcher would pick for this shape (it leaves badly padding maps to the implicit GEMM)
    direct = torch.empty((N, Cout, Ho, Wo), device=dev)
    ops._call("rfx_conv3x3_s2_f32", x.device, ops._p(x), ops._p(plan.wP), ops._p(plan.scale), ops._p(plan.shift), ops._p(r), ops._p(direct),
              N, Cin, H, W, Cout, ops.ACT_RELU)
    assert torch.equal(plan(x, residual=r), direct)                          # ConvPlan: either kernel, same bits
    generic = torch.empty_like(direct)
    ops._call("rfx_conv2d_f32", x.device, ops._p(x), ops._p(plan.wT), ops._p(plan.ktab), ops._p(plan.scale), ops._p(plan.shift),
              ops._p(r), ops._p(generic), N, Cin, H, W, Cout, 3, 3, 2, 1, ops.ACT_RELU)
    torch.cuda.synchronize()
    assert direct.shape == (N, Cout, Ho, Wo)
    ref = F.relu(F.batch_norm(F.conv2d(x.cpu(), w, stride=2, padding=1), bnd["running_mean"], bnd["running_var"], bnd["weight"],
                              bnd["bias"], False, 0.0, 1e-5) + r.cpu())
    assert relerr(direct, ref) < 2e-5
    if Cin * 9 >= 1152:
        # round 5: the long-K strided layers (ResNet-50 layer2.0 / layer3.0 conv2) close a chunk every 4 K steps in this kernel, the
        # implicit GEMM sums one chain: not the same bits any more -- and the blocked sum is the one closer to float64
        ref64 = F.relu(F.batch_norm(F.conv2d(x.cpu().double(), w.double(), stride=2, padding=1), bnd["running_mean"].double(),
                                    bnd["running_var"].double(), bnd["weight"].double(), bnd["bias"].double(), False, 0.0, 1e-5) + r.cpu().double())
        e_chunk = float(((direct.cpu().double() - ref64) ** 2).mean().sqrt())
        e_chain = float(((generic.cpu().double() - ref64) ** 2).mean().sqrt())
        print("stride-2 3x3, K = %d: rms error vs float64: chunked %.3e, chain %.3e" % (9 * Cin, e_chunk, e_chain))
        assert lib_kid(N, Cin, Cout, Ho, Wo) & 16384 and not torch.equal(direct, generic) and e_chunk < e_chain
    else:
        assert torch.equal(direct, generic)


def test_chunked_accumulation_of_the_long_k_layers(dev):
    """Round 4: the direct 3x3 kernel closes a chunk every 4 K steps (288 k) on layers with K = 9 Cin >= 2048 and adds the chunk
    sums to a running total -- the K-blocked sum of the CPU reference's GEMM / oneDNN kernels instead of ONE fma chain over 2304 /
    4608 products.  (1) the 64- and the 128-channel-tile instances (picked by launch size) agree bit for bit;
    (2) against float64 the chunked result carries clearly less round-off than the chain form, which rfx_conv2d_f32's
    implicit-GEMM kernel computes for the same layer; (3) layers below the threshold are untouched (bit-identical to the chain)."""
    g = torch.Generator().manual_seed(3)
    for Cin, Cout in ((256, 256), (512, 256)):
        x = torch.relu(torch.randn(48, Cin, 30, 40, generator=g)).to(dev)                  # post-ReLU activations, as in the trunk
        w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cout)) ** 0.5
        plan = ops.ConvPlan(w, None, 1, 1, ops.ACT_NONE, dev)
        big = plan(x)                                                                      # 48 images: 128-channel tiles
        small = plan(x[:2].contiguous())                                                   # 2 images: 64-channel tiles
        from rfx import _lib
        lib = _lib.load()
        assert lib.rfx_conv2d_kernel_id(48, Cin, Cout, 3, 3, 1, 1, 30, 40) & 3 == 0 and lib.rfx_conv2d_kernel_id(2, Cin, Cout, 3, 3, 1, 1, 30, 40) & 3 == 1
        assert torch.equal(big[:2], small)
        chain = torch.empty_like(small)
        ops._call("rfx_conv2d_f32", dev, ops._p(x[:2].contiguous()), ops._p(plan.wT), ops._p(plan.ktab), ops._p(None), ops._p(None), ops._p(None),
                  ops._p(chain), 2, Cin, 30, 40, Cout, 3, 3, 1, 1, ops.ACT_NONE)
        ref = F.conv2d(x[:2].cpu().double(), w.double(), padding=1)
        e_chunk = float(((small.cpu().double() - ref) ** 2).mean().sqrt())
        e_chain = float(((chain.cpu().double() - ref) ** 2).mean().sqrt())
        print("K = %d: rms error vs float64: chunked %.3e, chain %.3e (ratio %.2f)" % (9 * Cin, e_chunk, e_chain, e_chain / e_chunk))
        assert e_chunk < 0.7 * e_chain
        assert not torch.equal(small, chain)
    # the k-major 1x1 kernel at K = 1024 (layer3 conv1): chunks of 8 steps (256 k) on 64-channel tiles -- round-off on the level of
    # the CPU's own float32 convolution (one chain over 1024 products carries ~2x that: scripts/summation_order_model.py)
    x = torch.relu(torch.randn(2, 1024, 30, 40, generator=g)).to(dev)
    w = torch.randn(256, 1024, 1, 1, generator=g) * (2.0 / 1024) ** 0.5
    plan = ops.ConvPlan(w, None, 1, 0, ops.ACT_NONE, dev)
    assert lib.rfx_conv2d_kernel_id(2, 1024, 256, 1, 1, 1, 0, 30, 40) & 16384
    got = plan(x)
    ref = F.conv2d(x.cpu().double(), w.double())
    e_chunk = float(((got.cpu().double() - ref) ** 2).mean().sqrt())
    print("1x1 K = 1024: rms error vs float64: chunked %.3e" % e_chunk)
    x32 = x.cpu()
    e_cpu = float(((F.conv2d(x32, w) - ref) ** 2).mean().sqrt())
    print("              the CPU's own float32 conv: %.3e" % e_cpu)
    assert e_chunk < 2.0 * e_cpu
    # K = 1152 < 2048: the chain form, bit-identical to the implicit GEMM (also covered by test_direct_3x3_equals_implicit_gemm_bit_for_bit)
    x = torch.randn(2, 128, 30, 40, generator=g).to(dev)
    w = torch.randn(128, 128, 3, 3, generator=g) / (128 * 9) ** 0.5
    plan = ops.ConvPlan(w, None, 1, 1, ops.ACT_NONE, dev)
    chain = torch.empty((2, 128, 30, 40), device=dev)
    ops._call("rfx_conv2d_f32", dev, ops._p(x), ops._p(plan.wT), ops._p(plan.ktab), ops._p(None), ops._p(None), ops._p(None), ops._p(chain),
              2, 128, 30, 40, 128, 3, 3, 1, 1, ops.ACT_NONE)
    assert torch.equal(plan(x), chain)


@pytest.mark.parametrize("shape", [(2, 1024, 30, 40), (1, 256, 60, 80), (3, 1024, 25, 33), (2, 36, 7, 5), (1, 50, 9, 11), (70, 1024, 3, 5)])
def test_l2norm_equals_torch_normalize_bit_for_bit(dev, shape):
    """Round 5: rfx_l2norm_nchw_f32 sums a cell's squares in ATen's own order (ONE fma chain over the channels; pinned on the host by
    tests/test_oracle.py::test_l2norm_order_is_atens): ``ops.l2norm(x)`` == ``F.normalize(x.cpu())`` BIT FOR BIT -- both kernel forms
    (four loading wavefronts per 64 pixels / one thread per pixel), also when the result is scattered into the match matrix."""
    g = torch.Generator().manual_seed(shape[1] + shape[2])
    x = torch.relu(torch.randn(*shape, generator=g)) * torch.rand(1, shape[1], 1, 1, generator=g)
    x[0, :, 0, 0] = 0.0                                            # an all-zero cell: 0 / 1e-12 = 0
    want = F.normalize(x)
    got = ops.l2norm(x.to(dev)).cpu()
    assert torch.equal(got, want)
    N, C, H, W = shape
    ld = H * W + 12
    M = torch.zeros((N, C, ld), device=dev)
    ops.l2norm(x.to(dev), out=M[:, :, 5:], out_batch_stride=C * ld, out_chan_stride=ld)
    assert torch.equal(M[:, :, 5:5 + H * W].cpu(), want.reshape(N, C, H * W))


def test_pools_norm_head_resize(dev):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 19, 26, generator=g)
    xd = x.to(dev)
    assert torch.equal(ops.maxpool2d(xd, 3, 2, 1).cpu(), F.max_pool2d(x, 3, 2, 1))
    assert torch.equal(ops.maxpool2d(xd, 2, 1, 0).cpu(), F.max_pool2d(x, 2, 1))
    for s in (1, 2):
        assert (ops.blurpool2d(xd, s).cpu() - restate.blur_pool(x, s)).abs().max() < 1e-6
    for s in (1, 2):   # fused stem == the two separate kernels, bit for bit
        assert torch.equal(ops.maxblurpool2d(xd, s), ops.blurpool2d(ops.maxpool2d(xd, 2, 1, 0), s))
    assert (ops.maxblurpool2d(xd, 2).cpu() - restate.blur_pool(F.max_pool2d(x, 2, 1), 2)).abs().max() < 1e-6
    for shp in ((2, 3, 22, 28), (1, 2, 37, 40), (1, 2, 8, 8), (1, 1, 5, 4), (1, 1, 64, 96)):   # W % 4 == 0: 2x2-block kernel
        y = torch.randn(*shp, generator=g)
        y[0, 0, 3, 1] = float("nan")
        yd = y.to(dev)
        a, b = ops.maxblurpool2d(yd, 2), ops.blurpool2d(ops.maxpool2d(yd, 2, 1, 0), 2)
        assert a.shape == b.shape and torch.equal(torch.nan_to_num(a, nan=7.0), torch.nan_to_num(b, nan=7.0)), shp
    f = torch.randn(2, 1024, 3, 5, generator=g)
    assert (ops.l2norm(f.to(dev)).cpu() - F.normalize(f)).abs().max() < 1e-6
    z = torch.zeros(1, 8, 2, 2)
    assert torch.equal(ops.l2norm(z.to(dev)).cpu(), F.normalize(z))        # eps path: 0 / 1e-12
    # strided scatter into a concatenated (C, nA) matrix
    buf = torch.zeros(2, 1024, 40, device=dev)
    ops.l2norm(f.to(dev), out=buf[:, :, 7:], out_batch_stride=1024 * 40, out_chan_stride=40)
    assert (buf[:, :, 7:22].cpu() - F.normalize(f).view(2, 1024, 15)).abs().max() < 1e-6
    assert buf[:, :, :7].abs().max() == 0 and buf[:, :, 22:].abs().max() == 0
    lg = torch.randn(2, 49, 6, 9, generator=g) * 3
    p = F.softmax(lg, dim=1)
    off = torch.arange(-3, 4, dtype=torch.float32)
    fx = (p * off.view(1, 7).expand(7, 7).reshape(1, 49, 1, 1)).sum(1, keepdim=True) / 9 * 2
    fy = (p * off.view(7, 1).expand(7, 7).reshape(1, 49, 1, 1)).sum(1, keepdim=True) / 6 * 2
    assert (ops.flow_head(lg.to(dev)).cpu() - torch.cat((fx, fy), 1)).abs().max() < 1e-6
    assert ops.flow_head(torch.zeros(1, 49, 4, 4, device=dev)).abs().max() < 1e-7      # KAT: uniform logits -> 0
    m = torch.rand(2, 3, 6, 8, generator=g)
    for ac, size in ((False, (48, 64)), (False, (41, 67)), (True, (48, 64))):
        ref = F.interpolate(m, size=size, mode="bilinear", align_corners=ac)
        assert (ops.resize_bilinear(m.to(dev), size, ac).cpu() - ref).abs().max() < 1e-6


# ------------------------------------------------------------------ 7x7 local correlation


@pytest.mark.parametrize("shape", [(2, 256, 60, 80), (1, 16, 10, 12), (3, 64, 70, 20), (1, 8, 9, 11), (1, 256, 41, 136)])
def test_corr_neigh_matches_cpu(dev, shape):
    g = torch.Generator().manual_seed(shape[2])
    x = F.normalize(torch.randn(*shape, generator=g), dim=1)
    y = F.normalize(torch.randn(*shape, generator=g), dim=1)
    ref = restate.corr_neigh(x, y)
    out = ops.corr_neigh(x.to(dev), y.to(dev)).cpu()
    assert (out - ref).abs().max() < 1e-5, (out - ref).abs().max()
    same = ops.corr_neigh(x.to(dev), x.to(dev)).cpu()
    assert (same[:, 24] - 1).abs().max() < 1e-5                      # KAT: centre tap of normalised features
    assert same[0, 0, 0, 0] == 0                                     # zero padding


@pytest.mark.parametrize("shape", [(5, 32, 60, 80), (3, 16, 33, 44), (2, 8, 70, 100), (2, 8, 16, 48), (1, 8, 7, 32)])
def test_corr_neigh_tile_variants_are_bit_identical(dev, shape):
    """Every kernel configuration of rfx_corr_neigh_variant_f32 (16/32/64-row x 16-column tiles, full-width 16 x 80 plain
    and tuned -- 3 / 2 tap groups, hand-pipelined LDS reads, masked DMA, equal row tiles --, 16x48, 16x32, 32x32)
    accumulates each output in channel order -> bit-identical results, including on maps that are narrower / wider than
    the tile and ragged in both directions; an unknown variant is refused."""
    g = torch.Generator().manual_seed(shape[3])
    x = F.normalize(torch.randn(*shape, generator=g), dim=1).to(dev)
    y = F.normalize(torch.randn(*shape, generator=g), dim=1).to(dev)
    ref = restate.corr_neigh(x.cpu(), y.cpu())
    base = ops.corr_neigh(x, y, variant=3)
    assert (base.cpu() - ref).abs().max() < 1e-5
    for v in (0, 1, 2, 4, 5, 6, 7, 8, 9):
        assert torch.equal(ops.corr_neigh(x, y, variant=v), base), v
    if shape[3] <= 80:        # the DPP form (window quads exchanged between lanes inside v_fmac_f32_dpp; measured slower, kept as evidence)
        for v in (14, 15):
            assert torch.equal(ops.corr_neigh(x, y, variant=v), base), v
    from rfx import _lib
    with pytest.raises(_lib.RfxError):
        ops.corr_neigh(x, y, variant=77)


def test_corr_kernel_durations_from_dispatch_level_events(dev):
    """rfx_corr_timing / ops.Profiler.corr_durations (bench.py's roofline_corr since round 6): inside a Profiler every correlation
    launch carries a start / stop event attached to the DISPATCH (hipExtLaunchKernelGGL) -- the kernel's own duration, as
    rocprofv3 --kernel-trace reports it -- next to the event pair recorded AROUND the launch, whose interval also brackets the command
    processor's work.  Checked: one duration per launch and kind, in order; positive; never longer than the interval around the same
    launch; same results with and without the capture; nothing captured outside a Profiler."""
    from rfx import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(1)
    x = F.normalize(torch.randn(16, 256, 60, 80, generator=g), dim=1).to(dev)
    y = F.normalize(torch.randn(16, 256, 60, 80, generator=g), dim=1).to(dev)
    plain = ops.corr_neigh(x, y)
    assert lib.rfx_corr_timing_collect(None, 0) == 0                 # no capture outside a Profiler
    with ops.Profiler() as prof:
        a = ops.corr_neigh(x, y)
        b12, b21 = ops.corr_neigh_bidir(x, y)
        c = ops.corr_neigh(x[:4], y[:4])
    torch.cuda.synchronize()
    one, bid = prof.corr_durations()
    assert len(one) == 2 and len(bid) == 1 and len(prof.corr) == 2 and len(prof.corr_bidir) == 1
    assert torch.equal(a, plain) and torch.equal(b12, plain) and torch.equal(c, plain[:4])
    iv_one = [e0.elapsed_time(e1) * 1e3 for _, e0, e1 in prof.corr]
    iv_bid = [e0.elapsed_time(e1) * 1e3 for _, _, e0, e1 in prof.corr_bidir]
    for k, iv in zip(one + bid, iv_one + iv_bid):
        assert 5.0 < k <= iv + 1.0, (k, iv)                            # microseconds; the interval brackets the kernel
    assert one[1] < one[0]                                            # 4 pairs vs 16
    assert prof.corr_durations() == (one, bid)                        # collected once, cached
    assert lib.rfx_corr_timing(0) == 0 and lib.rfx_corr_timing_collect(None, 0) == 0     # switched off again on exit


def test_corr_neigh_golden(dev):
    g = gold("nets.npz")
    out = ops.corr_neigh(torch.from_numpy(g["fine_fa"]).to(dev), torch.from_numpy(g["fine_fb"]).to(dev)).cpu()
    assert np.abs(out.numpy() - g["fine_corr"]).max() < 1e-5


# ------------------------------------------------------------------ warping


def test_warp_grid_sample_compose(dev):
    g = gold("nets.npz")
    Hm = torch.from_numpy(g["warp_H"])
    wg = ops.warp_grid(Hm.to(dev), 48, 64)
    assert np.abs(wg.cpu().numpy() - g["warp_grid"]).max() < 1e-6
    xa = torch.from_numpy(g["fine_xa"])
    out = ops.grid_sample(xa.to(dev), wg)
    assert np.abs(out.cpu().numpy() - g["warp_sample"]).max() < 1e-6
    # random grids incl. out-of-range coordinates, both align modes, multi-batch
    gen = torch.Generator().manual_seed(4)
    img = torch.rand(2, 3, 17, 23, generator=gen)
    grid = torch.rand(2, 9, 11, 2, generator=gen) * 2.6 - 1.3
    for ac in (False, True):
        ref = F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=ac)
        assert (ops.grid_sample(img.to(dev), grid.to(dev), ac).cpu() - ref).abs().max() < 1e-6
    # fused compose vs the unfused recipe (align2images.py:92-95 and evalHpatch/evaluation.py:40-45,51)
    flowDown = (torch.rand(2, 2, 6, 8, generator=gen) - 0.5) * 0.3
    Hb = torch.stack((Hm[0], torch.tensor([[0.9, 0.1, 0.2], [0.05, 1.1, -0.3], [0.02, 0.01, 1.0]])))
    coarse = restate.warp_grid(Hb, 48, 64)
    for clamp in (False, True):
        up = F.interpolate(flowDown, size=(48, 64), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
        up = up + restate.identity_grid(48, 64)
        if clamp:
            up = torch.clamp(up, -1, 1)
        ref = restate.grid_sample(coarse.permute(0, 3, 1, 2), up).permute(0, 2, 3, 1)
        f12, inb, fup = ops.compose_flow(flowDown.to(dev), coarse.to(dev), clamp=clamp, want_inb=True, want_flow_up=True)
        assert (fup.cpu() - up).abs().max() < 1e-6
        assert (f12.cpu() - ref).abs().max() < 2e-6
        refin = ((ref[..., 0] >= -1) & (ref[..., 0] <= 1) & (ref[..., 1] >= -1) & (ref[..., 1] <= 1)).float()
        # in-bounds flag may only differ where the reference value sits within round-off of +-1
        diff = inb.cpu() != refin
        if diff.any():
            edge = (ref.abs() - 1).abs().min(dim=-1).values
            assert (edge[diff] < 1e-5).all()


# ------------------------------------------------------------------ mutual nearest neighbours


def _mnn_check(dev, A, B, mask=None):
    Bm = B * mask if mask is not None else B
    r1, r2 = restate.mutual_matching(A, Bm)
    o1, o2 = ops.mutual_nn(A.to(dev), B.to(dev), None if mask is None else mask.to(dev))
    o1, o2 = o1.cpu(), o2.cpu()
    if torch.equal(o1, r1) and torch.equal(o2, r2):
        return 0
    # every disagreement must be a float32 near-tie of the arg-max (margin in float64)
    S = A.double().t() @ Bm.double()
    ref = set(zip(r1.tolist(), r2.tolist()))
    got = set(zip(o1.tolist(), o2.tolist()))
    bad = ref ^ got
    for i, j in bad:
        row, col = S[i], S[:, j]
        top2r = torch.topk(row, 2).values
        top2c = torch.topk(col, 2).values
        assert min((top2r[0] - top2r[1]).item(), (top2c[0] - top2c[1]).item()) < 1e-5, (i, j)
    assert len(bad) <= max(2, len(ref) // 100)
    assert (o1[1:] > o1[:-1]).all()
    return len(bad)


def test_mutual_nn_golden_and_random(dev):
    g = gold("mutual.npz")
    o1, o2 = ops.mutual_nn(torch.from_numpy(g["A"]).to(dev), torch.from_numpy(g["B"]).to(dev))
    assert np.array_equal(o1.cpu().numpy(), g["index1"]) and np.array_equal(o2.cpu().numpy(), g["index2"])
    gen = torch.Generator().manual_seed(5)
    for (C, nA, nB) in ((1024, 2107, 300), (1024, 531, 1200), (40, 130, 129)):
        A = F.normalize(torch.relu(torch.randn(C, nA, generator=gen)), dim=0)
        # plant structure: B columns are noisy copies of some A columns -> many mutual matches
        pick = torch.randint(nA, (nB,), generator=gen)
        B = F.normalize(torch.relu(A[:, pick] + 0.3 * torch.randn(C, nB, generator=gen)), dim=0)
        _mnn_check(dev, A, B)
        mask = (torch.rand(nB, generator=gen) > 0.3).float()
        _mnn_check(dev, A, B, mask)


def test_mutual_nn_kmajor_form_equals_the_transposed_image_form(dev, monkeypatch):
    """Round 4: mnn_tile_kmajor_kernel (k-major LDS images, straight 16-byte staging, mask applied to the finished
    accumulators) accumulates every score in the same k order and pairing as mnn_tile_kernel -> identical match lists on
    ragged sizes, with and without a 0/1 column mask, with 16-byte rows (vector staging) and without (scalar staging), and
    on a padded leading dimension whose padding columns hold NaNs (they may only pollute rows / columns the arg-max skips)."""
    gen = torch.Generator().manual_seed(11)
    for (C, nA, nB, ld_pad) in ((1024, 2107, 300, 0), (256, 533, 1201, 0), (1024, 13065, 1200, 3), (64, 130, 129, 0), (96, 1000, 260, 4)):
        A = F.normalize(torch.relu(torch.randn(C, nA, generator=gen)), dim=0)
        pick = torch.randint(nA, (nB,), generator=gen)
        B = F.normalize(torch.relu(A[:, pick] + 0.3 * torch.randn(C, nB, generator=gen)), dim=0)
        mask = (torch.rand(nB, generator=gen) > 0.3).float().to(dev)
        Ad, Bd = A.to(dev), B.to(dev)
        kw = {}
        if ld_pad:                                                 # (C, ld) storage with ld = nA + pad, padding = NaN
            ldA = nA + ld_pad
            Ap = torch.full((C, ldA), float("nan"), device=dev)
            Ap[:, :nA] = Ad
            Ad, kw = Ap, dict(ldA=ldA, nA=nA)
        res = {}
        for form in ("0", "1"):
            monkeypatch.setenv("RFX_MNN_FORM", form)
            res[form] = [ops.mutual_nn(Ad, Bd, m, **kw) for m in (None, mask)]
        for (a1, a2), (b1, b2) in zip(res["0"], res["1"]):
            assert torch.equal(a1, b1) and torch.equal(a2, b2), (C, nA, nB, ld_pad)
        assert len(res["0"][0][0]) > 10


def test_mutual_nn_scores_are_accumulated_in_chunks_of_256_products(dev, monkeypatch):
    """Round 4: a score (utils/outil.py:34, featA.t() @ featB) is a sum of C = 1024 NON-NEGATIVE products (post-ReLU, L2-normalised
    features).  One fma chain over them carries 2.8x the round-off of the CPU reference's K-blocked sgemm -- and that error, larger
    than what the trunk features contribute, is what decides a float64 near-tie of the arg-max.  Both tile kernels close a chunk
    every 8 K steps (256 k) and add it to a running total (score_chunk < 0: the chain).  Read back: the per-row maxima the tile
    kernel leaves in the workspace (csrc/mutual_nn.hip::layout), compared with the float64 maxima."""
    gen = torch.Generator().manual_seed(5)
    C, nA, nB = 1024, 3000, 1100
    A = F.normalize(torch.relu(torch.randn(C, nA, generator=gen)), dim=0)
    B = F.normalize(torch.relu(A[:, torch.randint(nA, (nB,), generator=gen)] + 0.5 * torch.randn(C, nB, generator=gen)), dim=0)
    S64 = A.double().t() @ B.double()
    ref = S64.max(dim=1).values
    cpu = (A.t() @ B).max(dim=1).values.double()                  # the reference's own float32 product
    from rfx import _lib
    lib = _lib.load()
    al = lambda x: (x + 255) & ~255
    tA, tB = (nA + 127) // 128, (nB + 127) // 128
    o_rowval = 2 * al(tB * nA * 4) + 2 * al(tA * nB * 4)
    Ad, Bd = A.to(dev), B.to(dev)
    err, lists = {}, {}
    for form in ("0", "1"):
        monkeypatch.setenv("RFX_MNN_FORM", form)
        for chunk in ("8", "0"):
            ws = torch.zeros(lib.rfx_mutual_nn_ws_bytes(nA, nB), dtype=torch.uint8, device=dev)
            i1 = torch.empty(nB, dtype=torch.int64, device=dev); i2 = torch.empty_like(i1)
            cnt = torch.zeros(1, dtype=torch.int32, device=dev)
            ops._call("rfx_mutual_nn_f32", dev, ops._p(Ad), nA, nA, ops._p(Bd), nB, nB, C, ops._p(None), ops._p(i1), ops._p(i2), ops._p(cnt),
                      ops._p(ws), 256 if chunk == "8" else -1)
            rowval = ws[o_rowval:o_rowval + 4 * nA].view(torch.float32).cpu().double()
            err[form, chunk] = float(((rowval - ref) ** 2).mean().sqrt())
            n = int(cnt.item())
            lists[form, chunk] = (i1[:n].cpu(), i2[:n].cpu())
    e_cpu = float(((cpu - ref) ** 2).mean().sqrt())
    print("row-maximum round-off vs float64: chunked %.3e, chain %.3e, the CPU's torch.mm %.3e" % (err["0", "8"], err["0", "0"], e_cpu))
    assert err["0", "8"] == err["1", "8"] and err["0", "0"] == err["1", "0"]            # both kernel forms: the same sums
    assert torch.equal(lists["0", "8"][0], lists["1", "8"][0]) and torch.equal(lists["0", "8"][1], lists["1", "8"][1])
    assert err["0", "8"] < 0.6 * err["0", "0"]
    assert err["0", "8"] < 1.5 * e_cpu


def test_scores_equal_the_hosts_torch_mm_bit_for_bit_with_the_hosts_chunk(dev):
    """The reference's score is torch.mm on the HOST (utils/outil.py:34): MKL's sgemm sums k as an fma chain inside blocks of KC
    products and adds the block sums (KC = 192 on the GPU box's EPYC, 384 on a Xeon).  A pipeline built with score_chunk="host"
    resolves KC once at construction (ops.resolve_score_chunk -> ops.host_sgemm_k_block) and passes it to every mutual-NN launch
    (ABI 8: an argument, no process-wide state): the per-row score maxima the tile kernel leaves in its workspace then EQUAL
    torch.mm's on the same features.  Asserted: the argument is honoured per call (two chunk lengths interleaved give their own
    sums), a bad value is refused, the default (0) is 256 products, and the host's chunk is at least as close to torch.mm as the
    default; the share of bit-equal rows is printed (measured 1.0 on the round-4 box, profiles/r04_mm_blocking_probe.json) -- it is
    a property of the host's BLAS build, so it is only asserted when the probe found the host's blocking."""
    from rfx import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(9)
    C, nA, nB = 1024, 3000, 1100
    A = F.normalize(torch.relu(torch.randn(C, nA, generator=gen)), dim=0)
    B = F.normalize(torch.relu(A[:, torch.randint(nA, (nB,), generator=gen)] + 0.5 * torch.randn(C, nB, generator=gen)), dim=0)
    cpu = (A.t() @ B).max(dim=1).values
    al = lambda x: (x + 255) & ~255
    o_rowval = 2 * al(((nB + 127) // 128) * nA * 4) + 2 * al(((nA + 127) // 128) * nB * 4)
    Ad, Bd = A.to(dev), B.to(dev)

    def rowmax(chunk):
        ws = torch.zeros(lib.rfx_mutual_nn_ws_bytes(nA, nB), dtype=torch.uint8, device=dev)
        i1 = torch.empty(nB, dtype=torch.int64, device=dev); i2 = torch.empty_like(i1)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        ops._call("rfx_mutual_nn_f32", dev, ops._p(Ad), nA, nA, ops._p(Bd), nB, nB, C, ops._p(None), ops._p(i1), ops._p(i2), ops._p(cnt), ops._p(ws),
                  chunk)
        return ws[o_rowval:o_rowval + 4 * nA].view(torch.float32).cpu()
    r8 = rowmax(256)
    chain = rowmax(-1)
    assert torch.equal(rowmax(0), r8)                                    # 0 = the library default = 256 products
    assert torch.equal(rowmax(1024), chain)                              # one chunk over all of C = the chain
    assert not torch.equal(chain, r8) and torch.equal(rowmax(256), r8)   # per call: nothing sticks between launches
    with pytest.raises(_lib.RfxError):
        rowmax(100)                                                      # not a multiple of 32 products
    kc, src = ops.resolve_score_chunk("host")
    share8 = float((r8 == cpu).float().mean())
    print("score chunk for this host: %d (%s); rows bit-equal to torch.mm with chunks of 256: %.3f" % (kc, src, share8))
    if "host probe" in src:
        rk = rowmax(kc)
        share = float((rk == cpu).float().mean())
        print("... with chunks of %d (the host's): %.3f, max |d| %.2e" % (kc, share, float((rk - cpu).abs().max())))
        assert share >= 0.99 and share >= share8


# ------------------------------------------------------------------ RANSAC


def test_dlt_sign_and_value(dev):
    g = gold("ransac.npz")
    m1, m2 = torch.from_numpy(g["m1_1"]), torch.from_numpy(g["m2_1"])
    torch.manual_seed(17)
    s = restate.filter_samples(torch.randint(len(m1), (4000, 4)))
    X, Y = m1[s].contiguous(), m2[s].contiguous()
    ref = restate.homography_svd(X, Y)
    out = ops.dlt4_homography(X.to(dev), Y.to(dev)).cpu()
    A = restate.dlt_matrix(X.numpy(), Y.numpy())
    sv = np.linalg.svd(A, compute_uv=False)
    good = torch.from_numpy(sv[:, 7] > 1e-9)
    assert ((out * ref).sum((1, 2))[good] > 0).all()                 # LAPACK's sign on every rank-8 system
    assert (out - ref)[good].abs().max() <= 1.2e-7
    assert ((out.view(-1, 9).norm(dim=1) - 1).abs()[good] < 1e-6).all()


def test_det_gate_decides_like_torch_det_at_the_threshold(dev, tmp_path_factory):
    """utils/outil.py:113 zeroes the count of every hypothesis with det(H) <= 1e-6 (torch.det = float32 LU).  Hypotheses built
    to land within +-10 % of that gate (tests/test_oracle.py::det_gate_cases).  The kernel's determinant follows the LU order
    of torch.det as pinned in the authoring container (dlt.h: rfx_det3_lu_f32; tests/test_oracle.py::test_det3_lu_is_torch_det:
    bit-equal to torch 2.10 + MKL on an Intel host).  Here: (1) the device gate IS that function on the device's own float32 H
    (host build of the same header: exact, machine independent); (2) against torch.det of THIS host -- MKL picks another
    code path on the GPU box's AMD CPUs, so the library value itself moves by ~1e-8 from machine to machine -- any differing
    decision lies within float32 round-off (3e-8) of the gate, and they are rare."""
    import ctypes
    from test_oracle import det_gate_cases, _host_dlt_lib
    X, Y = det_gate_cases(6000, seed=5)
    N = len(X)
    m1, m2 = torch.from_numpy(X.reshape(-1, 3)).to(dev), torch.from_numpy(Y.reshape(-1, 3)).to(dev)
    samples = torch.arange(4 * N, dtype=torch.int64).view(N, 4).to(dev)
    H21, cnt = ops.score_hypotheses(m1, m2, samples, 0.05)
    Hc = H21.cpu().contiguous()
    d32, d64 = torch.det(Hc), torch.det(Hc.double())
    band = (d64.abs() - 1e-6).abs() < 1e-7
    pos = band & (d64 > 0)
    assert int(pos.sum()) > 1000, int(pos.sum())                       # the family straddles the gate (LAPACK's sign splits it)
    gate_dev = cnt.cpu() > 0      # the 4 sample points are inliers of their own hypothesis: count > 0 <=> the gate passed
    lib = _host_dlt_lib(tmp_path_factory)
    dlu = np.zeros(N, dtype=np.float32)
    hn = Hc.numpy().reshape(-1, 9)
    lib.rfx_host_det3(hn.ctypes.data_as(ctypes.c_void_p), N, dlu.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(gate_dev.numpy(), dlu > 1e-6)                 # (1) exact
    dis = gate_dev != (d32 > 1e-6)
    print("det gate: %d hypotheses within +-10 %% of 1e-6 (%d positive), %d decisions differ from this host's torch.det, "
          "bit-equal determinants %.4f" % (int(band.sum()), int(pos.sum()), int(dis.sum()), float(np.mean(dlu == d32.numpy()))))
    assert int(dis.sum()) <= N // 100                                   # (2) rare ...
    assert int(dis.sum()) == 0 or float((d64[dis] - 1e-6).abs().max()) < 3e-8   # ... and only within round-off of the gate
    assert 0.2 < float((d32 > 1e-6)[pos].float().mean()) < 0.8         # both outcomes occur


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_score_and_ransac_match_reference_golden(dev, seed):
    g = gold("ransac.npz")
    m1, m2 = torch.from_numpy(g["m1_%d" % seed]), torch.from_numpy(g["m2_%d" % seed])
    samples = torch.from_numpy(g["samples_%d" % seed])
    uniq = restate.filter_samples(samples)[:300]
    H21, cnt = ops.score_hypotheses(m1.to(dev), m2.to(dev), uniq.to(dev), 0.05)
    assert np.abs(H21.cpu().numpy() - g["score_H_%d" % seed]).max() <= 1.2e-7
    assert np.array_equal(cnt.cpu().numpy(), g["score_counts_%d" % seed])          # per-hypothesis counts: exact
    bestH, inl, res = ops.ransac_h4(m1.to(dev), m2.to(dev), samples.to(dev), 0.05)
    res = res.cpu().tolist()
    assert res[0] == 0 and res[1] == int(g["count_%d" % seed])
    assert res[3] == len(restate.filter_samples(samples))
    assert np.array_equal(inl.cpu().numpy(), g["inlier_%d" % seed])                # inlier indices: bit exact
    assert np.abs(bestH.cpu().numpy() - g["H_%d" % seed]).max() <= 1.2e-7
    assert samples[res[2]].tolist() in restate.filter_samples(samples).tolist()


def test_ransac_sentinels_and_many_seeds(dev):
    g = gold("ransac.npz")
    m1, m2 = torch.from_numpy(g["abort_m1"]), torch.from_numpy(g["abort_m2"])
    s = torch.from_numpy(g["abort_samples"])
    _, inl, res = ops.ransac_h4(m1.to(dev), m2.to(dev), s.to(dev), -1.0)
    assert res.cpu().tolist()[0] == 1 and not inl.any()                            # zero-chunk abort
    _, _, res = ops.ransac_h4(m1.to(dev), m2.to(dev), s[:50].to(dev), -1.0)
    assert res.cpu().tolist()[0] == 2                                              # the reference's TypeError case
    # fresh seeds against the oracle run on this host
    m1, m2 = torch.from_numpy(g["m1_2"]), torch.from_numpy(g["m2_2"])
    mism = 0
    for seed in range(20):
        torch.manual_seed(1000 + seed)
        s = torch.randint(len(m1), (700, 4))
        Hb, cnt, inl_ref, _ = restate.ransac(m1, m2, 0.05, s)
        bestH, inl, res = ops.ransac_h4(m1.to(dev), m2.to(dev), s.to(dev), 0.05)
        ok = np.array_equal(inl.cpu().numpy(), inl_ref) and res.cpu().tolist()[1] == int(cnt)
        mism += 0 if ok else 1
        if ok:
            assert np.abs(bestH.cpu().numpy() - Hb).max() <= 1.2e-7
    assert mism == 0, "%d / 20 seeds disagree with the oracle" % mism
    err = ops.prediction(m1.to(dev), m2.to(dev), torch.from_numpy(g["score_H_2"][:7]).to(dev)).cpu()
    ref = restate.prediction(m1[None], m2[None], torch.from_numpy(g["score_H_2"][:7]))
    # errors grow without bound where the projective denominator approaches 0: relative tolerance
    rel = (err - ref).abs() / ref.clamp_min(1.0)
    assert rel.max() <= 1e-3 and (rel <= 4e-7).float().mean() > 0.95  # host BLAS may round the K=3 chain differently


def _lattice_few_matches(seed, n=13, n_line=6):
    """A late multi-homography round in miniature: a handful of matches left on the 30x40 cell lattice -- ``n_line`` of them on
    ONE lattice row that maps to a shifted row (collinear in both images, same spacing), the rest unrelated.  A sample with three
    points of that line has a rank-7 DLT system, and such samples WIN here (they are the only ones that can explain the line):
    23 of 40 seeds."""
    W, Hh = restate.get_wh(30, 40)
    cells = torch.stack((Hh, W, torch.ones_like(W)), 1).view(30, 40, 3)
    g = torch.Generator().manual_seed(seed)
    row = int(torch.randint(3, 27, (1,), generator=g))
    cols = torch.randperm(36, generator=g)[:n_line]
    flat = cells.view(-1, 3)
    n_out = n - n_line
    m1 = torch.cat((cells[row + 1, cols + 2], flat[torch.randperm(1200, generator=g)[:n_out]]))
    m2 = torch.cat((cells[row, cols], flat[torch.randperm(1200, generator=g)[:n_out]]))
    p = torch.randperm(n, generator=g)
    return m1[p].clone(), m2[p].clone(), torch.randint(n, (600, 4), generator=g)


def test_rank_deficient_winners_equal_the_host_lapack_in_exact_mode(dev):
    """VERDICT r3 #4.  utils/outil.py:84 takes Vh[8] of an 8x9 system; when a sample's system has rank 7 (three matched points
    collinear in both images) that vector is whatever the HOST's LAPACK rounds to.  ops.ransac_h4(degenerate="lapack") flags those
    hypotheses on the device, re-solves exactly them with numpy's LAPACK and counts / selects on the patched set: bestH, the
    count and the inlier indices equal the oracle's on THIS host bit for bit on every seed -- including seeds whose WINNER is
    such a sample, which the device's own null vector ("device" mode) cannot reproduce.  Same for ScoreRANSAC / Homography and
    for the batched form."""
    n_degenerate_winner, n_device_differs = 0, 0
    for seed in range(40):
        m1, m2, s = _lattice_few_matches(seed)
        try:
            Hb, cnt, inl_ref, _ = restate.ransac(m1, m2, 0.05, s)
        except TypeError:
            continue
        if Hb is None:
            continue
        info = {}
        bestH, inl, res = ops.ransac_h4(m1.to(dev), m2.to(dev), s.to(dev), 0.05, degenerate="lapack", info=info)
        r = res.cpu().tolist()
        assert r[0] == 0 and r[1] == int(cnt), (seed, r, int(cnt))
        assert np.array_equal(inl.cpu().numpy(), inl_ref), seed
        assert np.array_equal(bestH.cpu().numpy(), Hb), (seed, np.abs(bestH.cpu().numpy() - Hb).max())
        win = s[r[2]]
        sv = np.linalg.svd(restate.dlt_matrix(m1[win][None].numpy(), m2[win][None].numpy()), compute_uv=False)[0]
        if sv[7] / sv[0] < 1e-10:
            n_degenerate_winner += 1
            assert info["n_degenerate"][0] > 0
            dH, dinl, _ = ops.ransac_h4(m1.to(dev), m2.to(dev), s.to(dev), 0.05)          # the device's own null vector
            if not np.array_equal(dH.cpu().numpy(), Hb) or not np.array_equal(dinl.cpu().numpy(), inl_ref):
                n_device_differs += 1
    assert n_degenerate_winner >= 10, n_degenerate_winner        # the case is exercised ...
    assert n_device_differs >= 1                                 # ... and it is one the device mode cannot pin
    # ScoreRANSAC / Homography on the same kind of samples
    m1, m2, s = _lattice_few_matches(5, n=40, n_line=12)
    uniq = restate.filter_samples(s)
    Ho, co = restate.score_ransac(m1, m2, 0.05, uniq)
    Hd, cd = ops.score_hypotheses(m1.to(dev), m2.to(dev), uniq.to(dev), 0.05, degenerate="lapack")
    flagged = np.abs(ops.score_hypotheses(m1.to(dev), m2.to(dev), uniq.to(dev), 0.05)[0].cpu().numpy() - Ho.numpy()).reshape(len(uniq), -1).max(1) > 1e-3
    assert flagged.sum() >= 1
    assert np.abs(Hd.cpu().numpy() - Ho.numpy()).max() <= 2.4e-7 and np.array_equal(Hd.cpu().numpy()[flagged], Ho.numpy()[flagged])
    assert np.array_equal(cd.cpu().numpy(), co.numpy())
    info = {}
    Hh_ = ops.dlt4_homography(m1[uniq].to(dev), m2[uniq].to(dev), degenerate="lapack", info=info)
    assert info["n_degenerate"] >= flagged.sum() and np.array_equal(Hh_.cpu().numpy()[flagged], Ho.numpy()[flagged])
    # batched: three pairs of different sizes in one chain, per pair identical to the single-pair call
    cases = [_lattice_few_matches(sd, n=nn) for sd, nn in ((3, 13), (8, 30), (11, 9))]
    cap = max(len(c[0]) for c in cases)
    M1 = torch.zeros((3, cap, 3)); M2 = torch.zeros((3, cap, 3))
    for b, (a, c, _) in enumerate(cases):
        M1[b, :len(a)], M2[b, :len(a)] = a, c
    nd = torch.tensor([len(c[0]) for c in cases], dtype=torch.int32, device=dev)
    S = torch.stack([c[2] % len(c[0]) for c in cases]).to(dev)
    bH, bI, bR = ops.ransac_h4_batched(M1.to(dev), M2.to(dev), nd, S, 0.05, degenerate="lapack")
    for b, (a, c, sm) in enumerate(cases):
        h1, i1, r1 = ops.ransac_h4(a.to(dev), c.to(dev), (sm % len(a)).to(dev), 0.05, degenerate="lapack")
        assert torch.equal(bH[b], h1) and torch.equal(bI[b, :len(a)], i1) and torch.equal(bR[b], r1)


# ------------------------------------------------------------------ whole networks


def test_networks_match_reference_golden(dev):
    g = gold("nets.npz")
    trunk = nets.ResNet50Trunk(weights.resnet50_trunk_sd(seed=31, randomize_bn=True), dev)
    out = trunk(torch.from_numpy(g["trunk_in"]).to(dev)).cpu().numpy()
    assert np.abs(out - g["trunk_out"]).max() <= 5e-5 * np.abs(g["trunk_out"]).max()
    fe = nets.FeatureExtractorNet(weights.feature_extractor_sd(seed=32, randomize_bn=True), dev)
    fa = ops.l2norm(fe(torch.from_numpy(g["fine_xa"]).to(dev))).cpu().numpy()
    assert np.abs(fa - g["fine_fa"]).max() < 1e-5
    nf = nets.NetFlowCoarseNet(weights.net_flow_coarse_sd(seed=33, randomize_bn=True), 7, dev)
    nm = nets.NetMatchabilityNet(weights.net_matchability_sd(seed=34, randomize_bn=True, last_std=0.02), 7, dev)
    c = torch.from_numpy(g["fine_corr"]).to(dev)
    assert np.abs(nf(c, False).cpu().numpy() - g["fine_flow"]).max() < 1e-5
    assert np.abs(nf(c, True).cpu().numpy() - g["fine_flow8"]).max() < 1e-5
    assert np.abs(nm(c, False).cpu().numpy() - g["fine_match"]).max() < 1e-5


# ------------------------------------------------------------------ device pre-processing


def test_lanczos_and_to_tensor_bit_exact_vs_pillow(dev):
    """The device resampler reproduces PIL.Image.resize(LANCZOS) byte for byte and ToTensor/Normalize bit for bit,
    so prepare_device() == prepare()."""
    import PIL.Image as Image
    from rfx import synth
    from rfx.pipeline import AlignPipeline
    rng = np.random.RandomState(3)
    for (w, h, ow, oh) in [(640, 480, 768, 576), (640, 480, 528, 400), (333, 217, 640, 416), (640, 480, 320, 480),
                           (500, 300, 500, 150), (64, 48, 64, 48)]:
        imgs = (rng.rand(2, h, w, 3) * 255).astype(np.uint8)
        ref = np.stack([np.asarray(Image.fromarray(im).resize((ow, oh), resample=Image.LANCZOS)) for im in imgs])
        got = ops.lanczos_resize_u8(torch.from_numpy(imgs).to(dev), ow, oh).cpu().numpy()
        assert np.array_equal(got, ref), (w, h, ow, oh)
    pairs = [synth.make_pair(240, 320, seed=s) for s in (0, 1)]
    for variant, minSize in (("A", 320), ("B", 240)):
        pipe = AlignPipeline(dict(trunk=weights.resnet50_trunk_sd(0)), nbScale=7, nbIter=10, tolerance=0.05, minSize=minSize,
                             scaleR=1.2, variant=variant, device=dev)
        host = pipe.prepare(pairs)
        devp = pipe.prepare_device(*pipe.upload_raw(pairs))
        assert len(host["src"]) == len(devp["src"])
        for a, b in zip(host["src"], devp["src"]):
            assert a.shape == b.shape and torch.equal(a, b)
        for k in ("tgt", "IsTensor", "ItTensor"):
            assert torch.equal(host[k], devp[k]), k


def test_batched_ransac_equals_per_pair(dev):
    """rfx_gather_matches_f32 + rfx_ransac_h4_batched against the single-pair entry point, pair by pair: identical H bits,
    inlier masks and result records; a pair with < 4 matches reports status 3, a hopeless pair the abort status."""
    g = torch.Generator().manual_seed(9)
    B, cap, N = 6, 300, 700
    rows, cols = 15, 20
    xs = ((torch.arange(cols) + 0.5) / cols - 0.5) * 2
    ys = ((torch.arange(rows) + 0.5) / rows - 0.5) * 2
    xb = xs.repeat(rows); yb = ys.repeat_interleave(cols)                    # target lattice (x, y)
    Hm = torch.tensor([[1.05, .02, .03], [-.01, .97, -.02], [.01, .02, 1.]])
    pts = torch.stack((xb, yb, torch.ones_like(xb)), 1) @ Hm.T
    xa, ya = pts[:, 0] / pts[:, 2], pts[:, 1] / pts[:, 2]                    # source coordinates of the same cells
    nb = [300, 257, 3, 120, 4, 64]
    idx1 = torch.zeros(B, cap, dtype=torch.int64); idx2 = torch.zeros(B, cap, dtype=torch.int64)
    for b, n in enumerate(nb):
        i2 = torch.randperm(rows * cols, generator=g)[:n]
        i1 = i2.clone()
        bad = torch.rand(n, generator=g) < (1.0 if b == 3 else 0.5)          # pair 3: every match wrong
        i1[bad] = torch.randint(rows * cols, (int(bad.sum()),), generator=g)
        idx1[b, :n], idx2[b, :n] = i1, i2
    smp = torch.stack([torch.randint(max(n, 1), (N, 4), generator=g) for n in nb])
    tol = 0.02
    n_dev = torch.tensor(nb, dtype=torch.int32, device=dev)
    M1, M2 = ops.gather_matches(idx1.to(dev), idx2.to(dev), n_dev, xa.to(dev), ya.to(dev), xb.to(dev), yb.to(dev))
    Hb, Ib, Rb = ops.ransac_h4_batched(M1, M2, n_dev, smp.to(dev), tol)
    Rb = Rb.cpu()
    for b, n in enumerate(nb):
        m1 = torch.stack((xa[idx1[b, :n]], ya[idx1[b, :n]], torch.ones(n)), 1)
        m2 = torch.stack((xb[idx2[b, :n]], yb[idx2[b, :n]], torch.ones(n)), 1)
        assert torch.equal(M1[b, :n].cpu(), m1) and torch.equal(M2[b, :n].cpu(), m2)
        assert float(M1[b, n:].abs().max() if n < cap else 0) == 0.0
        if n < 4:
            assert Rb[b, 0].item() == 3 and not bool(Ib[b].any())
            continue
        h, inl, r = ops.ransac_h4(m1.to(dev), m2.to(dev), smp[b].to(dev), tol)
        assert torch.equal(Rb[b], r.cpu()), (b, Rb[b], r)
        assert torch.equal(Hb[b].cpu().view(torch.int32), h.cpu().view(torch.int32))
        assert torch.equal(Ib[b, :n].cpu(), inl.cpu()) and not bool(Ib[b, n:].any())
    assert Rb[0, 0].item() == 0 and Rb[0, 1].item() > 50


@pytest.mark.parametrize("shape", [(2, 40, 56), (1, 37, 45), (3, 16, 16), (1, 7, 9), (1, 3, 3), (2, 64, 130)])
def test_fused_stem_equals_conv_then_maxblurpool(dev, shape):
    """rfx_stem_conv3x3_maxblur_f32 == rfx_conv2d_f32(ReLU) -> rfx_maxblurpool2d_f32, bit for bit (same MFMA k order,
    same pooling order), including ragged tiles, reflected borders and a NaN pixel."""
    N, H, W = shape
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(N, 3, H, W, generator=g)
    if H > 5:
        x[0, 1, 4, 2] = float("nan")
    w = torch.randn(64, 3, 3, 3, generator=g) / 27 ** 0.5
    bn = dict(weight=1 + 0.3 * torch.randn(64, generator=g), bias=0.2 * torch.randn(64, generator=g),
              running_mean=0.2 * torch.randn(64, generator=g), running_var=0.5 + torch.rand(64, generator=g))
    plan = ops.ConvPlan(w, bn, 1, 1, ops.ACT_RELU, dev)
    xd = x.to(dev)
    ref = ops.maxblurpool2d(plan(xd), 2)
    out = ops.stem_conv_maxblur(xd, plan)
    assert out.shape == ref.shape
    assert torch.equal(torch.nan_to_num(out, nan=7.0), torch.nan_to_num(ref, nan=7.0))


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 37, 45), (3, 16, 16), (1, 7, 9), (1, 1, 1), (1, 130, 70)])
def test_fused_resnet_stem_equals_conv_then_maxpool(dev, shape):
    """rfx_stem_conv7x7_maxpool_f32 == rfx_conv2d_f32(7x7/2, ReLU) -> rfx_maxpool2d_f32(3, 2, 1), bit for bit."""
    N, H, W = shape
    g = torch.Generator().manual_seed(H * 100 + W + 1)
    x = torch.randn(N, 3, H, W, generator=g)
    if H > 5:
        x[0, 2, 3, 5] = float("nan")
    w = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
    bn = dict(weight=1 + 0.3 * torch.randn(64, generator=g), bias=0.2 * torch.randn(64, generator=g),
              running_mean=0.2 * torch.randn(64, generator=g), running_var=0.5 + torch.rand(64, generator=g))
    plan = ops.ConvPlan(w, bn, 2, 3, ops.ACT_RELU, dev)
    xd = x.to(dev)
    ref = ops.maxpool2d(plan(xd), 3, 2, 1)
    out = ops.stem_conv7_maxpool(xd, plan)
    assert out.shape == ref.shape
    assert torch.equal(torch.nan_to_num(out, nan=7.0), torch.nan_to_num(ref, nan=7.0))


@pytest.mark.parametrize("cout", [32, 96, 128])
def test_fused_resnet_stem_channel_group_passes(dev, cout):
    """The fused stem with 1, 3 and 4 channel groups of 32: bit for bit equal to the separate kernels."""
    g = torch.Generator().manual_seed(cout)
    x = torch.randn(2, 3, 70, 58, generator=g)
    w = torch.randn(cout, 3, 7, 7, generator=g) / 147 ** 0.5
    bn = dict(weight=1 + 0.3 * torch.randn(cout, generator=g), bias=0.2 * torch.randn(cout, generator=g),
              running_mean=0.2 * torch.randn(cout, generator=g), running_var=0.5 + torch.rand(cout, generator=g))
    plan = ops.ConvPlan(w, bn, 2, 3, ops.ACT_RELU, dev)
    xd = x.to(dev)
    assert torch.equal(ops.stem_conv7_maxpool(xd, plan), ops.maxpool2d(plan(xd), 3, 2, 1))


@pytest.mark.parametrize("case", [
    # N, Cin, H, W, Cmid, Cexp, residual
    (2, 64, 30, 40, 64, 256, True),       # layer1-like, 16x8 patch
    (1, 64, 25, 33, 64, 256, True),       # 32x4 patch, ragged
    (3, 128, 19, 21, 128, 512, True),     # layer2-like, 8x16 patch, ragged
    (1, 64, 16, 48, 64, 128, False),      # no residual, single pass
    (1, 8, 5, 3, 128, 384, True),         # one K step of the 3x3, three passes of the 1x1
    (24, 64, 30, 40, 64, 256, True),      # big enough for the unfused 1x1 to run on the k-major kernel (conv1x1.hip)
    (20, 128, 17, 23, 128, 512, True),    # ... with an odd plane (scalar pixel loads)
    (32, 64, 120, 160, 64, 256, True),    # 64-channel tail on a launch large enough for the 256-pixel (16x16) patch
    (200, 64, 30, 44, 64, 128, False),    # ... ragged in both directions, images straddling the patches, single pass
])
def test_fused_bottleneck_tail_equals_two_convs(dev, case):
    """rfx_conv3x3_conv1x1_f32 == rfx_conv3x3_f32(3x3, k_chunk = 4) -> rfx_conv2d_f32(1x1 + residual), bit for bit: since round 5 the
    tail's 3x3 phase (K = 576 / 1152) closes a chunk every 4 K steps (288 products) into a second accumulator set -- the K-blocked
    sum of the reference's oneDNN kernels instead of ONE fma chain -- and the stand-alone 3x3 kernel does the same when asked
    (ConvPlan.k_chunk = 4, which rfx/nets.py sets on every tail).  Against float64 the chunked mid tile carries less round-off
    than the chain form (rfx_conv2d_f32's implicit-GEMM kernel on the same layer)."""
    N, Cin, H, W, Cmid, Cexp, with_res = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    x = torch.randn(N, Cin, H, W, generator=g)
    def bn(c):
        return dict(weight=1 + 0.3 * torch.randn(c, generator=g), bias=0.2 * torch.randn(c, generator=g),
                    running_mean=0.2 * torch.randn(c, generator=g), running_var=0.5 + torch.rand(c, generator=g))
    p2 = ops.ConvPlan(torch.randn(Cmid, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5, bn(Cmid), 1, 1, ops.ACT_RELU, dev)
    p3 = ops.ConvPlan(torch.randn(Cexp, Cmid, 1, 1, generator=g) / Cmid ** 0.5, bn(Cexp), 1, 0, ops.ACT_RELU, dev)
    assert ops.bottleneck_tail_eligible(p2, p3)
    xd = x.to(dev)
    r = torch.randn(N, Cexp, H, W, generator=g).to(dev) if with_res else None
    p2.k_chunk = 4
    mid = p2(xd)
    ref = p3(mid, residual=r)
    out = ops.bottleneck_tail(xd, p2, p3, residual=r)
    assert out.shape == ref.shape and torch.equal(out, ref)
    # the chain form of the same 3x3 layer (implicit GEMM), and both against float64 on the first images
    n2 = min(N, 2)
    chain = torch.empty((n2, Cmid, H, W), device=dev)
    ops._call("rfx_conv2d_f32", dev, ops._p(xd[:n2].contiguous()), ops._p(p2.wT), ops._p(p2.ktab), ops._p(p2.scale), ops._p(p2.shift), ops._p(None),
              ops._p(chain), n2, Cin, H, W, Cmid, 3, 3, 1, 1, ops.ACT_RELU)
    w2 = p2.wT[:Cin * 9, :Cmid].t().reshape(Cmid, Cin, 3, 3).cpu().double()
    ref64 = F.relu(F.conv2d(x[:n2].double(), w2, padding=1) * p2.scale.cpu().double().view(1, -1, 1, 1) + p2.shift.cpu().double().view(1, -1, 1, 1))
    e_chunk = float(((mid[:n2].cpu().double() - ref64) ** 2).mean().sqrt())
    e_chain = float(((chain.cpu().double() - ref64) ** 2).mean().sqrt())
    print("tail 3x3, K = %d: rms error vs float64: chunked %.3e, chain %.3e" % (9 * Cin, e_chunk, e_chain))
    if Cin * 9 > 288:                                         # more than one chunk: the sums differ, and the blocked one is closer
        assert not torch.equal(mid[:n2], chain) and e_chunk < e_chain
    else:
        assert torch.equal(mid[:n2], chain)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 256, 1024, 20, 28, True, "relu"), (2, 1024, 256, 25, 33, False, "relu"), (1, 128, 64, 9, 11, False, "none"),
                                   (2, 64, 200, 16, 16, True, "relu"), (1, 32, 24, 5, 5, False, "sigmoid"), (5, 512, 128, 30, 40, False, "relu"),
                                   (3, 256, 512, 25, 33, False, "none", 2), (2, 512, 1024, 20, 28, False, "none", 2)])
def test_split_1x1_convolution_is_closer_to_float64_than_the_fp32_kernel(dev, shape, monkeypatch):
    """rfx_conv1x1_split_f32 (csrc/conv1x1s.hip): float32 sums from the three exact bf16 pieces of both operands on the bf16 matrix
    cores.  Every product is exact, the accumulators round once per 16 k: against a float64 convolution its rms error is not larger
    than the fp32-MFMA kernel's (it is 0.45-0.7x), and both agree to float32 round-off.  Shapes: full / ragged pixel tiles, 64- and
    128-channel tiles, a Cout that fills neither, every activation, with and without residual."""
    monkeypatch.setenv("RFX_CONV_SPLIT", "1")            # the kernels under test, whatever the environment routes
    N, Cin, Cout, H, W, has_res, act = shape[:7]
    stride = shape[7] if len(shape) > 7 else 1            # 2: the projection shortcuts (rfx_conv1x1_split_strided_f32)
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cout) ** 0.5
    bn = dict(weight=1.0 + 0.2 * (torch.rand(Cout, generator=g) - 0.5), bias=0.1 * torch.randn(Cout, generator=g),
              running_mean=0.1 * torch.randn(Cout, generator=g), running_var=1.0 + 0.4 * (torch.rand(Cout, generator=g) - 0.5))
    a = dict(relu=ops.ACT_RELU, none=ops.ACT_NONE, sigmoid=ops.ACT_SIGMOID)[act]
    p32 = ops.ConvPlan(w, bn, stride, 0, a, dev)
    psp = ops.ConvPlan(w, bn, stride, 0, a, dev, split=True)
    assert psp.wS is not None and p32.wS is None
    x = torch.relu(torch.randn(N, Cin, H, W, generator=g)).to(dev)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = torch.randn(N, Cout, Ho, Wo, generator=g).to(dev) if has_res else None
    y64 = torch.einsum("mk,nkhw->nmhw", w.view(Cout, Cin).double().to(dev), x[:, :, ::stride, ::stride].double())
    y64 = y64 * p32.scale.double().view(1, -1, 1, 1) + p32.shift.double().view(1, -1, 1, 1)
    if has_res:
        y64 = y64 + res.double()
    y64 = torch.relu(y64) if act == "relu" else (torch.sigmoid(y64) if act == "sigmoid" else y64)
    y32, ysp = p32(x, residual=res), psp(x, residual=res)
    rms = float(y64.pow(2).mean().sqrt())
    e32 = float((y32.double() - y64).pow(2).mean().sqrt()) / rms
    esp = float((ysp.double() - y64).pow(2).mean().sqrt()) / rms
    assert esp <= 1.05 * e32 + 1e-9, (esp, e32)
    assert esp < 3e-7 and float((ysp - y32).abs().max()) / rms < 2e-5
    # the pieces are exact: hi + mid + lo == w
    pc = psp.wS.view(torch.bfloat16).float().cpu()
    Mpad = (Cout + 127) // 128 * 128
    back = pc.view(Cin // 16, 3, 2, Mpad, 8).permute(1, 3, 0, 2, 4).reshape(3, Mpad, Cin).double().sum(0)[:Cout]
    assert torch.equal(back.float(), w.view(Cout, Cin))
    # grouped launches record the same kernel
    with ops.launch_group(dev, False):
        yg = psp(x, residual=res)
    assert torch.equal(yg, ysp)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 256, 256, 30, 40, False, "relu"), (2, 64, 64, 25, 33, True, "relu"), (1, 128, 128, 9, 11, True, "none"),
                                   (2, 512, 256, 17, 19, False, "relu"), (1, 16, 24, 5, 7, False, "sigmoid"), (4, 256, 128, 8, 16, False, "relu"),
                                   (3, 49, 512, 15, 20, False, "relu"), (2, 40, 64, 33, 17, True, "none")])
def test_split_3x3_convolution_against_float64_and_the_fp32_kernel(dev, shape, monkeypatch):
    """rfx_conv3x3_split_f32 (csrc/conv3x3s.hip): the 3x3 / stride 1 / pad 1 convolution from exact bf16 operand pieces.  Against a
    float64 convolution its rms error stays within 1.6x of the fp32 kernel's K-blocked sum (it is 0.4-1.0x up to K = 2304, 1.5x at
    K = 4608) and far below the fp32 MFMA's single fma chain; borders (zero padding), images that straddle a workgroup's patch
    (the batch is tiled as one tall map), ragged column tiles, 64- and 128-channel tiles, partial channel tiles, every activation."""
    monkeypatch.setenv("RFX_CONV_SPLIT", "1")            # the kernels under test, whatever the environment routes
    N, Cin, Cout, H, W, has_res, act = shape
    g = torch.Generator().manual_seed(Cin * 5 + Cout + H)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cout)) ** 0.5
    bn = dict(weight=1.0 + 0.2 * (torch.rand(Cout, generator=g) - 0.5), bias=0.1 * torch.randn(Cout, generator=g),
              running_mean=0.1 * torch.randn(Cout, generator=g), running_var=1.0 + 0.4 * (torch.rand(Cout, generator=g) - 0.5))
    a = dict(relu=ops.ACT_RELU, none=ops.ACT_NONE, sigmoid=ops.ACT_SIGMOID)[act]
    p32 = ops.ConvPlan(w, bn, 1, 1, a, dev)
    psp = ops.ConvPlan(w, bn, 1, 1, a, dev, split=True)
    assert psp.wS is not None and tuple(psp.wS.shape[:3]) == ((Cin + 15) // 16, 9, 3)        # a ragged last block: 49 / 40 channels
    x = torch.relu(torch.randn(N, Cin, H, W, generator=g)).to(dev)
    res = torch.randn(N, Cout, H, W, generator=g).to(dev) if has_res else None
    y64 = F.conv2d(x.double(), w.double().to(dev), padding=1)
    y64 = y64 * p32.scale.double().view(1, -1, 1, 1) + p32.shift.double().view(1, -1, 1, 1)
    if has_res:
        y64 = y64 + res.double()
    y64 = torch.relu(y64) if act == "relu" else (torch.sigmoid(y64) if act == "sigmoid" else y64)
    y32, ysp = p32(x, residual=res), psp(x, residual=res)
    rms = float(y64.pow(2).mean().sqrt())
    e32 = float((y32.double() - y64).pow(2).mean().sqrt()) / rms
    esp = float((ysp.double() - y64).pow(2).mean().sqrt()) / rms
    assert esp <= 1.6 * e32 + 1e-9 and esp < 5e-7, (esp, e32)
    assert float((ysp - y32).abs().max()) / rms < 3e-5
    with ops.launch_group(dev, False):
        yg = psp(x, residual=res)
    assert torch.equal(yg, ysp)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 128, 128, 30, 40, "relu"), (2, 256, 256, 25, 33, "relu"), (2, 64, 128, 17, 19, "none"), (5, 128, 256, 8, 16, "relu"),
                                   (1, 64, 200, 9, 7, "sigmoid")])
def test_split_3x3_stride2_convolution_against_float64_and_the_fp32_kernel(dev, shape, monkeypatch):
    """rfx_conv3x3_split_s2_f32: the stride-2 form (parity-de-interleaved 17 x 33 patch, odd and even input sizes, images straddling a
    workgroup's output rows, ragged column tiles, a Cout that does not fill its last channel tile)."""
    monkeypatch.setenv("RFX_CONV_SPLIT", "1")            # the kernels under test, whatever the environment routes
    N, Cin, Cout, H, W, act = shape
    g = torch.Generator().manual_seed(Cin * 3 + Cout + H)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cout)) ** 0.5
    bn = dict(weight=1.0 + 0.2 * (torch.rand(Cout, generator=g) - 0.5), bias=0.1 * torch.randn(Cout, generator=g),
              running_mean=0.1 * torch.randn(Cout, generator=g), running_var=1.0 + 0.4 * (torch.rand(Cout, generator=g) - 0.5))
    a = dict(relu=ops.ACT_RELU, none=ops.ACT_NONE, sigmoid=ops.ACT_SIGMOID)[act]
    p32 = ops.ConvPlan(w, bn, 2, 1, a, dev)
    psp = ops.ConvPlan(w, bn, 2, 1, a, dev, split=True)
    assert psp.wS is not None
    x = torch.relu(torch.randn(N, Cin, H, W, generator=g)).to(dev)
    y64 = F.conv2d(x.double(), w.double().to(dev), stride=2, padding=1)
    y64 = y64 * p32.scale.double().view(1, -1, 1, 1) + p32.shift.double().view(1, -1, 1, 1)
    y64 = torch.relu(y64) if act == "relu" else (torch.sigmoid(y64) if act == "sigmoid" else y64)
    y32, ysp = p32(x), psp(x)
    assert ysp.shape == y64.shape
    rms = float(y64.pow(2).mean().sqrt())
    e32 = float((y32.double() - y64).pow(2).mean().sqrt()) / rms
    esp = float((ysp.double() - y64).pow(2).mean().sqrt()) / rms
    assert esp <= 1.6 * e32 + 1e-9 and esp < 5e-7, (esp, e32)
    assert float((ysp - y32).abs().max()) / rms < 3e-5
    with ops.launch_group(dev, False):
        yg = psp(x)
    assert torch.equal(yg, ysp)
