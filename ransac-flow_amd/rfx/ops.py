"""Tensor-level wrappers over the librfx C ABI.

PyTorch is used for device memory and streams only: every function takes ``torch`` tensors that already
live on a HIP device, passes ``data_ptr()`` / sizes / that device's current torch stream to the C entry point
(``_call``: device-guarded) and returns freshly allocated output tensors.  No op has a CPU or eager-PyTorch fallback.
"""
import collections
import ctypes
import os
import threading

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2


def _stream(dev=None):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _call(name, dev, *args):
    """lib.<name>(*args, stream) on ``dev``: the stream is torch's current stream OF THAT DEVICE and the HIP
    current device is switched for the launch when the tensors do not live on it (ctypes bypasses torch's
    device guard)."""
    fn = getattr(_lib.load(), name)
    if dev.index is not None and dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev):
            rc = fn(*args, _stream(dev))
    else:
        rc = fn(*args, _stream())
    _lib.check(rc, name)


def _one_device(*tensors):
    """All operands of an op must share one HIP device (foreign pointers fault or, worse, do not)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("rfx op operands live on different devices: %s and %s" % (dev, t.device))
    return dev


class Profiler:
    """Explicit, thread-local launch profiler (bench.py): inside ``with ops.Profiler() as prof:`` every
    convolution-class launch and every correlation launch of THIS thread is bracketed by two HIP events on the
    launch stream; nothing is recorded -- and no event is created -- outside such a block.

        prof.conv: list of (kernel_id, flops, ev0, ev1, shape, algorithmic_bytes)
        prof.corr: list of (algorithmic_bytes, ev0, ev1)                       one-direction launches
        prof.corr_bidir: list of (minimal_bytes, per_direction_bytes, ev0, ev1)   both directions of a pair in one launch
    """
    _tls = threading.local()

    def __init__(self):
        self.conv, self.corr, self.corr_bidir = [], [], []
        self.corr_kernel_us = None      # kernel durations of the correlation launches in launch order (corr_durations())
        self._corr_order = []           # "one" / "bidir" per captured launch

    def __enter__(self):
        self._prev = getattr(Profiler._tls, "active", None)
        Profiler._tls.active = self
        # the correlation launches of this thread carry dispatch-level start / stop events from here on (rfx_corr_timing)
        self._prev_timing = _lib.load().rfx_corr_timing(1)
        _lib.load().rfx_corr_timing_collect(None, 0)          # drop captures nobody collected
        return self

    def __exit__(self, *exc):
        Profiler._tls.active = self._prev
        _lib.load().rfx_corr_timing(self._prev_timing)
        return False

    def corr_durations(self):
        """Kernel durations (us) of the captured correlation launches: (one-direction list, bidir list), each in launch order.
        Call after the profiled region (synchronises on the captured events)."""
        if self.corr_kernel_us is None:
            n = len(self._corr_order)
            buf = (ctypes.c_float * max(n, 1))()
            got = _lib.load().rfx_corr_timing_collect(buf, n)
            us = [float(buf[i]) for i in range(n)] if got == n else []      # a launch that bypassed the DMA kernels: no pairing possible
            self.corr_kernel_us = ([u for u, k in zip(us, self._corr_order) if k == "one"],
                                   [u for u, k in zip(us, self._corr_order) if k == "bidir"])
        return self.corr_kernel_us

    @staticmethod
    def active():
        return getattr(Profiler._tls, "active", None)

    @staticmethod
    def begin(dev=None):
        """Start event on the current stream OF THE DEVICE THE LAUNCH GOES TO (``dev``: a tensor of the op or its device;
        default: the current device) -- the stream _call() launches on."""
        if getattr(Profiler._tls, "active", None) is None:
            return None
        dev = dev.device if isinstance(dev, torch.Tensor) else dev
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
        e0._rfx_dev = dev
        return e0

    @staticmethod
    def end(e0):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(torch.cuda.current_stream(getattr(e0, "_rfx_dev", None)))
        return e1


class launch_group:
    """``with ops.launch_group(device):`` -- the convolution ops issued inside are RECORDED by the library and launched at exit
    as ONE launch per kernel instance (rfx_group_begin / rfx_group_end: blockIdx.y selects the problem).  For the same layer
    of several independent inputs of different sizes (the 8 images of a single pair's trunk pass); the calls inside must not
    depend on each other, and their tensors must stay referenced until the block ends (the returned outputs do).  Not used
    under a Profiler (per-launch events are meaningless for a recorded launch)."""

    def __init__(self, device, side_streams=True):
        # side_streams=False: the group's kernel instances stay on the caller's stream (rfx_group_side_streams) -- for callers that
        # run several grouped chains on streams of their own
        self.dev, self.side = torch.device(device), bool(side_streams)

    def __enter__(self):
        _lib.check(_lib.load().rfx_group_begin(), "rfx_group_begin")
        return self

    def __exit__(self, et, ev, tb):
        lib = _lib.load()
        if et is not None:
            lib.rfx_group_abort()
            return False
        prev = lib.rfx_group_side_streams(1 if self.side else 0)
        try:
            if self.dev.index is not None and self.dev.index != torch.cuda.current_device():
                with torch.cuda.device(self.dev):
                    rc = lib.rfx_group_end(_stream(self.dev))
            else:
                rc = lib.rfx_group_end(_stream())
        finally:
            lib.rfx_group_side_streams(prev)
        _lib.check(rc, "rfx_group_end")
        return False


def _dev(t, name="tensor", dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s is on %s: rfx ops run only on a HIP device (no CPU fallback)" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


KID_SPLIT_3X3 = 65536      # ... of rfx_conv3x3_split_f32
KID_SPLIT_1X1 = 32768      # Profiler kernel id of rfx_conv1x1_split_f32 (| 1: 64-channel tiles, | 2: 128-channel tiles)


def conv_split_enabled():
    """RFX_CONV_SPLIT=0: every convolution on the fp32 MFMA kernels (A/B runs; the float32-only figure of the bench line)."""
    return os.environ.get("RFX_CONV_SPLIT", "1") != "0"


def split_weights(w2d):
    """(Cout, Cin) float32 -> rfx_conv1x1_split_f32's wS: the three bf16 pieces of every weight (hi = bf16(w), mid = bf16(w - hi),
    lo = bf16(w - hi - mid); round to nearest even, hi + mid + lo == w exactly) in fragment order
    [k / 16][piece][(k % 16) / 8][m (Mpad)][k % 8], as int16 bit patterns."""
    w = w2d.detach().float().cpu()
    Cout, Cin = w.shape
    if Cin % 16:       # a ragged last block (rfx_conv3x3_split_f32 stages the missing channels as zeros): zero weights
        w = torch.cat((w, torch.zeros(Cout, 16 - Cin % 16)), dim=1)
        Cin = w.shape[1]
    hi = w.bfloat16()
    r1 = w - hi.float()
    mid = r1.bfloat16()
    lo = (r1 - mid.float()).bfloat16()
    Mpad = (Cout + 127) // 128 * 128
    pieces = torch.zeros(3, Mpad, Cin, dtype=torch.bfloat16)
    pieces[:, :Cout] = torch.stack((hi, mid, lo))
    return pieces.view(3, Mpad, Cin // 16, 2, 8).permute(2, 0, 3, 1, 4).contiguous().view(torch.int16)


class ConvPlan:
    """Packed weights + folded BatchNorm of one convolution (see rfx_conv2d_f32 in include/rfx_api.h)."""

    def __init__(self, weight, bn=None, stride=1, pad=0, act=ACT_NONE, device=None, eps=1e-5, dilation=1, bias=None, split=False):
        # weight: (Cout, Cin, KH, KW) float32 (any device); bn: dict(weight,bias,running_mean,running_var) or None;
        # dilation > 1: rfx_conv2d_dilated_f32 (the sky-segmentation encoder, segNet/segModel.py:196-205); bias: the convolution's own
        # bias (segModel.py:243,245: the classifier convolutions), only without bn
        w = weight.detach().float().cpu()
        self.Cout, self.Cin, self.KH, self.KW = w.shape
        self.stride, self.pad, self.act, self.dilation = stride, pad, act, int(dilation)
        if bias is not None and bn is not None:
            raise ValueError("ConvPlan: a convolution bias together with a BatchNorm is not folded here")
        # rfx_conv3x3_f32's k_chunk: 0 = the library's rule (chunks for K >= 2048); 4 = chunks of 4 K steps whatever K is -- set by
        # the nets on the 3x3 convolution of a Bottleneck tail, so that the two-kernel form equals the fused kernel bit for bit
        self.k_chunk = 0
        K = self.Cin * self.KH * self.KW
        Kpad, Mpad = (K + 31) // 32 * 32, (self.Cout + 127) // 128 * 128
        wT = torch.zeros(Kpad, Mpad, dtype=torch.float32)
        wT[:K, :self.Cout] = w.reshape(self.Cout, K).t()
        self.w2d = w.reshape(self.Cout, K) if (self.KH == 1 and self.KW == 1) else None   # kept for quad_weights()
        self._wq = None
        self.wP = None
        # split=True: a 1x1 / stride 1 convolution runs on rfx_conv1x1_split_f32 (float32 sums from exact bf16 operand pieces on the
        # bf16 matrix cores: csrc/conv1x1s.hip) where the shape allows it and RFX_CONV_SPLIT != 0
        self.wS = None
        if (split and conv_split_enabled() and self.KH == 1 and self.KW == 1 and stride in (1, 2) and pad == 0 and self.dilation == 1
                and self.Cin % 16 == 0):
            self.wS = split_weights(w.reshape(self.Cout, K)).to(device or "cuda")
        elif (split and conv_split_enabled() and self.KH == 3 and self.KW == 3 and pad == 1 and self.dilation == 1
                and ((stride == 1 and self.Cin >= 16) or (stride == 2 and self.Cin % 16 == 0 and self.Cout >= 128))):
            # rfx_conv3x3_split_f32's wS3: [c / 16][tap][piece][h][m][8]
            self.wS = torch.stack([split_weights(w[:, :, kh, kw]) for kh in range(3) for kw in range(3)], dim=1).contiguous().to(device or "cuda")
        if (self.KH == 3 and self.KW == 3 and pad == 1 and self.dilation == 1 and self.Cin >= 8
                and (stride == 1 or (stride == 2 and self.Cin % 8 == 0))):
            # rfx_conv3x3_f32's order: wP[mt][s][h][m][kk] = W[mt*128 + m][s*72 + 2*kk + h]; a Cin that is not a multiple of 8
            # (the 49-channel correlation volume) gets zero rows for the missing channels of its last K step
            Kp = (self.Cin + 7) // 8 * 72
            wp = torch.zeros(Mpad, Kp, dtype=torch.float32)
            wp[:self.Cout, :K] = w.reshape(self.Cout, K)
            wp = wp.view(Mpad // 128, 128, Kp // 72, 36, 2).permute(0, 2, 4, 1, 3).contiguous()
            self.wP = wp.to(device or "cuda")
        k = torch.arange(K)
        c, r = k // (self.KH * self.KW), k % (self.KH * self.KW)
        ktab = torch.full((Kpad,), -1, dtype=torch.int32)
        # the gather adds these offsets to the window origin as they are: a dilated convolution stores kh*d / kw*d
        ktab[:K] = ((c << 8) | (((r // self.KW) * self.dilation) << 4) | ((r % self.KW) * self.dilation)).int()
        # per 32-k block: [16 even k | 16 odd k] (the order in which one MFMA lane-half consumes them)
        ktab = ktab.view(-1, 16, 2).permute(0, 2, 1).reshape(-1).contiguous()
        dev = device or "cuda"
        self.wT, self.ktab = wT.to(dev), ktab.to(dev)
        if bn is not None:
            # eval-mode BatchNorm as ATen's CPU kernel evaluates it: alpha = w * (1/sqrt(var+eps)), beta = b - mean*alpha
            invstd = 1.0 / torch.sqrt(bn["running_var"].detach().float().cpu() + eps)
            alpha = bn["weight"].detach().float().cpu() * invstd
            beta = bn["bias"].detach().float().cpu() - bn["running_mean"].detach().float().cpu() * alpha
            self.scale, self.shift = alpha.to(dev), beta.to(dev)
        elif bias is not None:
            self.scale, self.shift = torch.ones(self.Cout, dtype=torch.float32).to(dev), bias.detach().float().cpu().to(dev)
        else:
            self.scale = self.shift = None

    def quad_weights(self):
        """1x1 weights in the order rfx_conv3x3_conv1x1_f32 reads them: wQ[q][h][m][j] = W[m][8q + 2j + h]."""
        if getattr(self, "_wq", None) is None:
            w = self.w2d                                            # (Cout, Cin) on the CPU
            Cout, Cin = w.shape
            q = w.t().reshape(Cin // 8, 4, 2, Cout)                  # [q][j][h][m]
            self._wq = q.permute(0, 2, 3, 1).contiguous().to(self.wT.device)   # [q][h][m][j]
        return self._wq

    def out_hw(self, H, W):
        eh, ew = (self.KH - 1) * self.dilation + 1, (self.KW - 1) * self.dilation + 1
        return (H + 2 * self.pad - eh) // self.stride + 1, (W + 2 * self.pad - ew) // self.stride + 1

    def __call__(self, x, residual=None, act=None):
        x = _dev(x, "conv input")
        N, C, H, W = x.shape
        if C != self.Cin:
            raise ValueError("conv expects %d input channels, got %d" % (self.Cin, C))
        Ho, Wo = self.out_hw(H, W)
        out = torch.empty((N, self.Cout, Ho, Wo), dtype=torch.float32, device=x.device)
        res = _dev(residual, "residual") if residual is not None else None
        if res is not None and res.shape != out.shape:
            raise ValueError("residual shape %s != output shape %s" % (tuple(res.shape), tuple(out.shape)))
        lib = _lib.load()
        if self.dilation != 1:
            _call("rfx_conv2d_dilated_f32", _one_device(x, res, self.wT), _p(x), _p(self.wT), _p(self.ktab), _p(self.scale), _p(self.shift),
                  _p(res), _p(out), N, C, H, W, self.Cout, self.KH, self.KW, self.stride, self.pad, self.dilation,
                  self.act if act is None else act)
            return out
        if self.wS is not None and self.KH == 3:
            e0 = Profiler.begin(x)
            xin, Cp = x, C
            if C % 16:
                # the 49-channel correlation volume: the kernel takes whole blocks of 16 channels -- the input goes into a zero-padded
                # buffer (one device copy, 5 % of the layer's time; the padded weights are zero: split_weights)
                Cp = (C + 15) // 16 * 16
                xin = torch.zeros((N, Cp, H, W), dtype=torch.float32, device=x.device)
                xin[:, :C].copy_(x)
            _call("rfx_conv3x3_split_f32" if self.stride == 1 else "rfx_conv3x3_split_s2_f32", _one_device(xin, res, self.wS), _p(xin),
                  _p(self.wS), _p(self.scale), _p(self.shift), _p(res), _p(out), N, Cp, H, W, self.Cout, self.act if act is None else act)
            if e0 is not None:
                e1 = Profiler.end(e0)
                Profiler.active().conv.append((KID_SPLIT_3X3 | (2 if self.Cout > 64 else 1) | (4 if self.stride != 1 else 0),
                                               2.0 * N * Ho * Wo * self.Cout * self.Cin * 9, e0, e1, (N, self.Cin, H, W, self.Cout, 3, self.stride),
                                               4.0 * (N * C * H * W + N * self.Cout * Ho * Wo * (2 if res is not None else 1)) + 54.0 * self.Cout * self.Cin))
            return out
        if self.wS is not None:
            e0 = Profiler.begin(x)
            if self.stride == 1:
                _call("rfx_conv1x1_split_f32", _one_device(x, res, self.wS), _p(x), _p(self.wS), _p(self.scale), _p(self.shift), _p(res), _p(out),
                      N, C, H * W, self.Cout, self.act if act is None else act)
            else:
                _call("rfx_conv1x1_split_strided_f32", _one_device(x, res, self.wS), _p(x), _p(self.wS), _p(self.scale), _p(self.shift), _p(res),
                      _p(out), N, C, H, W, self.Cout, self.stride, self.act if act is None else act)
            if e0 is not None:
                e1 = Profiler.end(e0)
                Profiler.active().conv.append((KID_SPLIT_1X1 | (2 if self.Cout > 64 else 1) | (4 if self.stride != 1 else 0),
                                               2.0 * N * Ho * Wo * self.Cout * self.Cin, e0, e1, (N, self.Cin, H, W, self.Cout, 1, self.stride),
                                               4.0 * (N * C * Ho * Wo + N * self.Cout * Ho * Wo * (2 if res is not None else 1)) + 6.0 * self.Cout * self.Cin))
            return out
        kid0 = 0
        if self.wP is not None:
            kid0 = (lib.rfx_conv3x3_kernel_id(N, self.Cin, self.Cout, Ho, Wo, self.k_chunk) if self.stride == 1 else
                    lib.rfx_conv2d_kernel_id(N, self.Cin, self.Cout, self.KH, self.KW, self.stride, self.pad, Ho, Wo))
        e0 = Profiler.begin(x)
        if kid0 & 32:
            _call("rfx_conv3x3_f32", _one_device(x, res, self.wP), _p(x), _p(self.wP), _p(self.scale), _p(self.shift),
                  _p(res), _p(out), N, C, H, W, self.Cout, self.act if act is None else act, self.k_chunk)
        elif kid0 & 8192:
            _call("rfx_conv3x3_s2_f32", _one_device(x, res, self.wP), _p(x), _p(self.wP), _p(self.scale), _p(self.shift),
                  _p(res), _p(out), N, C, H, W, self.Cout, self.act if act is None else act)
        else:
            _call("rfx_conv2d_f32", _one_device(x, res, self.wT), _p(x), _p(self.wT), _p(self.ktab), _p(self.scale),
                  _p(self.shift), _p(res), _p(out), N, C, H, W, self.Cout, self.KH, self.KW, self.stride, self.pad,
                  self.act if act is None else act)
        if e0 is not None:
            e1 = Profiler.end(e0)
            flops = 2.0 * N * Ho * Wo * self.Cout * self.Cin * self.KH * self.KW
            nbytes = 4.0 * (N * C * H * W + N * self.Cout * Ho * Wo * (2 if res is not None else 1)
                            + self.Cout * self.Cin * self.KH * self.KW)
            kid = kid0 if (kid0 & 32) else lib.rfx_conv2d_kernel_id(N, self.Cin, self.Cout, self.KH, self.KW, self.stride, self.pad, Ho, Wo)
            Profiler.active().conv.append((kid, flops, e0, e1, (N, self.Cin, H, W, self.Cout, self.KH, self.stride), nbytes))
        return out


def bottleneck_tail_shape(plan2, plan3):
    """Do conv2 (3x3) + conv3 (1x1 expansion) of a Bottleneck have the shape rfx_conv3x3_conv1x1_f32 serves?"""
    return (plan2.KH == 3 and plan2.KW == 3 and plan2.stride == 1 and plan2.pad == 1 and plan2.Cin % 8 == 0
            and plan2.Cout in (64, 128) and plan2.act in (ACT_NONE, ACT_RELU) and plan3.KH == 1 and plan3.KW == 1
            and plan3.stride == 1 and plan3.pad == 0 and plan3.Cin == plan2.Cout and plan3.Cout % 128 == 0
            and plan3.act in (ACT_NONE, ACT_RELU) and plan3.scale is not None)


def bottleneck_tail_eligible(plan2, plan3):
    """Can conv2 (3x3) + conv3 (1x1 expansion) of a Bottleneck run as one kernel (rfx_conv3x3_conv1x1_f32)?"""
    return (bottleneck_tail_shape(plan2, plan3) and os.environ.get("RFX_FUSE_BOTTLENECK", "1") != "0"
            and getattr(plan2, "wS", None) is None and getattr(plan3, "wS", None) is None)        # split plans run as two kernels


def bottleneck_tail(x, plan2, plan3, residual=None):
    """plan3(plan2(x), residual) in one kernel: the plan2.Cout-channel intermediate stays in LDS."""
    x = _dev(x, "bottleneck input")
    N, C, H, W = x.shape
    if C != plan2.Cin:
        raise ValueError("conv expects %d input channels, got %d" % (plan2.Cin, C))
    res = _dev(residual, "residual") if residual is not None else None
    out = torch.empty((N, plan3.Cout, H, W), dtype=torch.float32, device=x.device)
    if res is not None and res.shape != out.shape:
        raise ValueError("residual shape %s != output shape %s" % (tuple(res.shape), tuple(out.shape)))
    e0 = Profiler.begin(x)
    _call("rfx_conv3x3_conv1x1_f32", _one_device(x, res, plan2.wP, plan3.scale), _p(x), _p(plan2.wP), _p(plan2.scale),
          _p(plan2.shift), plan2.act, _p(plan3.quad_weights()), _p(plan3.scale), _p(plan3.shift), _p(res), plan3.act,
          _p(out), N, C, H, W, plan2.Cout, plan3.Cout)
    if e0 is not None:
        e1 = Profiler.end(e0)
        flops = 2.0 * N * H * W * (plan2.Cout * plan2.Cin * 9 + plan3.Cout * plan3.Cin)
        nbytes = 4.0 * N * H * W * (C + plan3.Cout * (2 if res is not None else 1))
        kid = _lib.load().rfx_conv3x3_conv1x1_kernel_id(N, H, W, plan2.Cout)
        Profiler.active().conv.append((kid, flops, e0, e1,
                                       (N, C, H, W, plan3.Cout, 3, 1), nbytes))
    return out


def maxpool2d(x, k, stride, pad=0):
    x = _dev(x, "maxpool input")
    N, C, H, W = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=x.device)
    _call("rfx_maxpool2d_f32", x.device, _p(x), _p(out), N * C, H, W, k, stride, pad)
    return out


def blurpool2d(x, stride=2):
    x = _dev(x, "blurpool input")
    N, C, H, W = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=x.device)
    _call("rfx_blurpool2d_f32", x.device, _p(x), _p(out), N * C, H, W, stride)
    return out


def maxblurpool2d(x, stride=2):
    """MaxPool2d(2, stride 1) + BlurPool(stride) fused (FeatureExtractor stem)."""
    x = _dev(x, "maxblurpool input")
    N, C, H, W = x.shape
    Ho, Wo = (H - 2) // stride + 1, (W - 2) // stride + 1
    out = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=x.device)
    _call("rfx_maxblurpool2d_f32", x.device, _p(x), _p(out), N * C, H, W, stride)
    return out


def stem_conv_maxblur(x, plan):
    """FeatureExtractor stem: plan = ConvPlan of the 3x3 / stride 1 / pad 1 / ReLU convolution with 3 input channels;
    returns maxblurpool2d(plan(x), 2) without materialising plan(x)."""
    x = _dev(x, "stem input")
    N, C, H, W = x.shape
    if not (C == 3 and plan.Cin == 3 and plan.KH == 3 and plan.KW == 3 and plan.stride == 1 and plan.pad == 1
            and plan.act == ACT_RELU and plan.Cout % 32 == 0):
        raise ValueError("stem_conv_maxblur: not a 3x3/s1/p1 ReLU convolution of a 3-channel image")
    Ho, Wo = (H - 2) // 2 + 1, (W - 2) // 2 + 1
    out = torch.empty((N, plan.Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    e0 = Profiler.begin(x)
    _call("rfx_stem_conv3x3_maxblur_f32", _one_device(x, plan.wT), _p(x), _p(plan.wT), _p(plan.scale), _p(plan.shift), _p(out), N, H, W,
                                                       plan.Cout)
    if e0 is not None:
        e1 = Profiler.end(e0)
        Profiler.active().conv.append((256, 2.0 * N * H * W * plan.Cout * 27, e0, e1, (N, 3, H, W, plan.Cout, 3, 1),
                      4.0 * (N * 3 * H * W + N * plan.Cout * Ho * Wo)))
    return out


def stem_conv7_maxpool(x, plan):
    """ResNet-50 stem: plan = ConvPlan of the 7x7 / stride 2 / pad 3 / ReLU convolution with 3 input channels; returns
    maxpool2d(plan(x), 3, 2, 1) without materialising plan(x)."""
    x = _dev(x, "stem input")
    N, C, H, W = x.shape
    if not (C == 3 and plan.Cin == 3 and plan.KH == 7 and plan.KW == 7 and plan.stride == 2 and plan.pad == 3
            and plan.act == ACT_RELU and plan.Cout % 32 == 0):
        raise ValueError("stem_conv7_maxpool: not a 7x7/s2/p3 ReLU convolution of a 3-channel image")
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    out = torch.empty((N, plan.Cout, Hp, Wp), dtype=torch.float32, device=x.device)
    e0 = Profiler.begin(x)
    _call("rfx_stem_conv7x7_maxpool_f32", _one_device(x, plan.wT), _p(x), _p(plan.wT), _p(plan.scale), _p(plan.shift), _p(out), N, H, W,
                                                       plan.Cout)
    if e0 is not None:
        e1 = Profiler.end(e0)
        Profiler.active().conv.append((257, 2.0 * N * Hc * Wc * plan.Cout * 147, e0, e1, (N, 3, H, W, plan.Cout, 7, 2),
                      4.0 * (N * 3 * H * W + N * plan.Cout * Hp * Wp)))
    return out


def adaptive_avgpool2d(x, size):
    """nn.AdaptiveAvgPool2d(size) (segNet/segModel.py:227)."""
    x = _dev(x, "adaptive_avgpool input")
    N, C, H, W = x.shape
    Ho, Wo = (int(size), int(size)) if isinstance(size, int) else (int(size[0]), int(size[1]))
    out = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=x.device)
    _call("rfx_adaptive_avgpool2d_f32", x.device, _p(x), _p(out), N * C, H, W, Ho, Wo)
    return out


def softmax_accum(logits, scores=None, div=1.0):
    """scores (+)= softmax(logits, dim=1) / div for (N,C,H,W) logits (segNet/segModel.py:258 + segNet/segEval.py:34-35); a new
    tensor when ``scores`` is None, else updated IN PLACE."""
    logits = _dev(logits, "logits")
    N, C = logits.shape[0], logits.shape[1]
    HW = logits.numel() // (N * C)
    acc = scores is not None
    if acc:
        if not scores.is_contiguous() or scores.shape != logits.shape or scores.dtype != torch.float32 or not scores.is_cuda:
            raise ValueError("scores must be a contiguous float32 device tensor of the logits' shape (updated in place)")
    else:
        scores = torch.empty_like(logits)
    _call("rfx_softmax_accum_f32", _one_device(logits, scores), _p(logits), _p(scores), N, C, HW, float(div), 1 if acc else 0)
    return scores


def argmax_mask(scores, class_id, complement=False, want_pred=False):
    """torch.max(scores, dim=1) -> float32 mask (N,H,W) = (pred == class_id) (``complement``: 1 - that) [, pred int32]
    (segNet/segEval.py:37-43)."""
    scores = _dev(scores, "scores")
    N, C, H, W = scores.shape
    mask = torch.empty((N, H, W), dtype=torch.float32, device=scores.device)
    pred = torch.empty((N, H, W), dtype=torch.int32, device=scores.device) if want_pred else None
    _call("rfx_argmax_mask_f32", scores.device, _p(scores), N, C, H * W, int(class_id), 1 if complement else 0, _p(mask), _p(pred))
    return (mask, pred) if want_pred else mask


def l2norm(x, out=None, out_batch_stride=0, out_chan_stride=0):
    """F.normalize(x, dim=1) for (N,C,H,W) or (N,C,L).  With ``out`` (a float32 device tensor/view start) the
    result is scattered with the given strides (elements)."""
    x = _dev(x, "l2norm input")
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C)
    if out is None:
        out = torch.empty_like(x)
        obs = ocs = 0
    else:
        if not out.is_cuda or out.dtype != torch.float32:
            raise TypeError("l2norm out must be a float32 device tensor")
        obs, ocs = out_batch_stride, out_chan_stride
    _call("rfx_l2norm_nchw_f32", _one_device(x, out), _p(x), _p(out), N, C, HW, obs, ocs)
    return out


def flow_head(logits, K=7):
    logits = _dev(logits, "flow logits")
    N, C, R, Cc = logits.shape
    if C != K * K:
        raise ValueError("flow head expects %d logits, got %d" % (K * K, C))
    out = torch.empty((N, 2, R, Cc), dtype=torch.float32, device=logits.device)
    _call("rfx_flow_head_f32", logits.device, _p(logits), _p(out), N, K, R, Cc)
    return out


def resize_bilinear(x, size, align_corners=False):
    x = _dev(x, "resize input")
    N, C, H, W = x.shape
    Ho, Wo = int(size[0]), int(size[1])
    out = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=x.device)
    _call("rfx_resize_bilinear_f32", x.device, _p(x), _p(out), N * C, H, W, Ho, Wo, 1 if align_corners else 0)
    return out


_LANCZOS_TABLES = collections.OrderedDict()      # (sizes, device) -> coefficient tables, LRU-bounded: a stream of variable-size images
_LANCZOS_MAX = 64                                # must not grow it for ever (a table set is a few KB)


def lanczos_resize_u8(img, out_w, out_h):
    """PIL.Image.resize((out_w, out_h), LANCZOS) for a uint8 (N,H,W,3) device tensor -- bit-identical to Pillow."""
    from . import lanczos
    img = _dev(img, "image", torch.uint8)
    N, H, W, C = img.shape
    p = lanczos.plan(W, H, out_w, out_h)
    if p is None:
        return img.clone()
    key = (W, H, out_w, out_h, str(img.device))
    tabs = _LANCZOS_TABLES.get(key)
    if tabs is None:
        tabs = {k: torch.from_numpy(p[k]).to(img.device) for k in ("bounds_h", "kk_h", "bounds_v", "kk_v")}
        _LANCZOS_TABLES[key] = tabs
        while len(_LANCZOS_TABLES) > _LANCZOS_MAX:
            # the evicted tensors stay alive for as long as a kernel that reads them is queued (stream-ordered allocator); a captured
            # HIP graph that baked their addresses in holds its own references (pipeline._features_graphed keeps `prep`)
            _LANCZOS_TABLES.popitem(last=False)
    else:
        _LANCZOS_TABLES.move_to_end(key)
    lib = _lib.load()
    cur, curH = img, H
    if p["need_h"]:
        tmp = torch.empty((N, p["rows_h"], out_w, C), dtype=torch.uint8, device=img.device)
        _call("rfx_lanczos_pass_u8", img.device, _p(cur), _p(tmp), N, H, W, p["rows_h"], out_w, C, _p(tabs["bounds_h"]),
                                           _p(tabs["kk_h"]), p["ks_h"], 0, p["y_first"])
        cur, curH = tmp, p["rows_h"]
    if p["need_v"]:
        out = torch.empty((N, out_h, out_w, C), dtype=torch.uint8, device=img.device)
        _call("rfx_lanczos_pass_u8", img.device, _p(cur), _p(out), N, curH, out_w, out_h, out_w, C, _p(tabs["bounds_v"]),
                                           _p(tabs["kk_v"]), p["ks_v"], 1, 0)
        cur = out
    return cur


def u8_to_f32(img, mean=None, std=None, want_raw=True):
    """ToTensor (/255) and, with mean/std, Normalize: uint8 (N,H,W,3) -> float32 (N,3,H,W) raw and/or normalised."""
    img = _dev(img, "image", torch.uint8)
    N, H, W, C = img.shape
    if C != 3:
        raise ValueError("u8_to_f32 expects 3 channels")
    raw = torch.empty((N, 3, H, W), dtype=torch.float32, device=img.device) if want_raw else None
    norm = torch.empty((N, 3, H, W), dtype=torch.float32, device=img.device) if mean is not None else None
    m = (ctypes.c_float * 3)(*mean) if mean is not None else None
    s = (ctypes.c_float * 3)(*std) if std is not None else None
    _call("rfx_u8_to_f32_chw", img.device, _p(img), _p(raw), _p(norm), N, H, W, m, s)
    return raw, norm


def copy_cols(src, w_dst):
    """(rows, w_src) -> (rows, w_dst): copy min(w_src, w_dst) columns per row, zero-fill the rest."""
    src = _dev(src, "matrix")
    rows, ws = src.shape
    dst = torch.empty((rows, int(w_dst)), dtype=torch.float32, device=src.device)
    _call("rfx_copy_cols_f32", src.device, _p(src), _p(dst), rows, ws, int(w_dst))
    return dst


def corr_neigh(x, y, K=7, variant=None):
    """``variant``: force a tile shape of rfx_corr_neigh_variant_f32 (tuning / tests); default = RFX_CORR_VARIANT or
    the library's automatic choice.  All variants are bit-identical."""
    x, y = _dev(x, "corr x"), _dev(y, "corr y")
    if x.shape != y.shape:
        raise ValueError("corr_neigh: x %s and y %s differ" % (tuple(x.shape), tuple(y.shape)))
    N, C, H, W = x.shape
    if W % 4 != 0 and variant is None and os.environ.get("RFX_CORR_PAD", "1") == "1":
        # the LDS-DMA kernels move 16-byte quads: run them on copies zero-padded to a multiple of 4 columns and crop -- the
        # window's own padding is zero (model/model.py:135,143), so the result is the unpadded one, bit for bit
        Wp = (W + 3) // 4 * 4
        xp, yp = copy_cols(x.view(-1, W), Wp).view(N, C, H, Wp), copy_cols(y.view(-1, W), Wp).view(N, C, H, Wp)
        return copy_cols(corr_neigh(xp, yp, K).view(-1, Wp), W).view(N, K * K, H, W)
    out = torch.empty((N, K * K, H, W), dtype=torch.float32, device=x.device)
    e0 = Profiler.begin(x)
    v = int(os.environ.get("RFX_CORR_VARIANT", "0")) if variant is None else int(variant)
    _call("rfx_corr_neigh_variant_f32", _one_device(x, y), _p(x), _p(y), _p(out), N, C, H, W, K, v)
    if e0 is not None:
        Profiler.active().corr.append(((2 * C + K * K) * 4.0 * N * H * W, e0, Profiler.end(e0)))  # algorithmic bytes
        Profiler.active()._corr_order.append("one")
    return out


def corr_neigh_bidir(x, y, K=7, out=None):
    """Both directions of a pair in ONE launch: returns (corr_neigh(x, y), corr_neigh(y, x)), each (N,49,H,W).  The reverse
    volume is the forward one at mirrored taps and shifted pixels -- corr(y,x)[(6-i)*7+(6-j), r+i-3, c+j-3] =
    corr(x,y)[i*7+j, r, c], the same channel-ordered sums -- so it costs a second store, not a second 2*C*H*W read
    (evaluation/evalHpatch/evaluation.py:29-35 computes both).  ``out``: optional (2N,49,H,W) buffer; the two results are
    its halves (what the matchability head consumes as one batch)."""
    x, y = _dev(x, "corr x"), _dev(y, "corr y")
    if x.shape != y.shape:
        raise ValueError("corr_neigh_bidir: x %s and y %s differ" % (tuple(x.shape), tuple(y.shape)))
    N, C, H, W = x.shape
    if W % 4 != 0:
        Wp = (W + 3) // 4 * 4
        xp, yp = copy_cols(x.view(-1, W), Wp).view(N, C, H, Wp), copy_cols(y.view(-1, W), Wp).view(N, C, H, Wp)
        both = torch.empty((2 * N, K * K, H, Wp), dtype=torch.float32, device=x.device)
        corr_neigh_bidir(xp, yp, K, out=both)
        res = copy_cols(both.view(-1, Wp), W).view(2 * N, K * K, H, W)
        if out is not None:
            out.copy_(res)
            res = out
        return res[:N], res[N:]
    if out is None:
        out = torch.empty((2 * N, K * K, H, W), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (2 * N, K * K, H, W) or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("corr_neigh_bidir: out must be a contiguous float32 (2N,49,H,W) tensor")
    e0 = Profiler.begin(x)
    _call("rfx_corr_neigh_bidir_f32", _one_device(x, y, out), _p(x), _p(y), _p(out[:N]), _p(out[N:]), N, C, H, W, K)
    if e0 is not None:
        Profiler.active().corr_bidir.append(((2 * C + 2 * K * K) * 4.0 * N * H * W, 2 * (2 * C + K * K) * 4.0 * N * H * W, e0,
                                             Profiler.end(e0)))
        Profiler.active()._corr_order.append("bidir")
    return out[:N], out[N:]


def warp_grid(Hm, h, w):
    Hm = _dev(Hm, "homography")
    B = Hm.shape[0]
    grid = torch.empty((B, h, w, 2), dtype=torch.float32, device=Hm.device)
    _call("rfx_warp_grid_f32", Hm.device, _p(Hm), _p(grid), B, h, w)
    return grid


def grid_sample(inp, grid, align_corners=False):
    inp, grid = _dev(inp, "grid_sample input"), _dev(grid, "grid")
    N, C, Hi, Wi = inp.shape
    if grid.shape[0] != N or grid.shape[3] != 2:
        raise ValueError("grid must be (N,Ho,Wo,2)")
    Ho, Wo = grid.shape[1], grid.shape[2]
    out = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=inp.device)
    _call("rfx_grid_sample_f32", _one_device(inp, grid), _p(inp), _p(grid), _p(out), N, C, Hi, Wi, Ho, Wo,
                                              1 if align_corners else 0)
    return out


def compose_flow(flowDown, coarseGrid, clamp=False, want_inb=False, want_flow_up=False, out_hw=None):
    """flow12 = grid_sample(coarseGrid, [clamp](upsample(flowDown) + identity grid)).  ``out_hw``: output resolution when
    it differs from the coarse grid's (KITTI full-resolution pass); default = the coarse grid's."""
    flowDown, coarseGrid = _dev(flowDown, "flowDown"), _dev(coarseGrid, "coarse grid")
    N, two, hd, wd = flowDown.shape
    _, Hc, Wc, _ = coarseGrid.shape
    H, W = (Hc, Wc) if out_hw is None else (int(out_hw[0]), int(out_hw[1]))
    flow12 = torch.empty((N, H, W, 2), dtype=torch.float32, device=flowDown.device)
    inb = torch.empty((N, H, W), dtype=torch.float32, device=flowDown.device) if want_inb else None
    fup = torch.empty((N, H, W, 2), dtype=torch.float32, device=flowDown.device) if want_flow_up else None
    _call("rfx_compose_flow_f32", _one_device(flowDown, coarseGrid), _p(flowDown), _p(coarseGrid), _p(flow12), _p(inb), _p(fup),
          N, hd, wd, Hc, Wc, H, W, 1 if clamp else 0)
    return flow12, inb, fup


def flow_grad_clamp(flowCoarse, grid, want_grad=True):
    """model.predFlowCoarse's tail (model/model.py:333-340): -> (flowGrad (B,1,H-1,W-1) or None, clamp(flow^T + grid, -1, 1))."""
    f = _dev(flowCoarse, "flowCoarse")
    g = _dev(grid, "grid")
    B, C, H, W = f.shape
    if C != 2 or g.dim() != 4 or tuple(g.shape[1:]) != (H, W, 2) or g.shape[0] not in (1, B):
        raise ValueError("flowCoarse (B,2,H,W) / grid (1|B,H,W,2) expected, got %s / %s" % (tuple(f.shape), tuple(g.shape)))
    flow = torch.empty((B, H, W, 2), dtype=torch.float32, device=f.device)
    fg = torch.empty((B, 1, H - 1, W - 1), dtype=torch.float32, device=f.device) if want_grad else None
    _call("rfx_flow_grad_clamp_f32", _one_device(f, g), _p(f), _p(g), _p(fg), _p(flow), B, H, W, int(g.shape[0]))
    return fg, flow


def _match_plane(match12, n, H, W):
    if not isinstance(match12, torch.Tensor) or not match12.is_cuda or match12.dtype != torch.float32:
        raise RuntimeError("match12 must be a float32 tensor on a HIP device (no CPU fallback)")
    if tuple(match12.shape) != (n, H, W) or match12.stride(2) != 1 or match12.stride(1) != W:
        raise ValueError("match12 must be (n,H,W) with contiguous planes")
    return match12.stride(0) if n > 1 else H * W


def match_score(match12, cyc=None, inb=None):
    """match12 (n,H,W) (channel slice allowed) * cyc * inb -> (n,H,W)."""
    n, H, W = match12.shape
    stride = _match_plane(match12, n, H, W)
    c = _dev(cyc, "cyc") if cyc is not None else None
    b = _dev(inb, "inb") if inb is not None else None
    out = torch.empty((n, H, W), dtype=torch.float32, device=match12.device)
    _call("rfx_match_score_f32", _one_device(match12, c, b), match12.data_ptr(), stride, _p(c), _p(b), n, H * W, _p(out))
    return out


def merge_multi_h(flow, match12, th, multiH, cyc=None, inb=None):
    """flow (n,H,W,2); match12 (n,H,W) -- may be a channel slice of an (n,C,H,W) tensor (batch stride honoured);
    cyc / inb (n,H,W) contiguous or None  ->  flowGlobal (1,H,W,2), matchGlobal (1,H,W), binary (1,H,W) bool."""
    flow = _dev(flow, "flow")
    n, H, W, _ = flow.shape
    HW = H * W
    stride = _match_plane(match12, n, H, W)
    c = _dev(cyc, "cyc") if cyc is not None else None
    b = _dev(inb, "inb") if inb is not None else None
    fg = torch.empty((1, H, W, 2), dtype=torch.float32, device=flow.device)
    mg = torch.empty((1, H, W), dtype=torch.float32, device=flow.device)
    binary = torch.empty((1, H, W), dtype=torch.uint8, device=flow.device)
    _call("rfx_merge_multi_h_f32", _one_device(flow, match12, c, b), _p(flow), match12.data_ptr(), stride, _p(c),
                                                _p(b), n, HW, float(th), 1 if multiH else 0, _p(fg), _p(mg),
                                                _p(binary))
    return fg, mg, binary.bool()


def cc_max_area(size, cc_th):
    """Largest pixel count a with ``a / float(size) <= cc_th`` in float64 -- the reference's test
    (``np.mean(lab == i) <= cc_th`` / ``area <= cc_th`` on area fractions, evalKITTI/evaluation.py:96) as an integer bound."""
    a = int(cc_th * size)
    while (a + 1) / float(size) <= cc_th:
        a += 1
    while a > 0 and a / float(size) > cc_th:
        a -= 1
    return a


def remove_small_cc(match, cc_th, match_th=0.99):
    """evaluation/evalKITTI/evaluation.py:85-100 / getResults.py:66-84 on the device: zero every 8-connected component of
    (match > match_th) whose area fraction is <= cc_th.  match (N,H,W) or (H,W) float32; returns a new tensor."""
    m = _dev(match, "match map")
    if cc_th == 0:
        return m
    squeeze = m.dim() == 2
    m3 = (m[None] if squeeze else m).contiguous()
    N, H, W = m3.shape
    lib = _lib.load()
    ws = torch.empty(lib.rfx_remove_small_cc_ws_bytes(N, H, W), dtype=torch.uint8, device=m3.device)
    out = torch.empty_like(m3)
    _call("rfx_remove_small_cc_f32", _one_device(m3), _p(m3), _p(out), N, H, W, float(match_th), cc_max_area(H * W, float(cc_th)),
          _p(ws))
    return out[0] if squeeze else out


_HOST_KC = None


def host_sgemm_k_block(K=1024):
    """How the HOST's float32 matrix product -- torch.mm, the reference's score product (utils/outil.py:34) -- sums over k: MKL's
    sgemm accumulates an fma chain inside blocks of KC products and adds each block's sum to C; KC depends on the code path MKL
    picks for the CPU (192 on the GPU box's EPYC 9575F, 384 on a Xeon; MKL 2024.2; scripts/mm_blocking_probe.py).  Returns the
    KC whose float32 emulation reproduces torch.mm on a 384 x 256 post-ReLU, L2-normalised problem (MKL's small-matrix paths sum
    differently: the probe has to be large enough to take the blocked path the real 8 531 ... 25 747 x 1 200 ... 8 250 products
    take) bit for bit on all but <= 1e-4 of the elements (an edge kernel), or None when no candidate does -- another BLAS: the
    device then keeps its default chunking.  Pure host arithmetic, ~0.3 s per candidate, cached."""
    global _HOST_KC
    if _HOST_KC is not None:
        return _HOST_KC or None
    import numpy as np
    g = torch.Generator().manual_seed(7)
    A = torch.relu(torch.randn(K, 384, generator=g))
    B = torch.relu(torch.randn(K, 256, generator=g))
    A, B = A / A.norm(dim=0, keepdim=True), B / B.norm(dim=0, keepdim=True)
    mm = (A.t() @ B).numpy()
    a, b = A.numpy().astype(np.float64), B.numpy().astype(np.float64)
    found = 0
    for kc in (192, 384, 256, 128, 512, K):
        tot = np.zeros(mm.shape, np.float32)
        for k0 in range(0, K, kc):
            acc = np.zeros(mm.shape, np.float32)
            for k in range(k0, min(K, k0 + kc)):
                acc = (acc.astype(np.float64) + np.outer(a[k], b[k])).astype(np.float32)      # fma: exact product, one rounding
            tot = (tot.astype(np.float64) + acc).astype(np.float32)
        if float((tot != mm).mean()) <= 1e-4:
            found = kc
            break
    _HOST_KC = found
    return found or None


DEFAULT_SCORE_CHUNK = 256        # products per accumulation chunk of a mutual-NN score when nothing else is asked for


def resolve_score_chunk(spec=None):
    """The score chunk of ONE pipeline / drop-in module -> (products per chunk, where the value came from).  ``spec``:
      None      -> the environment variable RFX_SCORE_CHUNK if set (same grammar), else the fixed default of 256 products;
      an int    -> that many products per chunk (a positive multiple of 32); 0 = the library default of 256 (what 0 means in
                   rfx_api.h and ops.mutual_nn, one convention in every layer); < 0: ONE fma chain over all products (-1);
      "default" -> 256;
      "host"    -> the K blocking of THIS host's sgemm (host_sgemm_k_block: ~0.3-2 s of host arithmetic, cached per process), which
                   makes the device's scores equal the reference's torch.mm on this host bit for bit (the parity modes: drop-in
                   modules, parity sweeps, bench.py); 256 when no candidate reproduces the host's torch.mm (another BLAS).
    Called where a pipeline is CONSTRUCTED, never inside a launch: the value is an explicit argument of every
    rfx_mutual_nn*_f32 call (ABI 8) and travels with the results (AlignPipeline.score_chunk / .score_chunk_source)."""
    src = "argument"
    if spec is None:
        spec, src = os.environ.get("RFX_SCORE_CHUNK"), "RFX_SCORE_CHUNK"
        if spec is None or spec == "":
            return DEFAULT_SCORE_CHUNK, "default (fixed 256 products)"
    if isinstance(spec, str):
        if spec == "default":
            return DEFAULT_SCORE_CHUNK, "default (fixed 256 products)"
        if spec == "host":
            kc = host_sgemm_k_block()
            if kc and kc % 32 == 0:
                return int(kc), "host probe (host_sgemm_k_block: this host's torch.mm sums k in blocks of %d)" % kc
            return DEFAULT_SCORE_CHUNK, "fallback (no K blocking reproduced this host's torch.mm: fixed 256 products)"
        spec = int(spec)
    spec = int(spec)
    if spec == 0:
        return DEFAULT_SCORE_CHUNK, "%s = 0 (the library default: fixed 256 products)" % src
    if spec < 0:
        return -1, "%s (one chain)" % src
    if spec % 32:
        raise ValueError("score_chunk must be a multiple of 32 products, got %d" % spec)
    return spec, src


def mutual_nn(featA, featB, maskB=None, ldA=None, ldB=None, nA=None, nB=None, score_chunk=0):
    """featA (C,nA), featB (C,nB) -> (index1, index2) int64 device tensors (ascending index1).
    ``score_chunk``: products per accumulation chunk of a score (rfx_api.h; 0 = the library default of 256, < 0 = one chain).
    Synchronises once to read the match count (the reference's nonzero() does the same)."""
    featA, featB = _dev(featA, "featA"), _dev(featB, "featB")
    C = featA.shape[0]
    nA = nA or featA.shape[1]
    nB = nB or featB.shape[1]
    ldA = ldA or featA.stride(0)
    ldB = ldB or featB.stride(0)
    lib = _lib.load()
    ws = torch.empty(lib.rfx_mutual_nn_ws_bytes(nA, nB), dtype=torch.uint8, device=featA.device)
    cap = min(nA, nB)
    idx1 = torch.empty(cap, dtype=torch.int64, device=featA.device)
    idx2 = torch.empty(cap, dtype=torch.int64, device=featA.device)
    count = torch.zeros(1, dtype=torch.int32, device=featA.device)
    m = _dev(maskB, "maskB") if maskB is not None else None
    _call("rfx_mutual_nn_f32", _one_device(featA, featB, m), _p(featA), ldA, nA, _p(featB), ldB, nB, C, _p(m), _p(idx1), _p(idx2), _p(count),
                                     _p(ws), int(score_chunk))
    n = int(count.item())
    return idx1[:n], idx2[:n]


def lapack_dlt(X, Y):
    """The reference's own solve (utils/outil.py:68-87: float32 products stored into a float64 8x9 system, numpy's LAPACK SVD,
    Vh[8], float32) for the FEW hypotheses the device flags as rank deficient.  For those systems the null space is two-
    dimensional and "the reference's value" is by definition whatever LAPACK returns on THIS host (it differs between LAPACK
    builds): no device restatement can pin it, the host's own LAPACK does.  X, Y: (k,4,3) float32 numpy -> (k,3,3) float32."""
    import numpy as np
    from . import _lapack
    # the system is built and solved by ONE source text (rfx/_lapack.py: float32 products stored into a float64 8x9 matrix, numpy's
    # full SVD, row 8 of Vh) -- in this process for small batches, dealt to worker processes for large ones (numpy's batched SVD is
    # a serial loop of one dgesdd per system): per system the same LAPACK call on the same data, the same bits
    return _lapack.null_vectors(X, Y).reshape(X.shape[0], 3, 3).astype(np.float32)


def dlt4_homography(X, Y, degenerate="device", info=None):
    """outil.Homography.  ``degenerate="lapack"``: systems flagged rank deficient (csrc/dlt.h) are re-solved with the host's
    LAPACK (lapack_dlt) -- one sync; ``info`` (a dict) receives n_degenerate."""
    X, Y = _dev(X, "X"), _dev(Y, "Y")
    N = X.shape[0]
    H = torch.empty((N, 3, 3), dtype=torch.float32, device=X.device)
    if degenerate != "lapack":
        _call("rfx_dlt4_homography", _one_device(X, Y), _p(X), _p(Y), N, _p(H))
        return H
    fl = torch.empty(N, dtype=torch.uint8, device=X.device)
    _call("rfx_dlt4_homography_flags", _one_device(X, Y), _p(X), _p(Y), N, _p(H), _p(fl))
    bad = ((fl & 4) != 0).nonzero()[:, 0]
    if info is not None:
        info["n_degenerate"] = int(bad.numel())
    if bad.numel():
        H[bad] = torch.from_numpy(lapack_dlt(X[bad].cpu().numpy(), Y[bad].cpu().numpy())).to(H.device)
    return H


def prediction(match1, match2, Hs):
    """outil.Prediction for match1/match2 (n,3) and Hs (N,3,3) -> (N,n) float32 errors."""
    match1, match2, Hs = _dev(match1, "match1"), _dev(match2, "match2"), _dev(Hs, "H21")
    n, N = match1.shape[0], Hs.shape[0]
    err = torch.empty((N, n), dtype=torch.float32, device=match1.device)
    _call("rfx_prediction_f32", _one_device(match1, match2, Hs), _p(match1), _p(match2), n, _p(Hs), N, _p(err))
    return err


def score_hypotheses(match1, match2, samples, tol, degenerate="device"):
    """outil.ScoreRANSAC -> (H (N,3,3), counts (N,) int64).  ``degenerate="lapack"``: rank-deficient samples get the host
    LAPACK's homography (lapack_dlt) and their counts are re-evaluated (rfx_prediction_f32 + torch.det on the host, like the
    reference on a CPU run)."""
    match1, match2 = _dev(match1, "match1"), _dev(match2, "match2")
    samples = _dev(samples, "samples", torch.int64)
    n, N = match1.shape[0], samples.shape[0]
    lib = _lib.load()
    H = torch.empty((N, 3, 3), dtype=torch.float32, device=match1.device)
    counts = torch.empty(N, dtype=torch.int64, device=match1.device)
    ws = torch.empty(lib.rfx_ransac_ws_bytes(n, N), dtype=torch.uint8, device=match1.device)
    fl = torch.empty(N, dtype=torch.uint8, device=match1.device) if degenerate == "lapack" else None   # flags of the SAME DLT pass
    _call("rfx_score_hypotheses", _one_device(match1, match2, samples), _p(match1), _p(match2), n, _p(samples), N, float(tol), _p(H), _p(counts),
          _p(fl), _p(ws))
    if fl is not None:
        bad = ((fl & 4) != 0).nonzero()[:, 0]
        if bad.numel():
            sb = samples[bad]
            Hp = torch.from_numpy(lapack_dlt(match1[sb].cpu().numpy(), match2[sb].cpu().numpy()))
            H[bad] = Hp.to(H.device)
            cnt = torch.cat([(prediction(match1, match2, H[bad[i:i + 4096]]) < tol).sum(dim=1) for i in range(0, bad.numel(), 4096)])
            counts[bad] = cnt * (torch.det(Hp) > 1e-6).long().to(cnt.device)
    return H, counts


def ransac_h4(match1, match2, samples, tol, degenerate="device", info=None):
    """Device RANSAC.  Returns (bestH (3,3) f32, inlier (n,) bool, result int32[4]) as device tensors
    (result = [status, best count, winning hypothesis index, #hypotheses after the duplicate filter]).
    ``degenerate="lapack"``: the two-stage form of ransac_h4_batched for this one pair."""
    match1, match2 = _dev(match1, "match1"), _dev(match2, "match2")
    samples = _dev(samples, "samples", torch.int64)
    n, N = match1.shape[0], samples.shape[0]
    if degenerate == "lapack" and n >= 4:
        nd = torch.tensor([n], dtype=torch.int32, device=match1.device)
        bH, inl, res = ransac_h4_batched(match1[None], match2[None], nd, samples[None], tol, degenerate="lapack", info=info)
        return bH[0], inl[0], res[0]
    lib = _lib.load()
    dev = match1.device
    bestH = torch.empty((3, 3), dtype=torch.float32, device=dev)
    inl = torch.empty(n, dtype=torch.uint8, device=dev)
    res = torch.empty(4, dtype=torch.int32, device=dev)
    ws = torch.empty(lib.rfx_ransac_ws_bytes(n, N), dtype=torch.uint8, device=dev)
    _call("rfx_ransac_h4", _one_device(match1, match2, samples), _p(match1), _p(match2), n, _p(samples), N, float(tol), _p(bestH), _p(inl), _p(res),
                                 _p(ws))
    return bestH, inl.bool(), res


def gather_matches(idx1, idx2, n, xa, ya, xb, yb):
    """idx1/idx2 (B,cap) int64, n (B,) int32 (device) -> match1, match2 (B,cap,3) float32 (rows >= n[b] are zero)."""
    idx1, idx2 = _dev(idx1, "idx1", torch.int64), _dev(idx2, "idx2", torch.int64)
    n = _dev(n, "n", torch.int32)
    B, cap = idx1.shape
    m1 = torch.empty((B, cap, 3), dtype=torch.float32, device=idx1.device)
    m2 = torch.empty((B, cap, 3), dtype=torch.float32, device=idx1.device)
    _call("rfx_gather_matches_f32", _one_device(idx1, idx2, n, xa, ya, xb, yb), _p(idx1), _p(idx2), _p(n), cap, _p(_dev(xa, "xa")), _p(_dev(ya, "ya")),
                                                 _p(_dev(xb, "xb")), _p(_dev(yb, "yb")), _p(m1), _p(m2), B)
    return m1, m2


def ransac_h4_batched(match1, match2, n, samples, tol, degenerate="device", info=None):
    """match1/match2 (B,cap,3), n (B,) int32 device, samples (B,N,4) int64 -> bestH (B,3,3), inlier (B,cap) bool,
    result (B,4) int32 [status (3 = fewer than 4 matches), count, winner, nUnique] -- all device tensors.
    ``degenerate``: what to do with 4-point samples whose 8x9 DLT system is rank deficient (three matched points collinear in
    both images; ~1 % of the draws on the cell lattices).  "device" (default, no host round trip): the Householder sweep's
    own null vector -- a valid unit vector of the 2-D null space, but not the one the host LAPACK's rounding noise picks.
    "lapack": the search runs in two stages around ONE sync: the flagged hypotheses (rfx_ransac_degenerate_list) are re-solved
    with the host's LAPACK (lapack_dlt: utils/outil.py:68-87 itself) and patched in before counting, so that inlier counts,
    the winner and its inlier indices equal a CPU run of the reference on this host bit for bit even when such a sample wins
    (late multi-homography rounds with a handful of matches left).  ``info`` (dict) receives n_degenerate per pair."""
    match1, match2 = _dev(match1, "match1"), _dev(match2, "match2")
    n = _dev(n, "n", torch.int32)
    samples = _dev(samples, "samples", torch.int64)
    B, cap, _ = match1.shape
    N = samples.shape[1]
    if samples.shape[0] != B or n.shape[0] != B:
        raise ValueError("batch sizes differ")
    lib = _lib.load()
    dev = match1.device
    bestH = torch.empty((B, 3, 3), dtype=torch.float32, device=dev)
    inl = torch.empty((B, cap), dtype=torch.uint8, device=dev)
    res = torch.empty((B, 4), dtype=torch.int32, device=dev)
    ws = torch.empty(lib.rfx_ransac_batched_ws_bytes(cap, N, B), dtype=torch.uint8, device=dev)
    odev = _one_device(match1, match2, n, samples)
    if degenerate != "lapack":
        _call("rfx_ransac_h4_batched", odev, _p(match1), _p(match2), _p(n), cap, _p(samples), N, float(tol), _p(bestH), _p(inl),
              _p(res), _p(ws), B)
        return bestH, inl.bool(), res
    return ransac_h4_batched_finish(ransac_h4_batched_begin(match1, match2, n, samples, tol, _out=(bestH, inl, res, ws)), info=info)


# ---- the exact mode's search, split at its one host wait so that a caller can put other work between the halves -------------

_PINNED = {}           # (device index, stream) -> dict(off, xy, H): pinned exchange buffers of the exact mode, LRU-bounded


def _exact_buffers(dev, B, rows):
    """Pinned host buffers of one (device, stream): off (B+1) int32, xy (rows,16) f32, H (rows,9) f32 -- written by the gather kernels /
    read by the patch kernel in place (device-visible, coherent memory), grown on demand, at most 8 sets kept."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ent = _PINNED.pop(key, None)
    if ent is None or ent["off"].numel() < B + 1 or ent["xy"].shape[0] < rows:
        rows = max(rows, ent["xy"].shape[0] if ent is not None else 0)
        nb = max(B + 1, ent["off"].numel() if ent is not None else 0)
        ent = dict(off=torch.zeros(nb, dtype=torch.int32).pin_memory(), xy=torch.empty((rows, 16), dtype=torch.float32).pin_memory(),
                   H=torch.empty((rows, 9), dtype=torch.float32).pin_memory())
    _PINNED[key] = ent                       # most recently used last
    while len(_PINNED) > 8:
        _PINNED.pop(next(iter(_PINNED)))
    return ent


class _ExactSearch:
    __slots__ = ("odev", "args", "bestH", "inl", "res", "ws", "idx", "cnt", "off", "buf", "event", "B", "N", "cap", "gather_args", "t_begin")


def ransac_h4_batched_begin(match1, match2, n, samples, tol, _out=None):
    """First half of ransac_h4_batched(degenerate="lapack"): stage 1 (pack + DLT with rank flags) and rfx_ransac_degenerate_gather --
    the flagged samples of all pairs, one 64-byte row each, written by the device straight into pinned host memory -- then an event
    on the launch stream.  Nothing waits here: the caller may enqueue more device work (or run another batch's host work) before
    ransac_h4_batched_finish."""
    match1, match2 = _dev(match1, "match1"), _dev(match2, "match2")
    n = _dev(n, "n", torch.int32)
    samples = _dev(samples, "samples", torch.int64)
    B, cap, _ = match1.shape
    N = samples.shape[1]
    if samples.shape[0] != B or n.shape[0] != B:
        raise ValueError("batch sizes differ")
    from . import _lapack
    _lapack.load()
    lib = _lib.load()
    dev = match1.device
    c = _ExactSearch()
    if _out is None:
        c.bestH = torch.empty((B, 3, 3), dtype=torch.float32, device=dev)
        c.inl = torch.empty((B, cap), dtype=torch.uint8, device=dev)
        c.res = torch.empty((B, 4), dtype=torch.int32, device=dev)
        c.ws = torch.empty(lib.rfx_ransac_batched_ws_bytes(cap, N, B), dtype=torch.uint8, device=dev)
    else:
        c.bestH, c.inl, c.res, c.ws = _out
    c.odev = _one_device(match1, match2, n, samples)
    c.B, c.N, c.cap = B, N, cap
    c.args = (_p(match1), _p(match2), _p(n), cap, _p(samples), N, float(tol), _p(c.bestH), _p(c.inl), _p(c.res), _p(c.ws), B)
    c.gather_args = (match1, match2, n, samples)                           # kept alive until finish
    _call("rfx_ransac_h4_batched_stage", c.odev, *c.args, 1)
    c.idx = torch.empty((B, N), dtype=torch.int32, device=dev)
    c.cnt = torch.empty(B, dtype=torch.int32, device=dev)
    c.off = torch.empty(B + 1, dtype=torch.int32, device=dev)
    c.buf = _exact_buffers(dev, B, min(B * N, max(1 << 16, B * N // 8)))
    _exact_gather(c)
    return c


def _exact_gather(c):
    m1, m2, n, smp = c.gather_args
    _call("rfx_ransac_degenerate_gather", c.odev, _p(c.ws), _p(m1), _p(m2), _p(n), c.cap, _p(smp), c.N, c.B, _p(c.idx), _p(c.cnt),
          _p(c.off), _p(c.buf["off"]), _p(c.buf["xy"]), int(c.buf["xy"].shape[0]))
    c.event = torch.cuda.Event()
    c.event.record(torch.cuda.current_stream(c.odev))


def ransac_h4_batched_finish(c, info=None):
    """Second half: wait for the gather (the exact mode's ONE host wait), re-solve the flagged systems with the host's LAPACK
    (rfx/_lapack.py -> librfxhost.so: numpy's own dgesdd on std::threads, utils/outil.py:68-87), patch them in
    (rfx_ransac_patch_h_rows reads the pinned result rows in place), stage 2 (count + select).  Must run under the stream
    ransac_h4_batched_begin ran under.  ``info`` (dict) receives n_degenerate per pair, n_solved, host_ms."""
    import time
    from . import _lapack
    c.event.synchronize()
    B = c.B
    off = c.buf["off"].numpy()
    total = int(off[B])
    if total > c.buf["xy"].shape[0]:                                          # more flagged samples than rows: grow, gather again
        c.buf = _exact_buffers(c.odev, B, total)
        _exact_gather(c)
        c.event.synchronize()
        off = c.buf["off"].numpy()
    t0 = time.perf_counter()
    n_solved = 0
    if total > 0:
        _, n_solved = _lapack.solve_rows(c.buf["xy"].numpy()[:total], out=c.buf["H"].numpy())
        _call("rfx_ransac_patch_h_rows", c.odev, _p(c.ws), c.cap, c.N, B, _p(c.idx), _p(c.cnt), _p(c.off), _p(c.buf["H"]),
              int(c.buf["H"].shape[0]))
    if info is not None:
        info["n_degenerate"] = (off[1:B + 1] - off[:B]).tolist()
        info["n_solved"] = n_solved
        info["host_ms"] = (time.perf_counter() - t0) * 1e3
    _call("rfx_ransac_h4_batched_stage", c.odev, *c.args, 2)
    c.gather_args = None
    return c.bestH, c.inl.bool(), c.res


# ------------------------------------------------------------------------------------------------ multi-homography rounds (multih.hip)

def draw_samples(n, nb_iter, seed, stream_id, pair_ids=None):
    """RANSAC index draw on the device (utils/outil.py:120 on a GPU run): n (B,) int32 device match counts ->
    samples (B, nb_iter, 4) int64, Philox4x32-10 keyed by (seed, stream_id, pair id, hypothesis) modulo n[b]; pair id =
    pair_ids[b] ((B,) int32 device tensor: the caller's absolute ids) or b.  No host sync."""
    n = _dev(n, "n", torch.int32)
    B = n.shape[0]
    ids = _dev(pair_ids, "pair_ids", torch.int32) if pair_ids is not None else None
    if ids is not None and ids.numel() != B:
        raise ValueError("pair_ids must hold one id per count")
    smp = torch.empty((B, int(nb_iter), 4), dtype=torch.int64, device=n.device)
    _call("rfx_draw_samples_i64", _one_device(n, ids), _p(n), _p(smp), int(nb_iter), B, int(seed) & (2 ** 64 - 1),
          int(stream_id) & (2 ** 64 - 1), _p(ids))
    return smp


def filter_matches(idx1, idx2, count, active, mask, bg, rt, ct, xa, ya, xb, yb, want_kept=False):
    """The cached matches of the active pairs that lie outside the explained region, compacted in order (one launch):
    -> match1, match2 (a,cap,3) float32, n (a,) int32[, kept (a,cap) int32].  ``active``: (a,) int32 device tensor or None."""
    idx1, idx2 = _dev(idx1, "idx1", torch.int64), _dev(idx2, "idx2", torch.int64)
    count = _dev(count, "count", torch.int32)
    mask = _dev(mask, "mask")
    bg = _dev(bg, "bg") if bg is not None else None
    act = _dev(active, "active", torch.int32) if active is not None else None
    B, cap = idx1.shape
    a = B if act is None else act.shape[0]
    h, w = mask.shape[1], mask.shape[2]
    dev = idx1.device
    m1 = torch.empty((a, cap, 3), dtype=torch.float32, device=dev)
    m2 = torch.empty((a, cap, 3), dtype=torch.float32, device=dev)
    n = torch.empty(a, dtype=torch.int32, device=dev)
    kept = torch.empty((a, cap), dtype=torch.int32, device=dev) if want_kept else None
    _call("rfx_filter_matches_f32", _one_device(idx1, idx2, count, act, mask, bg, xa, ya, xb, yb), _p(idx1), _p(idx2), _p(count), cap,
          _p(act), a, _p(mask), _p(bg), h, w, int(rt), int(ct), _p(_dev(xa, "xa")), _p(_dev(ya, "ya")), _p(_dev(xb, "xb")),
          _p(_dev(yb, "yb")), _p(m1), _p(m2), _p(n), _p(kept))
    return (m1, m2, n, kept) if want_kept else (m1, m2, n)


def keep_mask(mask, bg, active, rt, ct, n=None, hw=None, device=None):
    """The (a, rt*ct) 0/1 keep map of variant A / C's getCoarse for the active pairs (rfx_keep_mask_f32): what the per-call
    mutual matching takes as its column mask.  ``mask`` (B,h,w) float32 or None (nothing explained yet: a zero mask is used when
    a background map ``bg`` is given; without ``bg`` either, ``n`` = number of pairs, ``hw`` = (h, w) and ``device`` are
    required)."""
    bgd = _dev(bg, "bg") if bg is not None else None
    act = _dev(active, "active", torch.int32) if active is not None else None
    if mask is None:
        if bgd is None and (n is None or hw is None or device is None):
            raise ValueError("keep_mask without a mask and without a background map needs n, hw = (h, w) and device")
        dev_ = bgd.device if bgd is not None else torch.device(device)
        h, w = (bgd.shape[1], bgd.shape[2]) if bgd is not None else (int(hw[0]), int(hw[1]))
        B = bgd.shape[0] if bgd is not None else int(n)
        m = torch.zeros((B, h, w), dtype=torch.float32, device=dev_) if bgd is not None else None
    else:
        m = _dev(mask, "mask")
        B, h, w = m.shape
        dev_ = m.device
    a = B if act is None else act.shape[0]
    keep = torch.empty((a, int(rt) * int(ct)), dtype=torch.float32, device=dev_)
    _call("rfx_keep_mask_f32", _one_device(keep, m, bgd, act), _p(m), _p(bgd), _p(act), a, h, w, int(rt), int(ct), _p(keep))
    return keep


class MultiHRecords:
    """The fixed-size per-pair result records of the multi-homography drivers (what
    evaluation/evalHpatch/evaluation.py:254-260 saves per pair), one float32 row per pair so that ONE all_gather moves them:
    [0] nbH (<= max_h) | [1] status (0 ok, 1 no homography, 3 more homographies accepted than the record holds, 4 stopped by the KITTI driver at the record's capacity -- the reference's ``while True`` has no limit) | H (max_h,9) | flowDown8 (max_h,2,h8,w8) | matchDown8 (max_h,2,h8,w8)
    [| flowD2 (max_h,2,hd2,wd2): the half-resolution /8 flow of the KITTI driver].  Filled on the device by multih_accept."""

    def __init__(self, B, h8, w8, device, max_h=11, hd2=0, wd2=0):
        self.B, self.h8, self.w8, self.hd2, self.wd2, self.max_h = B, h8, w8, hd2, wd2, max_h
        r4 = lambda x: (x + 3) // 4 * 4
        self.off_H = 4
        self.off_flow = self.off_H + r4(9 * max_h)
        self.off_match = self.off_flow + 2 * h8 * w8 * max_h
        self.off_d2 = self.off_match + 2 * h8 * w8 * max_h
        self.width = r4(self.off_d2 + 2 * hd2 * wd2 * max_h)
        self.rec = torch.zeros((B, self.width), dtype=torch.float32, device=device)
        self.rec[:, 1] = 1.0

    def rows(self, lo, hi):
        """The records of pairs [lo, hi) as a MultiHRecords of their own over the SAME storage (a lock-step group of the batch)."""
        import copy
        r = copy.copy(self)
        r.B, r.rec = hi - lo, self.rec[lo:hi]
        return r

    def views(self):
        """(nbH (B,), status (B,), H (B,max_h,3,3), flowDown8 (B,max_h,2,h8,w8), matchDown8 (B,max_h,2,h8,w8), flowD2 or None)."""
        r, B, m = self.rec, self.B, self.max_h
        n8 = 2 * self.h8 * self.w8
        d2 = r[:, self.off_d2:self.off_d2 + 2 * self.hd2 * self.wd2 * m].view(B, m, 2, self.hd2, self.wd2) if self.hd2 else None
        return (r[:, 0], r[:, 1], r[:, self.off_H:self.off_H + 9 * m].view(B, m, 3, 3),
                r[:, self.off_flow:self.off_flow + n8 * m].view(B, m, 2, self.h8, self.w8),
                r[:, self.off_match:self.off_match + n8 * m].view(B, m, 2, self.h8, self.w8), d2)


def multih_accept(match, mask, bg, active, ransac_result, n_match, nbH, th, mode, bestH=None, flowDown8=None, match12Down8=None,
                  match21Down8=None, flowD2=None, records=None):
    """Accept rule + mask update + record store of one round (rfx_multih_accept_f32) -> (accept (a,) int32, gain (a,) f32),
    both on the device; ``mask`` (B,h,w) and ``nbH`` (B,) int32 are updated in place."""
    for t, name in ((mask, "mask"), (nbH, "nbH")):          # updated IN PLACE: a silent .contiguous() copy would lose the update
        if isinstance(t, torch.Tensor) and not t.is_contiguous():
            raise ValueError("%s must be contiguous (updated in place)" % name)
    match, mask = _dev(match, "match"), _dev(mask, "mask")
    bg = _dev(bg, "bg") if bg is not None else None
    act = _dev(active, "active", torch.int32) if active is not None else None
    res, n_match, nbH = _dev(ransac_result, "ransac result", torch.int32), _dev(n_match, "n", torch.int32), _dev(nbH, "nbH", torch.int32)
    a = match.shape[0]
    h, w = mask.shape[1], mask.shape[2]
    if match.numel() != a * h * w:
        raise ValueError("match must be (a,h,w) / (a,1,h,w) at the mask's resolution")
    dev = match.device
    accept = torch.empty(a, dtype=torch.int32, device=dev)
    gain = torch.empty(a, dtype=torch.float32, device=dev)
    lib = _lib.load()
    ws = torch.empty(lib.rfx_multih_accept_ws_bytes(a), dtype=torch.uint8, device=dev)
    h8 = w8 = hd2 = wd2 = 0
    f8 = m12 = m21 = fd2 = None
    if flowDown8 is not None:
        f8 = _dev(flowDown8, "flowDown8")
        h8, w8 = f8.shape[2], f8.shape[3]
        m12, m21 = _dev(match12Down8, "match12Down8"), _dev(match21Down8, "match21Down8")
    if flowD2 is not None:
        fd2 = _dev(flowD2, "flowD2")
        hd2, wd2 = fd2.shape[2], fd2.shape[3]
    R = records
    if R is not None and (R.h8, R.w8, R.hd2, R.wd2) != (h8, w8, hd2, wd2):
        raise ValueError("record layout does not match the round's tensors")
    bh = _dev(bestH, "bestH") if bestH is not None else None
    _call("rfx_multih_accept_f32", _one_device(match, mask, bg, act, res, n_match, nbH, bh, f8, m12, m21, fd2), _p(match), _p(mask), _p(bg),
          _p(act), a, h, w, _p(res), _p(n_match), _p(nbH), float(th), int(mode), _p(accept), _p(gain), _p(ws), _p(bh), _p(f8), _p(m12),
          _p(m21), h8, w8, _p(fd2), hd2, wd2, _p(R.rec) if R is not None else ctypes.c_void_p(0), R.width if R is not None else 0,
          R.max_h if R is not None else 0, R.off_H if R is not None else 0, R.off_flow if R is not None else 0,
          R.off_match if R is not None else 0, R.off_d2 if R is not None else 0)
    return accept, gain
