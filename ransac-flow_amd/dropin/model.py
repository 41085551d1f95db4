"""Drop-in for the reference's ``model/model.py``: the same four ``nn.Module`` classes (constructor
arguments, parameter / buffer names, so reference checkpoints ``load_state_dict`` unchanged -- the dict of
four state dicts written by train/train.py:293-297 and read by quick_start/align2images.py:47-50) and the
``predFlowCoarse`` / ``predFlowCoarseNoGrad`` / ``predMatchability`` helpers, with the forward pass executed by
librfx HIP kernels.

Forward-only: ``.eval()`` mode on a HIP device.  ``.train()``-mode forward (the backward pass, SSIM loss)
is out of the north-star scope and raises.  The kernel plans (packed weights, folded BatchNorm) are built
lazily from the module's own parameters and rebuilt whenever those change (load_state_dict, .cuda(), in-place
updates).
"""
import torch
import torch.nn as nn

from rfx import nets, ops


def _bn(c):
    return nn.BatchNorm2d(c, eps=1e-05)


def _conv(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False)


class Downsample(nn.Module):
    """Parameter container for the anti-aliased down-sampler (model/downsample.py:12-46): buffer ``filt``."""

    def __init__(self, channels, stride=2):
        super().__init__()
        a = torch.tensor([1.0, 2.0, 1.0])
        f = a[:, None] * a[None, :]
        self.stride = stride
        self.register_buffer("filt", (f / f.sum())[None, None].repeat(channels, 1, 1, 1))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = _conv(inplanes, planes, 3, stride), _bn(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2, self.bn2 = _conv(planes, planes, 3), _bn(planes)
        self.downsample, self.stride = downsample, stride


def _kaiming_init(module):
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


class _HipModule(nn.Module):
    """Shared plumbing: lazily (re)build the rfx kernel plan from the current parameters."""

    _plan = None
    _plan_key = None

    def _key(self):
        ts = list(self.parameters()) + list(self.buffers())
        return tuple((t.data_ptr(), t._version, str(t.device)) for t in ts)

    def _get_plan(self):
        key = self._key()
        if self._plan is None or key != self._plan_key:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("%s: parameters are on %s; call .cuda() first (the HIP path has no CPU fallback)"
                                   % (type(self).__name__, dev))
            self._plan = self._build(self.state_dict(), dev)
            self._plan_key = key
        return self._plan

    def _check_eval(self):
        if self.training:
            raise NotImplementedError("%s: training-mode forward/backward is outside the MI355X hot path; call .eval()"
                                      % type(self).__name__)


class FeatureExtractor(_HipModule):
    """model/model.py:59-125."""

    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.conv1, self.bn1 = _conv(3, 64, 3), _bn(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.Sequential(nn.MaxPool2d(kernel_size=2, stride=1), Downsample(64, 2))
        self.layer1 = self._make_layer(64, 2, 1)
        self.layer2 = self._make_layer(128, 2, 2)
        self.layer3 = self._make_layer(256, 2, 2)
        _kaiming_init(self)

    def _make_layer(self, planes, blocks, stride):
        ds = None
        if stride != 1 or self.inplanes != planes:
            mods = ([Downsample(self.inplanes, stride)] if stride != 1 else []) + \
                   [nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=1, bias=False), nn.BatchNorm2d(planes)]
            ds = nn.Sequential(*mods)
        layers = [BasicBlock(self.inplanes, planes, stride, ds)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def _build(self, sd, dev):
        return nets.FeatureExtractorNet(sd, dev)

    def forward(self, x):
        self._check_eval()
        with torch.no_grad():
            return self._get_plan()(x)


class CorrNeigh(nn.Module):
    """model/model.py:129-160; forward(x, y): y is the windowed operand (zero padded)."""

    def __init__(self, kernelSize):
        super().__init__()
        assert kernelSize % 2 == 1
        self.kernelSize = kernelSize
        self.paddingSize = kernelSize // 2

    def forward(self, x, y):
        with torch.no_grad():
            return ops.corr_neigh(x, y, self.kernelSize)


class _Head(_HipModule):
    def __init__(self, kernelSize, cout_last):
        super().__init__()
        k2 = kernelSize * kernelSize
        self.conv1, self.bn1 = _conv(k2, 512, 3), _bn(512)
        self.relu = nn.ReLU(inplace=True)
        self.conv2, self.bn2 = _conv(512, 256, 3), _bn(256)
        self.conv3, self.bn3 = _conv(256, 128, 3), _bn(128)
        self.conv4 = _conv(128, cout_last, 3)
        self.kernelSize = kernelSize
        self.paddingSize = kernelSize // 2
        _kaiming_init(self)


class NetFlowCoarse(_Head):
    """model/model.py:167-249.  ``gridX`` / ``gridY`` (tap offsets) are plain attributes as in the reference
    (:190-191, not part of the state dict); the fused softmax-expectation kernel regenerates them."""

    def __init__(self, kernelSize):
        assert kernelSize % 2 == 1
        super().__init__(kernelSize, kernelSize * kernelSize)
        off = torch.arange(-self.paddingSize, self.paddingSize + 1).float()
        self.gridY = off.view(-1, 1).expand(kernelSize, kernelSize).reshape(1, -1, 1, 1).clone()
        self.gridX = off.view(1, -1).expand(kernelSize, kernelSize).reshape(1, -1, 1, 1).clone()
        self.softmax = nn.Softmax(dim=1)

    def cuda(self, device=None):
        super().cuda(device)
        self.gridX, self.gridY = self.gridX.cuda(device), self.gridY.cuda(device)
        return self

    def _build(self, sd, dev):
        return nets.NetFlowCoarseNet(sd, self.kernelSize, dev)

    def forward(self, coef, up8X=True):
        self._check_eval()
        with torch.no_grad():
            return self._get_plan()(coef, up8X)


class NetMatchability(_Head):
    """model/model.py:254-322."""

    def __init__(self, kernelSize):
        super().__init__(kernelSize, 1)
        self.sigmoid = nn.Sigmoid()
        nn.init.normal_(self.conv4.weight, mean=0.0, std=0.0001)

    def _build(self, sd, dev):
        return nets.NetMatchabilityNet(sd, self.kernelSize, dev)

    def forward(self, feat, up8X=True):
        self._check_eval()
        with torch.no_grad():
            return self._get_plan()(feat, up8X)


def SSIM(*a, **k):
    raise NotImplementedError("SSIM is a training loss (model/model.py:327) and outside the hot path")


def predFlowCoarse(corrKernel21, NetFlowCoarse, grid, up8X=True):
    """model/model.py:331-340 -> (flowGrad (B,1,H-1,W-1), clamp(flow.permute(0,2,3,1) + grid, -1, 1))."""
    flowCoarse = NetFlowCoarse(corrKernel21, up8X)
    return ops.flow_grad_clamp(flowCoarse.contiguous(), grid.contiguous(), want_grad=True)      # rfx_flow_grad_clamp_f32


def predFlowCoarseNoGrad(corrKernel21, NetFlowCoarse, grid, up8X=True):
    """model/model.py:342-350."""
    flowCoarse = NetFlowCoarse(corrKernel21, up8X)
    return ops.flow_grad_clamp(flowCoarse.contiguous(), grid.contiguous(), want_grad=False)[1]


def predMatchability(corrKernel21, NetMatchability, up8X=True):
    """model/model.py:353-357."""
    return NetMatchability(corrKernel21, up8X)
