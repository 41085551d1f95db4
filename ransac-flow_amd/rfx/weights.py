"""Deterministic random-init state dicts with the reference's parameter names.

No checkpoints are reachable offline (model/pretrained/download_model.sh needs the network), so BASELINE
configs use random-init weights of the reference architectures.  The keys/shapes below are the ones
``load_state_dict`` of the reference modules expects (model/resnet50.py:107-169 trunk slice conv1..layer3,
model/model.py:59-104 FeatureExtractor, :167-203 NetFlowCoarse, :254-287 NetMatchability), so the same dict
drives the reference (oracle), the CPU restatement and the HIP nets.

Initialisation follows the reference: conv weights ~ N(0, sqrt(2/(k*k*Cout))) (kaiming fan_out,
model/resnet50.py:128-131, model/model.py:77-79), BN weight 1 / bias 0 / running stats (0,1);
NetMatchability.conv4 ~ N(0, 1e-4) (model/model.py:286).  ``randomize_bn=True`` perturbs the BN affine and
running statistics so that tests exercise the folded scale/shift path with non-trivial values.
"""
import math

import torch


def _conv(g, cout, cin, k, std=None):
    std = math.sqrt(2.0 / (k * k * cout)) if std is None else std
    return torch.randn(cout, cin, k, k, generator=g) * std


def _bn(sd, prefix, c, g, randomize):
    if randomize:
        sd[prefix + ".weight"] = 1.0 + 0.2 * (torch.rand(c, generator=g) - 0.5)
        sd[prefix + ".bias"] = 0.1 * torch.randn(c, generator=g)
        sd[prefix + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[prefix + ".running_var"] = 1.0 + 0.4 * (torch.rand(c, generator=g) - 0.5)
    else:
        sd[prefix + ".weight"] = torch.ones(c)
        sd[prefix + ".bias"] = torch.zeros(c)
        sd[prefix + ".running_mean"] = torch.zeros(c)
        sd[prefix + ".running_var"] = torch.ones(c)
    sd[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def resnet50_trunk_sd(seed=0, randomize_bn=False):
    """conv1, bn1, layer1 (3 Bottlenecks, 64), layer2 (4, 128, /2), layer3 (6, 256, /2)."""
    g = torch.Generator().manual_seed(seed)
    sd = {"conv1.weight": _conv(g, 64, 3, 7)}
    _bn(sd, "bn1", 64, g, randomize_bn)
    inplanes = 64
    for layer, planes, nblk in (("layer1", 64, 3), ("layer2", 128, 4), ("layer3", 256, 6)):
        for b in range(nblk):
            p = "%s.%d" % (layer, b)
            sd[p + ".conv1.weight"] = _conv(g, planes, inplanes, 1)
            _bn(sd, p + ".bn1", planes, g, randomize_bn)
            sd[p + ".conv2.weight"] = _conv(g, planes, planes, 3)
            _bn(sd, p + ".bn2", planes, g, randomize_bn)
            sd[p + ".conv3.weight"] = _conv(g, planes * 4, planes, 1)
            _bn(sd, p + ".bn3", planes * 4, g, randomize_bn)
            if b == 0:
                sd[p + ".downsample.0.weight"] = _conv(g, planes * 4, inplanes, 1)
                _bn(sd, p + ".downsample.1", planes * 4, g, randomize_bn)
            inplanes = planes * 4
    return sd


def _blur_filt(c):
    a = torch.tensor([1.0, 2.0, 1.0])
    f = a[:, None] * a[None, :]
    return (f / f.sum())[None, None].repeat(c, 1, 1, 1)


def feature_extractor_sd(seed=1, randomize_bn=False):
    g = torch.Generator().manual_seed(seed)
    sd = {"conv1.weight": _conv(g, 64, 3, 3)}
    _bn(sd, "bn1", 64, g, randomize_bn)
    sd["maxpool.1.filt"] = _blur_filt(64)
    inplanes = 64
    for layer, planes, stride in (("layer1", 64, 1), ("layer2", 128, 2), ("layer3", 256, 2)):
        for b in range(2):
            p = "%s.%d" % (layer, b)
            cin = inplanes if b == 0 else planes
            sd[p + ".conv1.weight"] = _conv(g, planes, cin, 3)
            _bn(sd, p + ".bn1", planes, g, randomize_bn)
            sd[p + ".conv2.weight"] = _conv(g, planes, planes, 3)
            _bn(sd, p + ".bn2", planes, g, randomize_bn)
            if b == 0 and stride != 1:
                sd[p + ".downsample.0.filt"] = _blur_filt(inplanes)
                sd[p + ".downsample.1.weight"] = _conv(g, planes, inplanes, 1)
                _bn(sd, p + ".downsample.2", planes, g, randomize_bn)
        inplanes = planes
    return sd


def _head_sd(g, cout_last, randomize_bn, last_std=None, k=7):
    sd = {"conv1.weight": _conv(g, 512, k * k, 3)}
    _bn(sd, "bn1", 512, g, randomize_bn)
    sd["conv2.weight"] = _conv(g, 256, 512, 3)
    _bn(sd, "bn2", 256, g, randomize_bn)
    sd["conv3.weight"] = _conv(g, 128, 256, 3)
    _bn(sd, "bn3", 128, g, randomize_bn)
    sd["conv4.weight"] = _conv(g, cout_last, 128, 3, std=last_std)
    return sd


def net_flow_coarse_sd(seed=2, randomize_bn=False, k=7):
    return _head_sd(torch.Generator().manual_seed(seed), k * k, randomize_bn, k=k)


def net_matchability_sd(seed=3, randomize_bn=False, k=7, last_std=1e-4):
    return _head_sd(torch.Generator().manual_seed(seed), 1, randomize_bn, last_std=last_std, k=k)


# ------------------------------------------------------------------------------------------------ sky segmentation

def seg_encoder_sd(seed=4, randomize_bn=False):
    """segNet/segModel.py:59-127,156-215: the ResNet-50 with the three-convolution stem (3->64 /2, 64->64, 64->128; inplanes 128)
    whose layer3 / layer4 ResnetDilated turns into dilated stride-1 layers -- the keys of ``net_encoder.load_state_dict``
    (conv1..bn3, layer1..layer4).  Kaiming fan_out initialisation as the reference's constructor (:82-88)."""
    g = torch.Generator().manual_seed(seed)
    sd = {"conv1.weight": _conv(g, 64, 3, 3)}
    _bn(sd, "bn1", 64, g, randomize_bn)
    sd["conv2.weight"] = _conv(g, 64, 64, 3)
    _bn(sd, "bn2", 64, g, randomize_bn)
    sd["conv3.weight"] = _conv(g, 128, 64, 3)
    _bn(sd, "bn3", 128, g, randomize_bn)
    inplanes = 128
    for layer, planes, nblk in (("layer1", 64, 3), ("layer2", 128, 4), ("layer3", 256, 6), ("layer4", 512, 3)):
        for b in range(nblk):
            p = "%s.%d" % (layer, b)
            sd[p + ".conv1.weight"] = _conv(g, planes, inplanes, 1)
            _bn(sd, p + ".bn1", planes, g, randomize_bn)
            sd[p + ".conv2.weight"] = _conv(g, planes, planes, 3)
            _bn(sd, p + ".bn2", planes, g, randomize_bn)
            sd[p + ".conv3.weight"] = _conv(g, planes * 4, planes, 1)
            _bn(sd, p + ".bn3", planes * 4, g, randomize_bn)
            if b == 0:
                sd[p + ".downsample.0.weight"] = _conv(g, planes * 4, inplanes, 1)
                _bn(sd, p + ".downsample.1", planes * 4, g, randomize_bn)
            inplanes = planes * 4
    return sd


def seg_decoder_sd(seed=5, randomize_bn=False, num_class=150, fc_dim=2048, logit_std=None):
    """segNet/segModel.py:218-248 PPMDeepsup: ppm.{0..3}.{1: conv 1x1 fc_dim->512, 2: BN}, conv_last.{0: conv 3x3 (fc_dim+4*512)->512,
    1: BN, 4: conv 1x1 512->num_class + bias}; the deep-supervision branch (cbr_deepsup, conv_last_deepsup) holds parameters the
    inference path never reads.  ``logit_std``: std of the classifier weights (default kaiming): larger values spread the 150 class
    scores so that a synthetic image gets a few distinct regions."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k in range(4):
        sd["ppm.%d.1.weight" % k] = _conv(g, 512, fc_dim, 1)
        _bn(sd, "ppm.%d.2" % k, 512, g, randomize_bn)
    sd["cbr_deepsup.0.weight"] = _conv(g, fc_dim // 4, fc_dim // 2, 3)
    _bn(sd, "cbr_deepsup.1", fc_dim // 4, g, randomize_bn)
    sd["conv_last.0.weight"] = _conv(g, 512, fc_dim + 4 * 512, 3)
    _bn(sd, "conv_last.1", 512, g, randomize_bn)
    sd["conv_last.4.weight"] = _conv(g, num_class, 512, 1, std=logit_std)
    sd["conv_last.4.bias"] = 0.1 * torch.randn(num_class, generator=g)
    sd["conv_last_deepsup.weight"] = _conv(g, num_class, fc_dim // 4, 1)
    sd["conv_last_deepsup.bias"] = torch.zeros(num_class)
    return sd
