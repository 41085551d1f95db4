"""Synthetic smooth-texture image pairs (SURVEY.md 8d "Config N -> concrete"): a random low-resolution
colour field is bicubic-upsampled; source and target are two crops offset by (12, 8) pixels, optionally
with a seeded random homography applied to the target crop window."""
import numpy as np
import PIL.Image as Image


def make_pair(H, W, seed=0):
    """Returns (I1, I2) PIL RGB images of size W x H."""
    rng = np.random.RandomState(seed)
    base = (rng.rand(H // 8 + 4, W // 8 + 4, 3) * 255).astype(np.uint8)
    big = Image.fromarray(base).resize((W + 32, H + 32), resample=Image.BICUBIC)
    I1 = big.crop((0, 0, W, H))
    I2 = big.crop((12, 8, W + 12, H + 8))
    return I1, I2


def assembly_arrays(seed, n, hd, wd):
    """Seeded stand-ins for the arrays the evaluation scripts save for one pair: fine flow (n,2,hd,wd),
    half-resolution flow (n,2,hd/2,wd/2) (KITTI), homographies (n,3,3), matchability (n,2,hd,wd) in [0,1]."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    flowDown = (torch.randn(n, 2, hd, wd, generator=g) * 0.04).numpy()
    flowd2 = (torch.randn(n, 2, hd // 2, wd // 2, generator=g) * 0.05).numpy()
    param = (torch.eye(3)[None].repeat(n, 1, 1) + 0.04 * torch.randn(n, 3, 3, generator=g)).numpy().astype(np.float32)
    # smooth matchability so that thresholded maps have connected blobs of several sizes
    md = F.avg_pool2d(torch.rand(n, 2, hd + 2, wd + 2, generator=g), 3, 1).numpy().astype(np.float32) * 1.9
    return flowDown, flowd2, param, np.clip(md, 0, 1)
