"""Synthetic smooth-texture image pairs (SURVEY.md 8d "Config N -> concrete"): a random low-resolution
colour field is bicubic-upsampled; source and target are two crops offset by (12, 8) pixels, optionally
(``homography=True``) with a seeded random homography applied to the target crop window."""
import numpy as np
import PIL.Image as Image


def random_homography(seed, amp=0.05):
    """identity + amp * U(-1, 1) on the 8 free parameters, in normalised [-1, 1] coordinates
    (row-major 3x3, h33 = 1).  Maps TARGET normalised coordinates to SOURCE normalised coordinates -- the direction
    of the reference's own homographies (outil.Prediction: est = Y H^T, utils/outil.py:97-100)."""
    rng = np.random.RandomState(1_000_003 + seed)
    Hn = np.eye(3)
    Hn.flat[:8] += amp * (rng.rand(8) * 2 - 1)
    return Hn


def make_pair(H, W, seed=0, homography=False, amp=0.05):
    """Returns (I1, I2) PIL RGB images of size W x H.  I1 = crop of a smooth random texture; I2 = the same texture
    seen through the window shifted by (12, 8) pixels and, with ``homography=True``, additionally warped by the
    seeded ``random_homography(seed, amp)`` (so that a multi-homography driver has perspective to explain)."""
    rng = np.random.RandomState(seed)
    base = (rng.rand(H // 8 + 4, W // 8 + 4, 3) * 255).astype(np.uint8)
    big = Image.fromarray(base).resize((W + 32, H + 32), resample=Image.BICUBIC)
    I1 = big.crop((0, 0, W, H))
    if not homography:
        return I1, big.crop((12, 8, W + 12, H + 8))
    # pixel-space map target pixel (x, y) -> texture pixel: normalise, apply Hn, de-normalise, shift by the crop offset
    Hn = random_homography(seed, amp)
    N = np.array([[2.0 / W, 0, -1.0], [0, 2.0 / H, -1.0], [0, 0, 1.0]])        # pixel -> normalised
    D = np.array([[W / 2.0, 0, W / 2.0 + 12], [0, H / 2.0, H / 2.0 + 8], [0, 0, 1.0]])   # normalised -> texture pixel
    P = D @ Hn @ N
    P = P / P[2, 2]
    I2 = big.transform((W, H), Image.PERSPECTIVE, tuple(P.flatten()[:8]), resample=Image.BICUBIC)
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
