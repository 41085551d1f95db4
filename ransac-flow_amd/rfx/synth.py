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
