"""TEST INFRASTRUCTURE -- numpy restatement of Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy
as 1, 2, 3", SC'11) and of rfx_draw_samples_i64's mapping (include/rfx_api.h): the device-side RANSAC index draw of
ransac-flow_amd/csrc/multih.hip is checked against this, and this against the published known-answer vectors of the
Random123 distribution (kat_vectors: philox4x32 10)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: (..., 4) uint32 array-like, key: (k0, k1) -> (..., 4) uint32."""
    c = np.asarray(ctr, dtype=np.uint64).copy()
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c[..., 0]
        p1 = M1 * c[..., 2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        n0 = hi1 ^ c[..., 1] ^ np.uint64(k0)
        n2 = hi0 ^ c[..., 3] ^ np.uint64(k1)
        c = np.stack((n0, lo1, n2, lo0), axis=-1)
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c.astype(np.uint32)


def draw_samples(n, nb_iter, seed, stream_id, pair_ids=None):
    """rfx_draw_samples_i64: samples[b,h,p] = Philox(ctr = (h, id(b), stream lo, stream hi), key = (seed lo, seed hi))[p] % n[b],
    id(b) = pair_ids[b] or b."""
    n = np.asarray(n, dtype=np.int64)
    B = len(n)
    ctr = np.zeros((B, nb_iter, 4), dtype=np.uint64)
    ctr[..., 0] = np.arange(nb_iter, dtype=np.uint64)[None, :]
    ctr[..., 1] = (np.arange(B, dtype=np.uint64) if pair_ids is None else np.asarray(pair_ids, dtype=np.uint64))[:, None]
    ctr[..., 2] = int(stream_id) & 0xFFFFFFFF
    ctr[..., 3] = (int(stream_id) >> 32) & 0xFFFFFFFF
    r = philox4x32_10(ctr, (int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)).astype(np.int64)
    out = np.zeros((B, nb_iter, 4), dtype=np.int64)
    for b in range(B):
        if n[b] > 0:
            out[b] = r[b] % n[b]
    return out
