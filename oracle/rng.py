"""Oracle: the counter-based generator behind the in-graph random draws (numpy, CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference draws its randomness with torch's global / per-call generators INSIDE the step
(models/diffusion_prior.py:337,349-351 - DDPM noise; train_diffusion_prior.py:449 -> models/diffusion_prior.py:445,453,
255-259 and the Dropout layers of BrainNetwork :62-75 - timesteps, q_sample noise, cond-drop masks, dropout masks).  A
captured hipGraph cannot call torch's generator, so the library carries its own: Philox-4x32-10 (Salmon et al., "Parallel
random numbers: as easy as 1, 2, 3", SC'11 - the generator family torch's CUDA backend also uses), keyed by a seed and
addressed by (step offset, subsequence, element), with the transforms below.  The NUMBERS differ from torch's for the
same seed (torch's offset bookkeeping is not part of any contract the reference relies on - its CPU and CUDA generators
already disagree); what is pinned here is the algorithm: `philox4x32_10` against the known-answer vectors of the
Random123 distribution (tests/test_oracle_golden.py), the device kernel against this file bit for bit (raw words, masks,
integers) or to float rounding (normals).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
KIND_RAW, KIND_NORMAL, KIND_KEEP_SCALED, KIND_BERNOULLI_U8, KIND_RANDINT_I32, KIND_UNIFORM = 0, 1, 2, 3, 4, 5


def philox4x32_10(ctr, key):
    """ctr (..., 4) uint32, key (..., 2) uint32 -> (..., 4) uint32."""
    c = [np.asarray(ctr[..., i], dtype=np.uint32).copy() for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint32).copy()
    k1 = np.asarray(key[..., 1], dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        for r in range(10):
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            if r < 9:
                k0 = k0 + W0
                k1 = k1 + W1
    return np.stack(c, -1)


def raw_words(seed, offset, subsequence, n_words):
    """The word stream of one fill: word i = word (i & 3) of block (i >> 2); counter = (block lo, block hi | subsequence << 16,
    offset lo, offset hi), key = (seed lo, seed hi)."""
    nb = (n_words + 3) // 4
    blk = np.arange(nb, dtype=np.uint64)
    ctr = np.empty((nb, 4), dtype=np.uint32)
    ctr[:, 0] = (blk & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = ((blk >> np.uint64(32)).astype(np.uint32)) | np.uint32((subsequence & 0xFFFF) << 16)
    ctr[:, 2] = np.uint32(offset & 0xFFFFFFFF)
    ctr[:, 3] = np.uint32((offset >> 32) & 0xFFFFFFFF)
    key = np.empty((nb, 2), dtype=np.uint32)
    key[:, 0] = np.uint32(seed & 0xFFFFFFFF)
    key[:, 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    return philox4x32_10(ctr, key).reshape(-1)[:n_words]


def uniform24(w):
    """[0, 1): the top 24 bits of a word."""
    return (w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def fill(seed, offset, subsequence, kind, n, param=0.0):
    if kind == KIND_RAW:
        return raw_words(seed, offset, subsequence, n)
    w = raw_words(seed, offset, subsequence, (n + 3) // 4 * 4)
    if kind == KIND_NORMAL:                       # Box-Muller on word pairs: (w0, w1) -> elements 0, 1; (w2, w3) -> 2, 3
        w = w.reshape(-1, 2)
        u1 = ((w[:, 0] >> np.uint32(8)).astype(np.float64) + 1.0) * 2.0 ** -24      # (0, 1]
        u2 = (w[:, 1] >> np.uint32(8)).astype(np.float64) * 2.0 ** -24               # [0, 1)
        r = np.sqrt(-2.0 * np.log(u1))
        out = np.stack([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)], -1).reshape(-1)
        return out[:n].astype(np.float32)
    u = uniform24(w)[:n]
    if kind == KIND_UNIFORM:
        return u
    if kind == KIND_KEEP_SCALED:                  # dropout keep mask, pre-scaled: (u >= p) / (1 - p)
        return np.where(u >= np.float32(param), np.float32(1.0) / (np.float32(1.0) - np.float32(param)), np.float32(0.0))
    if kind == KIND_BERNOULLI_U8:                 # prob_mask_like(p): u < p
        return (u < np.float32(param)).astype(np.uint8)
    if kind == KIND_RANDINT_I32:                  # uniform integer in [0, param)
        return ((w[:n].astype(np.uint64) * np.uint64(int(param))) >> np.uint64(32)).astype(np.int32)
    raise ValueError(kind)
