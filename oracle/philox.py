"""numpy Philox4x32-10 and the bits->float maps used by the HIP kernels (dposer_amd/csrc/rng.h).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  This is *our* counter-based RNG contract
(the reference draws from torch's CPU generator, which a GPU kernel cannot reproduce; SURVEY.md
§7 "RNG"), restated on the CPU so tests can feed the oracle the very numbers the kernels draw.

Counter / key contract (all uint32):
    counter = (index_lo, index_hi, stream, offset)      key = (seed_lo, seed_hi)
    index  : which group of random numbers inside one draw (see each kernel)
    stream : what the numbers are for (STREAM_* below)
    offset : optimisation / sampler step number
"""
from __future__ import annotations

import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)

STREAM_TRAIN_T = 1          # t ~ U(eps, T): one u32 per sample          (index = sample)
STREAM_TRAIN_Z = 2          # z ~ N(0, I): 4 normals per call            (index = flat_elem // 4)
STREAM_EM_NOISE = 3         # predictor noise                             (index = flat_elem // 4)
STREAM_IMPUTE_A = 4         # imputation noise after the corrector
STREAM_IMPUTE_B = 5         # imputation noise after the predictor
STREAM_LANGEVIN = 6         # corrector noise
STREAM_PRIOR = 7            # x_T ~ N(0, I) prior draw / prior-loss z
STREAM_DROPOUT0 = 16        # + dropout-site id (0..4): 8 x 16-bit lanes per call, Philox4x32-7


def philox4x32_10(c0, c1, c2, c3, k0, k1, rounds=10):
    """Vectorised Philox4x32-``rounds`` (10 = the standard generator; the dropout streams use 7, see dropout_keep_mask).
    All inputs broadcastable uint32 arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(a, dtype=np.uint32) for a in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    for _ in range(rounds):
        p0 = M0 * c0.astype(np.uint64)
        p1 = M1 * c2.astype(np.uint64)
        hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
        lo0 = p0.astype(np.uint32)
        hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
        lo1 = p1.astype(np.uint32)
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
        k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def u01_open_low(bits):
    """(0, 1]: ((bits >> 8) + 1) * 2^-24 -- safe argument for log()."""
    return ((bits >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(2.0 ** -24)


def u01(bits):
    """[0, 1): (bits >> 8) * 2^-24 -- same support as torch.rand for fp32."""
    return (bits >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def normals4(index, stream, offset, seed):
    """Four N(0,1) numbers per counter (Box-Muller on (r0,r1) and (r2,r3)); fp32.
    Returns array [..., 4]."""
    index = np.asarray(index, dtype=np.uint64)
    lo = (index & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (index >> np.uint64(32)).astype(np.uint32)
    r = philox4x32_10(lo, hi, np.uint32(stream), np.uint32(offset),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    out = []
    for a, b in ((r[0], r[1]), (r[2], r[3])):
        rad = np.sqrt(np.float32(-2.0) * np.log(u01_open_low(a)))
        ang = np.float32(2.0 * np.pi) * u01(b)
        out += [rad * np.cos(ang), rad * np.sin(ang)]
    return np.stack(out, axis=-1).astype(np.float32)


def normal_matrix(rows, cols, stream, offset, seed):
    """[rows, cols] N(0,1) matrix as the kernels draw it: one counter per (sample, quad of 4 channels),
    counter index = r * ceil(cols/4) + c // 4, lane c % 4 (dposer_amd/csrc/elementwise.hip)."""
    qd = (cols + 3) // 4
    idx = (np.arange(rows, dtype=np.uint64)[:, None] * np.uint64(qd) + np.arange(qd, dtype=np.uint64)[None, :])
    z = normals4(idx.reshape(-1), stream, offset, seed).reshape(rows, qd * 4)
    return z[:, :cols]


def uniform_t(rows, offset, seed, eps=1e-5, T=1.0):
    """t = u*(T-eps)+eps with u in [0,1) -- one counter per sample, lane 0."""
    idx = np.arange(rows, dtype=np.uint64)
    r = philox4x32_10((idx & np.uint64(0xFFFFFFFF)).astype(np.uint32),
                      (idx >> np.uint64(32)).astype(np.uint32),
                      np.uint32(STREAM_TRAIN_T), np.uint32(offset),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return u01(r[0]) * np.float32(T - eps) + np.float32(eps)


def dropout_keep_mask(rows, channels, site, offset, seed, p):
    """{0,1} keep mask [rows, channels] for dropout site ``site`` (0 = after pre_gnorm, 1.. = block
    layers in order) exactly as dposer_amd/csrc/epilogues.h dropout_mask16 draws it.  Within a
    GroupNorm group g of 32 channels, channel cl = 8q + 4hi + r belongs to lane-half ``hi`` and quad q;
    counter index = sample*(channels/8) + 4g + 2hi + q//2; the 8 16-bit lanes of the call are
    (low, high) halves of r0..r3; lane = (q%2)*4 + r.  keep <=> lane16 < floor((1-p)*65536)."""
    thr = np.uint32(int((1.0 - p) * 65536.0))
    s = np.arange(rows, dtype=np.uint64)[:, None]
    c = np.arange(channels, dtype=np.uint64)[None, :]
    g, cl = c // np.uint64(32), c % np.uint64(32)
    q, hi, r = cl // np.uint64(8), (cl % np.uint64(8)) // np.uint64(4), cl % np.uint64(4)
    idx = s * np.uint64(channels // 8) + g * np.uint64(4) + hi * np.uint64(2) + q // np.uint64(2)
    idx = np.broadcast_to(idx, (rows, channels)).reshape(-1)
    # 7 rounds for the dropout streams (dposer_amd/csrc/rng.h: philox_at_dropout; Crush-resistant per Salmon et al., and 30 % cheaper
    # in the training-forward epilogue, whose VALU time the draw dominates)
    w = philox4x32_10((idx & np.uint64(0xFFFFFFFF)).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32),
                      np.uint32(STREAM_DROPOUT0 + site), np.uint32(offset), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, rounds=7)
    w = np.stack(w, axis=-1)                                                  # [n, 4]
    lane = np.broadcast_to((q % np.uint64(2)) * np.uint64(4) + r, (rows, channels)).reshape(-1).astype(np.int64)
    word = np.take_along_axis(w, (lane // 2)[:, None], axis=1)[:, 0]
    bits = np.where(lane % 2 == 0, word & np.uint32(0xFFFF), word >> np.uint32(16))
    return (bits < thr).astype(np.float32).reshape(rows, channels)
