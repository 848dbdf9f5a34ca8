"""Philox4x32-10 counter RNG (Salmon et al., SC'11), numpy.  Mirrors the in-kernel generator bit for bit so that
fused-step parity tests can include resets, pushes and observation noise.
key = (seed_lo, seed_hi); counter = (env_id, step_lo, step_hi, stream_id)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

# stream ids (must match wl_rng.h)
S_RESET, S_TIMERS, S_PUSH_HF, S_PUSH_LF, S_NOISE0 = 0, 1, 2, 3, 4


def philox4x32(env_ids, step, stream_id, seed):
    c0 = np.asarray(env_ids, dtype=np.uint64) & MASK
    c1 = np.full_like(c0, np.uint64(step & 0xFFFFFFFF))
    c2 = np.full_like(c0, np.uint64((step >> 32) & 0xFFFFFFFF))
    c3 = np.full_like(c0, np.uint64(stream_id))
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3]).astype(np.uint32)


def uniform4(env_ids, step, stream_id, seed):
    """-> float32 [4, n] in [0, 1)"""
    x = philox4x32(env_ids, step, stream_id, seed)
    return ((x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def normal12(env_ids, step, seed):
    """12 standard normals per env from streams S_NOISE0..+2 via Box-Muller -> float32 [12, n]"""
    out = []
    for s in range(3):
        u = uniform4(env_ids, step, S_NOISE0 + s, seed)
        for j in (0, 2):
            r = np.sqrt(np.float32(-2.0) * np.log(np.float32(1.0) - u[j])).astype(np.float32)
            th = np.float32(2.0 * np.pi) * u[j + 1]
            out += [r * np.cos(th), r * np.sin(th)]
    return np.stack(out).astype(np.float32)
