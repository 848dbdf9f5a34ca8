"""Philox4x32 counter RNG (Salmon et al., SC'11), numpy, ROUNDS rounds (7: the smallest Crush-resistant round count the paper
reports for this width; wheeledlab_amd/csrc/wl_rng.h says why).  Mirrors the in-kernel generator bit for bit so that fused-step
parity tests can include resets, pushes and observation noise.  Pinned against the Random123 distribution's known answers
for 7 rounds and (through `rounds=10`) for 10 (tests/test_oracle_golden_drift.py).
key = (seed_lo, seed_hi); counter = (env_id, step_lo, step_hi, stream_id)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

ROUNDS = 7
# stream ids (must match wl_rng.h); the drift step draws from S_DRIFT_EVENTS, S_NOISE0 and S_NOISE1 (layout: drift_draws below)
S_RESET, S_DRIFT_EVENTS, S_NOISE0, S_NOISE1 = 0, 0, 4, 5


def philox4x32(env_ids, step, stream_id, seed, rounds=ROUNDS):
    c0 = np.asarray(env_ids, dtype=np.uint64) & MASK
    c1 = np.full_like(c0, np.uint64(step & 0xFFFFFFFF))
    c2 = np.full_like(c0, np.uint64((step >> 32) & 0xFFFFFFFF))
    c3 = np.full_like(c0, np.uint64(stream_id))
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for _ in range(rounds):
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


def uniform8(env_ids, step, stream_id, seed):
    """eight 16-bit uniforms per block, (h + 1/2) / 65536 in (0, 1): index 2k = low half of word k, 2k + 1 = its high half
    (wl_rng.h u16_lo / u16_hi) -> float32 [8, n]"""
    x = philox4x32(env_ids, step, stream_id, seed)
    lo, hi = (x & np.uint32(0xFFFF)).astype(np.float32), (x >> np.uint32(16)).astype(np.float32)
    h = np.stack([lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], lo[3], hi[3]])
    return (h * np.float32(2.0 ** -16) + np.float32(2.0 ** -17)).astype(np.float32)


def _pair(u0, u1):
    r = np.sqrt(np.float32(-2.0) * np.log(u0)).astype(np.float32)
    th = np.float32(2.0 * np.pi) * u1
    return [r * np.cos(th), r * np.sin(th)]


def normal12(env_ids, step, seed):
    """12 standard normals per env: one Box-Muller pair per word (radius from the low half, angle from the high half) of the four
    words of stream S_NOISE0 and the first two of S_NOISE1 (wl_drift_env.h obs_noise) -> float32 [12, n]"""
    a, b = uniform8(env_ids, step, S_NOISE0, seed), uniform8(env_ids, step, S_NOISE1, seed)
    out = []
    for k in range(4):
        out += _pair(a[2 * k], a[2 * k + 1])
    for k in range(2):
        out += _pair(b[2 * k], b[2 * k + 1])
    return np.stack(out).astype(np.float32)
