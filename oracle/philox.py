"""CPU restatement of the in-kernel noise generator (TEST INFRASTRUCTURE ONLY).

The reference draws the per-prompt noise with `torch.randn_like` (`src/attack_model.py:316-321`,
`src/crossattack_models.py:357-362`); bit-parity with torch's stream is not a goal (parity tests feed
the noise as an input tensor).  What IS pinned here is that the HIP generator is the published
Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) and
that its counter addressing is the documented one:

    counter = (quad q of the sample, batch row b, offset_lo, offset_hi),  key = (seed_lo, seed_hi)
    element 4q+k of row b  <-  k-th Box-Muller output of that block

`KAT` holds the three known-answer vectors of the Random123 distribution (`kat_vectors`,
lines "philox4x32 10 ..."); `tests/test_oracle_philox.py` checks this file against them and the
`-m gpu` tests check the device against this file.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85

# (counter[4], key[2]) -> expected[4]
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF),
     (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


def philox4x32_10(counter, key, rounds=10):
    """counter: uint32 [..., 4]; key: (k0, k1).  Returns uint32 [..., 4]."""
    c = np.asarray(counter, dtype=np.uint32)
    c0, c1, c2, c3 = (c[..., k].astype(np.uint64) for k in range(4))
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    sh = np.uint64(32)
    for _ in range(rounds):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> sh, p0 & mask
        hi1, lo1 = p1 >> sh, p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def box_muller4(r):
    """uint32 [..., 4] -> float64 [..., 4]: the device's uniform construction (one fp32 convert and one
    fp32 fma / multiply per word), then exact Box-Muller in float64 (the device evaluates log2 / sqrt /
    sin / cos with the hardware approximations, so it is compared at a tolerance)."""
    rf = r.astype(np.float32).astype(np.float64)                 # v_cvt_f32_u32: round to nearest even
    u1 = (rf[..., 0] * 2.0 ** -32 + 2.0 ** -33).astype(np.float32).astype(np.float64)   # fma: one rounding
    u2 = (rf[..., 1] * 2.0 ** -32).astype(np.float32).astype(np.float64)
    u3 = (rf[..., 2] * 2.0 ** -32 + 2.0 ** -33).astype(np.float32).astype(np.float64)
    u4 = (rf[..., 3] * 2.0 ** -32).astype(np.float32).astype(np.float64)
    ra = np.sqrt(-2.0 * np.log(u1))
    rb = np.sqrt(-2.0 * np.log(u3))
    return np.stack([ra * np.cos(2 * np.pi * u2), ra * np.sin(2 * np.pi * u2),
                     rb * np.cos(2 * np.pi * u4), rb * np.sin(2 * np.pi * u4)], axis=-1)


def unit_noise(batch, n, seed, offset):
    """[batch, n] float64: the unit normals k_emit / k_fused_fwd add (times sigma) to sample b, element i.
    n is padded up to a multiple of four inside a block; the tail of the last block is dropped."""
    n4 = (n + 3) // 4
    q = np.arange(n4, dtype=np.uint32)
    out = np.empty((batch, n4 * 4), np.float64)
    for b in range(batch):
        c = np.stack([q, np.full(n4, b, np.uint32), np.full(n4, offset & 0xFFFFFFFF, np.uint32),
                      np.full(n4, (offset >> 32) & 0xFFFFFFFF, np.uint32)], axis=-1)
        out[b] = box_muller4(philox4x32_10(c, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))).reshape(-1)
    return out[:, :n]
