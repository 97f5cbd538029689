"""CPU restatement of the masking-noise generator pm_mae_noise (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 reference
implementation): 10 rounds of c = (hi(M1 c2) ^ c1 ^ k0, lo(M1 c2), hi(M0 c0) ^ c3 ^ k1, lo(M0 c0)) with M0 = 0xD2511F53,
M1 = 0xCD9E8D57 and the Weyl key schedule k0 += 0x9E3779B9, k1 += 0xBB67AE85.  Pinned by the known-answer vectors the Random123
distribution publishes (tests/test_host_cpu.py).  The reference draws this noise with torch.rand on the device
(models_mae.py:132), a stream that is not reproducible elsewhere; the build owns its generator ("Masking RNG").
"""
import numpy as np

_M = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: 4 arrays (or ints) of uint32 values, key: 2 uint32 values -> 4 uint32 arrays."""
    c = [np.asarray(x, dtype=np.uint64) & _M for x in counter]
    k0, k1 = np.uint64(key[0]) & _M, np.uint64(key[1]) & _M
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & _M, p1 >> np.uint64(32), p1 & _M
        c = [(hi1 ^ c[1] ^ k0) & _M, lo1, (hi0 ^ c[3] ^ k1) & _M, lo0]
        k0 = (k0 + np.uint64(0x9E3779B9)) & _M
        k1 = (k1 + np.uint64(0xBB67AE85)) & _M
    return [x.astype(np.uint32) for x in c]


def mae_noise(n: int, seed: int, stream_id: int) -> np.ndarray:
    """float32 [n]: element i = (word (i % 4) of philox(counter = (i // 4 lo, i // 4 hi, stream_id, 0), key = seed lo, hi) >> 8) * 2^-24."""
    q = np.arange((n + 3) // 4, dtype=np.uint64)
    w = philox4x32_10([q & _M, q >> np.uint64(32), np.full_like(q, stream_id), np.zeros_like(q)],
                      [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF])
    vals = np.stack(w, axis=1).reshape(-1)[:n]
    return ((vals >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)
