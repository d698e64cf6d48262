"""TEST INFRASTRUCTURE ONLY (never imported by the product): numpy restatement of `ivid_randn` (include/ivid_hip.h) --
Philox4x32-10 as published (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123
known-answer vectors pin it in tests/test_oracle_golden.py) + the kernel's 24-bit uniforms and Box-Muller transform.  The
reference itself draws its noise with torch.randn (ddim.py:101, ddpm.py:128, inpaint_cfg.py:36-45): there is no reference stream
to match, only the N(0,1) law and the counter-based contract (a value depends on (seed, stream_id, index) alone)."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(counter, key):
    """counter [...,4] uint32, key [...,2] uint32 (broadcastable) -> [...,4] uint32."""
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(M0) * c[0], np.uint64(M1) * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0, k1 = (k0 + np.uint64(W0)) & mask, (k1 + np.uint64(W1)) & mask
    return np.stack(c, axis=-1).astype(np.uint32)


def randn(seed, stream_id, n):
    """The first n values of stream `stream_id` under `seed` (float32; the kernel's logf / sincospif differ by a few ulp)."""
    nb = (n + 3) // 4
    j = np.arange(nb, dtype=np.uint64)
    ctr = np.stack([j & np.uint64(0xFFFFFFFF), j >> np.uint64(32), np.full(nb, stream_id & 0xFFFFFFFF, np.uint64),
                    np.full(nb, (stream_id >> 32) & 0xFFFFFFFF, np.uint64)], axis=-1)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    u = ((r >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    out = np.empty((nb, 4), dtype=np.float32)
    for h in range(2):
        rad = np.sqrt(np.float32(-2.0) * np.log(u[:, 2 * h].astype(np.float64))).astype(np.float32)
        ang = 2.0 * np.pi * u[:, 2 * h + 1].astype(np.float64)
        out[:, 2 * h] = rad * np.cos(ang).astype(np.float32)
        out[:, 2 * h + 1] = rad * np.sin(ang).astype(np.float32)
    return out.reshape(-1)[:n]
