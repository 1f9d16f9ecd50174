#!/usr/bin/env python3
"""
Lane-level NumPy model of the wave64 FFT / MDCT / real-FFT index math used by
audio-codec_amd/csrc (development aid: the HIP code is a transcription of this
file; run it to check the decomposition against np.fft).

Model: a wave is 64 lanes; "registers" are arrays [64, R]; LDS is a flat
complex array; every exchange is one scatter + one gather.
"""
import numpy as np

L = np.arange(64)


def dft8(v):
    """v[64, 8] complex -> DFT over axis 1 (what each lane does in registers)."""
    k = np.arange(8)
    W = np.exp(-2j * np.pi * np.outer(k, k) / 8)
    return v @ W.T


def fft64x8(v):
    """8 independent 64-point FFTs, one per 8-lane group g = lane>>3.
    In : lane (g, r) reg j  = x_g[r + 8 j]
    Out: lane (g, r) reg k3 = X_g[r + 8 k3]"""
    g, r = L >> 3, L & 7
    B = dft8(v)                                           # over j -> k2
    k2 = np.arange(8)
    B = B * np.exp(-2j * np.pi * np.outer(r, k2) / 64)    # W64^(r*k2)
    # exchange inside the group: element (g, k2, n3=r) stored at row g,
    # column 8*k2 + (r ^ k2)   (xor swizzle, conflict-free both ways)
    lds = np.zeros(8 * 72, dtype=complex)
    for kk in range(8):
        lds[g * 72 + 8 * kk + (r ^ kk)] = B[:, kk]
    c = np.zeros((64, 8), dtype=complex)
    for n3 in range(8):                                   # lane (g, r''=k2) reads n3
        c[:, n3] = lds[g * 72 + 8 * r + (n3 ^ r)]
    return dft8(c)                                        # over n3 -> k3


def fft512(v):
    """In : lane l reg n1 = x[l + 64 n1]
    Out: lane l = 8*k1 + k2, reg k3 = X[k1 + 8 k2 + 64 k3]"""
    A = dft8(v)
    k1 = np.arange(8)
    A = A * np.exp(-2j * np.pi * np.outer(L, k1) / 512)   # W512^(l*k1)
    lds = np.zeros(8 * 72, dtype=complex)
    for kk in range(8):
        lds[kk * 72 + L] = A[:, kk]                       # row k1, col l
    g, r = L >> 3, L & 7
    b = np.zeros((64, 8), dtype=complex)
    for n2 in range(8):
        b[:, n2] = lds[g * 72 + 8 * n2 + r]               # row k1'=g, col 8 n2 + n3'
    return fft64x8(b)


def out_index_512():
    """k held by (lane, reg) after fft512."""
    g, r = L >> 3, L & 7
    return g[:, None] + 8 * r[:, None] + 64 * np.arange(8)[None, :]


def mdct_long(xw):
    """MDCT of a windowed 2048 block via one 512-point complex FFT
    (the single twiddle table d[n] =
    exp(-j*pi*(8n+1)/8192) serves as pre- and post-twiddle)."""
    N, M, Q = 2048, 1024, 512
    n = L[:, None] + 64 * np.arange(8)[None, :]           # t index per (lane, reg)
    lo = n < Q // 2
    m = 2 * n - Q
    re = np.where(lo, -xw[np.where(lo, 3 * Q - 1 - 2 * n, 0)] - xw[np.where(lo, 3 * Q + 2 * n, 0)],
                  xw[np.where(lo, 0, m)] - xw[np.where(lo, 0, M - 1 - m)])
    im = np.where(lo, xw[np.where(lo, Q - 1 - 2 * n, 0)] - xw[np.where(lo, Q + 2 * n, 0)],
                  -xw[np.where(lo, 0, 2 * Q + m)] - xw[np.where(lo, 0, 4 * Q - 1 - m)])
    d = lambda i: np.exp(-1j * np.pi * (8 * i + 1) / 8192)
    t = (re + 1j * im) * d(n)
    T = fft512(t)
    k = out_index_512()
    y = T * d(k) * (2.0 / N)
    out = np.zeros(M)
    out[2 * k] = y.real
    out[M - 1 - 2 * k] = -y.imag
    return out


def mdct_short8(xw8):
    """Eight 128-line MDCTs (windowed 256 blocks, xw8[8, 256]) in one wave via
    fft64x8: lane (g, r) handles sub-block g."""
    N, M, Q = 256, 128, 64
    g, r = L >> 3, L & 7
    n = r[:, None] + 8 * np.arange(8)[None, :]            # 0..63 within sub-block
    lo = n < Q // 2
    m = 2 * n - Q
    X = xw8[g]                                            # [64, 256]
    take = lambda idx: np.take_along_axis(X, idx, axis=1)
    z = np.zeros_like(n)
    re = np.where(lo, -take(np.where(lo, 3 * Q - 1 - 2 * n, z)) - take(np.where(lo, 3 * Q + 2 * n, z)),
                  take(np.where(lo, z, m)) - take(np.where(lo, z, M - 1 - m)))
    im = np.where(lo, take(np.where(lo, Q - 1 - 2 * n, z)) - take(np.where(lo, Q + 2 * n, z)),
                  -take(np.where(lo, z, 2 * Q + m)) - take(np.where(lo, z, 4 * Q - 1 - m)))
    d = lambda i: np.exp(-1j * np.pi * (8 * i + 1) / (8 * M))
    t = (re + 1j * im) * d(n)
    T = fft64x8(t)
    k = r[:, None] + 8 * np.arange(8)[None, :]
    y = T * d(k) * (2.0 / N)
    out = np.zeros((8, M))
    out[g[:, None], 2 * k] = y.real
    out[g[:, None], M - 1 - 2 * k] = -y.imag
    return out


def rfft2048(xh):
    """2048-point real FFT via two 512-point complex FFTs.
    z[m] = xh[2m] + j xh[2m+1] (1024 complex); E = FFT512(z[0::2]),
    O = FFT512(z[1::2]); Z[k] = E[k] + W1024^k O[k], Z[k+512] = E[k] - W1024^k O[k];
    X[k] = (Z[k] + conj Z[1024-k])/2 - j/2 W2048^k (Z[k] - conj Z[1024-k])."""
    n = L[:, None] + 64 * np.arange(8)[None, :]
    e = xh[4 * n] + 1j * xh[4 * n + 1]
    o = xh[4 * n + 2] + 1j * xh[4 * n + 3]
    E, O = fft512(e), fft512(o)
    k = out_index_512()
    w = np.exp(-2j * np.pi * k / 1024)
    Z = np.zeros(1025, dtype=complex)                     # LDS image
    Z[k] = E + w * O
    Z[k + 512] = E - w * O
    Z[1024] = Z[0]
    kk = np.arange(1025)
    a, b = Z[kk], np.conj(Z[1024 - kk])
    return 0.5 * (a + b) - 0.5j * np.exp(-2j * np.pi * kk / 2048) * (a - b)


def rfft256x8(xh8):
    """Eight 256-point real FFTs (xh8[8,256]) via fft64x8 on evens and odds."""
    g, r = L >> 3, L & 7
    n = r[:, None] + 8 * np.arange(8)[None, :]
    X = xh8[g]
    take = lambda idx: np.take_along_axis(X, idx, axis=1)
    e = take(4 * n) + 1j * take(4 * n + 1)
    o = take(4 * n + 2) + 1j * take(4 * n + 3)
    E, O = fft64x8(e), fft64x8(o)
    k = n
    w = np.exp(-2j * np.pi * k / 128)
    Z = np.zeros((8, 129), dtype=complex)
    Z[g[:, None], k] = E + w * O
    Z[g[:, None], k + 64] = E - w * O
    Z[:, 128] = Z[:, 0]
    kk = np.arange(129)
    a, b = Z[:, kk], np.conj(Z[:, 128 - kk])
    return 0.5 * (a + b) - 0.5j * np.exp(-2j * np.pi * kk / 256) * (a - b)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    x = rng.standard_normal(512) + 1j * rng.standard_normal(512)
    v = x[L[:, None] + 64 * np.arange(8)[None, :]]
    X = np.zeros(512, dtype=complex)
    X[out_index_512()] = fft512(v)
    print("fft512  err", np.max(np.abs(X - np.fft.fft(x))))

    x8 = rng.standard_normal((8, 64)) + 1j * rng.standard_normal((8, 64))
    g, r = L >> 3, L & 7
    v = x8[g[:, None], r[:, None] + 8 * np.arange(8)[None, :]]
    Y = fft64x8(v)
    X8 = np.zeros((8, 64), dtype=complex)
    X8[g[:, None], r[:, None] + 8 * np.arange(8)[None, :]] = Y
    print("fft64x8 err", np.max(np.abs(X8 - np.fft.fft(x8, axis=1))))

    def sine_window(n):
        return np.sin(np.pi * (np.arange(n) + 0.5) / n)

    def mdct_direct(x, half_n):
        """X[k] = (2/N) sum x[n] cos(2 pi/N (n + n0)(k + 1/2)), n0 = N/4 + 1/2 (the definition)"""
        n_total = 2 * half_n
        n = np.arange(n_total)[None, :] + (half_n / 2 + 0.5)
        k = np.arange(half_n)[:, None] + 0.5
        return (2.0 / n_total) * (np.cos(2 * np.pi / n_total * n * k) @ np.atleast_2d(x).T).T.squeeze()

    xw = sine_window(2048) * rng.standard_normal(2048)
    ref = mdct_direct(xw, 1024)
    got = mdct_long(xw)
    print("mdct2048 rel err", np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
    xw8 = sine_window(256)[None, :] * rng.standard_normal((8, 256))
    ref = mdct_direct(xw8, 128)
    got = mdct_short8(xw8)
    print("mdct256x8 rel err", np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
    xh = rng.standard_normal(2048)
    print("rfft2048 err", np.max(np.abs(rfft2048(xh) - np.fft.rfft(xh))))
    xh8 = rng.standard_normal((8, 256))
    print("rfft256x8 err", np.max(np.abs(rfft256x8(xh8) - np.fft.rfft(xh8, axis=1))))


# ---------------------------------------------------------------------------
# v2: 512-point FFT with NATURAL-order output (lane L, reg k3 holds X[L + 64 k3])
# on an unpadded 512-complex tile; both exchanges conflict-free via XOR swizzles
# (searched exhaustively against the ds_read_b128 / ds_write_b128 lane groups).
def fft512n(v):
    A = dft8(v)
    k1 = np.arange(8)
    A = A * np.exp(-2j * np.pi * np.outer(L, k1) / 512)
    lds = np.zeros(512, dtype=complex)
    for kk in range(8):
        lds[64 * kk + (L ^ (8 * kk))] = A[:, kk]                  # X1 write
    g, r = L >> 3, L & 7
    b = np.zeros((64, 8), dtype=complex)
    for n2 in range(8):
        b[:, n2] = lds[64 * g + 8 * (n2 ^ g) + r]                  # X1 read
    B = dft8(b)
    k2 = np.arange(8)
    B = B * np.exp(-2j * np.pi * np.outer(r, k2) / 64)
    lds = np.zeros(512, dtype=complex)
    for kk in range(8):
        lds[64 * g + 8 * kk + (r ^ g)] = B[:, kk]                  # X2 write
    c = np.zeros((64, 8), dtype=complex)
    for n3 in range(8):
        c[:, n3] = lds[64 * (L & 7) + 8 * (L >> 3) + (n3 ^ (L & 7))]   # X2 read, natural lanes
    return dft8(c)


def mdct_long_v2(xw):
    """MDCT with fft512n: pre- and post-twiddle use the same per-lane table
    entries d[L + 64 j]; out[2k+1] comes from the mirrored lane / register."""
    N, M, Q = 2048, 1024, 512
    n = L[:, None] + 64 * np.arange(8)[None, :]
    lo = n < Q // 2
    m = 2 * n - Q
    re = np.where(lo, -xw[np.where(lo, 3 * Q - 1 - 2 * n, 0)] - xw[np.where(lo, 3 * Q + 2 * n, 0)],
                  xw[np.where(lo, 0, m)] - xw[np.where(lo, 0, M - 1 - m)])
    im = np.where(lo, xw[np.where(lo, Q - 1 - 2 * n, 0)] - xw[np.where(lo, Q + 2 * n, 0)],
                  -xw[np.where(lo, 0, 2 * Q + m)] - xw[np.where(lo, 0, 4 * Q - 1 - m)])
    d = np.exp(-1j * np.pi * (8 * n + 1) / 8192)
    y = fft512n((re + 1j * im) * d) * d * (2.0 / N)
    a = y.real                      # -> out[2k]
    bb = -y.imag                    # -> out[M-1-2k]
    partner = bb[63 - L][:, ::-1]   # lane 63-L, register 7-k3
    out = np.zeros(M)
    out[2 * n] = a
    out[2 * n + 1] = partner
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    x = rng.standard_normal(512) + 1j * rng.standard_normal(512)
    v = x[L[:, None] + 64 * np.arange(8)[None, :]]
    X = np.zeros(512, dtype=complex)
    X[L[:, None] + 64 * np.arange(8)[None, :]] = fft512n(v)
    print("fft512n err", np.max(np.abs(X - np.fft.fft(x))))
    xw = po.sine_window(2048) * rng.standard_normal(2048)
    ref = po.mdct_forward(xw, 1024, 1024)
    print("mdct v2 rel err", np.max(np.abs(mdct_long_v2(xw) - ref)) / np.max(np.abs(ref)))


# ---------------------------------------------------------------------------
# IMDCT through the SAME DCT-IV pipeline as the forward transform: the lines are
# fed where the folded input u went, the result is unfolded (transpose of the fold)
def dct4_pipeline(u, M):
    """what the forward kernel computes from the folded input u (without 2/N)."""
    Q = M // 2
    n = np.arange(Q)
    d = np.exp(-1j * np.pi * (8 * n + 1) / (8 * M))
    t = (u[2 * n] + 1j * u[M - 1 - 2 * n]) * d
    T = np.fft.fft(t)
    y = T * d
    out = np.zeros(M)
    out[2 * n] = y.real
    out[M - 1 - 2 * n] = -y.imag
    return out


def imdct_via_dct4(X, N):
    M, Q = N // 2, N // 4
    v = dct4_pipeline(X, M)
    y = np.zeros(N)
    n = np.arange(Q)
    y[3 * Q - 1 - n] = -v[n]
    y[3 * Q + n] = -v[n]
    y[n] = v[Q + n]
    y[M - 1 - n] = -v[Q + n]
    return y


if __name__ == "__main__":
    rng = np.random.default_rng(5)
    for N in (2048, 256):
        X = rng.standard_normal(N // 2)
        ref = po.mdct_inverse(X, N // 2, N // 2)
        got = imdct_via_dct4(X, N)
        scale = np.dot(ref, got) / np.dot(got, got)
        print("imdct", N, "scale", scale, "rel err", np.max(np.abs(ref - scale * got)) / np.max(np.abs(ref)))
