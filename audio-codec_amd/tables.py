"""Static float64 tables of the encode path, evaluated on the host with the same
NumPy expressions the reference uses, so that what is uploaded to HBM is
bit-identical to what the reference multiplies by.  (Host set-up code: runs once
per handle, not per frame.)

Reference: coder/window.py:14-92, coder/psychoac.py:35-48, 100-160, 171-184.
"""
import numpy as np

SHORT_WINDOW = 256            # coder/codec.py:27
SHORT_LINES = 128             # coder/pacfile.py:490

# Zwicker critical-band upper edges, coder/psychoac.py:100-103
cbFreqLimits = np.array([
    100, 200, 300, 400, 510, 630, 770, 920, 1080, 1270, 1480, 1720, 2000, 2320,
    2700, 3150, 3700, 4400, 5300, 6400, 7700, 9500, 12000, 15500, 24000
])


def sine(n):
    return np.sin(np.pi * (np.arange(n) + 0.5) / n)


def hann(n):
    return 0.5 * (1 - np.cos(2 * np.pi * (np.arange(n) + 0.5) / n))


def kbd(n, alpha=4.):
    """coder/window.py:53-57 (what the reference calls its KBD window: the Kaiser form
    i0(pi alpha sqrt(1 - ((2n+1)/N - 1)^2)) / i0(pi alpha))."""
    k = np.arange(n)
    numerator = np.i0(np.pi * alpha * np.sqrt(1 - ((2 * k + 1) / n - 1) ** 2))
    denominator = np.i0(np.pi * alpha)
    return numerator / denominator


def start(n_long, n_short):
    pad = n_long // 4 - n_short // 4
    return np.concatenate((sine(n_long)[:n_long // 2], np.ones(pad),
                           sine(n_short)[n_short // 2:], np.zeros(pad)))


def stop(n_long, n_short):
    return np.flip(start(n_long, n_short))


def start_stop(n_long, n_short):
    pad = n_long // 4 - n_short // 4
    s = sine(n_short)
    return np.concatenate((np.zeros(pad), s[:n_short // 2], np.ones(2 * pad),
                           s[n_short // 2:], np.zeros(pad)))


def long_windows(n_long):
    """[4][n_long] in PACX_WIN_* order: sine, start, stop, start-stop."""
    return np.ascontiguousarray(np.stack([
        sine(n_long), start(n_long, SHORT_WINDOW), stop(n_long, SHORT_WINDOW),
        start_stop(n_long, SHORT_WINDOW)]))


def line_freqs(n_lines, sample_rate):
    return sample_rate / (2 * n_lines) * (np.arange(n_lines) + 0.5)


def bark(f):
    return 13.0 * np.arctan(0.76 * f / 1000.0) + 3.5 * np.arctan((f / 7500.0) ** 2)


def thresh(f):
    f = np.array(f, dtype=np.float64)
    f[f < 10] = 10
    return 3.64 * (f / 1000) ** (-0.8) - 6.5 * np.exp(-0.6 * (f / 1000 - 3.3) ** 2) \
        + 10 ** (-3) * (f / 1000) ** 4


def fft_norm(n):
    """coder/psychoac.py:172-173 (np.hanning: the symmetric Hann)."""
    return 4 / (n ** 2 * np.mean(np.hanning(n) ** 2))


def fft_freq_step(n, sample_rate):
    """Step of np.fft.rfftfreq(n, d=1/sample_rate)."""
    return 1.0 / (n * (1 / sample_rate))


def band_line_counts(n_lines, sample_rate, flimit=cbFreqLimits):
    """coder/psychoac.py:106-124."""
    width = sample_rate / (2 * n_lines)
    centers = np.floor(flimit / width - 0.5)
    counts = centers - np.concatenate([[-1], centers[:-1]])
    for i in range(len(counts)):
        if flimit[i] > sample_rate / 2:
            counts[i] = n_lines - np.sum(counts[0:i])
            counts[i + 1:] = 0
            break
    return counts


def half_log2(l_max):
    """0.5*np.log2(L) for L = 0..l_max (gain_shape_alloc,
    coder/gain_shape_quantize.py:59); entry 0 is unused."""
    out = np.zeros(l_max + 1)
    out[1:] = 0.5 * np.log2(np.arange(1, l_max + 1))
    return out


def log_mu1(mu=255):
    """np.log(1 + mu) of mu_law_fn, coder/gain_shape_quantize.py:294-295."""
    return float(np.log(1 + mu))


def sbr_gauss(sigma=200, truncate=4.0):
    """Weights of scipy.ndimage.gaussian_filter1d(x, sigma) as Decode_SBR calls it
    (coder/codec.py:147): exp(-0.5 (j/sigma)^2), j = -r..r, over their sum."""
    r = int(truncate * float(sigma) + 0.5)
    j = np.arange(-r, r + 1)
    w = np.exp(-0.5 / (sigma * sigma) * j ** 2)
    return w / w.sum(), r


VQ_THETA_TABLE_BITS = 12


def vq_log2_tan(max_bits=VQ_THETA_TABLE_BITS):
    """log2(tan(theta_q) + eps) for every quantised split angle of up to max_bits
    bits, evaluated exactly as bit_allocation_ms does
    (coder/gain_shape_quantize.py:302-309 with theta_q =
    DequantizeUniform(code, a) * (pi/2), coder/quantize.py:39-57).  Layout: the
    2^(a-1) non-negative codes of width a start at offset 2^(a-1) - 1."""
    eps = np.finfo(float).eps
    out = np.zeros((1 << max_bits) - 1)
    for a in range(1, max_bits + 1):
        base = (1 << (a - 1)) - 1
        for code in range(1 << (a - 1)):
            theta = (1 * 2 * code / (2 ** a - 1)) * (np.pi / 2)
            if theta != 0:
                out[base + code] = np.log2(np.tan(abs(theta)) + eps)
    return out
