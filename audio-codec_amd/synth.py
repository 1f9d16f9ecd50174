"""Synthetic 48 kHz workload of BASELINE.json configs[1]:
six tones of the reference's own test signal (coder/psychoac.py:338-339
amplitudes / frequencies) with random phases plus white noise, as int16."""
import numpy as np

SEED = 422
AMPS = np.array([.43, .24, .15, .09, .05, .04])
FREQS = np.array([440, 550, 660, 880, 4400, 8800])


def stream(n_hops, n_ch=2, sample_rate=48000, seed=SEED, hop=1024, gain=1.0):
    """int16 [n_hops*hop, n_ch]."""
    rng = np.random.default_rng(seed)
    n = np.arange(n_hops * hop)
    out = np.zeros((n_hops * hop, n_ch), dtype=np.int16)
    for ch in range(n_ch):
        ph = rng.uniform(0, 2 * np.pi, size=6)
        x = 0.5 * np.sum(AMPS[:, None] * np.cos(
            2 * np.pi * FREQS[:, None] * n[None, :] / sample_rate + ph[:, None]), axis=0)
        x = x + 0.01 * rng.standard_normal(len(n))
        out[:, ch] = np.rint(32767 * np.clip(gain * x, -1, 1)).astype(np.int16)
    return out


def planar_with_halo(pcm, hop=1024):
    """[n, n_ch] interleaved -> [n_ch, n + hop] planar with a leading hop of
    zeros: the layout PcmView.stream expects (frame f = hops f, f+1)."""
    n, n_ch = pcm.shape
    out = np.zeros((n_ch, n + hop), dtype=pcm.dtype)
    out[:, hop:] = pcm.T
    return out
