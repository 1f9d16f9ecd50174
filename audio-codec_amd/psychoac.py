"""psychoac.py mirror: same names and argument meaning as coder/psychoac.py.

ScaleFactorBands / AssignMDCTLinesFromFreqLimits / cbFreqLimits are static
tables (host).  CalcSMRs and getMaskedThreshold run on the GPU
(k_side_* + k_mask in csrc/k_psy.hip).
"""
import numpy as np

from . import tables
from .tables import cbFreqLimits  # noqa: F401  (re-exported, coder/psychoac.py:100)


def AssignMDCTLinesFromFreqLimits(nMDCTLines, sampleRate, flimit=cbFreqLimits):
    """coder/psychoac.py:106-124."""
    return tables.band_line_counts(nMDCTLines, sampleRate, flimit)


class ScaleFactorBands:
    """coder/psychoac.py:127-160: bands of <= 12 lines are merged into their
    right neighbour; nBands, nLines, lowerLine, upperLine."""

    def __init__(self, nLines):
        n = np.array(nLines, dtype=int)
        i = 1
        while i < len(n):
            if n[i - 1] <= 12:
                n[i] += n[i - 1]
                n = np.delete(n, i - 1)
            else:
                i += 1
        self.nLines = n
        self.nBands = len(n)
        self.lowerLine = np.zeros((self.nBands,), dtype=int)
        for i in range(1, self.nBands):
            self.lowerLine[i] = self.lowerLine[i - 1] + n[i - 1]
        self.upperLine = (self.lowerLine + n - 1).astype(int)


def bands_from_counts(lines):
    """A ScaleFactorBands holding exactly these line counts (no merge step): the private copy an Encoder keeps."""
    b = ScaleFactorBands.__new__(ScaleFactorBands)
    n = np.array(lines, dtype=int)
    b.nLines, b.nBands = n, len(n)
    b.lowerLine = np.concatenate(([0], np.cumsum(n)[:-1])).astype(int)
    b.upperLine = (b.lowerLine + n - 1).astype(int)
    return b


def true_line_counts(bands):
    """Line counts of a band table taken from its line RANGES where it has them.  The reference's BitAlloc_SBR
    overwrites sfBands.nLines of the SBR-omitted bands with 1 and leaves it that way for the rest of the file
    (coder/bitalloc.py:141-143); lowerLine / upperLine keep the layout, and they are what its slicing uses."""
    if bands is None:
        return None
    lo, up = getattr(bands, "lowerLine", None), getattr(bands, "upperLine", None)
    if lo is not None and up is not None and len(lo) == len(up) == len(bands.nLines):
        return tuple(int(u) - int(l) + 1 for l, u in zip(lo, up))
    return tuple(int(v) for v in bands.nLines)


_generic = {}


def _generic_tables(n, sample_rate, sf_bands, device):
    """device copies of the tables pacx_smr_generic_batch wants, evaluated with NumPy the way the reference
    evaluates them (window.HanningWindow, np.hanning for the norm, rfftfreq, the Bark values and thresholds in
    quiet of the MDCT lines: coder/psychoac.py:45-46, 28-42, 171-179, 183-186), cached per (N, rate, band layout)"""
    import torch
    from . import tables
    counts = true_line_counts(sf_bands)
    key = (int(n), float(sample_rate), counts, str(device))
    t = _generic.get(key)
    if t is None:
        if min(counts) < 1:
            raise ValueError("zero-size array to reduction operation maximum which has no identity")   # np.amax, coder/psychoac.py:289
        m = np.arange(n) * (2.0 * np.pi / n)
        f = tables.line_freqs(n // 2, sample_rate)
        lower = np.concatenate(([0], np.cumsum(counts)[:-1]))
        as_t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=device)
        t = {"hann": as_t(tables.hann(n), np.float64), "tw_cos": as_t(np.cos(m), np.float64),
             "tw_sin": as_t(np.sin(m), np.float64), "fft_norm": float(tables.fft_norm(n)),
             "fft_freq_step": float(tables.fft_freq_step(n, sample_rate)), "bark": as_t(tables.bark(f), np.float64),
             "quiet": as_t(tables.thresh(f), np.float64), "band_lower": as_t(lower, np.int32),
             "band_lines": as_t(counts, np.int32)}
        _generic[key] = t
    return t


def _run_generic(data, MDCTdata, MDCTscale, sampleRate, sfBands, want_threshold):
    """block lengths the tuned kernels are not built for (nMDCTLines other than 1024 / 128): the function-level
    kernel k_smr_generic, one workgroup per block"""
    import torch
    from . import context
    enc = context.any_encoder()
    n = len(data)
    if n % 2 or len(MDCTdata) != n // 2:
        raise ValueError("CalcSMRs: data holds 2 * len(MDCTdata) samples")
    t = _generic_tables(n, sampleRate, sfBands, enc.device)
    lines = np.asarray(MDCTdata, dtype=np.float64) / 2 ** MDCTscale
    res = enc.smr_generic(torch.as_tensor(data, device=enc.device).view(1, n),
                          torch.as_tensor(lines, device=enc.device).view(1, n // 2), t, want_threshold=want_threshold)
    if want_threshold:
        return res[1][0].cpu().numpy()
    return res[0].cpu().numpy()


def _run(data, MDCTdata, MDCTscale, sampleRate, sfBands, want_threshold):
    import torch
    from . import context
    from .engine import PcmView
    data = np.ascontiguousarray(data, dtype=np.float64)
    n_lines = len(MDCTdata)
    short = (n_lines == 128)
    if n_lines not in (1024, 128) or len(data) != 2 * n_lines:
        return _run_generic(data, MDCTdata, MDCTscale, sampleRate, sfBands, want_threshold)
    enc = context.encoder_for_bands(sampleRate, sfBands, short)
    lines = np.asarray(MDCTdata, dtype=np.float64) / 2 ** MDCTscale
    if short:
        # a lone short block: park it in sub-block 0 of a frame
        frame = np.zeros(2048)
        frame[448:448 + 256] = data
        full = np.zeros(1024)
        full[:128] = lines
        data, lines = frame, full
    pcm = PcmView.frames(torch.as_tensor(data, device=enc.device).view(1, 1, 2048))
    res = enc.smr(pcm, torch.as_tensor(lines, device=enc.device).view(1, 1024), short=short,
                  want_threshold=want_threshold)
    nb = sfBands.nBands
    if want_threshold:
        return res[1][0, :n_lines].cpu().numpy()
    return res[0, :nb].cpu().numpy()


def CalcSMRs(data, MDCTdata, MDCTscale, sampleRate, sfBands):
    """coder/psychoac.py:220-291 on the GPU.  MDCTdata are the lines scaled by
    2^MDCTscale, as the reference passes them."""
    return _run(data, MDCTdata, MDCTscale, sampleRate, sfBands, False)


def getMaskedThreshold(data, MDCTdata, MDCTscale, sampleRate, sfBands):
    """coder/psychoac.py:163-217 on the GPU: masked threshold (dB SPL) at the
    MDCT line frequencies."""
    return _run(data, MDCTdata, MDCTscale, sampleRate, sfBands, True)
