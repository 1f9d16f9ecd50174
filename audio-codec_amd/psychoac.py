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


def _run(data, MDCTdata, MDCTscale, sampleRate, sfBands, want_threshold):
    import torch
    from . import context
    from .engine import PcmView
    data = np.ascontiguousarray(data, dtype=np.float64)
    n_lines = len(MDCTdata)
    short = (n_lines == 128)
    if n_lines not in (1024, 128) or len(data) != 2 * n_lines:
        raise NotImplementedError("GPU CalcSMRs handles 1024- and 128-line blocks")
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
