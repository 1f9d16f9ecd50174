"""Cache of Encoder handles for the function-level mirrors (one per distinct
sample rate / bit rate / band layout), so that repeated calls such as
codec.Encode(data, codingParams, ...) reuse resident tables."""
from . import engine

_cache = {}


def encoder(sample_rate, target_bits_per_sample, n_scale_bits=4, n_mant_size_bits=12,
            sf_bands=None, sf_bands_short=None, use_vq=False, use_sbr=False):
    # the layout by its line ranges, and a private copy for the handle: a caller's sfBands.nLines may have been
    # overwritten by BitAlloc_SBR (1 for the omitted bands, coder/bitalloc.py:141-143) -- the reference's own
    # slicing goes by lowerLine / upperLine, which stay
    from .psychoac import bands_from_counts, true_line_counts
    long_lines, short_lines = true_line_counts(sf_bands), true_line_counts(sf_bands_short)
    key = (int(sample_rate), float(target_bits_per_sample), int(n_scale_bits), int(n_mant_size_bits),
           bool(use_vq), bool(use_sbr), long_lines, short_lines)
    enc = _cache.get(key)
    if enc is None:
        enc = engine.Encoder(sample_rate, target_bits_per_sample, n_scale_bits, n_mant_size_bits,
                             None if long_lines is None else bands_from_counts(long_lines),
                             None if short_lines is None else bands_from_counts(short_lines),
                             use_vq=use_vq, use_sbr=use_sbr)
        _cache[key] = enc
    return enc


def encoder_for_params(cp):
    """From a reference-style CodingParams bag (coder/pacfile.py:699-707,323-330)."""
    return encoder(cp.sampleRate, cp.targetBitsPerSample, cp.nScaleBits, cp.nMantSizeBits,
                   getattr(cp, "sfBands", None), getattr(cp, "sfBandsShort", None),
                   use_vq=bool(getattr(cp, "useVQ", False)), use_sbr=bool(getattr(cp, "useSBR", False)))


def encoder_for_bands(sample_rate, sf_bands, short):
    """For psychoac.CalcSMRs(..., sampleRate, sfBands): only the band layout of
    the block kind in use matters."""
    if short:
        return encoder(sample_rate, 128 / (sample_rate / 1000), sf_bands_short=sf_bands)
    return encoder(sample_rate, 128 / (sample_rate / 1000), sf_bands=sf_bands)


def any_encoder():
    """For functions that need no codec configuration (windows, quantisers)."""
    if _cache:
        return next(iter(_cache.values()))
    return encoder(48000, 128 / 48.0)


def clear():
    for e in _cache.values():
        e.close()
    _cache.clear()
