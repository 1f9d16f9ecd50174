"""Cache of Encoder handles for the function-level mirrors (one per distinct
sample rate / bit rate / band layout), so that repeated calls such as
codec.Encode(data, codingParams, ...) reuse resident tables."""
from . import engine

_cache = {}


def encoder(sample_rate, target_bits_per_sample, n_scale_bits=4, n_mant_size_bits=12,
            sf_bands=None, sf_bands_short=None, use_vq=False, use_sbr=False):
    key = (int(sample_rate), float(target_bits_per_sample), int(n_scale_bits), int(n_mant_size_bits),
           bool(use_vq), bool(use_sbr),
           None if sf_bands is None else tuple(int(v) for v in sf_bands.nLines),
           None if sf_bands_short is None else tuple(int(v) for v in sf_bands_short.nLines))
    enc = _cache.get(key)
    if enc is None:
        enc = engine.Encoder(sample_rate, target_bits_per_sample, n_scale_bits, n_mant_size_bits,
                             sf_bands, sf_bands_short, use_vq=use_vq, use_sbr=use_sbr)
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
