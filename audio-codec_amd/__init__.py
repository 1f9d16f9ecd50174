"""MI355X-native batched encode path for the Abhipray/audio-codec perceptual
coder: the reference's window -> MDCT -> psychoacoustic SMR -> bit allocation ->
scale-factor/mantissa quantisation hot loop as hand-written HIP kernels behind
a C ABI (include/pacx.h), plus Python modules that mirror the reference's own
interface (codec.Encode, PACFile.WriteDataBlock, psychoac.CalcSMRs, ...).

The directory name carries a hyphen; import it as `audio_codec_amd`
(audio_codec_amd.py at the repository root aliases it)."""
from . import _lib                                            # noqa: F401
from ._lib import PacxError, load                             # noqa: F401


def __getattr__(name):
    # engine / mirrors import torch; keep `import audio_codec_amd` light
    import importlib
    if name in ("engine", "codec", "window", "mdct", "psychoac", "bitalloc", "quantize", "pacfile",
                "pcmfile", "audiofile", "detect_transients", "synth", "tables", "context", "dist", "build", "streaming"):
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
