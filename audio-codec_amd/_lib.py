"""ctypes binding of include/pacx.h.  The HIP library is the only compute path:
if libpacx.so is missing (or no GPU is visible at create time) this raises."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PACX_LIB") or os.path.join(HERE, "libpacx.so")   # PACX_LIB: kernel-variant experiments

PACX_ABI_VERSION = 7
PCM_I16, PCM_F64 = 0, 1
FLAG_LAST, FLAG_CUR, FLAG_NEXT = 1, 2, 4
ST_SHORT, ST_ZERO_SUBBLOCK, ST_ALLOC_CAP, ST_VQ_UNDEFINED, ST_GUARD, ST_MALFORMED = 1, 2, 4, 8, 16, 32
ST_REF_RAISES = 64
# what the reference raises where PACX_ST_REF_RAISES is set (coder/quantize.py:74, see include/pacx.h)
REF_SCALAR_SBR_ERROR = "'numpy.int64' object does not support item assignment"
MAX_BANDS = 32
SUB = 8

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)


class PacxSmrTables(ctypes.Structure):
    """pacx_smr_tables (include/pacx.h): device pointers to caller-evaluated tables of pacx_smr_generic_batch"""
    _fields_ = [
        ("hann", ctypes.c_void_p),
        ("tw_cos", ctypes.c_void_p),
        ("tw_sin", ctypes.c_void_p),
        ("fft_norm", ctypes.c_double),
        ("fft_freq_step", ctypes.c_double),
        ("bark", ctypes.c_void_p),
        ("quiet", ctypes.c_void_p),
        ("band_lower", ctypes.c_void_p),
        ("band_lines", ctypes.c_void_p),
        ("n_bands", ctypes.c_int32),
    ]


class PacxConfig(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("sample_rate", ctypes.c_int32),
        ("n_lines_long", ctypes.c_int32),
        ("n_lines_short", ctypes.c_int32),
        ("n_scale_bits", ctypes.c_int32),
        ("n_mant_size_bits", ctypes.c_int32),
        ("n_bands_long", ctypes.c_int32),
        ("n_bands_short", ctypes.c_int32),
        ("target_bits_per_sample", ctypes.c_double),
        ("band_lines_long", c_int32_p),
        ("band_lines_short", c_int32_p),
        ("win_long", c_double_p),
        ("win_short", c_double_p),
        ("hann_long", c_double_p),
        ("hann_short", c_double_p),
        ("bark_long", c_double_p),
        ("thresh_long", c_double_p),
        ("bark_short", c_double_p),
        ("thresh_short", c_double_p),
        ("fft_norm_long", ctypes.c_double),
        ("fft_norm_short", ctypes.c_double),
        ("fft_freq_step_long", ctypes.c_double),
        ("fft_freq_step_short", ctypes.c_double),
        ("use_vq", ctypes.c_int32),
        ("use_sbr", ctypes.c_int32),
        ("half_log2", c_double_p),
        ("vq_log2_tan", c_double_p),
        ("log_mu1", ctypes.c_double),
        ("sbr_gauss", c_double_p),
        ("sbr_gauss_radius", ctypes.c_int32),
        ("line_freq_long", c_double_p),
        ("kbd_long", c_double_p),
        ("kbd_short", c_double_p),
        ("guard", ctypes.c_int32),
    ]


class PacxVqEntry(ctypes.Structure):
    _fields_ = [("value", ctypes.c_uint64), ("width", ctypes.c_int32), ("band", ctypes.c_int32)]


class PacxPcm(ctypes.Structure):
    _fields_ = [
        ("data", ctypes.c_void_p),
        ("dtype", ctypes.c_int32),
        ("n_channels", ctypes.c_int32),
        ("n_frames", ctypes.c_int64),
        ("frame_stride", ctypes.c_int64),
        ("channel_stride", ctypes.c_int64),
        ("sample_stride", ctypes.c_int64),
    ]


# name -> (restype, argtypes); this is the full export list of include/pacx.h
_P = ctypes.c_void_p
SIGNATURES = {
    "pacx_abi_version": (ctypes.c_int, []),
    "pacx_create": (ctypes.c_int, [ctypes.POINTER(PacxConfig), ctypes.POINTER(_P)]),
    "pacx_destroy": (None, [_P]),
    "pacx_last_error": (ctypes.c_char_p, [_P]),
    "pacx_band_stride": (ctypes.c_int, [_P]),
    "pacx_payload_stride": (ctypes.c_int, [_P]),
    "pacx_reserve": (ctypes.c_int, [_P, ctypes.c_int64]),
    "pacx_tables_exact": (ctypes.c_int, [_P]),
    "pacx_default_bands": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_int32_p, c_int32_p]),
    "pacx_mdct_batch": (ctypes.c_int, [_P, ctypes.POINTER(PacxPcm), _P, ctypes.c_int, _P, _P, _P]),
    "pacx_smr_batch": (ctypes.c_int, [_P, ctypes.POINTER(PacxPcm), _P, ctypes.c_int, _P, _P, _P, _P]),
    "pacx_bitalloc_batch": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int, _P, ctypes.c_int, _P, _P, _P, _P]),
    "pacx_quantize_batch": (ctypes.c_int, [_P, ctypes.c_int64, _P, _P, _P, ctypes.c_int, _P, _P, _P]),
    "pacx_encode_batch": (ctypes.c_int, [_P, ctypes.POINTER(PacxPcm), _P, _P, _P, _P, _P, _P, _P]),
    "pacx_encode_pack_batch": (ctypes.c_int, [_P, ctypes.POINTER(PacxPcm), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pacx_encode_vq_batch": (ctypes.c_int, [_P, ctypes.POINTER(PacxPcm), _P, _P, _P, _P, _P, _P, _P, _P,
                                            ctypes.c_int32, _P]),
    "pacx_pack_batch": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pacx_gather_body": (ctypes.c_int, [_P, ctypes.c_int64, _P, _P, _P, ctypes.c_int64, _P, _P]),
    "pacx_window_batch": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int64, _P, _P, _P]),
    "pacx_window_table_batch": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int64, _P, _P, _P]),
    "pacx_quantize_uniform": (ctypes.c_int, [_P, ctypes.c_int64, _P, ctypes.c_int, _P, _P]),
    "pacx_scale_factor": (ctypes.c_int, [_P, ctypes.c_int64, _P, ctypes.c_int, ctypes.c_int, _P, _P]),
    "pacx_mantissa": (ctypes.c_int, [_P, ctypes.c_int64, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    "pacx_dequantize_uniform": (ctypes.c_int, [_P, ctypes.c_int64, _P, ctypes.c_int, _P, _P]),
    "pacx_dequantize": (ctypes.c_int, [_P, ctypes.c_int64, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    "pacx_mantissa_fp": (ctypes.c_int, [_P, ctypes.c_int64, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    "pacx_dequantize_fp": (ctypes.c_int, [_P, ctypes.c_int64, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    "pacx_imdct_batch": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int, _P, _P, _P]),
    "pacx_mdct_direct_batch": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P, _P]),
    "pacx_transient_detect_f64": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _P, ctypes.c_double, _P, _P]),
    "pacx_unpack_batch": (ctypes.c_int, [_P, ctypes.c_int64, _P, ctypes.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pacx_set_side_fork": (ctypes.c_int, [_P, ctypes.c_int]),
    "pacx_smr_generic_batch": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int, _P, _P, ctypes.POINTER(PacxSmrTables), _P, _P,
                                              _P, _P]),
    "pacx_decode_batch": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pacx_decode_sbr_batch": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int, _P, _P, _P, _P, _P, ctypes.c_int, _P, _P,
                                             _P, _P, _P]),
    "pacx_decode_vq_batch": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int, _P, ctypes.c_int, _P, _P, _P, _P, _P,
                                            _P, _P, _P, _P, _P]),
    "pacx_transient_flags": (ctypes.c_int, [_P, ctypes.POINTER(PacxPcm), _P, _P, _P]),
    "pacx_bitalloc_generic": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int, _P, _P, ctypes.c_int, _P, _P, _P]),
}
(WIN_SINE, WIN_START, WIN_STOP, WIN_STARTSTOP, WIN_SINE_SHORT, WIN_HANN, WIN_HANN_SHORT, WIN_KBD,
 WIN_KBD_SHORT) = range(9)
MDCT_SHORT, MDCT_PREWINDOWED, MDCT_KBD = 1, 2, 4

_lib = None


class PacxError(RuntimeError):
    pass


def load():
    """Load libpacx.so and declare every prototype.  No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # The library must belong to the sources next to it: build.py keys that on a content hash
    # (sources + headers + flags) stored beside the .so.  A stale or missing library is an ERROR here,
    # not a silent ten-compile rebuild inside whatever process happened to import the package (a
    # torchrun rank, a run under rocprofv3, a timed test).  PACX_AUTOBUILD=1 opts in to the rebuild
    # (__graft_entry__, bench.py and the test suite set it; the rebuild says so on stderr).
    from . import build as _build
    if os.environ.get("PACX_LIB"):
        # a kernel-variant or debug library (build.py --variant / --phase-debug): it carries the hash of the
        # sources it was built from as well
        if _build._read(LIB_PATH + ".hash") != _build.library_hash():
            raise PacxError(f"{LIB_PATH} (PACX_LIB) was not built from the sources in this tree: rebuild it with "
                            "`python audio-codec_amd/build.py --variant ...` / `--phase-debug`")
    elif not _build.is_current():
        if os.environ.get("PACX_AUTOBUILD") == "1":
            import sys
            print("audio_codec_amd: libpacx.so is missing or stale -- rebuilding (PACX_AUTOBUILD=1)", file=sys.stderr,
                  flush=True)
            try:
                _build.build(verbose=False)
            except Exception as e:                   # no hipcc, compile error: say so, no fallback
                raise PacxError(f"libpacx.so is missing or stale and could not be rebuilt: {e}") from e
        else:
            raise PacxError("libpacx.so is missing or does not belong to the sources in this tree (content hash "
                            "mismatch): run `python audio-codec_amd/build.py` (or set PACX_AUTOBUILD=1).  "
                            "There is no CPU implementation of this path.")
    if not os.path.exists(LIB_PATH):
        raise PacxError(
            f"{LIB_PATH} not found: build it with `python audio-codec_amd/build.py` "
            "(hipcc, gfx950).  There is no CPU implementation of this path.")
    # One HIP runtime per process: PyTorch ships its own libamdhip64 (SONAME
    # libamdhip64.so.7).  Importing torch first makes the loader resolve this
    # library's NEEDED libamdhip64.so.7 to that already-loaded copy, so torch's
    # device pointers and streams are valid here.  (A C host without torch
    # resolves it through the RUNPATH to /opt/rocm.)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if lib.pacx_abi_version() != PACX_ABI_VERSION:
        raise PacxError("libpacx.so ABI version does not match the Python binding")
    _lib = lib
    return lib


def check(lib, handle, rc, what):
    if rc != 0:
        msg = lib.pacx_last_error(handle)
        raise PacxError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
