"""Batched encode engine: owns one pacx handle (HIP library, include/pacx.h) and
runs the per-frame hot path of codec.Encode for thousands of channel-frames per
launch.  PyTorch is used for device memory and streams only.

Reference path replaced: coder/codec.py:225-380 (Encode/EncodeSingleChannel)
and the window/mdct/psychoac/bitalloc/quantize functions it calls.
"""
import ctypes

import numpy as np
import torch

from . import _lib, tables
from .psychoac import ScaleFactorBands, AssignMDCTLinesFromFreqLimits

N_LONG, N_SHORT = 1024, 128          # MDCT lines of long / short blocks


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class PcmView:
    """Strided view of PCM on the device (see pacx_pcm in include/pacx.h)."""

    def __init__(self, tensor, n_channels, n_frames, frame_stride, channel_stride,
                 sample_stride=1):
        if tensor.dtype == torch.int16:
            dt = _lib.PCM_I16
        elif tensor.dtype == torch.float64:
            dt = _lib.PCM_F64
        else:
            raise TypeError("PCM must be int16 codes or float64 signed fractions")
        if not tensor.is_cuda:
            raise ValueError("PCM tensor must live on the GPU")
        self.tensor = tensor                      # keeps the memory alive
        self.n_channels, self.n_frames = int(n_channels), int(n_frames)
        self.c = _lib.PacxPcm(tensor.data_ptr(), dt, self.n_channels, self.n_frames,
                              int(frame_stride), int(channel_stride), int(sample_stride))

    @property
    def n_cf(self):
        return self.n_channels * self.n_frames

    @staticmethod
    def stream(planar, hop=N_LONG):
        """planar: [n_ch, (n_hops+1)*hop] -- hop 0 is the prior block (zeros at
        the start of a file); frame f spans hops f, f+1 (coder/pacfile.py:460-464)."""
        n_ch, n = planar.shape
        assert planar.is_contiguous() and n % hop == 0 and n >= 2 * hop
        return PcmView(planar, n_ch, n // hop - 1, hop, n, 1)

    @staticmethod
    def frames(blocks):
        """blocks: [n_frames, n_ch, 2*hop] independent full blocks."""
        n_f, n_ch, n = blocks.shape
        assert blocks.is_contiguous()
        return PcmView(blocks, n_ch, n_f, n_ch * n, n, 1)


class Encoder:
    """One handle = one (sampleRate, bit rate, band layout) on one GPU."""

    def __init__(self, sample_rate, target_bits_per_sample, n_scale_bits=4, n_mant_size_bits=12,
                 sf_bands=None, sf_bands_short=None, device=None, n_mdct_lines=N_LONG,
                 use_vq=False, use_sbr=False, guard=False):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.PacxError("no GPU visible: the encode path has no CPU implementation")
        if n_mdct_lines != N_LONG:
            raise _lib.PacxError("kernels are built for nMDCTLines = 1024")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.sample_rate = int(sample_rate)
        self.target_bits_per_sample = float(target_bits_per_sample)
        self.n_scale_bits, self.n_mant_size_bits = int(n_scale_bits), int(n_mant_size_bits)
        self.sfBands = sf_bands or ScaleFactorBands(
            AssignMDCTLinesFromFreqLimits(N_LONG, self.sample_rate))
        self.sfBandsShort = sf_bands_short or ScaleFactorBands(
            AssignMDCTLinesFromFreqLimits(N_SHORT, self.sample_rate))
        for bands in (self.sfBands, self.sfBandsShort):
            if np.any(np.asarray(bands.nLines) <= 0):
                # a band without lines (the default 25-band table below 31 kHz: bands above Nyquist are empty): the
                # reference gets as far as CalcSMRs' np.amax over the empty band (coder/psychoac.py:289) and raises
                raise ValueError("zero-size array to reduction operation maximum which has no identity")

        keep = self._host = {}
        def f64(name, arr):
            keep[name] = np.ascontiguousarray(arr, dtype=np.float64)
            return keep[name].ctypes.data_as(_lib.c_double_p)
        def i32(name, arr):
            keep[name] = np.ascontiguousarray(arr, dtype=np.int32)
            return keep[name].ctypes.data_as(_lib.c_int32_p)
        sr = self.sample_rate
        cfg = _lib.PacxConfig()
        cfg.abi_version = _lib.PACX_ABI_VERSION
        cfg.device = self.device.index
        cfg.sample_rate = sr
        cfg.n_lines_long, cfg.n_lines_short = N_LONG, N_SHORT
        cfg.n_scale_bits, cfg.n_mant_size_bits = self.n_scale_bits, self.n_mant_size_bits
        cfg.n_bands_long, cfg.n_bands_short = self.sfBands.nBands, self.sfBandsShort.nBands
        cfg.target_bits_per_sample = self.target_bits_per_sample
        cfg.band_lines_long = i32("bl", self.sfBands.nLines)
        cfg.band_lines_short = i32("bs", self.sfBandsShort.nLines)
        cfg.win_long = f64("wl", tables.long_windows(2 * N_LONG))
        cfg.win_short = f64("ws", tables.sine(2 * N_SHORT))
        cfg.hann_long = f64("hl", tables.hann(2 * N_LONG))
        cfg.hann_short = f64("hs", tables.hann(2 * N_SHORT))
        cfg.bark_long = f64("zl", tables.bark(tables.line_freqs(N_LONG, sr)))
        cfg.thresh_long = f64("tl", tables.thresh(tables.line_freqs(N_LONG, sr)))
        cfg.bark_short = f64("zs", tables.bark(tables.line_freqs(N_SHORT, sr)))
        cfg.thresh_short = f64("ts", tables.thresh(tables.line_freqs(N_SHORT, sr)))
        cfg.fft_norm_long = tables.fft_norm(2 * N_LONG)
        cfg.fft_norm_short = tables.fft_norm(2 * N_SHORT)
        cfg.fft_freq_step_long = tables.fft_freq_step(2 * N_LONG, sr)
        cfg.fft_freq_step_short = tables.fft_freq_step(2 * N_SHORT, sr)
        self.use_vq, self.use_sbr = bool(use_vq), bool(use_sbr)
        cfg.use_vq, cfg.use_sbr = int(self.use_vq), int(self.use_sbr)
        l_max = int(max(np.max(self.sfBands.nLines), np.max(self.sfBandsShort.nLines)))
        cfg.half_log2 = f64("hl2", tables.half_log2(l_max))
        cfg.vq_log2_tan = f64("lt", tables.vq_log2_tan())
        cfg.log_mu1 = tables.log_mu1()
        gw, gr = tables.sbr_gauss()
        cfg.sbr_gauss, cfg.sbr_gauss_radius = f64("gw", gw), gr
        cfg.line_freq_long = f64("lf", (np.arange(N_LONG) + 1 / 2) * (sr / (2 * N_LONG)))
        cfg.kbd_long = f64("kl", tables.kbd(2 * N_LONG))
        cfg.kbd_short = f64("ks", tables.kbd(2 * N_SHORT))
        cfg.guard = int(bool(guard))             # PACX_ST_GUARD in the status words (about 3 % of throughput)
        h = ctypes.c_void_p()
        rc = self.lib.pacx_create(ctypes.byref(cfg), ctypes.byref(h))
        _lib.check(self.lib, None, rc, "pacx_create")
        self.h = h
        self.band_stride = self.lib.pacx_band_stride(h)
        self.payload_stride = self.lib.pacx_payload_stride(h)

    def set_side_fork(self, enable):
        """all-long scalar batches: the side chain on the handle's second stream beside the transform (pays with several
        handles in the process, costs with one: include/pacx.h)"""
        self._call("pacx_set_side_fork", int(bool(enable)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.pacx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ----------------------------------------------------------------- helpers
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _call(self, name, *args):
        rc = getattr(self.lib, name)(self.h, *args)
        _lib.check(self.lib, self.h, rc, name)

    def reserve(self, n_cf):
        self._call("pacx_reserve", ctypes.c_int64(int(n_cf)))

    def flags_tensor(self, flags, n_frames):
        """flags: None, or per-frame iterable of (last, cur, next) / packed uint8."""
        if flags is None:
            return None
        if isinstance(flags, torch.Tensor):
            t = flags.to(device=self.device, dtype=torch.uint8)
        else:
            a = np.asarray(flags)
            if a.ndim == 2:
                a = (a[:, 0] != 0) * 1 + (a[:, 1] != 0) * 2 + (a[:, 2] != 0) * 4
            t = torch.as_tensor(a.astype(np.uint8), device=self.device)
        assert t.numel() == n_frames
        return t.contiguous()

    # ------------------------------------------------------------------ stages
    def mdct(self, pcm, flags=None, short=False, want_scale=False, prewindowed=False, kbd=False):
        """window + MDCT (+ overall scale factor).  lines: [n_cf, 1024] float64
        (short: [n_cf, 8, 128]).  kbd: KBDWindow instead of the sine window."""
        fl = self.flags_tensor(flags, pcm.n_frames)
        lines = self._empty((pcm.n_cf, N_LONG), torch.float64)
        scale = self._empty((pcm.n_cf, _lib.SUB) if short else (pcm.n_cf,), torch.int32) \
            if want_scale else None
        mode = (_lib.MDCT_SHORT if short else 0) | (_lib.MDCT_PREWINDOWED if prewindowed else 0) | \
            (_lib.MDCT_KBD if kbd else 0)
        self._call("pacx_mdct_batch", ctypes.byref(pcm.c), _ptr(fl), mode, _ptr(lines),
                   _ptr(scale), self._stream())
        if short:
            lines = lines.view(pcm.n_cf, _lib.SUB, N_SHORT)
        return (lines, scale) if want_scale else lines

    def smr(self, pcm, lines, short=False, want_threshold=False, want_peaks=False):
        """CalcSMRs.  lines are the unscaled MDCT lines.  smr: [n_cf, band_stride]."""
        lines = lines.contiguous().view(pcm.n_cf, N_LONG)
        smr = torch.zeros((pcm.n_cf, self.band_stride), dtype=torch.float64, device=self.device)
        thr = self._empty((pcm.n_cf, N_LONG), torch.float64) if want_threshold else None
        npk = self._empty((pcm.n_cf, _lib.SUB) if short else (pcm.n_cf,), torch.int32) \
            if want_peaks else None
        self._call("pacx_smr_batch", ctypes.byref(pcm.c), _ptr(lines), int(bool(short)), _ptr(smr),
                   _ptr(thr), _ptr(npk), self._stream())
        out = [smr]
        if want_threshold:
            out.append(thr)
        if want_peaks:
            out.append(npk)
        return out[0] if len(out) == 1 else tuple(out)

    def smr_generic(self, data, lines, t, want_threshold=False, want_peaks=False):
        """CalcSMRs / getMaskedThreshold for ANY block length (pacx_smr_generic_batch): data [n, N] float64 time
        blocks, lines [n, N/2] = MDCTdata / 2^MDCTscale; t: dict of device tensors hann, tw_cos, tw_sin [N], bark,
        quiet [N/2], band_lower, band_lines (int32) and the floats fft_norm, fft_freq_step (psychoac._generic_tables)."""
        data, lines = data.contiguous(), lines.contiguous()
        n_blocks, n = data.shape
        nb = int(t["band_lines"].numel())
        c = _lib.PacxSmrTables(t["hann"].data_ptr(), t["tw_cos"].data_ptr(), t["tw_sin"].data_ptr(), float(t["fft_norm"]),
                               float(t["fft_freq_step"]), t["bark"].data_ptr(), t["quiet"].data_ptr(),
                               t["band_lower"].data_ptr(), t["band_lines"].data_ptr(), nb)
        smr = self._empty((n_blocks, nb), torch.float64)
        thr = self._empty((n_blocks, n // 2), torch.float64) if want_threshold else None
        npk = self._empty((n_blocks,), torch.int32) if want_peaks else None
        self._call("pacx_smr_generic_batch", ctypes.c_int64(n_blocks), int(n), _ptr(data), _ptr(lines), ctypes.byref(c),
                   _ptr(smr), _ptr(thr), _ptr(npk), self._stream())
        out = [smr] + ([thr] if want_threshold else []) + ([npk] if want_peaks else [])
        return out[0] if len(out) == 1 else tuple(out)

    def bit_alloc(self, smr, n_channels=1, flags=None, short=False):
        smr = smr.contiguous()
        n_cf = smr.shape[0]
        fl = self.flags_tensor(flags, n_cf // n_channels)
        ba = torch.zeros((n_cf, self.band_stride), dtype=torch.int32, device=self.device)
        status = torch.zeros((n_cf,), dtype=torch.int32, device=self.device)
        self._call("pacx_bitalloc_batch", ctypes.c_int64(n_cf), int(n_channels), _ptr(fl),
                   int(bool(short)), _ptr(smr), _ptr(ba), _ptr(status), self._stream())
        return ba, status

    def quantize(self, lines, overall_scale, bit_alloc, short=False):
        n_cf = bit_alloc.shape[0]
        lines = lines.contiguous().view(n_cf, N_LONG)
        sf = torch.zeros((n_cf, self.band_stride), dtype=torch.int32, device=self.device)
        mant = self._empty((n_cf, N_LONG), torch.int32)
        self._call("pacx_quantize_batch", ctypes.c_int64(n_cf), _ptr(lines),
                   _ptr(overall_scale.contiguous()), _ptr(bit_alloc.contiguous()), int(bool(short)),
                   _ptr(sf), _ptr(mant), self._stream())
        return sf, mant

    # -------------------------------------------------------------- whole path
    def encode(self, pcm, flags=None, out=None):
        """codec.Encode for every channel of every frame of `pcm`.
        Returns dict of device tensors: overall [n_cf,8], scale_factor / bit_alloc
        [n_cf, band_stride], mantissa [n_cf,1024] (line-indexed), status [n_cf]."""
        n_cf = pcm.n_cf
        fl = self.flags_tensor(flags, pcm.n_frames)
        if out is None:
            out = self.alloc_outputs(n_cf)
        self._call("pacx_encode_batch", ctypes.byref(pcm.c), _ptr(fl), _ptr(out["overall"]),
                   _ptr(out["scale_factor"]), _ptr(out["bit_alloc"]), _ptr(out["mantissa"]),
                   _ptr(out["status"]), self._stream())
        out["flags"] = fl
        return out

    def encode_pack(self, pcm, flags=None, out=None, want_mantissa=False):
        """encode() + pack() in one call (long frames: one fused kernel after the
        masking stage).  `out` as alloc_outputs(n_cf, with_payload=True)."""
        n_cf = pcm.n_cf
        fl = self.flags_tensor(flags, pcm.n_frames)
        if out is None:
            out = self.alloc_outputs(n_cf, with_payload=True)
        self._call("pacx_encode_pack_batch", ctypes.byref(pcm.c), _ptr(fl), _ptr(out["overall"]),
                   _ptr(out["scale_factor"]), _ptr(out["bit_alloc"]),
                   _ptr(out["mantissa"]) if want_mantissa else None, _ptr(out["status"]),
                   _ptr(out["payload"]), _ptr(out["n_bytes"]), self._stream())
        out["flags"] = fl
        return out

    def encode_vq(self, pcm, flags=None, out=None, want_entries=False, entries_per_band=160):
        """The shipped configuration (gain-shape PVQ, SBR if the handle has it) from
        PCM to finished payloads.  Returns dict: overall [n_cf,8], bit_alloc
        [n_cf, band_stride] (final), payload [n_cf, payload_stride], n_bytes, status;
        with want_entries also entries [n_cf,8,32,cap] (structured: value, width,
        band) and entry_count [n_cf,8,32]."""
        n_cf = pcm.n_cf
        fl = self.flags_tensor(flags, pcm.n_frames)
        if out is None:
            out = {
                "overall": self._empty((n_cf, _lib.SUB), torch.int32),
                "bit_alloc": torch.zeros((n_cf, self.band_stride), dtype=torch.int32, device=self.device),
                "status": self._empty((n_cf,), torch.int32),
                "payload": self._empty((n_cf, self.payload_stride), torch.uint8),
                "n_bytes": self._empty((n_cf,), torch.int32),
            }
        ent = cnt = None
        if want_entries:
            ent = torch.zeros((n_cf, _lib.SUB, _lib.MAX_BANDS, entries_per_band, 2), dtype=torch.int64,
                              device=self.device)
            cnt = torch.zeros((n_cf, _lib.SUB, _lib.MAX_BANDS), dtype=torch.int32, device=self.device)
        self._call("pacx_encode_vq_batch", ctypes.byref(pcm.c), _ptr(fl), _ptr(out["overall"]),
                   _ptr(out["bit_alloc"]), _ptr(out["payload"]), _ptr(out["n_bytes"]), _ptr(out["status"]),
                   _ptr(ent), _ptr(cnt), ctypes.c_int32(entries_per_band if want_entries else 0),
                   self._stream())
        out["flags"] = fl
        if want_entries:
            out["entries"], out["entry_count"] = ent, cnt
        return out

    def alloc_outputs(self, n_cf, with_payload=False):
        o = {
            "overall": self._empty((n_cf, _lib.SUB), torch.int32),
            "scale_factor": torch.zeros((n_cf, self.band_stride), dtype=torch.int32, device=self.device),
            "bit_alloc": torch.zeros((n_cf, self.band_stride), dtype=torch.int32, device=self.device),
            "mantissa": self._empty((n_cf, N_LONG), torch.int32),
            "status": self._empty((n_cf,), torch.int32),
        }
        if with_payload:
            o["payload"] = self._empty((n_cf, self.payload_stride), torch.uint8)
            o["n_bytes"] = self._empty((n_cf,), torch.int32)
        return o

    def pack(self, enc, n_channels, out=None):
        """.pac payload of every cf: payload [n_cf, payload_stride] uint8, n_bytes [n_cf]."""
        n_cf = enc["bit_alloc"].shape[0]
        payload = out["payload"] if out else self._empty((n_cf, self.payload_stride), torch.uint8)
        n_bytes = out["n_bytes"] if out else self._empty((n_cf,), torch.int32)
        self._call("pacx_pack_batch", ctypes.c_int64(n_cf), int(n_channels), _ptr(enc.get("flags")),
                   _ptr(enc["overall"]), _ptr(enc["scale_factor"]), _ptr(enc["bit_alloc"]),
                   _ptr(enc["mantissa"]), _ptr(enc["status"]), _ptr(payload), _ptr(n_bytes),
                   self._stream())
        return payload, n_bytes

    def gather_body(self, payload, n_bytes, capacity=None, out=None):
        """'<L nBytes' + payload of every cf, back to back (the .pac body).  `out`: a uint8 device
        tensor to write into (e.g. a gather slot of dist.BitstreamGather); its length is the capacity."""
        n_cf = n_bytes.shape[0]
        if out is not None:
            capacity = out.numel()
        elif capacity is None:
            capacity = n_cf * (self.payload_stride + 4)
        body = out if out is not None else self._empty((capacity,), torch.uint8)
        total = torch.zeros((1,), dtype=torch.int64, device=self.device)
        self._call("pacx_gather_body", ctypes.c_int64(n_cf), _ptr(payload), _ptr(n_bytes), _ptr(body),
                   ctypes.c_int64(capacity), _ptr(total), self._stream())
        return body, total

    def transient_flags(self, planar, n_hops, hop=N_LONG):
        """Block-switching flags on the GPU for a device stream laid out as
        pacfile.device_stream builds it ([nCh, (n_hops+3)*hop]: zeros, the hops, the
        last hop again, zeros).  Returns (transient uint8[n_hops], flags uint8[n_hops+2])."""
        n_ch, n = planar.shape
        view = _lib.PacxPcm(planar.data_ptr() + 2 * hop, _lib.PCM_I16, n_ch, int(n_hops), hop, n, 1)
        tr = self._empty((max(n_hops, 1),), torch.uint8)
        fl = self._empty((n_hops + 2,), torch.uint8)
        self._call("pacx_transient_flags", ctypes.byref(view), _ptr(tr), _ptr(fl), self._stream())
        return tr[:n_hops], fl

    # ------------------------------------------------------------ decode side
    def unpack(self, payload, n_bytes, offsets=None):
        """Parse packed channel-blocks (slot layout, or a byte stream + int64 offsets).
        out["status"] carries PACX_ST_MALFORMED for a record that is truncated or corrupt."""
        n_cf = n_bytes.shape[0]
        out = self.alloc_outputs(n_cf)
        out["flags"] = self._empty((n_cf,), torch.uint8)
        stride = 0 if offsets is not None else int(payload.shape[1])
        self._call("pacx_unpack_batch", ctypes.c_int64(n_cf), _ptr(payload), stride, _ptr(offsets),
                   _ptr(n_bytes), _ptr(out["flags"]), _ptr(out["overall"]), _ptr(out["scale_factor"]),
                   _ptr(out["bit_alloc"]), _ptr(out["mantissa"]), _ptr(out["status"]), self._stream())
        return out

    def decode(self, codes, n_channels, want_blocks=False, want_pcm=True, extra=None, every_long_block=False):
        """codec.Decode + overlap-and-add + PCM for blocks in stream order.
        codes: dict with flags (per cf), overall, scale_factor, bit_alloc, mantissa.
        extra (a dict): route the blocks of an SBR file with scalar mantissas as PACFile.Decode does
        (coder/pacfile.py:645-668: long blocks with a coded omitted band through Decode_SBR's scalar
        branch; every_long_block: Decode_SBR on all of them); it receives "status"
        (PACX_ST_VQ_UNDEFINED where Decode_SBR raises IndexError) and "lines" (dequantised,
        reconstructed, before / 2^overall)."""
        n_cf = codes["bit_alloc"].shape[0]
        n_blocks = n_cf // n_channels
        blocks = self._empty((n_cf, 2 * N_LONG), torch.float64) if want_blocks else None
        pcm = self._empty(((n_blocks + 1) * N_LONG, n_channels), torch.int16) if want_pcm else None
        if extra is not None:
            extra["status"] = self._empty((n_cf,), torch.int32)
            extra["lines"] = self._empty((n_cf, N_LONG), torch.float64)
            self._call("pacx_decode_sbr_batch", ctypes.c_int64(n_blocks), int(n_channels), _ptr(codes["flags"]),
                       _ptr(codes["overall"]), _ptr(codes["scale_factor"]), _ptr(codes["bit_alloc"]),
                       _ptr(codes["mantissa"]), int(bool(every_long_block)), _ptr(extra["lines"]), _ptr(blocks), _ptr(pcm),
                       _ptr(extra["status"]), self._stream())
        else:
            self._call("pacx_decode_batch", ctypes.c_int64(n_blocks), int(n_channels), _ptr(codes["flags"]),
                       _ptr(codes["overall"]), _ptr(codes["scale_factor"]), _ptr(codes["bit_alloc"]),
                       _ptr(codes["mantissa"]), _ptr(blocks), _ptr(pcm), self._stream())
        return (blocks, pcm) if want_blocks and want_pcm else (blocks if want_blocks else pcm)

    def decode_vq(self, payload, n_bytes, n_channels, offsets=None, want_lines=False, want_blocks=False,
                  want_pcm=True):
        """Gain-shape coded channel-blocks (slot layout, or byte stream + int64
        offsets) -> dict: flags, overall, bit_alloc, status and, as requested,
        lines [n_cf,1024], blocks [n_cf,2048], pcm int16 [(n_blocks+1)*1024, nCh]."""
        n_cf = n_bytes.shape[0]
        n_blocks = n_cf // n_channels
        out = {
            "flags": self._empty((n_cf,), torch.uint8),
            "overall": self._empty((n_cf, _lib.SUB), torch.int32),
            "bit_alloc": torch.zeros((n_cf, self.band_stride), dtype=torch.int32, device=self.device),
            "status": self._empty((n_cf,), torch.int32),
            "lines": self._empty((n_cf, N_LONG), torch.float64) if want_lines else None,
            "blocks": self._empty((n_cf, 2 * N_LONG), torch.float64) if want_blocks else None,
            "pcm": self._empty(((n_blocks + 1) * N_LONG, n_channels), torch.int16) if want_pcm else None,
        }
        stride = 0 if offsets is not None else int(payload.shape[1])
        self._call("pacx_decode_vq_batch", ctypes.c_int64(n_blocks), int(n_channels), _ptr(payload), stride,
                   _ptr(offsets), _ptr(n_bytes), _ptr(out["flags"]), _ptr(out["overall"]),
                   _ptr(out["bit_alloc"]), _ptr(out["lines"]), _ptr(out["blocks"]), _ptr(out["pcm"]),
                   _ptr(out["status"]), self._stream())
        return out

    # ------------------------------------------- function-level entry points
    def window(self, kind, x):
        """window * x for rows of x ([n, 2048] or [n, 256] float64 on the GPU)."""
        x = x.contiguous()
        y = torch.empty_like(x)
        self._call("pacx_window_batch", int(kind), ctypes.c_int64(x.shape[0]), _ptr(x), _ptr(y),
                   self._stream())
        return y

    def window_table(self, table, x):
        """table * x for rows of x ([n, len] float64 on the GPU) with a caller-evaluated
        table (host array or device tensor of len entries)."""
        x = x.contiguous()
        t = torch.as_tensor(np.ascontiguousarray(table, dtype=np.float64), device=self.device) \
            if not isinstance(table, torch.Tensor) else table.contiguous()
        assert t.numel() == x.shape[-1]
        y = torch.empty_like(x)
        self._call("pacx_window_table_batch", _ptr(t), int(t.numel()), ctypes.c_int64(x.numel() // t.numel()),
                   _ptr(x), _ptr(y), self._stream())
        return y

    def tables_exact(self):
        return bool(self.lib.pacx_tables_exact(self.h))

    def _elem(self, name, x, *ints):
        x = x.contiguous()
        out = self._empty(x.shape, torch.int64)
        self._call(name, ctypes.c_int64(x.numel()), _ptr(x), *[int(i) for i in ints], _ptr(out),
                   self._stream())
        return out

    def quantize_uniform(self, x, n_bits):
        return self._elem("pacx_quantize_uniform", x, n_bits)

    def scale_factor(self, x, n_scale_bits, n_mant_bits):
        return self._elem("pacx_scale_factor", x, n_scale_bits, n_mant_bits)

    def mantissa(self, x, scale, n_scale_bits, n_mant_bits):
        return self._elem("pacx_mantissa", x, scale, n_scale_bits, n_mant_bits)

    def mantissa_fp(self, x, scale, n_scale_bits, n_mant_bits):
        return self._elem("pacx_mantissa_fp", x, scale, n_scale_bits, n_mant_bits)

    def _delem(self, name, codes, *ints):
        codes = codes.contiguous()
        assert codes.dtype == torch.int64
        out = self._empty(codes.shape, torch.float64)
        self._call(name, ctypes.c_int64(codes.numel()), _ptr(codes), *[int(i) for i in ints], _ptr(out),
                   self._stream())
        return out

    def dequantize_uniform(self, codes, n_bits):
        """vDequantizeUniform (coder/quantize.py:82-95) of int64 codes on the device"""
        return self._delem("pacx_dequantize_uniform", codes, n_bits)

    def dequantize(self, mant, scale, n_scale_bits, n_mant_bits):
        """vDequantize (coder/quantize.py:254-274)"""
        return self._delem("pacx_dequantize", mant, scale, n_scale_bits, n_mant_bits)

    def dequantize_fp(self, mant, scale, n_scale_bits, n_mant_bits):
        return self._delem("pacx_dequantize_fp", mant, scale, n_scale_bits, n_mant_bits)

    def imdct(self, lines, short=False):
        """mdct.IMDCT for rows of 1024 lines (short: rows of 8 x 128) -> [n, 2048] samples, unwindowed
        (k_imdct_long / k_imdct_short of the decode path)."""
        lines = lines.contiguous().view(-1, N_LONG)
        out = self._empty((lines.shape[0], 2 * N_LONG), torch.float64)
        self._call("pacx_imdct_batch", ctypes.c_int64(lines.shape[0]), _lib.MDCT_SHORT if short else 0, _ptr(lines),
                   _ptr(out), self._stream())
        return out

    def mdct_direct(self, x, a, b, inverse=False):
        """MDCT / IMDCT by the defining sums for any a + b (rows of x): coder/mdct.py:14-77"""
        n_in = (a + b) // 2 if inverse else a + b
        n_out = a + b if inverse else (a + b) // 2
        x = x.contiguous().view(-1, n_in)
        out = self._empty((x.shape[0], n_out), torch.float64)
        self._call("pacx_mdct_direct_batch", ctypes.c_int64(x.shape[0]), int(a), int(b), int(bool(inverse)), _ptr(x),
                   _ptr(out), self._stream())
        return out

    def transient_detect(self, blocks, thresh=4.5):
        """parTransientDetect (coder/detect_transients.py:5-23, axis=1) of float64 blocks [n, nCh, len] on the
        device: uint8 [n], 2 = the mean is exactly zero (the reference returns 0), 1 / 0 = transient or not."""
        blocks = blocks.contiguous()
        n, n_ch, ln = blocks.shape
        out = self._empty((n,), torch.uint8)
        self._call("pacx_transient_detect_f64", ctypes.c_int64(n), int(n_ch), int(ln), _ptr(blocks),
                   ctypes.c_double(float(thresh)), _ptr(out), self._stream())
        return out

    def bit_alloc_generic(self, budget, max_mant_bits, n_lines, smr):
        """BitAlloc for rows of smr [n, nBands] with per-row budgets [n]."""
        smr = smr.contiguous()
        n, nb = smr.shape
        nl = torch.as_tensor(np.asarray(n_lines, dtype=np.int32), device=self.device)
        bits = self._empty((n, nb), torch.int32)
        self._call("pacx_bitalloc_generic", ctypes.c_int64(n), int(nb), _ptr(nl), _ptr(budget.contiguous()),
                   int(max_mant_bits), _ptr(smr), _ptr(bits), self._stream())
        return bits


class EncoderPool:
    """Several steps in flight: N handles (each with its own workspaces) on N HIP streams, for callers with many
    INDEPENDENT batches.  A handle serialises its calls -- its workspaces are shared between them -- and every kernel of
    the path leaves a third to a half of the vector issue slots idle, bound by latency chains at the occupancy its
    registers and LDS allow; kernels of another batch fill those slots (two batches side by side: +15-28 % on one
    MI355X; more than two add nothing).  What `bench.py` times by default.

        pool = EncoderPool(2, 48000, 128 / 48.0)
        for i, batch in enumerate(batches):
            k = pool.next()                       # round robin; waits (on the device) for what last ran in slot k
            with pool.slot(k) as enc:             # enc: that slot's Encoder; the body is queued on the slot's stream
                enc.encode_pack(batch.view, None, outs[k])
        pool.synchronize()

    The caller keeps one set of output buffers per slot and must not read slot k's outputs before pool.wait(k) (or
    synchronize()).  The reference runs its own workers side by side too (Pool(8), coder/pacfile.py:771-781)."""

    def __init__(self, n, *args, **kwargs):
        self.encs = [Encoder(*args, **kwargs) for _ in range(max(1, int(n)))]
        dev = self.encs[0].device
        self.streams = [torch.cuda.Stream(device=dev) for _ in self.encs]
        self.done = [torch.cuda.Event() for _ in self.encs]
        self._turn = 0
        self._gate = self._gate_stream = None
        self._used = [False] * len(self.encs)
        if len(self.encs) > 1:                       # two handles: their two streams each share a hardware queue, the fork pays
            for e in self.encs:
                e.set_side_fork(True)

    def __len__(self):
        return len(self.encs)

    def next(self):
        k = self._turn
        self._turn = (k + 1) % len(self.encs)
        return k

    def slot(self, k):
        pool = self

        class _Slot:
            def __enter__(self_inner):
                self_inner.ctx = torch.cuda.stream(pool.streams[k])
                self_inner.ctx.__enter__()
                return pool.encs[k]

            def __exit__(self_inner, *exc):
                pool.done[k].record(pool.streams[k])
                pool._used[k] = True
                return self_inner.ctx.__exit__(*exc)
        return _Slot()

    def align(self, delay_us=None):
        """Start the next calls of all slots at the same instant.  From an idle device the slots' first steps start as
        far apart as the host takes to queue one (tens of microseconds), and two free-running pipelines then settle
        into one of two stable relations -- in phase (like kernels side by side: 51.4 M cf/s on the headline batch)
        or in anti-phase (47.0 M), DESIGN.md 5.0.  A gate event behind a short spin on a third stream holds every
        slot's stream until all the first calls are queued; started together they stay in phase for a hundred steps or
        so (bench.py: 51.9 M cf/s over 100-step regions, 47.9 M over 200-step ones -- the pipelines drift into the other
        relation; calling this again mid-stream, where it is a barrier between the slots followed by the common start,
        did not bring the faster relation back in the measurements of round 3).  Costs delay_us (default
        PACX_POOL_GATE_US or 100) per call."""
        if len(self.encs) < 2:
            return
        if delay_us is None:
            delay_us = float(os.environ.get("PACX_POOL_GATE_US", "100"))
        if delay_us <= 0 or not hasattr(torch.cuda, "_sleep"):
            return
        dev = self.encs[0].device
        if self._gate is None:
            self._gate_stream = torch.cuda.Stream(device=dev)
            self._gate = torch.cuda.Event()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(self._gate_stream):          # calibrate the spin: cycles per microsecond
                torch.cuda._sleep(100_000)
                t0.record()
                torch.cuda._sleep(2_000_000)
                t1.record()
            t1.synchronize()
            self._cycles_per_us = 2_000_000 / max(t0.elapsed_time(t1) * 1e3, 1.0)
        with torch.cuda.stream(self._gate_stream):
            for k in range(len(self.encs)):             # mid-stream: a barrier between the slots (what they have queued
                if self._used[k]:                       # so far finishes), then the common start
                    self._gate_stream.wait_event(self.done[k])
            torch.cuda._sleep(int(delay_us * self._cycles_per_us))
            self._gate.record(self._gate_stream)
        for s in self.streams:
            s.wait_event(self._gate)

    def wait(self, k, stream=None):
        """make `stream` (default: the current one) wait for what was last queued in slot k"""
        (stream or torch.cuda.current_stream(self.encs[0].device)).wait_event(self.done[k])

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def close(self):
        for e in self.encs:
            e.close()
