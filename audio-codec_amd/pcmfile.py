"""pcmfile.py mirror (coder/pcmfile.py): 16-bit WAV reader with the reference's
input contract.  File I/O is host code; the int16 -> signed-fraction mapping
(+-2|c|/65535, coder/pcmfile.py:89-99 + coder/quantize.py:82-95) is also done
inside the GPU kernels when they are fed int16 directly."""
import struct

import numpy as np

from .audiofile import AudioFile, CodingParams

BYTESIZE = 8


def codes_to_fraction(codes):
    """int16 codes -> signed fractions, value for value what
    PCMFile.ReadDataBlock returns (-32768 -> -0.0)."""
    c = np.asarray(codes).astype(np.int64)
    neg = c < 0
    mag = np.where(neg, -c, c) & 32767
    val = (2 * mag) / 65535
    val[neg] *= -1.0
    return val


class PCMFile(AudioFile):
    def ReadFileHeader(self):
        """coder/pcmfile.py:32-64 (4-byte scan for 'fmt ' then 'data')."""
        tag = self.fp.read(12)
        if tag[0:4] != b"RIFF" or tag[8:12] != b"WAVE":
            raise RuntimeError("ERROR: File opened for PCMFile is not a RIFF file!")
        while True:
            tag = self.fp.read(4)
            if len(tag) < 4:
                raise RuntimeError("ERROR: Didn't find WAV file 'fmt ' chunk following RIFF file header")
            if tag == b"fmt ":
                break
        (_, formatTag, nChannels, sampleRate, _, _, bitsPerSample) = struct.unpack(
            "<LHHLLHH", self.fp.read(20))
        if formatTag != 1:
            raise IOError("Opened a non-PCM WAV file as a PCMFile")
        if bitsPerSample != 16:
            raise RuntimeError("PCMFile was not 16-bits per sample")
        while True:
            tag = self.fp.read(4)
            if len(tag) < 4:
                raise RuntimeError("Didn't find WAV file 'data' chunk following 'fmt ' chunk")
            if tag == b"data":
                break
        numSamples = struct.unpack("<L", self.fp.read(4))[0] // (nChannels * (bitsPerSample // BYTESIZE))
        p = CodingParams()
        p.nChannels, p.bitsPerSample = nChannels, bitsPerSample
        p.sampleRate, p.numSamples = sampleRate, numSamples
        p.bytesReadSoFar = 0
        return p

    def _read_codes(self, cp):
        """Next block as interleaved int16 codes or None; byte accounting of
        coder/pcmfile.py:66-80 (it trusts cp.numSamples, which PACFile's header
        writer has inflated by then -- the reference reads past the data chunk)."""
        want = cp.nSamplesPerBlock * cp.nChannels * (cp.bitsPerSample // BYTESIZE)
        left = cp.nChannels * cp.numSamples * (cp.bitsPerSample // BYTESIZE) - cp.bytesReadSoFar
        if left <= 0:
            raw = None
        elif left < want:
            raw = self.fp.read(left)
        else:
            raw = self.fp.read(want)
        cp.bytesReadSoFar += want
        if raw and len(raw) < want:
            raw += (want - len(raw)) * b"\0"
        elif not raw:
            return None
        return np.frombuffer(raw, dtype="<i2").reshape(-1, cp.nChannels)

    def ReadDataBlock(self, codingParams):
        """List (per channel) of float64 signed-fraction arrays, or None at the end."""
        codes = self._read_codes(codingParams)
        if codes is None:
            return None
        return [codes_to_fraction(codes[:, ch]) for ch in range(codingParams.nChannels)]


def wav_effective_stream(path, hop=1024):
    """All PCM the reference's driver would feed the coder for this file, as
    int16 [n, nCh] (n a multiple of hop), plus (sampleRate, declared numSamples).
    Includes the bytes it reads beyond the data chunk (see PCMFile._read_codes)."""
    f = PCMFile(path)
    cp = f.OpenForReading()
    declared = cp.numSamples
    cp.nSamplesPerBlock = hop
    if not cp.numSamples % hop:                      # coder/pacfile.py:309-313
        cp.numSamples += hop - cp.numSamples % hop
    cp.numSamples += hop                             # coder/pacfile.py:315
    blocks = []
    while True:
        b = f._read_codes(cp)
        if b is None:
            break
        blocks.append(b)
    f.Close(cp)
    pcm = np.concatenate(blocks) if blocks else np.zeros((0, cp.nChannels), dtype=np.int16)
    return cp.sampleRate, pcm, declared


def fraction_to_codes(x):
    """Signed fractions -> int16 codes, coder/pcmfile.py:127-134 (host I/O glue;
    the batched decoder does this on the GPU in k_ola_pcm)."""
    x = np.array(x, dtype=np.float64)
    neg = np.signbit(x)
    mag = np.abs(x)
    q = np.floor((65535 * mag + 1) / 2)
    q[mag >= 1] = 32767
    q = q.astype(np.int16)
    q[neg] *= -1
    return q
