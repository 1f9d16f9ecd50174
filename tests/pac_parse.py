"""Test helper: walk a scalar-path .pac byte stream (header + '<L nBytes' blocks,
coder/pacfile.py:404-447,552-577 layout) and produce a canonical form in which
the sign bit of zero-magnitude mantissas is cleared.  Such sign bits encode
'-0': where a line is pure FFT rounding noise (a constant block) the reference's
choice depends on NumPy's FFT and cannot be reproduced (known limit)."""
import hashlib

import numpy as np


def _bits(block):
    return np.unpackbits(np.frombuffer(block, dtype=np.uint8))


def _get(bits, pos, n):
    v = 0
    for b in bits[pos:pos + n]:
        v = (v << 1) | int(b)
    return v


def canonical_blocks(data, header_len, bands_long, bands_short, n_scale_bits=4, n_mant_size_bits=12):
    """Yield canonicalised payload bytes per channel-block, and the number of sign bits cleared."""
    pos = header_len
    cleared = 0
    out = []
    while pos < len(data):
        n = int.from_bytes(data[pos:pos + 4], "little")
        block = bytearray(data[pos + 4:pos + 4 + n])
        pos += 4 + n
        bits = _bits(bytes(block))
        cur = int(bits[1])
        p = 3
        for _ in range(8 if cur else 1):
            n_lines = bands_short if cur else bands_long
            p += n_scale_bits
            for nl in n_lines:
                a = _get(bits, p, n_mant_size_bits)
                a = a + 1 if a else 0
                p += n_mant_size_bits + n_scale_bits
                if a:
                    fields = bits[p:p + a * nl].reshape(nl, a)
                    zero_mag = ~fields[:, 1:].any(axis=1) & (fields[:, 0] == 1)
                    for j in np.nonzero(zero_mag)[0]:
                        bp = p + j * a
                        block[bp >> 3] &= ~(0x80 >> (bp & 7)) & 0xFF
                        cleared += 1
                    p += a * nl
        out.append(bytes(block))
    return out, cleared


def canonical_sha256(data, header_len, bands_long, bands_short):
    blocks, cleared = canonical_blocks(data, header_len, bands_long, bands_short)
    h = hashlib.sha256(data[:header_len])
    for b in blocks:
        h.update(len(b).to_bytes(4, "little"))
        h.update(b)
    return h.hexdigest(), cleared
