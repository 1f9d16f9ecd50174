"""
CPU restatement of the reference's gain-shape (PVQ) and SBR encode variants --
TEST INFRASTRUCTURE ONLY (see oracle/pac_oracle.py's header: nothing in the
product may import this package).

Covers SURVEY.md section 8f-3 / BASELINE config 4:
  coder/gain_shape_quantize.py   pyramid VQ of a band's shape, mu-law gain,
                                 recursive mid/side band splitting
  coder/codec.py:292-360         EncodeSingleChannel, useVQ branch
  coder/codec.py:426-531         EncodeSingleChannel_SBR, useVQ branch
  coder/bitalloc.py:123-145      BitAlloc_SBR (in-place nLines[omitted] = 1)
  coder/pacfile.py:342-402       size rule + WriiteEncodedBitsVQ
  coder/pacfile.py:699-757       the shipped driver: useVQ always, useSBR
                                 below 128 kb/s, block switching on

Parity is pinned on the reference's OWN committed outputs: the files under
/root/reference/test_decoded_full/*_coded_{96,128}.pac are what this driver
wrote for test_signals/*.wav, and tests/golden/make_golden.py --vq checks that
encode_stream_vq() reproduces them byte for byte (sha256 in
tests/golden/vqfile.json).

The same NumPy calls as the reference are used wherever the summation order
matters (np.sum, np.mean, np.linalg.norm), so this file inherits NumPy's
pairwise / BLAS orders by construction.
"""
import math
import struct

import numpy as np

from . import pac_oracle as po

SPLIT_BITS = 32          # coder/gain_shape_quantize.py:27
K_FINE = 0               # coder/codec.py:26
EPS = np.finfo(float).eps


# ------------------------------------------------------------------ N(L, K)
_rows = {0: [1]}         # _rows[l][k] = N(l, k); row 0 is 1, 0, 0, ...


def codebook_size(l, k):
    """Number of integer vectors of dimension l with sum |y_i| = k
    (coder/gain_shape_quantize.py:69-102: N(l,0)=1, N(0,k>=1)=0,
    N(l,k) = N(l-1,k) + N(l-1,k-1) + N(l,k-1)).  Python ints, grown on demand."""
    if l == 0:
        return 1 if k == 0 else 0
    for ll in range(1, l + 1):
        row = _rows.setdefault(ll, [1])
        if len(row) <= k:
            codebook_size(ll - 1, k)
            below = _rows[ll - 1] if ll > 1 else None
            while len(row) <= k:
                kk = len(row)
                if ll == 1:
                    up, upl = (0, 1 if kk == 1 else 0)
                else:
                    up, upl = below[kk], below[kk - 1]
                row.append(up + upl + row[kk - 1])
    return _rows[l][k]


_kcache = {}


def pulses_for_bits(l, max_bits):
    """Largest K with N(l,K) <= 2**max_bits and the index width the reference
    then uses, coder/gain_shape_quantize.py:275-291:
    ceil(log2(N + eps)) -- which is 1, not 0, for K = 0."""
    key = (l, max_bits)
    if key not in _kcache:
        cap = 2 ** max_bits
        k = 1
        while codebook_size(l, k) <= cap:
            k += 1
        n = codebook_size(l, k - 1)
        width = int(np.ceil(np.log2(n + EPS)))
        _kcache[key] = (k - 1, width)
    return _kcache[key]


# ------------------------------------------------------------- PVQ search / index
def pvq_search(x, k):
    """Project a unit vector on the pyramid sum|y| = k,
    coder/gain_shape_quantize.py:30-54: scale to L1 = k, floor, then hand the
    missing pulses one at a time to the largest remainder (first index wins),
    finally copy the signs (an exact zero of x erases its pulse)."""
    l1 = np.sum(np.abs(x))
    target = np.abs(k * x / l1)
    y = np.floor(target)
    missing = k - np.sum(abs(y))
    while missing > 0:
        y[np.argmax(abs(target) - y)] += 1
        missing -= 1
    y *= np.sign(x)
    return y.astype(int)


def pvq_index(y, k_total):
    """Enumeration index of a pyramid vector, coder/gain_shape_quantize.py:105-124.
    Walks the components with l dimensions and k pulses left; a component of
    magnitude a >= 1 skips N(l-1,k) (the a=0 vectors), both signs of magnitudes
    1..a-1, and the positive a-vectors if it is negative."""
    b = 0
    k = k_total
    l = len(y)
    for v in y:
        a = abs(int(v))
        if a >= 1:
            b += codebook_size(l - 1, k)
            for j in range(1, a):
                b += 2 * codebook_size(l - 1, k - j)
            if v < 0:
                b += codebook_size(l - 1, k - a)
        k -= a
        l -= 1
        if k == 0:
            break
    return int(b)


def quantize_shape_leaf(x, num_bits):
    """coder/gain_shape_quantize.py:244-256 -> (index, width)."""
    k, width = pulses_for_bits(len(x), num_bits)
    return pvq_index(pvq_search(x, k), k), width


# ---------------------------------------------------------- bit splits, mu-law
def gain_shape_alloc(r, l, k_fine=K_FINE):
    """coder/gain_shape_quantize.py:57-62."""
    r_gain = np.floor(r / l + 0.5 * np.log2(l) - k_fine)
    r_shape = max(r - r_gain, 0)
    return int(r_gain), int(r_shape)


def mid_side_alloc(a_mid_side, theta, length):
    """coder/gain_shape_quantize.py:302-312."""
    if theta == 0:
        a_mid = 0
    else:
        a_mid = int(np.floor(
            (a_mid_side - (length - 1) * np.log2(np.tan(abs(theta)) + EPS)) / 2))
    a_mid = min(max(a_mid, 0), a_mid_side)
    return a_mid, max(a_mid_side - a_mid, 0)


def mu_law(x, mu=255):
    """coder/gain_shape_quantize.py:294-295."""
    return np.sign(x) * np.log(1 + mu * np.abs(x)) / np.log(1 + mu)


def split_encode(x, bit_alloc, trace=None, depth=0):
    """coder/gain_shape_quantize.py:315-408.  More than 32 bits: fold the band
    into mid = (left+right)/2 and side = (left-right)/2 (an odd band pads the
    LEFT half with a trailing zero), send the angle atan(|S|/|M|) and recurse /
    PVQ the two unit vectors with the bits the angle leaves them."""
    if bit_alloc <= SPLIT_BITS:
        idx, width = quantize_shape_leaf(x, min(32, bit_alloc))
        if trace is not None:
            trace.append(('leaf', depth, len(x), bit_alloc, width))
        return [idx], [width]
    n = len(x)
    cut = n // 2
    half = int(np.ceil(n / 2))
    left = np.concatenate([x[:cut], [0]]) if half > cut else x[:cut]
    right = x[cut:]
    mid_v = (left + right) / 2
    side_v = (left - right) / 2
    mid_n = np.linalg.norm(mid_v)
    side_n = np.linalg.norm(side_v)
    m = mid_v / mid_n if mid_n != 0 else mid_v
    s = side_v / side_n if side_n != 0 else side_v
    theta = 0 if mid_n == 0 else np.arctan(side_n / mid_n)
    a_theta, a_rest = gain_shape_alloc(bit_alloc, half)
    theta_idx = po.quantize_uniform(theta / (np.pi / 2), a_theta)
    theta_q = po.dequantize_uniform(theta_idx, a_theta) * (np.pi / 2)
    a_mid, a_side = mid_side_alloc(a_rest, theta_q, half)
    if trace is not None:
        trace.append(('split', depth, n, bit_alloc, a_theta, a_mid, a_side))
    indices, bits = [theta_idx], [a_theta]
    for vec, a in ((m, a_mid), (s, a_side)):
        if a > SPLIT_BITS:
            i2, b2 = split_encode(vec, a, trace, depth + 1)
        elif a > 0:
            i1, b1 = quantize_shape_leaf(vec, a)
            if trace is not None:
                trace.append(('leaf', depth + 1, len(vec), a, b1))
            i2, b2 = [i1], [b1]
        else:
            i2, b2 = [], []
        indices += i2
        bits += b2
    return indices, bits


def quantize_gain_shape(x, bit_alloc, trace=None):
    """coder/gain_shape_quantize.py:476-512 -> (indices, widths); the gain
    index comes last and soaks up every bit the shape did not use."""
    l = len(x)
    bits_gain, bits_shape = gain_shape_alloc(bit_alloc, l)
    gain = np.linalg.norm(x)
    if gain == 0:
        return [0], [0]
    if bits_shape != 0:
        indices, bits = split_encode(x / gain, bits_shape, trace)
        bits_gain += bits_shape - sum(bits)
    else:
        indices, bits = [], []
    g = mu_law(gain / l)
    if bits_gain < 0:
        bits_gain = 0
    return indices + [po.quantize_uniform(g, bits_gain)], bits + [bits_gain]


# ------------------------------------------------------------ channel-frame encode
def make_params_vq(sample_rate, n_channels, kbps_per_channel):
    """The shipped driver's settings, coder/pacfile.py:699-707."""
    p = po.make_params(sample_rate, n_channels, kbps_per_channel)
    p.useVQ = True
    p.useSBR = kbps_per_channel < 128
    p.omittedBands = (list(po.omitted_bands(p.sfBands)) if p.useSBR else [])
    return p


def encode_channel_vq(data, p, last_trans=False, cur_trans=False,
                      next_trans=False, trace=None):
    """coder/codec.py:266-360 with useVQ: same MDCT / SMR / BitAlloc front end
    as the scalar path but the VQ budget rule (:292-294), then one gain-shape
    code per allocated band.  Returns (bitAlloc, indices, widths, overall)."""
    half_n = p.nMDCTLines
    max_mant = min(1 << p.nMantSizeBits, 16)
    bands = p.sfBandsShort if cur_trans else p.sfBands
    n_eff = int(1.45 * half_n) if cur_trans else half_n
    if last_trans or next_trans:
        n_eff = int(0.85 * n_eff)
    budget = p.targetBitsPerSample * n_eff
    budget -= p.nScaleBits
    budget -= p.nMantSizeBits * bands.nBands

    windowed = po.apply_window(data, last_trans, cur_trans, next_trans)
    lines = po.mdct_forward(windowed, half_n, half_n)[:half_n]
    overall = po.scale_factor(np.max(np.abs(lines)), p.nScaleBits)
    lines *= (1 << overall)
    smr = po.calc_smrs(data, lines, overall, p.sampleRate, bands)
    alloc = po.bit_alloc(budget, max_mant, bands.nBands, bands.nLines, smr)
    all_idx, all_bits = [], []
    for b in range(bands.nBands):
        if alloc[b]:
            lo, hi = bands.lowerLine[b], bands.upperLine[b] + 1
            idx, bits = quantize_gain_shape(
                lines[lo:hi], int(alloc[b] * bands.nLines[b]), trace)
            if sum(bits) == 0:
                alloc[b] = 0
            else:
                all_idx.append(idx)
                all_bits.append(bits)
    return alloc, all_idx, all_bits, overall


def encode_channel_sbr_vq(data, p, last_trans=False, cur_trans=False,
                          next_trans=False, trace=None, carries=None):
    """coder/codec.py:426-531 with useVQ (long blocks only; the driver sends
    short blocks through encode_channel_vq, coder/pacfile.py:639-643).
    Differences from encode_channel_vq: budget from the full halfN whatever the
    flags (:446), the overall scale also covers |rfft(Hann x)|/halfN (:459-472),
    omitted bands count as ONE line from the first call on (BitAlloc_SBR writes
    nLines in place, coder/bitalloc.py:141-143) and are coded as the gain of
    the mean FFT magnitude over the band (:503-505), and the spill rule
    (:522-524) that compares a band's total bits with its per-line allocation."""
    half_n = p.nMDCTLines
    max_mant = min(1 << p.nMantSizeBits, 16)
    bands = p.sfBands
    omitted = p.omittedBands
    budget = p.targetBitsPerSample * half_n
    budget -= p.nScaleBits
    budget -= p.nMantSizeBits * bands.nBands

    windowed = po.apply_window(data, last_trans, cur_trans, next_trans)
    lines = po.mdct_forward(windowed, half_n, half_n)[:half_n]
    fft_mag = np.abs(np.fft.rfft(po.hann_window(len(data)) * data)) / half_n
    peak = max(np.max(np.abs(lines)), np.max(fft_mag))
    overall = po.scale_factor(peak, p.nScaleBits)
    lines *= (1 << overall)
    fft_mag *= (1 << overall)
    smr = po.calc_smrs(data, lines, overall, p.sampleRate, bands)
    for b in omitted:
        bands.nLines[b] = 1
    alloc = po.bit_alloc(budget, max_mant, bands.nBands, bands.nLines, smr)
    all_idx, all_bits = [], []
    for b in range(bands.nBands):
        if not alloc[b]:
            continue
        lo, hi = bands.lowerLine[b], bands.upperLine[b] + 1
        if b in omitted:
            x = np.array([np.mean(np.abs(fft_mag[lo:hi]))])
        else:
            x = lines[lo:hi]
        idx, bits = quantize_gain_shape(x, int(alloc[b] * bands.nLines[b]), trace)
        if sum(bits) == 0:
            alloc[b] = 0
        else:
            all_idx.append(idx)
            all_bits.append(bits)
        if sum(bits) < alloc[b] and b < bands.nBands - 2:
            if carries is not None:
                carries.append(b)
            alloc[b + 1] += alloc[b] - sum(bits)
    return alloc, all_idx, all_bits, overall


# -------------------------------------------------------------------- .pac layout
def block_bits_vq(p, alloc, cur_trans=False):
    """coder/pacfile.py:342-361 as the VQ writer uses it: the size still counts
    a scale factor per band although none is written."""
    bands = p.sfBandsShort if cur_trans else p.sfBands
    n = p.nScaleBits
    for b in range(bands.nBands):
        n += p.nMantSizeBits + p.nScaleBits
        if alloc[b]:
            if b in p.omittedBands and not cur_trans:
                n += alloc[b]
            else:
                n += alloc[b] * bands.nLines[b]
    return int(n)


def write_block_body_vq(bw, p, overall, alloc, indices, widths, cur_trans=False):
    """coder/pacfile.py:363-402: overall scale, ALL allocations first, then the
    index lists of the allocated bands (an omitted band sends its last entry)."""
    bands = p.sfBandsShort if cur_trans else p.sfBands
    bw.put(overall, p.nScaleBits)
    for b in range(bands.nBands):
        ba = int(alloc[b])
        bw.put(ba - 1 if ba else 0, p.nMantSizeBits)
    at = 0
    for b in range(bands.nBands):
        if alloc[b]:
            if b in p.omittedBands and not cur_trans:
                bw.put(indices[at][-1], widths[at][-1])
            else:
                for v, w in zip(indices[at], widths[at]):
                    bw.put(v, w)
            at += 1


def pack_channel_block_vq(p, flags, parts):
    """parts = [(alloc, indices, widths, overall)] (1 long / 8 short)."""
    last_t, cur_t, next_t = flags
    n_bits = sum(block_bits_vq(p, a, bool(cur_t)) for (a, _, _, _) in parts) + 4
    n_bytes = n_bits // 8 if n_bits % 8 == 0 else n_bits // 8 + 1
    bw = po.BitWriter(n_bytes)
    bw.put(last_t, 1)
    bw.put(cur_t, 1)
    bw.put(next_t, 1)
    for (alloc, indices, widths, overall) in parts:
        write_block_body_vq(bw, p, overall, alloc, indices, widths, bool(cur_t))
    return n_bytes, bw.bytes()


def encode_hop_vq(p, prior, hop, flags, trace=None, carries=None):
    """coder/pacfile.py:460-520 + :627-643 routing: long blocks of an SBR file
    go to the SBR encoder, everything else to the plain VQ encoder."""
    last_t, cur_t, next_t = flags
    n_ch = p.nChannels
    full = [np.concatenate((prior[ch], hop[ch])) for ch in range(n_ch)]
    if not cur_t:
        if p.useSBR:
            return [[encode_channel_sbr_vq(full[ch], p, last_t, cur_t, next_t,
                                           trace, carries)]
                    for ch in range(n_ch)]
        return [[encode_channel_vq(full[ch], p, last_t, cur_t, next_t, trace)]
                for ch in range(n_ch)]
    long_n = p.nMDCTLines
    short_n = po.SHORT_LINES
    pad = long_n // 2 - short_n // 2
    per_ch = [[] for _ in range(n_ch)]
    p.nMDCTLines = p.nSamplesPerBlock = short_n
    try:
        for n in range(pad, 2 * long_n - short_n - pad, short_n):
            for ch in range(n_ch):
                if np.all(full[ch][n:n + 2 * short_n] == 0):
                    return None
            for ch in range(n_ch):
                per_ch[ch].append(encode_channel_vq(
                    full[ch][n:n + 2 * short_n], p, last_t, cur_t, next_t, trace))
    finally:
        p.nMDCTLines = p.nSamplesPerBlock = long_n
    return per_ch


def encode_stream_vq(pcm, sample_rate, kbps_per_channel, block_switching=True,
                     max_hops=None, collect=None, header_samples=None,
                     trace=None, carries=None):
    """Whole-file encode with the shipped driver's settings
    (coder/pacfile.py:699-757 + Close :612-625): useVQ on, useSBR below
    128 kb/s, transient detector on.  Same loop quirks as
    pac_oracle.encode_stream."""
    pcm = np.asarray(pcm)
    n_samples, n_ch = pcm.shape
    p = make_params_vq(sample_rate, n_ch, kbps_per_channel)
    hop_n = p.nMDCTLines
    out = [po.pac_header(p, n_samples if header_samples is None else header_samples)]
    n_hops = -(-n_samples // hop_n)
    if max_hops is not None:
        n_hops = min(n_hops, max_hops)
    prior = [np.zeros(hop_n) for _ in range(n_ch)]
    look = np.zeros((n_ch, 2 * hop_n))
    last_t = cur_t = False

    def emit(hop, flags):
        nonlocal prior
        parts = encode_hop_vq(p, prior, hop, flags, trace, carries)
        prior = hop
        if collect is not None:
            collect.append((flags, parts))
        if parts is None:
            return
        for ch in range(n_ch):
            n_bytes, payload = pack_channel_block_vq(p, flags, parts[ch])
            out.append(struct.pack('<L', int(n_bytes)))
            out.append(payload)

    for h in range(n_hops + 1):
        if h < n_hops:
            chunk = pcm[h * hop_n:(h + 1) * hop_n]
            if len(chunk) < hop_n:
                chunk = np.concatenate(
                    (chunk, np.zeros((hop_n - len(chunk), n_ch), pcm.dtype)))
            data = np.array([po.pcm16_to_fraction(chunk[:, ch])
                             for ch in range(n_ch)])
            look = np.concatenate((np.copy(data), look[:, hop_n:]), axis=1)
            nxt = po.transient_detect(look) if block_switching else False
        else:
            nxt = False
        hop = look[:, :hop_n]
        emit([hop[ch] for ch in range(n_ch)], (last_t, cur_t, nxt))
        last_t, cur_t = cur_t, nxt
    emit([np.zeros(hop_n) for _ in range(n_ch)], (False, False, False))
    return b''.join(out)


# =========================================================================
# decode side of the gain-shape / SBR variants
#   coder/gain_shape_quantize.py:127-176, 259-272, 411-473, 515-541
#   coder/codec.py:47-92 (useVQ branch), :95-222 (Decode_SBR)
#   coder/pacfile.py:177-229, 645-668
# Third-party arithmetic restated here (the reference calls SciPy, no version
# pinned; checked against scipy 1.15.3 in tests/test_oracle_vq.py):
#   scipy.ndimage.gaussian_filter1d(x, sigma=200)  -> gaussian_smooth()
#   scipy.interpolate.interp1d(kind='slinear')     -> slinear()
# =========================================================================
def pvq_decode(b, l_total, k_total):
    """Index -> pyramid vector: the five-step machine of
    coder/gain_shape_quantize.py:127-176 written as one loop.  At component i
    with l dimensions and k pulses left and xb the first index of the current
    group: b == xb ends the walk (the LAST component takes the k pulses left,
    positive); b below xb + N(l-1,k) is a zero; otherwise the magnitude j
    is the first whose two sign groups reach past b."""
    x = np.zeros(l_total, dtype=int)
    i, xb, k, l = 0, 0, k_total, l_total
    while True:
        if b == xb:                                    # step 1 -> step 5
            x[i] = 0
            if k > 0:
                x[l_total - 1] = k - abs(x[i])
            return x
        if b < xb + codebook_size(l - 1, k):           # step 2, zero
            x[i] = 0
        else:
            xb += codebook_size(l - 1, k)
            j = 1
            restart = False
            while True:                                # step 3
                j = min(j, k)
                if b < xb + 2 * codebook_size(l - 1, max(k - j, 0)):
                    if b >= xb + codebook_size(l - 1, k - j):
                        x[i] = -j
                    elif xb <= b:
                        x[i] = j
                    break
                xb += 2 * codebook_size(l - 1, k - j)
                if j < k:
                    j += 1
                else:
                    restart = True                     # -> step 1 with the same i
                    break
            if restart:
                continue
        k -= abs(x[i])                                 # step 4
        l -= 1
        i += 1
        if k <= 0:
            return x                                   # step 5 with k == 0 does nothing


def dequantize_leaf(idx, l, k):
    """coder/gain_shape_quantize.py:259-272: decoded vector over its L2 norm."""
    x = pvq_decode(idx, l, k).astype(float)
    n = np.linalg.norm(x)
    if n != 0:
        x = x / np.linalg.norm(x)
    return x


def split_decode(br, num_bits, band_size):
    """coder/gain_shape_quantize.py:411-473 -> (unit vector, bits read)."""
    if num_bits <= SPLIT_BITS:
        k, width = pulses_for_bits(band_size, num_bits)
        x = dequantize_leaf(br.get(width), band_size, k)
        n = np.linalg.norm(x)
        if n != 0:
            x /= n
        return x, width
    half = int(np.ceil(band_size / 2))
    a_theta, a_rest = gain_shape_alloc(num_bits, half)
    theta = po.dequantize_uniform(br.get(a_theta), a_theta) * (np.pi / 2)
    a_mid, a_side = mid_side_alloc(a_rest, theta, half)
    used = a_theta
    parts = []
    for a in (a_mid, a_side):
        if a > SPLIT_BITS:
            v, u = split_decode(br, a, half)
            used += u
        elif a > 0:
            k, width = pulses_for_bits(half, a)
            v = dequantize_leaf(br.get(width), half, k)
            used += width
        else:
            v = np.zeros((half,))
        parts.append(v)
    mid, side = parts
    left = mid * np.cos(theta) + side * np.sin(theta)
    right = mid * np.cos(theta) - side * np.sin(theta)
    left /= np.sqrt(2)
    right /= np.sqrt(2)
    if half > band_size // 2:
        left = left[:len(left) - 1]
    x = np.concatenate([left, right])
    n = np.linalg.norm(x)
    if n != 0:
        x /= n
    return x, used


def inv_mu_law(y, mu=255):
    """coder/gain_shape_quantize.py:298-299."""
    return np.sign(y) / mu * ((1 + mu) ** np.abs(y) - 1)


def dequantize_gain_shape(br, bit_alloc, l):
    """coder/gain_shape_quantize.py:515-541."""
    bits_gain, bits_shape = gain_shape_alloc(bit_alloc, l)
    if bits_shape != 0:
        shape, used = split_decode(br, bits_shape, l)
        bits_gain += bits_shape - used
    else:
        shape = np.ones((l,))
    g = po.dequantize_uniform(br.get(bits_gain), bits_gain)
    return (inv_mu_law(g) * l) * shape


def gaussian_smooth(x, sigma=200, truncate=4.0):
    """scipy.ndimage.gaussian_filter1d(x, sigma) with its defaults (order 0,
    mode 'reflect', truncate 4): weights exp(-0.5 (j/sigma)^2) normalised by
    their sum over j = -r..r, r = int(truncate*sigma + 0.5); the input is
    extended by mirroring about its edges (d c b a | a b c d | d c b a); each
    output is centre*w0 followed by the symmetric pairs from the far end
    inwards (NI_Correlate1D's symmetric branch)."""
    x = np.asarray(x, dtype=np.float64)
    r = int(truncate * float(sigma) + 0.5)
    j = np.arange(-r, r + 1)
    w = np.exp(-0.5 / (sigma * sigma) * j ** 2)
    w = w / w.sum()
    n = len(x)
    idx = np.arange(-r, n + r)
    period = 2 * n
    m = np.mod(idx, period)
    m = np.where(m >= n, period - 1 - m, m)
    ext = x[m]
    out = np.empty(n)
    for i in range(n):
        c = i + r
        acc = ext[c] * w[r]
        for jj in range(-r, 0):
            acc += (ext[c + jj] + ext[c - jj]) * w[r + jj]
        out[i] = acc
    return out


def gaussian_smooth_sparse(x, sigma=200, truncate=4.0):
    """Same result as gaussian_smooth for inputs that are mostly exact zeros
    (adding (0+0)*w leaves the accumulator unchanged): only the pairs that
    touch a non-zero sample are added, in the same far-to-near order."""
    x = np.asarray(x, dtype=np.float64)
    r = int(truncate * float(sigma) + 0.5)
    j = np.arange(-r, r + 1)
    w = np.exp(-0.5 / (sigma * sigma) * j ** 2)
    w = w / w.sum()
    n = len(x)
    period = 2 * n
    nz = np.nonzero(x)[0]
    out = np.zeros(n)
    for i in range(n):
        # source positions (in the mirrored extension) holding non-zero samples
        offs = set()
        for s in nz:
            for base in range(-period * ((r // period) + 2), period * ((r // period) + 3), period):
                for pos in (base + s, base + period - 1 - s):
                    d = pos - i
                    if -r <= d <= r:
                        offs.add(abs(d))
        acc = x[i] * w[r]
        for d in sorted(offs, reverse=True):
            if d == 0:
                continue
            def at(p):
                q = p % period
                return x[q] if q < n else x[period - 1 - q]
            acc += (at(i - d) + at(i + d)) * w[r - d]
        out[i] = acc
    return out


def slinear(xs, ys, xq):
    """scipy.interpolate.interp1d(xs, ys, kind='slinear')(xq): the order-1
    B-spline through the points (knots = xs with both ends doubled,
    coefficients = ys), evaluated by de Boor's recursion: on
    [xs[j], xs[j+1]]  w = 1/((xs[j+1]-x) + (x-xs[j])),
    value = ys[j]*(w*(xs[j+1]-x)) + ys[j+1]*(w*(x-xs[j]))."""
    xs = np.asarray(xs, dtype=np.float64)
    ys = np.asarray(ys, dtype=np.float64)
    out = np.empty(len(xq))
    for i, x in enumerate(xq):
        j = int(np.searchsorted(xs, x, side='right')) - 1
        j = min(max(j, 0), len(xs) - 2)
        xb = xs[j + 1] - x
        xa = x - xs[j]
        w = 1.0 / (xb + xa)
        out[i] = 0.0 + ys[j] * (w * xb) + ys[j + 1] * (w * xa)
    return out


def decode_lines_vq(br, p, alloc, cur_t, sbr):
    """The line loop of coder/codec.py:59-76 / :117-134 (useVQ)."""
    bands = p.sfBandsShort if cur_t else p.sfBands
    lines = np.zeros(p.nMDCTLines, dtype=np.float64)
    at = 0
    for b in range(bands.nBands):
        n = bands.nLines[b]
        if sbr and b in p.omittedBands:
            n = 1
        if alloc[b]:
            lines[at:at + n] = dequantize_gain_shape(br, int(alloc[b] * n), n)
        at += n
    return lines


def sbr_reconstruct(lines, p):
    """coder/codec.py:136-198: the band values sit at the first lines above the
    cut (one per omitted band, the rest of the 'envelope' is zero), get
    smoothed, the lower half of the spectrum is transposed up by linear
    interpolation and each omitted band is scaled to the smoothed envelope
    over its mean magnitude."""
    bands = p.sfBands
    cut = bands.lowerLine[p.omittedBands[0]]
    n_omit = len(lines) - cut
    env = np.append(lines[cut:], np.zeros(n_omit))
    smooth = gaussian_smooth(env) if np.count_nonzero(env) > 8 else gaussian_smooth_sparse(env)
    up = int(math.floor(len(lines) / n_omit))
    spacing = p.sampleRate / (2 * p.nMDCTLines)
    freqs = (np.arange(p.nMDCTLines) + 1 / 2) * spacing
    ii = np.arange(cut // up - 1, len(lines) // up + 1)
    lines[cut:] = slinear(freqs[ii], lines[ii], freqs[cut:] / up)
    for b in p.omittedBands:
        lo, hi = bands.lowerLine[b], bands.upperLine[b] + 1
        if np.max(np.abs(lines[lo:hi])) > 0:
            lines[lo:hi] *= smooth[lo - cut:hi - cut] / np.mean(np.abs(lines[lo:hi]))
    return lines


def decode_block_vq(br, p, last_t, cur_t, next_t):
    """getDecodedBlock + Decode / Decode_SBR for one (sub-)block of a VQ stream
    (coder/pacfile.py:177-229, 645-668)."""
    bands = p.sfBandsShort if cur_t else p.sfBands
    overall = br.get(p.nScaleBits)
    alloc = []
    for b in range(bands.nBands):
        a = br.get(p.nMantSizeBits)
        alloc.append(a + 1 if a else 0)
    sbr = bool(p.useSBR and not cur_t and
               np.any(np.array(alloc)[np.array(p.omittedBands, dtype=int)] != 0))
    lines = decode_lines_vq(br, p, alloc, cur_t, sbr)
    if sbr:
        lines = sbr_reconstruct(lines, p)
    lines /= 1. * (1 << overall)
    half_n = p.nMDCTLines
    win = po.window_table(po.window_kind(last_t, cur_t, next_t), 2 * half_n)
    return win * po.mdct_inverse(lines, half_n, half_n)


def decode_stream_vq(data, max_blocks=None):
    """Whole VQ / VQ+SBR .pac -> int16 [n, nCh] (coder/pacfile.py:231-298 +
    coder/pcmfile.py:127-134), as pac_oracle.decode_stream does for scalar files."""
    (sr, n_ch, n_samples, n_lines, n_scale, n_mant_size, use_sbr, use_vq) = struct.unpack(
        '<LHLLHHHH', data[4:4 + struct.calcsize('<LHLLHHHH')])
    assert data[:4] == b'PAC ' and use_vq and n_lines == 1024
    pos = 4 + struct.calcsize('<LHLLHHHH')
    n_bands = struct.unpack('<L', data[pos:pos + 4])[0]
    pos += 4 + 2 * n_bands
    p = po.make_params(sr, n_ch, 128, n_lines, n_scale, n_mant_size)
    p.useVQ, p.useSBR = True, bool(use_sbr)
    p.omittedBands = list(po.omitted_bands(p.sfBands)) if p.useSBR else []
    hop = p.nMDCTLines
    ola = [np.zeros(hop) for _ in range(n_ch)]
    out = []
    n_done = 0
    while pos < len(data) and (max_blocks is None or n_done < max_blocks):
        hop_out = []
        for ch in range(n_ch):
            n_bytes = struct.unpack('<L', data[pos:pos + 4])[0]
            br = po.BitReader(data[pos + 4:pos + 4 + n_bytes] + b'\0' * 8)
            pos += 4 + n_bytes
            last_t, cur_t, next_t = br.get(1), br.get(1), br.get(1)
            if not cur_t:
                block = decode_block_vq(br, p, last_t, cur_t, next_t)
            else:
                block = np.zeros(2 * hop)
                p.nMDCTLines = p.nSamplesPerBlock = po.SHORT_LINES
                try:
                    pad = hop // 2 - po.SHORT_LINES // 2
                    for n in range(pad, 2 * hop - po.SHORT_LINES - pad, po.SHORT_LINES):
                        block[n:n + 2 * po.SHORT_LINES] += decode_block_vq(br, p, last_t, cur_t, next_t)
                finally:
                    p.nMDCTLines = p.nSamplesPerBlock = hop
            hop_out.append(np.add(ola[ch], block[:hop]))
            ola[ch] = block[hop:]
        out.append(np.stack([po.fraction_to_pcm16(h) for h in hop_out], axis=1))
        n_done += 1
    out.append(np.stack([po.fraction_to_pcm16(o) for o in ola], axis=1))
    return np.concatenate(out)
