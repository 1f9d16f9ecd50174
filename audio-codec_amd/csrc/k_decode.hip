/*
 * k_decode.hip -- the decode side, scalar-mantissa coder:
 *
 *   k_unpack   : MSB-first payload of one channel-block -> flags, overall scale,
 *                bit allocation, scale factors, line-indexed mantissas
 *                (coder/pacfile.py:185-213, 264-266; the inverse of k_pack)
 *   k_imdct<M> : vDequantize (coder/quantize.py:254-274) -> / 2^overall -> IMDCT
 *                (coder/mdct.py:56-62) -> window (coder/codec.py:81-89); short
 *                frames: eight sub-blocks overlap-added at n = 448 + 128 j
 *                (coder/pacfile.py:272-287).  Output: the 2048-sample block
 *                before overlap-and-add.
 *   k_ola_pcm  : hop h = second half of block h-1 + first half of block h
 *                (coder/pacfile.py:289-295; the last half is flushed at EOF,
 *                :245-249), then the 16-bit PCM mapping of coder/pcmfile.py:127-134.
 *
 * The IMDCT reuses the forward machinery: y = 2 * unfold(DCT4(X)) where DCT4 is the
 * same "pre-twiddle, N/4-point complex FFT, post-twiddle" pipeline the MDCT
 * kernels apply to the folded input (tools/proto_wave_fft.py:imdct_via_dct4).
 */
#include "pacx_dev.h"
#include "wave_fft.h"

#define UNPACK_WORDS 552

__device__ __forceinline__ unsigned get_bits(const unsigned *words, int pos, int width)
{
    /* stream bit p lives in word p>>5 at bit 31-(p&31) (MSB first) */
    if (width <= 0)
        return 0u;
    const int w = pos >> 5, o = pos & 31;
    const unsigned long long two = ((unsigned long long)words[w] << 32) | words[w + 1];
    return (unsigned)((two >> (64 - o - width)) & ((1ull << width) - 1ull));
}

/* ------------------------------------------------------------------ unpack */
/* The bit cursor is driven by what the payload says (a 12-bit allocation field, band after
 * band), so a truncated or corrupt record must not steer it: every field is checked against
 * the record's own length before it is read, an allocation above maxMantBits = 16
 * (coder/codec.py:296) cannot have been written by the coder, and nothing beyond n_bytes is
 * ever fetched.  A record that fails gets PACX_ST_MALFORMED and all-zero codes -- the
 * reference stops with "Only read a partial block of coded PACFile data"
 * (coder/pacfile.py:203-205); the host mirrors raise the same on the status bit. */
__global__ __launch_bounds__(64) void k_unpack(PacxTables T, long long n_cf, const uint8_t *__restrict__ payload,
                                              int payload_stride, const long long *__restrict__ offsets,
                                              const int32_t *__restrict__ n_bytes,
                                              uint8_t *__restrict__ flags_out, int32_t *__restrict__ overall,
                                              int32_t *__restrict__ scale_factor, int32_t *__restrict__ bit_alloc,
                                              int32_t *__restrict__ mantissa, uint32_t *__restrict__ status)
{
    __shared__ unsigned words[UNPACK_WORDS];
    __shared__ int offs[PACX_SUB][PACX_MAX_BANDS], bas[PACX_SUB][PACX_MAX_BANDS];
    __shared__ int is_short_s, bad_s;
    const int lane = threadIdx.x;
    const long long cf = blockIdx.x;
    if (cf >= n_cf)
        return;
    int nbytes = n_bytes[cf];
    bool bad = nbytes < 1 || nbytes > 4 * (UNPACK_WORDS - 2);     /* no coder output is that long */
    if (bad)
        nbytes = 0;
    const uint8_t *src = payload + (offsets ? offsets[cf] : cf * (long long)payload_stride);
    const int n_words = (nbytes + 3) >> 2;
    for (int i = lane; i < UNPACK_WORDS; i += 64) {
        unsigned v = 0;
        if (i < n_words) {
            /* byte loads: the record may start at any byte offset of a file body */
            const int b0 = 4 * i;
            v = ((unsigned)src[b0] << 24) | ((b0 + 1 < nbytes ? (unsigned)src[b0 + 1] : 0u) << 16) |
                ((b0 + 2 < nbytes ? (unsigned)src[b0 + 2] : 0u) << 8) | (b0 + 3 < nbytes ? (unsigned)src[b0 + 3] : 0u);
        }
        words[i] = v;
    }
    __syncthreads();
    int32_t *ba_o = bit_alloc + cf * T.band_stride, *sf_o = scale_factor + cf * T.band_stride;
    if (lane == 0) {
        const int limit = 8 * nbytes;                     /* bits the record really holds */
        const unsigned fl = bad ? 0u
                                : get_bits(words, 0, 1) | (get_bits(words, 1, 1) << 1) | (get_bits(words, 2, 1) << 2);
        const int shrt = (fl >> 1) & 1;
        const int nb = shrt ? T.nb_short : T.nb_long;
        const int32_t *cnt = shrt ? T.band_lines_short : T.band_lines_long;
        const int head = T.n_mant_size_bits + T.n_scale_bits;
        const bool one_code = T.use_sbr && !T.use_vq && !shrt;
        int pos = 3;
        for (int s = 0; s < (shrt ? PACX_SUB : 1) && !bad; ++s) {
            if (pos + T.n_scale_bits > limit) {
                bad = true;
                break;
            }
            overall[cf * PACX_SUB + s] = (int)get_bits(words, pos, T.n_scale_bits);
            pos += T.n_scale_bits;
            for (int b = 0; b < nb; ++b) {
                if (pos + head > limit) {
                    bad = true;
                    break;
                }
                int a = (int)get_bits(words, pos, T.n_mant_size_bits);
                if (a)
                    a += 1;
                const int sf = (int)get_bits(words, pos + T.n_mant_size_bits, T.n_scale_bits);
                pos += head;
                /* scalar mantissas in an SBR file: a coded omitted band of a long block carries ONE mantissa,
                   which the reader assigns to every line of the band (coder/pacfile.py:203-205, 212-213) */
                const int n_codes = (one_code && b >= T.first_omitted) ? 1 : cnt[b];
                if (a > 16 || pos + a * n_codes > limit) {
                    bad = true;
                    break;
                }
                offs[s][b] = pos;
                bas[s][b] = a;
                if (b == nb - 1 && nb < PACX_MAX_BANDS)
                    bas[s][nb] = 0;                        /* dummy band of the lines no band covers */
                ba_o[s * nb + b] = a;
                sf_o[s * nb + b] = sf;
                pos += a * n_codes;
            }
        }
        if (bad) {                                         /* all-zero codes for the whole record */
            for (int s = 0; s < PACX_SUB; ++s) {
                overall[cf * PACX_SUB + s] = 0;
                for (int b = 0; b <= nb && b < PACX_MAX_BANDS; ++b)
                    bas[s][b] = 0;
            }
            for (int i = 0; i < T.band_stride; ++i) {
                ba_o[i] = 0;
                sf_o[i] = 0;
            }
        } else if (!shrt) {
            for (int s = 1; s < PACX_SUB; ++s)
                overall[cf * PACX_SUB + s] = 0;
        }
        flags_out[cf] = (uint8_t)fl;
        is_short_s = shrt;
        bad_s = bad ? 1 : 0;
        if (status)
            status[cf] = bad ? 32u : 0u;                   /* PACX_ST_MALFORMED */
    }
    __syncthreads();
    const int shrt = is_short_s;
    const uint8_t *band_of = shrt ? T.line_band_short : T.line_band_long;
    const int32_t *lower = shrt ? T.band_lower_short : T.band_lower_long;
    const int m_lines = shrt ? PACX_M_SHORT : PACX_M_LONG;
    for (int k = lane; k < PACX_M_LONG; k += 64) {
        const int s = shrt ? k / PACX_M_SHORT : 0;
        const int kk = k - s * m_lines;
        const int b = band_of[kk];
        const int a = bad_s ? 0 : bas[s][b];
        const int idx = (T.use_sbr && !T.use_vq && !shrt && b >= T.first_omitted) ? 0 : kk - lower[b];
        mantissa[cf * PACX_M_LONG + k] = a ? (int32_t)get_bits(words, offs[s][b] + idx * a, a) : 0;
    }
}

/* one dequantised line: vDequantize (coder/quantize.py:260-274, pacx_exact.h) then / 2^overall */
__device__ __forceinline__ double dequant_line(int mant, int scale, int ba, int n_scale_bits, int overall)
{
    if (!ba)
        return 0.0;
    return pacx_dequantize(mant, scale, n_scale_bits, ba) / (double)(1 << overall);
}

/* ------------------------------------------- scalar mantissas in an SBR file */
/* PACFile.Decode (coder/pacfile.py:645-668) sends a long block with bits in an omitted band to
 * Decode_SBR, whose line loop (coder/codec.py:117-134, useVQ off) counts ONE line for an omitted
 * band: band b's value lands on line cut + (b - first omitted), dequantised from the mantissa the
 * reader left at THAT line (the reader broadcast each coded omitted band's one mantissa over the
 * band's own lines, k_unpack).  Every other block is codec.Decode's.  This kernel writes the
 * dequantised lines of every channel-block BEFORE the division by 2^overall (k_sbr_recon works on
 * those, then k_imdct_* divide, as for gain-shape streams) and says which blocks k_sbr_recon takes. */
__global__ __launch_bounds__(64) void k_sbr_scalar_lines(PacxTables T, long long n_cf, const uint8_t *__restrict__ cf_flags,
                                                        const int32_t *__restrict__ scale_factor,
                                                        const int32_t *__restrict__ bit_alloc,
                                                        const int32_t *__restrict__ mantissa,
                                                        double *__restrict__ lines, uint8_t *__restrict__ sbr_flag,
                                                        int routing)
{
    /* routing 0: codec.Decode for every block; 1: PACFile.Decode's rule; 2: Decode_SBR for every long block */
    const int lane = threadIdx.x;
    const long long cf = blockIdx.x;
    if (cf >= n_cf)
        return;
    const int32_t *ba = bit_alloc + cf * T.band_stride, *sf = scale_factor + cf * T.band_stride;
    const int32_t *mant = mantissa + cf * PACX_M_LONG;
    double *out = lines + cf * PACX_M_LONG;
    if (cf_flags[cf] & 2u) {
        for (int k = lane; k < PACX_M_LONG; k += 64) {
            const int s = k / PACX_M_SHORT, kk = k % PACX_M_SHORT;
            const int b = T.line_band_short[kk];
            const int a = (b < T.nb_short) ? ba[s * T.nb_short + b] : 0;
            out[k] = a ? pacx_dequantize(mant[k], sf[s * T.nb_short + b], T.n_scale_bits, a) : 0.0;
        }
        if (lane == 0)
            sbr_flag[cf] = 0;
        return;
    }
    const bool coded = lane >= T.first_omitted && lane < T.nb_long && ba[lane] != 0;
    const bool sbr = routing == 2 || (routing == 1 && __builtin_amdgcn_ballot_w64(coded) != 0ull);
    const int cut = T.band_lower_long[T.first_omitted];
    for (int k = lane; k < PACX_M_LONG; k += 64) {
        int b = T.line_band_long[k];
        if (sbr && k >= cut)
            b = T.first_omitted + (k - cut);               /* one line per omitted band, then nothing */
        const int a = (b < T.nb_long) ? ba[b] : 0;
        out[k] = a ? pacx_dequantize(mant[k], sf[b], T.n_scale_bits, a) : 0.0;
    }
    if (lane == 0)
        sbr_flag[cf] = sbr ? 1 : 0;
}

/* -------------------------------------------------------------- IMDCT long */
__global__ __launch_bounds__(64) void k_imdct_long(PacxTables T, long long n_cf, const uint8_t *__restrict__ cf_flags,
                                                  const int32_t *__restrict__ overall,
                                                  const int32_t *__restrict__ scale_factor,
                                                  const int32_t *__restrict__ bit_alloc,
                                                  const int32_t *__restrict__ mantissa,
                                                  const double *__restrict__ lines_in,
                                                  double *__restrict__ blocks, int plain)
{
    /* plain (mdct.IMDCT, coder/mdct.py:73-77): lines_in are the lines themselves -- no flags, no overall
       scale, no window: the bare 2 * unfold(DCT-IV).
       The lines, the FFT exchange tile and the DCT-IV output follow one another in time (the lines are in
       registers before the first exchange, the spectrum is in registers before the output is written) and
       share their LDS: 9 KB per wave instead of 17 -- a one-wave kernel bound by how many of its waves fit a CU */
    __shared__ __attribute__((aligned(16))) cplx tile[WFFT_TILE];
    static_assert(sizeof(cplx) * WFFT_TILE >= sizeof(double) * PACX_M_LONG, "the tile holds the lines");
    double *buf = (double *)tile;
    const int lane = threadIdx.x;
    const long long cf = blockIdx.x;
    if (cf >= n_cf)
        return;
    const unsigned fl = plain ? 0u : cf_flags[cf];
    if (fl & 2u)
        return;                                            /* short frame: k_imdct_short */
    const int ov = plain ? 0 : overall[cf * PACX_SUB];
    if (lines_in) {                                        /* gain-shape streams: lines come from k_vq_dec */
        const double rescale = (double)(1 << ov);
        for (int k = lane; k < PACX_M_LONG; k += 64)
            buf[k] = lines_in[cf * PACX_M_LONG + k] / rescale;
    } else {
        const int32_t *ba = bit_alloc + cf * T.band_stride, *sf = scale_factor + cf * T.band_stride;
        for (int k = lane; k < PACX_M_LONG; k += 64) {
            const int b = T.line_band_long[k];
            buf[k] = (b < T.nb_long) ? dequant_line(mantissa[cf * PACX_M_LONG + k], sf[b], ba[b], T.n_scale_bits, ov)
                                     : 0.0;
        }
    }
    __syncthreads();
    const int M = PACX_M_LONG, Q = M / 2;
    cplx v[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int n = lane + 64 * n1;
        v[n1] = c_mul(make_double2(buf[2 * n], buf[M - 1 - 2 * n]), T.tw_long[n]);
    }
    __syncthreads();
    fft512(v, tile, T.w512, lane);
    __syncthreads();                                       /* the tile is done with: it becomes the output buffer */
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) {
        const int k = fft512_out_index(lane, k3);
        const cplx y = c_mul(v[k3], T.tw_long[k]);
        buf[2 * k] = y.x;                                  /* DCT-IV of the lines */
        buf[M - 1 - 2 * k] = -y.y;
    }
    __syncthreads();
    /* y = 2 * unfold(DCT4), then the block's window */
    const double *__restrict__ w = T.win_long + pacx_window_kind(fl) * PACX_N_LONG;
    double *__restrict__ out = blocks + cf * PACX_N_LONG;
    for (int i = lane; i < PACX_N_LONG; i += 64) {
        double t;
        if (i < Q)
            t = buf[Q + i];
        else if (i < M)
            t = -buf[Q + (M - 1 - i)];
        else if (i < 3 * Q)
            t = -buf[3 * Q - 1 - i];
        else
            t = -buf[i - 3 * Q];
        out[i] = plain ? 2.0 * t : w[i] * (2.0 * t);
    }
}

/* ------------------------------------------------------------- IMDCT short */
__global__ __launch_bounds__(64) void k_imdct_short(PacxTables T, long long n_cf, const uint8_t *__restrict__ cf_flags,
                                                   const int32_t *__restrict__ overall,
                                                   const int32_t *__restrict__ scale_factor,
                                                   const int32_t *__restrict__ bit_alloc,
                                                   const int32_t *__restrict__ mantissa,
                                                   const double *__restrict__ lines_in,
                                                   double *__restrict__ blocks, int plain)
{
    /* plain: the eight 128-line rows of lines_in through the bare IMDCT, unwindowed, overlap-added at their
       positions 448 + 128 s (a lone row in sub-block 0 comes out at samples 448..703) */
    __shared__ __attribute__((aligned(16))) cplx tile[WFFT_TILE];      /* shared with the lines, as in k_imdct_long */
    double (*buf)[PACX_M_SHORT] = (double (*)[PACX_M_SHORT])tile;
    const int lane = threadIdx.x;
    const long long cf = blockIdx.x;
    if (cf >= n_cf)
        return;
    const unsigned fl = plain ? 2u : cf_flags[cf];
    if (!(fl & 2u))
        return;
    if (lines_in) {
        for (int k = lane; k < PACX_M_LONG; k += 64) {
            const int s = k / PACX_M_SHORT, kk = k % PACX_M_SHORT;
            buf[s][kk] = plain ? lines_in[cf * PACX_M_LONG + k]
                               : lines_in[cf * PACX_M_LONG + k] / (double)(1 << overall[cf * PACX_SUB + s]);
        }
    } else {
        const int32_t *ba = bit_alloc + cf * T.band_stride, *sf = scale_factor + cf * T.band_stride;
        for (int k = lane; k < PACX_M_LONG; k += 64) {
            const int s = k / PACX_M_SHORT, kk = k % PACX_M_SHORT;
            const int b = T.line_band_short[kk];
            buf[s][kk] = (b < T.nb_short)
                             ? dequant_line(mantissa[cf * PACX_M_LONG + k], sf[s * T.nb_short + b],
                                            ba[s * T.nb_short + b], T.n_scale_bits, overall[cf * PACX_SUB + s])
                             : 0.0;
        }
    }
    __syncthreads();
    const int g = lane >> 3, r = lane & 7;
    const int M = PACX_M_SHORT, Q = M / 2;
    cplx v[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int n = r + 8 * n1;
        v[n1] = c_mul(make_double2(buf[g][2 * n], buf[g][M - 1 - 2 * n]), T.tw_short[n]);
    }
    __syncthreads();
    fft64x8(v, tile, T.w512, lane);
    __syncthreads();
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) {
        const int k = fft64_out_index(lane, k3);
        const cplx y = c_mul(v[k3], T.tw_short[k]);
        buf[g][2 * k] = y.x;
        buf[g][M - 1 - 2 * k] = -y.y;
    }
    __syncthreads();
    /* sample i of the 2048 block: sum over the (at most two) sub-blocks that cover it,
       in ascending sub-block order as the reference's += does */
    const double *__restrict__ w = T.win_short;
    double *__restrict__ out = blocks + cf * PACX_N_LONG;
    for (int i = lane; i < PACX_N_LONG; i += 64) {
        double acc = 0.0;
        for (int s = 0; s < PACX_SUB; ++s) {
            const int ii = i - (PACX_SHORT_FIRST + PACX_M_SHORT * s);
            if (ii < 0 || ii >= PACX_N_SHORT)
                continue;
            double t;
            if (ii < Q)
                t = buf[s][Q + ii];
            else if (ii < M)
                t = -buf[s][Q + (M - 1 - ii)];
            else if (ii < 3 * Q)
                t = -buf[s][3 * Q - 1 - ii];
            else
                t = -buf[s][ii - 3 * Q];
            acc += plain ? 2.0 * t : w[ii] * (2.0 * t);
        }
        out[i] = acc;
    }
}

/* ------------------------------------------------- overlap-add + PCM mapping */
__global__ void k_ola_pcm(long long n_blocks, int n_ch, const double *__restrict__ blocks,
                          int16_t *__restrict__ pcm)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   /* (hop, sample, channel) */
    const long long total = (n_blocks + 1) * PACX_M_LONG * n_ch;
    if (idx >= total)
        return;
    const int ch = (int)(idx % n_ch);
    const long long t = idx / n_ch;
    const long long h = t / PACX_M_LONG;
    const int i = (int)(t % PACX_M_LONG);
    double s;
    if (h < n_blocks) {
        const double prev = (h >= 1) ? blocks[((h - 1) * n_ch + ch) * PACX_N_LONG + PACX_M_LONG + i] : 0.0;
        s = prev + blocks[(h * n_ch + ch) * PACX_N_LONG + i];
    } else {
        s = (h >= 1) ? blocks[((h - 1) * n_ch + ch) * PACX_N_LONG + PACX_M_LONG + i] : 0.0;   /* EOF flush */
    }
    const bool neg = signbit(s);
    const long long q = pacx_quant_mag(fabs(s), 16);
    pcm[idx] = (int16_t)(neg ? -q : q);
}

/* ------------------------------------------------------------- launchers */
void pacx_launch_unpack(const PacxTables &T, long long n_cf, const uint8_t *payload, int payload_stride,
                        const long long *offsets, const int32_t *n_bytes, uint8_t *flags_out, int32_t *overall,
                        int32_t *scale_factor, int32_t *bit_alloc, int32_t *mantissa, uint32_t *status,
                        hipStream_t st)
{
    if (n_cf > 0)
        hipLaunchKernelGGL(k_unpack, dim3((unsigned)n_cf), dim3(64), 0, st, T, n_cf, payload, payload_stride,
                           offsets, n_bytes, flags_out, overall, scale_factor, bit_alloc, mantissa, status);
}

void pacx_launch_decode(const PacxTables &T, long long n_blocks, int n_ch, const uint8_t *cf_flags,
                        const int32_t *overall, const int32_t *scale_factor, const int32_t *bit_alloc,
                        const int32_t *mantissa, const double *lines_in, double *blocks, int16_t *pcm,
                        hipStream_t st)
{
    const long long n_cf = n_blocks * n_ch;
    if (n_cf > 0) {
        hipLaunchKernelGGL(k_imdct_long, dim3((unsigned)n_cf), dim3(64), 0, st, T, n_cf, cf_flags, overall,
                           scale_factor, bit_alloc, mantissa, lines_in, blocks, 0);
        hipLaunchKernelGGL(k_imdct_short, dim3((unsigned)n_cf), dim3(64), 0, st, T, n_cf, cf_flags, overall,
                           scale_factor, bit_alloc, mantissa, lines_in, blocks, 0);
    }
    if (pcm) {
        const long long total = (n_blocks + 1) * PACX_M_LONG * n_ch;
        hipLaunchKernelGGL(k_ola_pcm, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, n_blocks, n_ch, blocks,
                           pcm);
    }
}

void pacx_launch_sbr_scalar_lines(const PacxTables &T, long long n_cf, const uint8_t *cf_flags,
                                  const int32_t *scale_factor, const int32_t *bit_alloc, const int32_t *mantissa,
                                  double *lines, uint8_t *sbr_flag, int routing, hipStream_t st)
{
    if (n_cf > 0)
        hipLaunchKernelGGL(k_sbr_scalar_lines, dim3((unsigned)n_cf), dim3(64), 0, st, T, n_cf, cf_flags, scale_factor,
                           bit_alloc, mantissa, lines, sbr_flag, routing);
}

/* mdct.IMDCT for rows of 1024 lines (short_blocks: rows of 8 x 128 lines) -> 2048 samples each, unwindowed */
void pacx_launch_imdct_plain(const PacxTables &T, long long n_rows, int short_blocks, const double *lines,
                             double *blocks, hipStream_t st)
{
    if (n_rows <= 0)
        return;
    if (short_blocks)
        hipLaunchKernelGGL(k_imdct_short, dim3((unsigned)n_rows), dim3(64), 0, st, T, n_rows, nullptr, nullptr, nullptr,
                           nullptr, nullptr, lines, blocks, 1);
    else
        hipLaunchKernelGGL(k_imdct_long, dim3((unsigned)n_rows), dim3(64), 0, st, T, n_rows, nullptr, nullptr, nullptr,
                           nullptr, nullptr, lines, blocks, 1);
}
