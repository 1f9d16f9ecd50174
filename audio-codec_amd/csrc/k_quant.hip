/*
 * k_quant.hip -- bit allocation, scale factors + mantissas, bit packing.
 *
 *   k_bitalloc : BitAlloc (coder/bitalloc.py:62-121) with the budget rule of
 *                coder/codec.py:288-299; one lane per (sub-)block -- the loop
 *                is serial, data dependent and tiny (<= 25 bands)
 *   k_quantize : per band ScaleFactor + vMantissa (coder/codec.py:362-377,
 *                coder/quantize.py:99-125, 229-250); one wave per (sub-)block
 *   k_pack     : MSB-first payload of one channel-block
 *                (coder/pacfile.py:404-447, 552-577; coder/bitpack.py:37-102)
 *   k_scan_x, k_copy_body : "<L nBytes" + payload of every cf back to back
 *                (coder/pacfile.py:566-568, 608)
 *
 * All arithmetic that decides an integer code goes through pacx_exact.h and is
 * compiled with -ffp-contract=off.
 */
#include "pacx_dev.h"
#include "wave_fft.h"   /* wave_max */
#include "quant_dev.h"

__global__ __launch_bounds__(64) void k_bitalloc(PacxTables T, const uint8_t *__restrict__ flags, int n_ch,
                                                long long n_cf, int short_blocks, int mixed, int skip_long,
                                                const double *__restrict__ smr,
                                                int32_t *__restrict__ bit_alloc,
                                                uint32_t *__restrict__ status)
{
    __shared__ double cp[2][32];
    const int lane = threadIdx.x, half = lane >> 5, l = lane & 31;
    const bool dense = !short_blocks && !(mixed && flags);
    const long long unit = (long long)blockIdx.x * 2 + half;
    const long long cf = dense ? unit : unit / PACX_SUB;
    const int sb = dense ? 0 : (int)(unit % PACX_SUB);
    bool alive = cf < n_cf;
    const unsigned fl = (alive && flags) ? flags[cf / n_ch] : 0u;
    const bool is_short = mixed ? ((fl & 2u) != 0) : (short_blocks != 0);
    if (!is_short && (sb != 0 || skip_long == 1))
        alive = false;                                   /* long frames may belong to k_tail_long */
    if (is_short && skip_long == 2)
        alive = false;                                   /* 2: the long frames only (the short chain has its own launch) */
    const int nb = is_short ? T.nb_short : T.nb_long;
    /* an SBR file counts every omitted band of a long block as one line
       (BitAlloc_SBR, coder/bitalloc.py:141-143) */
    const int32_t *__restrict__ n_lines = is_short ? T.band_lines_short
                                                   : (T.use_sbr ? T.band_lines_long_alloc : T.band_lines_long);
    const double budget = pacx_bit_budget(T.target_bps, is_short ? PACX_M_SHORT : PACX_M_LONG,
                                          is_short ? 1 : 0, (fl & 5u) != 0, T.n_scale_bits,
                                          T.n_mant_size_bits, nb, T.use_vq, T.use_sbr && !is_short);
    int max_mant = 1 << T.n_mant_size_bits;
    if (max_mant > 16)
        max_mant = 16;
    const long long off = cf * T.band_stride + (is_short ? sb * nb : 0);
    const bool has = alive && l < nb;
    const double s = has ? smr[off + l] : 0.0;
    const int nl = has ? n_lines[l] : 0;
    int bits = 0, cap = 0;
    bitalloc_half(alive, has, s, nl, budget, max_mant, cp[half], half, l, bits, cap, T.guard != 0,
                  T.nb_long > T.nb_short ? T.nb_long : T.nb_short);
    if (has)
        bit_alloc[off + l] = bits;
    if (alive && cap && status && l == 0)
        atomicOr(&status[cf], ((cap & 1) ? 4u : 0u) | ((cap & 2) ? 16u : 0u));   /* ALLOC_CAP, GUARD */
}

template <int M>
__global__ __launch_bounds__(64) void k_quantize(PacxTables T, const uint8_t *__restrict__ flags, int n_ch,
                                                long long n_units, int mixed,
                                                const double *__restrict__ lines,
                                                const int32_t *__restrict__ overall, int overall_stride,
                                                const int32_t *__restrict__ bit_alloc,
                                                int32_t *__restrict__ scale_factor,
                                                int32_t *__restrict__ mantissa)
{
    constexpr bool SHORT = (M == PACX_M_SHORT);
    constexpr int PER = M / 64;
    __shared__ unsigned long long bmax[PACX_MAX_BANDS];
    __shared__ int ba_s[PACX_MAX_BANDS], sf_s[PACX_MAX_BANDS];
    const int lane = threadIdx.x;
    const long long unit = blockIdx.x;
    if (unit >= n_units)
        return;
    const long long cf = SHORT ? unit / PACX_SUB : unit;
    const int sb = SHORT ? (int)(unit % PACX_SUB) : 0;
    if (mixed) {
        const unsigned fl = flags ? flags[cf / n_ch] : 0u;
        if (SHORT != ((fl & 2u) != 0))
            return;
    }
    const int nb = SHORT ? T.nb_short : T.nb_long;
    const uint8_t *__restrict__ band_of = SHORT ? T.line_band_short : T.line_band_long;
    const long long boff = cf * T.band_stride + sb * T.nb_short;
    const long long loff = cf * PACX_M_LONG + sb * PACX_M_SHORT;
    const int ov = overall[SHORT ? (cf * overall_stride + (overall_stride == 1 ? 0 : sb)) : cf * overall_stride];
    const double up = (double)(1 << ov);            /* mdctLines *= (1 << overallScale) */
    if constexpr (!SHORT) {
        if (lane < nb)
            ba_s[lane] = bit_alloc[boff + lane];
        if (lane == nb && nb < PACX_MAX_BANDS)
            ba_s[nb] = 0;                                 /* dummy band of the lines no band covers */
        double xl[16];
        uint8_t bandl[16];
        int32_t mantl[16];
        bool near_unused;
        quantize_long_core(T, lines + loff, up, bmax, ba_s, sf_s, lane, xl, bandl, mantl, near_unused);
        if (lane < nb)
            scale_factor[boff + lane] = sf_s[lane];
#pragma unroll
        for (int j = 0; j < 16; j += 4)
            *(int4 *)(mantissa + loff + 16 * lane + j) = make_int4(mantl[j], mantl[j + 1], mantl[j + 2], mantl[j + 3]);
        return;
    }
    if (lane < PACX_MAX_BANDS)
        bmax[lane] = 0ull;
    if (lane < nb)
        ba_s[lane] = bit_alloc[boff + lane];
    if (lane == nb && nb < PACX_MAX_BANDS)
        ba_s[nb] = 0;                                     /* dummy band of the lines no band covers */
    const int k0 = PER * lane;
    double x[PER];
    uint8_t band[PER];
#pragma unroll
    for (int j = 0; j < PER; j += 2) {
        const double2 v = *(const double2 *)(lines + loff + k0 + j);
        x[j] = v.x * up;
        x[j + 1] = v.y * up;
    }
    if constexpr (SHORT) {
        const uchar2 b2 = *(const uchar2 *)(band_of + k0);
        band[0] = b2.x;
        band[1] = b2.y;
    } else {
        const uint4 b16 = *(const uint4 *)(band_of + k0);
        const unsigned w[4] = {b16.x, b16.y, b16.z, b16.w};
#pragma unroll
        for (int j = 0; j < PER; ++j)
            band[j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
    }
    __syncthreads();
    if constexpr (SHORT) {
        /* 6 bands, 2 lines per lane: one masked wave reduction per band */
        for (int b = 0; b < nb; ++b) {
            double m = 0.0;
#pragma unroll
            for (int j = 0; j < PER; ++j)
                if (band[j] == b)
                    m = fmax(m, fabs(x[j]));
            m = wave_max(m);
            if (lane == b)
                bmax[b] = (unsigned long long)__double_as_longlong(m);
        }
    }
    __syncthreads();
    if (lane < nb) {
        const int sf = pacx_scale_factor(__longlong_as_double((long long)bmax[lane]), T.n_scale_bits, ba_s[lane]);
        sf_s[lane] = sf;
        scale_factor[boff + lane] = sf;
    }
    __syncthreads();
    int32_t mant[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int b = band[j];
        const int ba = ba_s[b];
        mant[j] = ba ? pacx_mantissa(x[j], sf_s[b], T.n_scale_bits, ba) : 0;
    }
    if constexpr (SHORT) {
        *(int2 *)(mantissa + loff + k0) = make_int2(mant[0], mant[1]);
    } else {
#pragma unroll
        for (int j = 0; j < PER; j += 4)
            *(int4 *)(mantissa + loff + k0 + j) = make_int4(mant[j], mant[j + 1], mant[j + 2], mant[j + 3]);
    }
}

/* one (sub-)block body starting at stream bit `pos`; returns its bit length.
 * Lanes < nBands write the band headers (their offsets come from an exclusive
 * prefix sum across lanes); every lane then packs its M/64 consecutive lines
 * (one 16-byte load of mantissas at a time, band tables from LDS). */
template <int M>
__device__ __forceinline__ int pack_body(const PacxTables &T, unsigned *words, int *offs, int *ba_s,
                                         int *lower_s, int pos, int ov, const int32_t *__restrict__ ba,
                                         const int32_t *__restrict__ sf,
                                         const int32_t *__restrict__ mant, int lane)
{
    constexpr bool SHORT = (M == PACX_M_SHORT);
    constexpr int PER = M / 64;
    const int nb = SHORT ? T.nb_short : T.nb_long;
    const int32_t *__restrict__ lower = SHORT ? T.band_lower_short : T.band_lower_long;
    const int32_t *__restrict__ count = SHORT ? T.band_lines_short : T.band_lines_long;
    const uint8_t *__restrict__ band_of = SHORT ? T.line_band_short : T.line_band_long;
    const int k0 = PER * lane;
    /* loads first: mantissas and band ids of this lane's lines, band tables */
    int32_t m[PER];
    uint8_t band[PER];
    if constexpr (SHORT) {
        const int2 v = *(const int2 *)(mant + k0);
        m[0] = v.x; m[1] = v.y;
        const uchar2 b2 = *(const uchar2 *)(band_of + k0);
        band[0] = b2.x; band[1] = b2.y;
    } else {
#pragma unroll
        for (int j = 0; j < PER; j += 4) {
            const int4 v = *(const int4 *)(mant + k0 + j);
            m[j] = v.x; m[j + 1] = v.y; m[j + 2] = v.z; m[j + 3] = v.w;
        }
        const uint4 b16 = *(const uint4 *)(band_of + k0);
        const unsigned w[4] = {b16.x, b16.y, b16.z, b16.w};
#pragma unroll
        for (int j = 0; j < PER; ++j)
            band[j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
    }
    const int a_mine = (lane < nb) ? ba[lane] : 0;
    const int width = (lane < nb) ? T.n_mant_size_bits + T.n_scale_bits + a_mine * count[lane] : 0;
    int incl = width;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off)
            incl += t;
    }
    const int my_off = pos + T.n_scale_bits + incl - width;
    if (lane < nb) {
        offs[lane] = my_off + T.n_mant_size_bits + T.n_scale_bits;      /* first mantissa bit of the band */
        ba_s[lane] = a_mine;
        lower_s[lane] = lower[lane];
    }
    if (lane == nb && nb < PACX_MAX_BANDS)
        ba_s[nb] = 0;                                     /* dummy band of the lines no band covers */
    if (lane == nb - 1)
        offs[nb] = my_off + width;
    if (lane == 0)
        put_bits(words, pos, (unsigned)ov, T.n_scale_bits);
    if (lane < nb) {
        put_bits(words, my_off, (unsigned)(a_mine ? a_mine - 1 : 0), T.n_mant_size_bits);
        put_bits(words, my_off + T.n_mant_size_bits, (unsigned)sf[lane], T.n_scale_bits);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int b = band[j];
        const int a = ba_s[b];
        if (a)
            put_bits(words, offs[b] + (k0 + j - lower_s[b]) * a, (unsigned)m[j], a);
    }
    const int end = offs[nb];
    __syncthreads();
    return end - pos;
}

__global__ __launch_bounds__(64) void k_pack(PacxTables T, const uint8_t *__restrict__ flags, int n_ch,
                                            long long n_cf, const int32_t *__restrict__ overall,
                                            const int32_t *__restrict__ scale_factor,
                                            const int32_t *__restrict__ bit_alloc,
                                            const int32_t *__restrict__ mantissa,
                                            const uint32_t *__restrict__ status,
                                            uint8_t *__restrict__ payload, int payload_stride,
                                            int32_t *__restrict__ n_bytes, int only_short)
{
    __shared__ unsigned words[PACX_PACK_WORDS];
    __shared__ int offs[PACX_MAX_BANDS + 1], ba_s[PACX_MAX_BANDS], lower_s[PACX_MAX_BANDS];
    const int lane = threadIdx.x;
    const long long cf = blockIdx.x;
    if (cf >= n_cf)
        return;
    const long long f = cf / n_ch;
    const unsigned fl = flags ? flags[f] : 0u;
    const bool is_short = (fl & 2u) != 0;
    if (only_short && !is_short)
        return;                                          /* long frames were packed by k_tail_long */
    /* the reference drops the hop for every channel when any channel holds an
       all-zero short sub-block (coder/pacfile.py:530-533) */
    unsigned st = 0;
    if (status)
        for (int c = 0; c < n_ch; ++c)
            st |= status[f * n_ch + c];
    if (is_short && (st & 2u)) {
        if (lane == 0)
            n_bytes[cf] = 0;
        return;
    }
    for (int i = lane; i < PACX_PACK_WORDS; i += 64)
        words[i] = 0u;
    __syncthreads();
    if (lane == 0) {
        put_bits(words, 0, fl & 1u, 1);
        put_bits(words, 1, (fl >> 1) & 1u, 1);
        put_bits(words, 2, (fl >> 2) & 1u, 1);
    }
    int pos = 3;
    const int32_t *ba = bit_alloc + cf * T.band_stride;
    const int32_t *sf = scale_factor + cf * T.band_stride;
    const int32_t *mant = mantissa + cf * PACX_M_LONG;
    const int32_t *ov = overall + cf * PACX_SUB;
    if (!is_short) {
        pos += pack_body<PACX_M_LONG>(T, words, offs, ba_s, lower_s, pos, ov[0], ba, sf, mant, lane);
    } else {
        for (int sb = 0; sb < PACX_SUB; ++sb)
            pos += pack_body<PACX_M_SHORT>(T, words, offs, ba_s, lower_s, pos, ov[sb], ba + sb * T.nb_short,
                                           sf + sb * T.nb_short, mant + sb * PACX_M_SHORT, lane);
    }
    /* size rule of coder/pacfile.py:552-565: body bits + 4, rounded up */
    const int bits = (pos - 3) + 4;
    const int nbytes = (bits + 7) >> 3;
    unsigned *dst = (unsigned *)(payload + cf * (long long)payload_stride);
    for (int i = lane; i < (nbytes + 3) / 4; i += 64)
        dst[i] = __builtin_bswap32(words[i]);
    if (lane == 0)
        n_bytes[cf] = nbytes;
}

/* ------------------------------------------------- fused tail, long blocks */
/* BitAlloc -> scale factors + mantissas -> payload of LONG channel-frames, two per
 * wave: the three stages above back to back, the lines read once and the
 * mantissas packed straight from registers (same arithmetic, same helpers).
 * Short frames keep the separate kernels. */
#ifdef PACX_TAIL_DEBUG
/* phase stamps (s_memtime), summed over all blocks: a measuring aid read by
   tools/psy_phase_probe.py from a library built with --phase-debug */
__device__ long long g_tail_dbg[16];
#define TAIL_T(k) do { long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                       if (threadIdx.x == 0 && tail_last) atomicAdd((unsigned long long *)&g_tail_dbg[k], (unsigned long long)(t_ - tail_last)); \
                       tail_last = t_; } while (0)
extern "C" int pacx_debug_read_tail(long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_dbg), sizeof(long long) * n);
}
#else
#define TAIL_T(k) do { } while (0)
#endif

__global__ __launch_bounds__(64, 4) void k_tail_long(PacxTables T, const uint8_t *__restrict__ flags, int n_ch,
                                                 long long n_cf, int mixed, const double *__restrict__ smr,
                                                 const double *__restrict__ lines,
                                                 const int32_t *__restrict__ overall,
                                                 int32_t *__restrict__ bit_alloc,
                                                 int32_t *__restrict__ scale_factor,
                                                 int32_t *__restrict__ mantissa, uint32_t *__restrict__ status,
                                                 uint8_t *__restrict__ payload, int payload_stride,
                                                 int32_t *__restrict__ n_bytes)
{
    __shared__ unsigned words[PACX_PACK_WORDS];
    __shared__ double cp[2][32];
    __shared__ unsigned long long bmax[PACX_MAX_BANDS];
    __shared__ int ba_2[2][PACX_MAX_BANDS], sf_s[PACX_MAX_BANDS], offs[PACX_MAX_BANDS + 1], lower_s[PACX_MAX_BANDS];
    const int lane = threadIdx.x, half = lane >> 5, l = lane & 31;
    /* two channel-frames per wave: BitAlloc is a half-wave job (lanes = bands), so the two
       halves run it for the two frames at once; scale factors, mantissas and payload
       then take the whole wave, one frame after the other */
    const int per_block = mixed ? 1 : 2;                  /* see pacx_launch_tail */
    const long long cf0 = per_block * (long long)blockIdx.x;
    if (cf0 >= n_cf)
        return;
    if (mixed && flags && (flags[cf0 / n_ch] & 2u))
        return;                                           /* a short-coded frame: k_tail_short's */
    const int nb = T.nb_long;
#ifdef PACX_TAIL_DEBUG
    long long tail_last = 0;
#endif
    TAIL_T(15);
    unsigned long long raises = 0;                        /* lanes of the halves whose frame is flagged */
    /* 1. BitAlloc */
    {
        const long long cfh = cf0 + half;
        const unsigned flh = (flags && cfh < n_cf) ? flags[cfh / n_ch] : 0u;
        const bool alive = half < per_block && cfh < n_cf && !(mixed && (flh & 2u));
        const long long boff = cfh * T.band_stride;
        const int32_t *__restrict__ n_lines = T.use_sbr ? T.band_lines_long_alloc : T.band_lines_long;
        const double budget = pacx_bit_budget(T.target_bps, PACX_M_LONG, 0, (flh & 5u) != 0, T.n_scale_bits,
                                              T.n_mant_size_bits, nb, T.use_vq, T.use_sbr);
        int max_mant = 1 << T.n_mant_size_bits;
        if (max_mant > 16)
            max_mant = 16;
        const bool has = alive && l < nb;
        const double sv = has ? smr[boff + l] : 0.0;
        const int nl = has ? n_lines[l] : 0;
        int bits = 0, cap = 0;
        bitalloc_half(alive, has, sv, nl, budget, max_mant, cp[half], half, l, bits, cap, T.guard != 0, T.nb_long);
        if (has) {
            bit_alloc[boff + l] = bits;
            ba_2[half][l] = bits;
        }
        if (alive && l == nb && nb < PACX_MAX_BANDS)
            ba_2[half][nb] = 0;                           /* dummy band of the lines no band covers */
        if (alive && cap && status && l == 0)
            atomicOr(&status[cfh], ((cap & 1) ? 4u : 0u) | ((cap & 2) ? 16u : 0u));
        /* scalar mantissas + SBR: bits on an omitted band are where the reference raises
           (PACX_ST_REF_RAISES, include/pacx.h) */
        raises = T.use_sbr && !T.use_vq && __builtin_amdgcn_ballot_w64(has && l >= T.first_omitted && bits != 0);
        if (T.use_sbr && !T.use_vq && has && l >= T.first_omitted && bits != 0 && status)
            atomicOr(&status[cfh], 64u);
    }
    TAIL_T(0);
    for (int c = 0; c < per_block; ++c) {
        const long long cf = cf0 + c;
        if (cf >= n_cf)
            break;
        const unsigned fl = flags ? flags[cf / n_ch] : 0u;
        if (mixed && (fl & 2u))
            continue;
        const long long boff = cf * T.band_stride;
        int *ba_s = ba_2[c];
        __syncthreads();                                  /* ba_2 visible; words, offs free again */
        if (payload)
            for (int i = lane; i < PACX_PACK_WORDS; i += 64)
                words[i] = 0u;
        __syncthreads();
        /* 2. scale factors + mantissas */
        const int ov = overall[cf * PACX_SUB];
        double x[16];
        uint8_t band[16];
        int32_t mant[16];
        bool near;
        quantize_long_core(T, lines + cf * PACX_M_LONG, (double)(1 << ov), bmax, ba_s, sf_s, lane, x, band, mant, near);
        if (lane < nb)
            scale_factor[boff + lane] = sf_s[lane];
        if (status && __builtin_amdgcn_ballot_w64(near) && lane == 0)
            atomicOr(&status[cf], 16u);                   /* PACX_ST_GUARD */
        const int k0 = 16 * lane;
        if (mantissa) {
#pragma unroll
            for (int j = 0; j < 16; j += 4)
                *(int4 *)(mantissa + cf * PACX_M_LONG + k0 + j) = make_int4(mant[j], mant[j + 1], mant[j + 2], mant[j + 3]);
        }
        if (!payload)
            continue;
        if ((raises >> (32 * c)) & 0xffffffffull) {      /* PACX_ST_REF_RAISES: no payload is defined */
            if (lane == 0)
                n_bytes[cf] = 0;
            continue;
        }
        TAIL_T(1);
        /* 3. payload (layout of pack_body) */
        const int a_mine = (lane < nb) ? ba_s[lane] : 0;
        const int width = (lane < nb) ? T.n_mant_size_bits + T.n_scale_bits + a_mine * T.band_lines_long[lane] : 0;
        int incl = width;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off)
                incl += t;
        }
        const int my_off = 3 + T.n_scale_bits + incl - width;
        if (lane < nb) {
            offs[lane] = my_off + T.n_mant_size_bits + T.n_scale_bits;
            lower_s[lane] = T.band_lower_long[lane];
        }
        if (lane == nb - 1)
            offs[nb] = my_off + width;
        if (lane == 0) {
            put_bits(words, 0, fl & 1u, 1);
            put_bits(words, 1, (fl >> 1) & 1u, 1);
            put_bits(words, 2, (fl >> 2) & 1u, 1);
            put_bits(words, 3, (unsigned)ov, T.n_scale_bits);
        }
        if (lane < nb) {
            put_bits(words, my_off, (unsigned)(a_mine ? a_mine - 1 : 0), T.n_mant_size_bits);
            put_bits(words, my_off + T.n_mant_size_bits, (unsigned)sf_s[lane], T.n_scale_bits);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int b = band[j];
            const int a = ba_s[b];
            if (a)
                put_bits(words, offs[b] + (k0 + j - lower_s[b]) * a, (unsigned)mant[j], a);
        }
        const int end = offs[nb];
        __syncthreads();
        const int nbytes = ((end - 3) + 4 + 7) >> 3;
        unsigned *dst = (unsigned *)(payload + cf * (long long)payload_stride);
        for (int i = lane; i < (nbytes + 3) / 4; i += 64)
            dst[i] = __builtin_bswap32(words[i]);
        if (lane == 0)
            n_bytes[cf] = nbytes;
        TAIL_T(2);
    }
}

/* ------------------------------------------------ fused tail, short frames */
/* One workgroup of four waves per short-coded channel-frame: wave w owns
 * sub-blocks 2w and 2w+1.  BitAlloc of both at once (one per half wave), then
 * per sub-block the scale factors and mantissas (2 lines per lane, band maxima by
 * masked wave reductions), then the payload: sub-block bit lengths meet in LDS,
 * every wave ORs its two bodies into the frame's bit buffer at their offsets.
 * Same helpers and arithmetic as k_bitalloc / k_quantize<128> / k_pack. */
__global__ __launch_bounds__(256) void k_tail_short(PacxTables T, const uint8_t *__restrict__ flags, int n_ch,
                                                   const int32_t *__restrict__ cf_list,
                                                   const int32_t *__restrict__ cf_count,
                                                   const double *__restrict__ smr,
                                                   const double *__restrict__ lines,
                                                   const int32_t *__restrict__ overall,
                                                   int32_t *__restrict__ bit_alloc,
                                                   int32_t *__restrict__ scale_factor,
                                                   int32_t *__restrict__ mantissa, uint32_t *__restrict__ status,
                                                   uint8_t *__restrict__ payload, int payload_stride,
                                                   int32_t *__restrict__ n_bytes)
{
    __shared__ unsigned words[PACX_PACK_WORDS];
    __shared__ double cp[4][2][32];
    __shared__ int ba_s[PACX_SUB][9], sf_s[PACX_SUB][9], len_s[PACX_SUB];   /* [.][nb] = dummy band */
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, half = lane >> 5, l = lane & 31;
    if ((long long)blockIdx.x >= (long long)*cf_count)
        return;
    const long long cf = cf_list[blockIdx.x];
    const long long frame = cf / n_ch;
    const unsigned fl = flags[frame];
    const int nb = T.nb_short;
    /* 1. BitAlloc: sub-block 2 wv + half on each half wave */
    {
        const int sb = 2 * wv + half;
        const double budget = pacx_bit_budget(T.target_bps, PACX_M_SHORT, 1, (fl & 5u) != 0, T.n_scale_bits,
                                              T.n_mant_size_bits, nb, T.use_vq, 0);
        int max_mant = 1 << T.n_mant_size_bits;
        if (max_mant > 16)
            max_mant = 16;
        const bool has = l < nb;
        const long long off = cf * T.band_stride + sb * nb;
        const double sv = has ? smr[off + l] : 0.0;
        const int nl = has ? T.band_lines_short[l] : 0;
        int bits = 0, cap = 0;
        bitalloc_half(true, has, sv, nl, budget, max_mant, cp[wv][half], half, l, bits, cap, T.guard != 0, nb);
        if (has) {
            bit_alloc[off + l] = bits;
            ba_s[sb][l] = bits;
        }
        if (l == nb) {
            ba_s[sb][nb] = 0;
            sf_s[sb][nb] = 0;
        }
        if (cap && status && l == 0)
            atomicOr(&status[cf], ((cap & 1) ? 4u : 0u) | ((cap & 2) ? 16u : 0u));
    }
    for (int i = tid; i < PACX_PACK_WORDS; i += 256)
        words[i] = 0u;
    __syncthreads();
    /* 2. scale factors + mantissas of the wave's two sub-blocks */
    const uchar2 b2 = *(const uchar2 *)(T.line_band_short + 2 * lane);
    const int band0 = b2.x, band1 = b2.y;
    int32_t mant[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int sb = 2 * wv + q;
        const int ov = overall[cf * PACX_SUB + sb];
        const double up = (double)(1 << ov);
        const long long loff = cf * PACX_M_LONG + sb * PACX_M_SHORT;
        const double2 v = *(const double2 *)(lines + loff + 2 * lane);
        const double x0 = v.x * up, x1 = v.y * up;
        {
            /* ScaleFactor is non-increasing in the magnitude, so a band's scale factor is the
               minimum of its lines' own scale factors: small integers (a 64-bit wave maximum of
               the magnitudes per band was six dependent LDS round trips for each of the six bands;
               round 2 took the minimum by bisection with one ballot per bit). */
            const int nsb = T.n_scale_bits;
            /* round 3: the per-band minimum as a 32-bit LDS atomic min -- each band's owner lane posts ScaleFactor(0),
               the value of a band without lines, then every lane posts its two lines' own scale factors (one
               ds_min_i32 each, no return value): one LDS round trip for all six bands, where the bisection by
               ballots took five ballots per band (130 instructions per sub-block, a sixth of this kernel) */
            if (lane < nb)
                sf_s[sb][lane] = pacx_scale_factor(0.0, nsb, ba_s[sb][lane]);
            wave_lds_fence();
            if (band0 < nb)
                atomicMin(&sf_s[sb][band0], pacx_scale_factor(fabs(x0), nsb, ba_s[sb][band0]));
            if (band1 < nb)
                atomicMin(&sf_s[sb][band1], pacx_scale_factor(fabs(x1), nsb, ba_s[sb][band1]));
        }
        wave_lds_fence();
        const int a0 = ba_s[sb][band0], a1 = ba_s[sb][band1];
        mant[q][0] = a0 ? pacx_mantissa(x0, sf_s[sb][band0], T.n_scale_bits, a0) : 0;
        mant[q][1] = a1 ? pacx_mantissa(x1, sf_s[sb][band1], T.n_scale_bits, a1) : 0;
        if (status && T.guard) {                          /* PACX_ST_GUARD, see pacx_exact.h */
            const int r0 = (1 << T.n_scale_bits) - 1;
            const bool near = (band0 < nb && pacx_quant_guard(fabs(x0), r0 + a0, PACX_GUARD_LINE_ERR)) ||
                              (band1 < nb && pacx_quant_guard(fabs(x1), r0 + a1, PACX_GUARD_LINE_ERR));
            if (__builtin_amdgcn_ballot_w64(near) && lane == 0)
                atomicOr(&status[cf], 16u);
        }
        if (lane < nb)
            scale_factor[cf * T.band_stride + sb * nb + lane] = sf_s[sb][lane];
        if (mantissa)
            *(int2 *)(mantissa + loff + 2 * lane) = make_int2(mant[q][0], mant[q][1]);
        if (lane == 0) {
            int len = T.n_scale_bits;
            for (int b = 0; b < nb; ++b)
                len += T.n_mant_size_bits + T.n_scale_bits + ba_s[sb][b] * T.band_lines_short[b];
            len_s[sb] = len;
        }
    }
    if (!payload)
        return;
    /* the reference drops the hop for every channel when any channel holds an
       all-zero short sub-block (coder/pacfile.py:530-533) */
    unsigned st = 0;
    if (status)
        for (int c = 0; c < n_ch; ++c)
            st |= status[frame * n_ch + c];
    if (st & 2u) {
        if (tid == 0)
            n_bytes[cf] = 0;
        return;
    }
    __syncthreads();
    /* 3. payload: flags, then the eight bodies back to back */
    if (tid == 0) {
        put_bits(words, 0, fl & 1u, 1);
        put_bits(words, 1, (fl >> 1) & 1u, 1);
        put_bits(words, 2, (fl >> 2) & 1u, 1);
    }
    int total = 0;
#pragma unroll
    for (int q = 0; q < PACX_SUB; ++q)
        total += len_s[q];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int sb = 2 * wv + q;
        int pos = 3;
        for (int u = 0; u < sb; ++u)
            pos += len_s[u];
        if (lane == 0)
            put_bits(words, pos, (unsigned)overall[cf * PACX_SUB + sb], T.n_scale_bits);
        /* band header offsets: exclusive prefix over the bands (lanes < nb) */
        const int a_mine = (lane < nb) ? ba_s[sb][lane] : 0;
        const int width = (lane < nb) ? T.n_mant_size_bits + T.n_scale_bits + a_mine * T.band_lines_short[lane] : 0;
        int incl = width;
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off)
                incl += t;
        }
        const int my_off = pos + T.n_scale_bits + incl - width;
        if (lane < nb) {
            put_bits(words, my_off, (unsigned)(a_mine ? a_mine - 1 : 0), T.n_mant_size_bits);
            put_bits(words, my_off + T.n_mant_size_bits, (unsigned)sf_s[sb][lane], T.n_scale_bits);
        }
        /* first mantissa bit of the band of each of this lane's two lines */
        const int f0 = __shfl(my_off, band0, 64) + T.n_mant_size_bits + T.n_scale_bits;
        const int f1 = __shfl(my_off, band1, 64) + T.n_mant_size_bits + T.n_scale_bits;
        const int a0 = ba_s[sb][band0], a1 = ba_s[sb][band1];
        if (a0)
            put_bits(words, f0 + (2 * lane - T.band_lower_short[band0]) * a0, (unsigned)mant[q][0], a0);
        if (a1)
            put_bits(words, f1 + (2 * lane + 1 - T.band_lower_short[band1]) * a1, (unsigned)mant[q][1], a1);
    }
    __syncthreads();
    const int nbytes = (total + 4 + 7) >> 3;
    unsigned *dst = (unsigned *)(payload + cf * (long long)payload_stride);
    for (int i = tid; i < (nbytes + 3) / 4; i += 256)
        dst[i] = __builtin_bswap32(words[i]);
    if (tid == 0)
        n_bytes[cf] = nbytes;
}

/* ----------------------------------------------------------- body gather */
#define SCAN_CHUNK 256

__device__ __forceinline__ long long rec_len(int nb) { return nb > 0 ? (long long)nb + 4 : 0; }

/* per-chunk byte totals */
__global__ __launch_bounds__(256) void k_scan_partial(const int32_t *__restrict__ n_bytes, long long n,
                                                     long long *__restrict__ chunk_sum)
{
    __shared__ long long red[256];
    const long long i = (long long)blockIdx.x * SCAN_CHUNK + threadIdx.x;
    red[threadIdx.x] = (i < n) ? rec_len(n_bytes[i]) : 0;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        chunk_sum[blockIdx.x] = red[0];
}

/* exclusive scan of the chunk totals (one wave; chunks are few: n_cf/256) */
__global__ __launch_bounds__(64) void k_scan_chunks(long long *chunk_sum, long long n_chunks, long long *total)
{
    const int lane = threadIdx.x;
    long long run = 0;
    for (long long base = 0; base < n_chunks; base += 64) {
        const long long i = base + lane;
        const long long v = (i < n_chunks) ? chunk_sum[i] : 0;
        long long inc = v;                       /* inclusive scan across the wave */
        for (int off = 1; off < 64; off <<= 1) {
            const long long t = __shfl_up(inc, off, 64);
            if (lane >= off)
                inc += t;
        }
        if (i < n_chunks)
            chunk_sum[i] = run + inc - v;
        run += __shfl(inc, 63, 64);
    }
    if (total && lane == 0)
        *total = run;
}

/* byte offset of every record */
__global__ __launch_bounds__(256) void k_scan_offsets(const int32_t *__restrict__ n_bytes, long long n,
                                                     const long long *__restrict__ chunk_off,
                                                     long long *__restrict__ offs)
{
    __shared__ long long buf[256];
    const int t = threadIdx.x;
    const long long i = (long long)blockIdx.x * SCAN_CHUNK + t;
    const long long v = (i < n) ? rec_len(n_bytes[i]) : 0;
    buf[t] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const long long add = (t >= off) ? buf[t - off] : 0;
        __syncthreads();
        buf[t] += add;
        __syncthreads();
    }
    if (i < n)
        offs[i] = chunk_off[blockIdx.x] + buf[t] - v;
}

#define SCAN_SMALL_MAX 32768          /* up to here: k_gather_small, one launch */

/* one record by one wave: "<L nBytes" then the payload bytes */
__device__ __forceinline__ void copy_record(int nb, long long off, const uint8_t *__restrict__ src,
                                            uint8_t *__restrict__ body, long long capacity, int lane)
{
    if (nb <= 0 || off + nb + 4 > capacity)
        return;
    if (lane < 4)
        body[off + lane] = (uint8_t)((unsigned)nb >> (8 * lane));
    uint8_t *dst = body + off + 4;
    /* head bytes up to 4-byte alignment of dst, then dwords (src slots are 16-byte aligned), then tail */
    const int head = (int)((4 - ((uintptr_t)dst & 3)) & 3);
    if (lane < head && lane < nb)
        dst[lane] = src[lane];
    const int n_words = (nb - head) > 0 ? (nb - head) >> 2 : 0;
    for (int w = lane; w < n_words; w += 64) {
        const uint8_t *s = src + head + 4 * w;
        const unsigned v = (unsigned)s[0] | ((unsigned)s[1] << 8) | ((unsigned)s[2] << 16) | ((unsigned)s[3] << 24);
        *(unsigned *)(dst + head + 4 * w) = v;
    }
    const int done = head + 4 * n_words;
    if (done + lane < nb && lane < 4)
        dst[done + lane] = src[done + lane];
}

__global__ __launch_bounds__(64) void k_copy_body(const int32_t *__restrict__ n_bytes, long long n,
                                                 const long long *__restrict__ offs,
                                                 const uint8_t *__restrict__ payload, int payload_stride,
                                                 uint8_t *__restrict__ body, long long capacity)
{
    const long long i = blockIdx.x;
    if (i >= n)
        return;
    copy_record(n_bytes[i], offs[i], payload + i * (long long)payload_stride, body, capacity, threadIdx.x);
}

/* Small batches (the bench batch): scan and copy in ONE launch, no dependency between
 * workgroups.  A 4-wave workgroup owns GATHER_REC consecutive records (8); it finds its
 * byte offset by adding up the lengths of ALL earlier records itself (coalesced 4-byte
 * reads that hit L2: n/2 entries on average, 8 MB over the whole grid at n = 8192),
 * scans its own records in one wave and copies them, one wave per record.  Sums fit
 * 32 bits: n <= SCAN_SMALL_MAX records of at most payload_stride + 4 bytes. */
#define GATHER_REC 8
__global__ __launch_bounds__(256) void k_gather_small(const int32_t *__restrict__ n_bytes, long long n,
                                                     const uint8_t *__restrict__ payload, int payload_stride,
                                                     uint8_t *__restrict__ body, long long capacity,
                                                     long long *__restrict__ total)
{
    __shared__ int part[4];
    __shared__ int offs_s[GATHER_REC + 1];
    __shared__ int len_s[GATHER_REC];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const long long first = (long long)blockIdx.x * GATHER_REC;
    int acc = 0;
    const int4 *__restrict__ nb4 = (const int4 *)n_bytes;       /* first is a multiple of GATHER_REC >= 4 */
    const int n4 = (int)(first >> 2);
#pragma unroll 8
    for (int j = t; j < n4; j += 256) {
        const int4 v = nb4[j];
        acc += (int)(rec_len(v.x) + rec_len(v.y) + rec_len(v.z) + rec_len(v.w));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        acc += __shfl_xor(acc, off, 64);
    if (lane == 0)
        part[wv] = acc;
    __syncthreads();
    if (wv == 0) {
        const int base = part[0] + part[1] + part[2] + part[3];
        const int nb = (lane < GATHER_REC && first + lane < n) ? n_bytes[first + lane] : 0;
        const int len = (int)rec_len(nb);
        int incl = len;
#pragma unroll
        for (int off = 1; off < GATHER_REC; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off)
                incl += v;
        }
        if (lane < GATHER_REC) {
            offs_s[lane] = base + incl - len;
            len_s[lane] = nb;
        }
        if (lane == GATHER_REC - 1) {
            offs_s[GATHER_REC] = base + incl;
            if (total && first + GATHER_REC >= n)
                *total = (long long)(base + incl);
        }
    }
    __syncthreads();
    for (int r = wv; r < GATHER_REC && first + r < n; r += 4)
        copy_record(len_s[r], (long long)offs_s[r], payload + (first + r) * (long long)payload_stride, body,
                    capacity, lane);
}

/* ------------------------------------------------------------- launchers */
void pacx_launch_bitalloc(const PacxTables &T, const uint8_t *flags, int n_ch, long long n_cf,
                          int short_blocks, int mixed, int skip_long, const double *smr, int32_t *bit_alloc,
                          uint32_t *status, hipStream_t st)
{
    if (n_cf <= 0)
        return;
    const bool dense = !short_blocks && !(mixed && flags);
    if (dense && skip_long == 1)
        return;                                 /* every frame is long and was allocated by the mask kernel */
    const long long units = dense ? n_cf : n_cf * PACX_SUB;       /* two units per wave */
    hipLaunchKernelGGL(k_bitalloc, dim3((unsigned)((units + 1) / 2)), dim3(64), 0, st, T, flags, n_ch,
                       n_cf, short_blocks, mixed, skip_long, smr, bit_alloc, status);
}

void pacx_launch_quantize(const PacxTables &T, const uint8_t *flags, int n_ch, long long n_cf,
                          int short_blocks, int mixed, const double *lines, const int32_t *overall,
                          int overall_stride, const int32_t *bit_alloc, int32_t *scale_factor,
                          int32_t *mantissa, hipStream_t st)
{
    if (n_cf <= 0)
        return;
    if (!short_blocks || mixed)
        hipLaunchKernelGGL((k_quantize<PACX_M_LONG>), dim3((unsigned)n_cf), dim3(64), 0, st, T, flags, n_ch,
                           n_cf, mixed, lines, overall, overall_stride, bit_alloc, scale_factor, mantissa);
    if (short_blocks || mixed)
        hipLaunchKernelGGL((k_quantize<PACX_M_SHORT>), dim3((unsigned)(n_cf * PACX_SUB)), dim3(64), 0, st, T,
                           flags, n_ch, n_cf * PACX_SUB, mixed, lines, overall, overall_stride, bit_alloc,
                           scale_factor, mantissa);
}

void pacx_launch_pack(const PacxTables &T, const uint8_t *flags, int n_ch, long long n_cf,
                      const int32_t *overall, const int32_t *scale_factor, const int32_t *bit_alloc,
                      const int32_t *mantissa, const uint32_t *status, uint8_t *payload,
                      int payload_stride, int32_t *n_bytes, hipStream_t st)
{
    if (n_cf <= 0)
        return;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)n_cf), dim3(64), 0, st, T, flags, n_ch, n_cf, overall,
                       scale_factor, bit_alloc, mantissa, status, payload, payload_stride, n_bytes, 0);
}

/* BitAlloc + quantize (+ pack when payload != NULL) of a whole batch: long frames
 * through the fused kernel, short frames (flags with CUR) through the separate ones. */
void pacx_launch_tail(const PacxTables &T, const uint8_t *flags, int n_ch, long long n_cf, const double *smr,
                      const double *lines, const int32_t *overall, int32_t *bit_alloc, int32_t *scale_factor,
                      int32_t *mantissa, uint32_t *status, uint8_t *payload, int payload_stride,
                      int32_t *n_bytes, const int32_t *list_short, const int32_t *count_short, int skip_long,
                      hipStream_t st)
{
    if (n_cf <= 0)
        return;
    const int mixed = flags ? 1 : 0;
    /* skip_long: the long frames were finished inside the mask kernel (k_psy.hip, k_mask<1024, true>).
       Otherwise -- all-long batches: two frames per wave (BitAlloc of both on the two half
       waves).  Mixed streams: one frame per wave -- the waves of short-coded frames leave at
       once, and the long ones that remain fit the chip in one round, so the shorter chain per
       wave wins */
    /* skip_long: 0 = long and short frames, 1 = short frames only, 2 = long frames only */
    if (skip_long != 1)
        hipLaunchKernelGGL(k_tail_long, dim3((unsigned)(mixed ? n_cf : (n_cf + 1) / 2)), dim3(64), 0, st, T, flags,
                           n_ch, n_cf, mixed, smr, lines, overall, bit_alloc, scale_factor, mantissa, status, payload,
                           payload_stride, n_bytes);
    if (skip_long == 2)
        return;
    if (mixed && list_short && T.nb_short <= 8) {
        /* short frames: one fused workgroup each, over the compacted list */
        hipLaunchKernelGGL(k_tail_short, dim3((unsigned)n_cf), dim3(256), 0, st, T, flags, n_ch, list_short,
                           count_short, smr, lines, overall, bit_alloc, scale_factor, mantissa, status, payload,
                           payload_stride, n_bytes);
    } else if (mixed) {
        const long long units = n_cf * PACX_SUB;
        hipLaunchKernelGGL(k_bitalloc, dim3((unsigned)((units + 1) / 2)), dim3(64), 0, st, T, flags, n_ch, n_cf,
                           0, 1, 1, smr, bit_alloc, status);
        hipLaunchKernelGGL((k_quantize<PACX_M_SHORT>), dim3((unsigned)units), dim3(64), 0, st, T, flags, n_ch,
                           units, 1, lines, overall, PACX_SUB, bit_alloc, scale_factor, mantissa);
        if (payload)
            hipLaunchKernelGGL(k_pack, dim3((unsigned)n_cf), dim3(64), 0, st, T, flags, n_ch, n_cf, overall,
                               scale_factor, bit_alloc, mantissa, status, payload, payload_stride, n_bytes, 1);
    }
}

void pacx_launch_gather(long long n_cf, const uint8_t *payload, int payload_stride,
                        const int32_t *n_bytes, long long *chunk_buf, long long *offs_buf, uint8_t *body,
                        long long capacity, long long *total, hipStream_t st)
{
    if (n_cf <= 0)
        return;
    if (n_cf <= SCAN_SMALL_MAX) {
        hipLaunchKernelGGL(k_gather_small, dim3((unsigned)((n_cf + GATHER_REC - 1) / GATHER_REC)), dim3(256), 0, st,
                           n_bytes, n_cf, payload, payload_stride, body, capacity, total);
        return;
    }
    const long long n_chunks = (n_cf + SCAN_CHUNK - 1) / SCAN_CHUNK;
    hipLaunchKernelGGL(k_scan_partial, dim3((unsigned)n_chunks), dim3(256), 0, st, n_bytes, n_cf, chunk_buf);
    hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(64), 0, st, chunk_buf, n_chunks, total);
    hipLaunchKernelGGL(k_scan_offsets, dim3((unsigned)n_chunks), dim3(256), 0, st, n_bytes, n_cf, chunk_buf,
                       offs_buf);
    hipLaunchKernelGGL(k_copy_body, dim3((unsigned)n_cf), dim3(64), 0, st, n_bytes, n_cf, offs_buf, payload,
                       payload_stride, body, capacity);
}
