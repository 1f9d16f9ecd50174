/*
 * k_vq_dec.hip -- decode side of the gain-shape / SBR variants (SURVEY.md 8f-4
 * for VQ streams): channel-block payload -> MDCT lines.
 *
 *   k_vq_dec    one workgroup per channel-block.  Header fields first (flags,
 *               then per (sub-)block the overall scale and the allocations,
 *               coder/pacfile.py:185-197, 264-266); a coded band occupies
 *               exactly bitAlloc*nLines bits, so every band's position follows
 *               from the allocations and the bands are decoded independently,
 *               by ticket: dequantize_gain_shape
 *               (coder/gain_shape_quantize.py:515-541) = split_band_decode
 *               (:411-473) with pyramid-VQ leaves (decode_pvq_vector
 *               :127-176, dequantize_pvq :259-272) times the mu-law gain.
 *               Lines land where codec.Decode / Decode_SBR put them
 *               (coder/codec.py:59-76, 117-134: an omitted band of an SBR long
 *               block is ONE value, stored at the next line index).
 *   k_sbr_recon Decode_SBR's reconstruction (coder/codec.py:136-198) for the
 *               long blocks that carry omitted-band values: Gaussian-smoothed
 *               envelope (scipy.ndimage.gaussian_filter1d, sigma 200, reflect),
 *               transposition of the lower half by order-1 spline
 *               interpolation (scipy interp1d 'slinear'), per-band scaling.
 *               Both SciPy routines are restated with their summation order
 *               (the CPU tests check the same restatement against SciPy bit for bit).
 *
 * The lines then go through the IMDCT / window / overlap-add kernels of
 * k_decode.hip, which also divide by 2^overallScale.
 */
#include "../../include/pacx.h"
#include "pacx_dev.h"
#include "wave_np_sum.h"

#define VQD_WAVES 4
#define VQD_DEPTH 16
#define VQD_WORDS 552

struct VqDecView {
    const uint64_t *n_tab, *p_tab;
    const int32_t *row_off;
    const int32_t *k_of;
    const uint8_t *w_of;
    const double *half_log2;
    int l_max;
    const double *log2_tan;       /* as VqView::log2_tan                       */
    const double *gauss;          /* [2r+1] normalised weights, centre at r   */
    int gauss_r;
    const double *line_freq;      /* [1024] (k + 1/2) * sampleRate / 2048      */
    /* scratch of wave w = [scr_off[w], scr_off[w+1]) doubles: the first VQD_WAVES tickets are static (wave w
       takes the (w+1)-th band from the top), so only those waves need room for the biggest bands */
    int scr_off[VQD_WAVES + 1];
};

__device__ __forceinline__ uint64_t vqd_N(const VqDecView &V, int l, long long k)
{
    if (k < 0)
        return 0;
    if (l <= 0)
        return k == 0 ? 1ull : 0ull;
    if (k == 0)
        return 1ull;
    if (l == 1)
        return 2ull;
    if (l == 2)
        return 4ull * (uint64_t)k;
    return V.n_tab[V.row_off[l] + k];
}

__device__ __forceinline__ uint64_t vqd_P(const VqDecView &V, int l, long long k)
{
    if (k < 0)
        return 0;
    if (l <= 0)
        return 1ull;
    if (l == 1)
        return 1ull + 2ull * (uint64_t)k;
    if (l == 2)
        return 1ull + 2ull * (uint64_t)k * (uint64_t)(k + 1);
    return V.p_tab[V.row_off[l] + k];
}

__device__ __forceinline__ void vqd_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double vqd_wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v = v + __shfl_xor(v, off, 64);
    return v;
}

/* device-library polynomials as real calls: inlined, their coefficients are hoisted out of the
   band loops into registers of their own (see k_vq.hip) */
__device__ __attribute__((noinline)) double vqd_cos(double x) { return cos(x); }
__device__ __attribute__((noinline)) double vqd_sin(double x) { return sin(x); }
__device__ __attribute__((noinline)) double vqd_pow256(double y) { return pow(256.0, y); }

/* up to 64 bits from the MSB-first word array */
__device__ __forceinline__ unsigned long long vqd_get(const unsigned *words, int pos, int width)
{
    if (width <= 0)
        return 0ull;
    unsigned long long out = 0;
    int left = width;
    while (left > 0) {
        const int w = pos >> 5, o = pos & 31;
        const int take = (32 - o) < left ? (32 - o) : left;
        const unsigned chunk = (words[w] >> (32 - o - take)) & (take >= 32 ? 0xFFFFFFFFu : ((1u << take) - 1u));
        out = (out << take) | chunk;
        pos += take;
        left -= take;
    }
    return out;
}

/* decode_pvq_vector: index b -> integer vector y[0..L) (LDS, zeroed here).
 * Uniform across the wave; lane 0 stores. */
__device__ __forceinline__ void vqd_pvq(const VqDecView &V, unsigned long long b, int L, int K, double *y,
                                        int lane, unsigned &flags)
{
    for (int i = lane; i < L; i += 64)
        y[i] = 0.0;
    vqd_fence();
    unsigned long long xb = 0;
    long long k = K;
    int l = L;
    for (int i = 0; i < L && k > 0; ++i, --l) {
        if (b == xb) {                                  /* the rest is zero, the last component takes k */
            if (lane == 0)
                y[L - 1] = (double)k;
            k = 0;
            break;
        }
        const unsigned long long n0 = vqd_N(V, l - 1, k);
        if (b - xb < n0)
            continue;                                   /* this component is zero */
        xb += n0;
        const unsigned long long r = b - xb;
        const unsigned long long pk1 = vqd_P(V, l - 1, k - 1);
        long long lo = 1, hi = k;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            const unsigned long long c = 2ull * (pk1 - vqd_P(V, l - 1, k - mid - 1));
            if (r < c)
                hi = mid;
            else
                lo = mid + 1;
        }
        const long long j = lo;
        if (r >= 2ull * (pk1 - vqd_P(V, l - 1, k - j - 1))) {
            flags |= PACX_ST_VQ_UNDEFINED;              /* not an index of this codebook */
            break;
        }
        const unsigned long long base = 2ull * (pk1 - vqd_P(V, l - 1, k - j));
        /* the reference reaches the negative group by walking through the whole
           positive one at the next component (its step 3 -> step 1 restart) */
        const unsigned long long group = vqd_N(V, l - 1, k - j);
        const bool neg = (r - base) >= group;
        if (lane == 0)
            y[i] = neg ? -(double)j : (double)j;
        xb += base + (neg ? group : 0ull);
        k -= j;
    }
    if (k > 0)
        flags |= PACX_ST_VQ_UNDEFINED;                  /* pulses left over: the reference raises */
    vqd_fence();
}

/* x /= ||x|| when the norm is not zero (np.linalg.norm: sqrt of the dot product) */
__device__ __forceinline__ void vqd_normalize(double *x, int n, int lane)
{
    double acc = 0.0;
    for (int i = lane; i < n; i += 64)
        acc = fma(x[i], x[i], acc);
    const double nrm = sqrt(vqd_wave_sum(acc));
    if (nrm != 0.0)
        for (int i = lane; i < n; i += 64)
            x[i] = x[i] / nrm;
    vqd_fence();
}

/* one PVQ leaf into out[0..n) */
__device__ __forceinline__ void vqd_leaf(const VqDecView &V, const unsigned *words, int &pos, double *out, int n,
                                         int bits, int lane, unsigned &flags)
{
    const int K = V.k_of[n * 33 + bits];
    const int width = V.w_of[n * 33 + bits];
    if (K < 0) {
        flags |= PACX_ST_VQ_UNDEFINED;
        for (int i = lane; i < n; i += 64)
            out[i] = 0.0;
        vqd_fence();
        return;
    }
    const unsigned long long idx = vqd_get(words, pos, width);
    pos += width;
    vqd_pvq(V, idx, n, K, out, lane, flags);
    vqd_normalize(out, n, lane);
}

/* The two leaf children of a bottom split at once (93 % of all leaves): the mid
 * leaf on lanes 0-31, the side leaf on lanes 32-63.  The index walk is the one of
 * vqd_pvq with per-half state; each half's lane 0 stores.  n <= 32 components,
 * both indices are read before the walk (their widths come from the tables). */
__device__ __forceinline__ void vqd_leaf_pair(const VqDecView &V, const unsigned *words, int &pos, double *mid,
                                              double *side, int n, int bits_mid, int bits_side, int lane,
                                              unsigned &flags)
{
    const int h = lane >> 5, l = lane & 31;
    const int bits = h ? bits_side : bits_mid;
    const int K = V.k_of[n * 33 + bits];
    const int width = V.w_of[n * 33 + bits];
    const int w_mid = __shfl(width, 0, 64);
    const unsigned long long b = vqd_get(words, pos + (h ? w_mid : 0), width);
    pos += w_mid + __shfl(width, 32, 64);
    double *y = h ? side : mid;
    if (l < n)
        y[l] = 0.0;
    vqd_fence();
    unsigned long long xb = 0;
    long long k = K;
    int ld = n;
    bool bad = false;
    for (int i = 0; i < n; ++i, --ld) {
        const bool live = k > 0 && !bad;                 /* per half */
        if (!__ballot(live))
            break;
        if (!live)
            continue;
        if (b == xb) {                                   /* the rest is zero, the last component takes k */
            if (l == 0)
                y[n - 1] = (double)k;
            k = 0;
            continue;
        }
        const unsigned long long n0 = vqd_N(V, ld - 1, k);
        if (b - xb < n0)
            continue;                                    /* this component is zero */
        xb += n0;
        const unsigned long long r = b - xb;
        const unsigned long long pk1 = vqd_P(V, ld - 1, k - 1);
        long long lo = 1, hi = k;
        while (lo < hi) {
            const long long md = (lo + hi) >> 1;
            const unsigned long long c = 2ull * (pk1 - vqd_P(V, ld - 1, k - md - 1));
            if (r < c)
                hi = md;
            else
                lo = md + 1;
        }
        const long long j = lo;
        if (r >= 2ull * (pk1 - vqd_P(V, ld - 1, k - j - 1))) {
            bad = true;                                  /* not an index of this codebook */
            continue;
        }
        const unsigned long long base = 2ull * (pk1 - vqd_P(V, ld - 1, k - j));
        const unsigned long long group = vqd_N(V, ld - 1, k - j);
        const bool neg = (r - base) >= group;
        if (l == 0)
            y[i] = neg ? -(double)j : (double)j;
        xb += base + (neg ? group : 0ull);
        k -= j;
    }
    if (__ballot(bad || k > 0))
        flags |= PACX_ST_VQ_UNDEFINED;
    vqd_fence();
    /* x / ||x|| per half (sums of squares of integers: exact in any order) */
    const double v = (l < n) ? y[l] : 0.0;
    double acc = v * v;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1)
        acc = acc + __shfl_xor(acc, off, 32);
    const double nrm = sqrt(acc);
    if (l < n && nrm != 0.0)
        y[l] = v / nrm;
    vqd_fence();
}

/* frames of the split tree (per wave, in LDS) */
struct VqdFrame {
    double theta;
    int out, n, half, a_side, reg, state;
};

/* split_band_decode: unit vector of dimension n0 at scr[out0..], bits0 bits. */
__device__ __forceinline__ void vqd_shape(const VqDecView &V, const unsigned *words, int &pos, double *scr,
                                          int out0, int n0, int bits0, int reg0, VqdFrame *fr, int lane,
                                          unsigned &flags)
{
    const double half_pi = 1.5707963267948966;
    if (bits0 <= PACX_VQ_SPLIT_BITS) {
        /* the non-split branch normalises a second time (:468-472) */
        vqd_leaf(V, words, pos, scr + out0, n0, bits0, lane, flags);
        vqd_normalize(scr + out0, n0, lane);
        return;
    }
    int depth = 0;
    int cur_out = out0, cur_n = n0, cur_bits = bits0, cur_reg = reg0;
    int phase = 0;                  /* 0: enter split node, 1: side of frame[depth], 2: combine frame[depth] */
    for (;;) {
        if (phase == 0) {
            if (depth >= VQD_DEPTH) {
                flags |= PACX_ST_VQ_UNDEFINED;
                return;
            }
            const int half = cur_n - cur_n / 2;
            int a_theta = (int)floor((double)cur_bits / (double)half + V.half_log2[half]);
            int a_rest = cur_bits - a_theta;
            if (a_rest < 0)
                a_rest = 0;
            double theta = 0.0;
            unsigned long long theta_code = 0;
            if (a_theta > 0 && a_theta <= 62) {
                const unsigned long long code = vqd_get(words, pos, a_theta);
                theta_code = code;
                const unsigned long long mag = code & ((1ull << (a_theta - 1)) - 1ull);
                const double den = (a_theta <= 53) ? (double)((1ull << a_theta) - 1ull) : ldexp(1.0, a_theta);
                double dq = (double)(2ull * mag) / den;
                if (code >> (a_theta - 1))
                    dq = -dq;
                theta = dq * half_pi;
            } else if (a_theta > 62) {
                flags |= PACX_ST_VQ_UNDEFINED;
            }
            pos += a_theta > 0 ? a_theta : 0;
            int a_mid = 0;
            if (theta != 0.0) {
                double lt;
                if (a_theta <= PACX_VQ_THETA_TABLE_BITS && theta > 0.0)
                    lt = V.log2_tan[((1 << (a_theta - 1)) - 1) + (int)theta_code];
                else
                    lt = log2(tan(fabs(theta)) + PACX_EPS);
                const double v = ((double)a_rest - (double)(half - 1) * lt) / 2.0;
                const double f = floor(v);
                a_mid = (f < 0.0) ? 0 : ((f > (double)a_rest) ? a_rest : (int)f);
            }
            VqdFrame &F = fr[depth];
            if (lane == 0) {
                F.theta = theta;
                F.out = cur_out;
                F.n = cur_n;
                F.half = half;
                F.a_side = a_rest - a_mid;
                F.reg = cur_reg;
                F.state = 0;
            }
            vqd_fence();
            const int mid_slot = cur_reg;
            if (a_mid > PACX_VQ_SPLIT_BITS) {
                depth += 1;
                cur_out = mid_slot;
                cur_n = half;
                cur_bits = a_mid;
                cur_reg = cur_reg + 2 * half;
                continue;                                   /* phase 0 on the mid child */
            }
            if (a_mid > 0 && a_rest - a_mid > 0 && a_rest - a_mid <= PACX_VQ_SPLIT_BITS && half >= 2 && half <= 32) {
                /* both children are small leaves: decode them side by side */
                vqd_leaf_pair(V, words, pos, scr + mid_slot, scr + mid_slot + half, half, a_mid, a_rest - a_mid,
                              lane, flags);
                phase = 2;
                continue;
            }
            if (a_mid > 0) {
                vqd_leaf(V, words, pos, scr + mid_slot, half, a_mid, lane, flags);
            } else {
                for (int i = lane; i < half; i += 64)
                    scr[mid_slot + i] = 0.0;
                vqd_fence();
            }
            phase = 1;
            continue;
        }
        if (phase == 1) {
            const int half = fr[depth].half, a_side = fr[depth].a_side, reg = fr[depth].reg;
            const int side_slot = reg + half;
            if (lane == 0)
                fr[depth].state = 1;
            vqd_fence();
            if (a_side > PACX_VQ_SPLIT_BITS) {
                depth += 1;
                cur_out = side_slot;
                cur_n = half;
                cur_bits = a_side;
                cur_reg = reg + 2 * half;
                phase = 0;
                continue;
            }
            if (a_side > 0) {
                vqd_leaf(V, words, pos, scr + side_slot, half, a_side, lane, flags);
            } else {
                for (int i = lane; i < half; i += 64)
                    scr[side_slot + i] = 0.0;
                vqd_fence();
            }
            phase = 2;
            continue;
        }
        /* phase 2: left/right from mid/side (:455-466) */
        {
            const VqdFrame F = fr[depth];
            const double ct = vqd_cos(F.theta), st = vqd_sin(F.theta);
            const double root2 = sqrt(2.0);
            const int cut = F.n / 2;
            const double *mid = scr + F.reg, *side = scr + F.reg + F.half;
            double *out = scr + F.out;
            for (int i = lane; i < F.half; i += 64) {
                const double m = mid[i] * ct, s = side[i] * st;
                const double left = (m + s) / root2;
                const double right = (m - s) / root2;
                if (i < cut)
                    out[i] = left;                          /* an odd band drops the last left value */
                out[cut + i] = right;
            }
            vqd_fence();
            vqd_normalize(out, F.n, lane);
        }
        if (depth == 0)
            return;
        depth -= 1;
        phase = (fr[depth].state == 0) ? 1 : 2;
    }
}

struct VqDecArgs {
    long long n_cf;
    const uint8_t *payload;
    int payload_stride;
    const long long *offsets;
    const int32_t *n_bytes;
    uint8_t *cf_flags;
    int32_t *overall;          /* [cf][8]                                   */
    int32_t *bit_alloc;        /* [cf][band_stride]                         */
    double *lines;             /* [cf][1024] before SBR reconstruction and /2^overall */
    uint8_t *sbr_flag;         /* [cf] 1: Decode_SBR applies                */
    uint32_t *status;
    int scr_len;
    int redo;                  /* 1: decode only the blocks k_vq_dec_frame left (sbr_flag 0x80) */
};

__global__ __launch_bounds__(64 * VQD_WAVES) void k_vq_dec(PacxTables T, VqDecView V, VqDecArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *words = (unsigned *)smem;                                  /* VQD_WORDS            */
    double *lines_s = (double *)(smem + VQD_WORDS * 4);                  /* 1024                 */
    int *item_pos = (int *)(lines_s + PACX_M_LONG);                      /* [8*32] band bit position */
    int *item_ba = item_pos + PACX_SUB * PACX_MAX_BANDS;                 /* [8*32]               */
    int *misc = item_ba + PACX_SUB * PACX_MAX_BANDS;                     /* ticket, n_items, short, sbr */
    VqdFrame *frames = (VqdFrame *)(misc + 4);                           /* waves * DEPTH        */
    double *scr_all = (double *)(frames + VQD_WAVES * VQD_DEPTH);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long cf = blockIdx.x;
    if (cf >= A.n_cf)
        return;
    if (A.redo && A.sbr_flag[cf] != 0x80)
        return;                                            /* k_vq_dec_frame decoded this block */
    /* as k_unpack (k_decode.hip): a record that is truncated or carries an impossible
       allocation gets PACX_ST_MALFORMED and decodes to zeros; nothing beyond n_bytes is read */
    int nbytes = A.n_bytes[cf];
    bool bad = nbytes < 1 || nbytes > 4 * (VQD_WORDS - 2);
    if (bad)
        nbytes = 0;
    const uint8_t *src = A.payload + (A.offsets ? A.offsets[cf] : cf * (long long)A.payload_stride);
    const int n_words = (nbytes + 3) >> 2;
    for (int i = tid; i < VQD_WORDS; i += 64 * VQD_WAVES) {
        unsigned v = 0;
        if (i < n_words) {
            const int b0 = 4 * i;
            v = ((unsigned)src[b0] << 24) | ((b0 + 1 < nbytes ? (unsigned)src[b0 + 1] : 0u) << 16) |
                ((b0 + 2 < nbytes ? (unsigned)src[b0 + 2] : 0u) << 8) | (b0 + 3 < nbytes ? (unsigned)src[b0 + 3] : 0u);
        }
        words[i] = v;
    }
    for (int i = tid; i < PACX_M_LONG; i += 64 * VQD_WAVES)
        lines_s[i] = 0.0;
    __syncthreads();
    if (tid == 0) {
        const int limit = 8 * nbytes;
        const unsigned fl = bad ? 0u
                                : (unsigned)vqd_get(words, 0, 1) | ((unsigned)vqd_get(words, 1, 1) << 1) |
                                      ((unsigned)vqd_get(words, 2, 1) << 2);
        A.cf_flags[cf] = (uint8_t)fl;
        const int shrt = (fl >> 1) & 1;
        const int nb = shrt ? T.nb_short : T.nb_long;
        const int32_t *cnt = shrt ? T.band_lines_short : T.band_lines_long;
        int pos = 3;
        /* Decode_SBR is chosen per block: an SBR file, a long block, and some
           omitted band with a non-zero allocation (coder/pacfile.py:661-666) */
        int sbr = 0;
        if (3 + T.n_scale_bits + T.n_mant_size_bits * nb > limit)
            bad = true;
        if (T.use_sbr && !shrt && !bad) {
            int p2 = pos + T.n_scale_bits;
            for (int b = 0; b < nb; ++b) {
                if (b >= T.first_omitted && vqd_get(words, p2, T.n_mant_size_bits) != 0)
                    sbr = 1;
                p2 += T.n_mant_size_bits;
            }
        }
        for (int s = 0; s < (shrt ? PACX_SUB : 1) && !bad; ++s) {
            if (pos + T.n_scale_bits + T.n_mant_size_bits * nb > limit) {
                bad = true;
                break;
            }
            A.overall[cf * PACX_SUB + s] = (int)vqd_get(words, pos, T.n_scale_bits);
            pos += T.n_scale_bits;
            int body = pos + T.n_mant_size_bits * nb;
            for (int b = 0; b < nb; ++b) {
                int a = (int)vqd_get(words, pos, T.n_mant_size_bits);
                if (a)
                    a += 1;
                pos += T.n_mant_size_bits;
                const int span = a * ((sbr && b >= T.first_omitted) ? 1 : cnt[b]);
                if (a > 16 || body + span > limit) {
                    bad = true;
                    break;
                }
                A.bit_alloc[cf * T.band_stride + s * nb + b] = a;
                item_pos[s * PACX_MAX_BANDS + b] = body;
                item_ba[s * PACX_MAX_BANDS + b] = a;
                body += span;
            }
            pos = body;
        }
        if (bad) {
            for (int s = 0; s < PACX_SUB; ++s)
                A.overall[cf * PACX_SUB + s] = 0;
            for (int i = 0; i < T.band_stride; ++i)
                A.bit_alloc[cf * T.band_stride + i] = 0;
            atomicOr(&A.status[cf], 32u);                          /* PACX_ST_MALFORMED */
            sbr = 0;
        } else if (!shrt) {
            for (int s = 1; s < PACX_SUB; ++s)
                A.overall[cf * PACX_SUB + s] = 0;
        }
        misc[0] = VQD_WAVES;                               /* tickets 0 .. VQD_WAVES-1 are the waves' own */
        misc[1] = bad ? 0 : (shrt ? PACX_SUB : 1) * nb;
        misc[2] = shrt;
        misc[3] = sbr;
        A.sbr_flag[cf] = (uint8_t)sbr;
    }
    __syncthreads();
    const int shrt = misc[2], sbr = misc[3], n_items = misc[1];
    const int nb = shrt ? T.nb_short : T.nb_long;
    const int32_t *__restrict__ lower = shrt ? T.band_lower_short : T.band_lower_long;
    const int32_t *__restrict__ count = shrt ? T.band_lines_short : T.band_lines_long;
    double *scr = scr_all + V.scr_off[wave];
    VqdFrame *fr = frames + wave * VQD_DEPTH;
    unsigned raised = 0;
    for (int tk = wave;; tk = -1) {
        if (tk < 0) {
            if (lane == 0)
                tk = atomicAdd(&misc[0], 1);
            tk = __builtin_amdgcn_readfirstlane(tk);
        }
        if (tk >= n_items)
            break;
        const int s = tk / nb, b = nb - 1 - (tk % nb);           /* large bands first */
        const int ba = item_ba[s * PACX_MAX_BANDS + b];
        if (!ba)
            continue;
        int pos = item_pos[s * PACX_MAX_BANDS + b];
        /* where the band's lines go: iMant advances by 1 over an omitted band of
           an SBR block (coder/codec.py:121-134) */
        int n = count[b], at = lower[b];
        if (sbr && b >= T.first_omitted) {
            n = 1;
            at = lower[T.first_omitted] + (b - T.first_omitted);
        }
        const int r_bits = ba * n;
        int bits_gain = (int)floor((double)r_bits / (double)n + V.half_log2[n]);
        int bits_shape = r_bits - bits_gain;
        if (bits_shape < 0)
            bits_shape = 0;
        if (bits_shape != 0) {
            const int before = pos;
            vqd_shape(V, words, pos, scr, 0, n, bits_shape, n, fr, lane, raised);
            bits_gain += bits_shape - (pos - before);
        } else {
            for (int i = lane; i < n; i += 64)
                scr[i] = 1.0;
            vqd_fence();
        }
        /* gain: DequantizeUniform then the inverse mu-law, times L (:533-537) */
        double deq = 0.0;
        if (bits_gain > 64) {
            raised |= PACX_ST_VQ_UNDEFINED;
        } else if (bits_gain > 0) {
            const unsigned long long code = vqd_get(words, pos, bits_gain);
            const unsigned long long mag = (bits_gain == 64) ? (code & 0x7FFFFFFFFFFFFFFFull)
                                                              : (code & ((1ull << (bits_gain - 1)) - 1ull));
            const double den = (bits_gain <= 53) ? (double)((1ull << bits_gain) - 1ull) : ldexp(1.0, bits_gain);
            deq = (2.0 * (double)mag) / den;
            if (code >> (bits_gain - 1))
                deq = -deq;
        }
        const double sgn = (deq > 0.0) ? 1.0 : ((deq < 0.0) ? -1.0 : 0.0);
        const double gain = (sgn / 255.0 * (vqd_pow256(fabs(deq)) - 1.0)) * (double)n;
        double *dst = lines_s + (shrt ? s * PACX_M_SHORT : 0) + at;
        for (int i = lane; i < n; i += 64)
            dst[i] = gain * scr[i];
        vqd_fence();
    }
    if (raised && lane == 0)
        atomicOr(&A.status[cf], raised);
    __syncthreads();
    double *out = A.lines + cf * PACX_M_LONG;
    for (int i = tid; i < PACX_M_LONG; i += 64 * VQD_WAVES)
        out[i] = lines_s[i];
}

/* ------------------------------------------------ frame-level decode (k_vq_dec_frame) */
/* k_vq_dec walks a band's tree depth first with a whole wave per node and decodes the pyramid indices
 * one component at a time with wave-uniform 64-bit arithmetic: 84 k vector instructions per channel-block,
 * the vector unit 80 % busy.  Here, as in k_vq_frame of the encoder, one workgroup takes a channel-block
 * through four stages over ALL its bands:
 *   1. parse -- one band per LANE: the band's fields are read in the stream's (depth-first) order, but only
 *      the angles are evaluated; every node of the tree goes into a store in LDS (splits with their angle,
 *      leaves with the position of their index), its vector gets room in the buffer of its depth;
 *   2. leaves -- one leaf per LANE (all four waves): index -> pulses -> unit vector;
 *   3. combine -- level by level from the deepest: left/right from mid/side and the normalisation, several
 *      nodes per pass (one per aligned block of Q lanes, so that the xor butterfly adds in the order of
 *      vqd_normalize's 64-lane one);
 *   4. gains and the lines.
 * Same arithmetic per node as k_vq_dec.  A block whose trees do not fit is left to k_vq_dec (sbr_flag 0x80). */
#define VQDF_NCAP 304
#define VQDF_BUF 1536                  /* doubles: every node's vector lives in a region of pow2ceil(n) of them */
#define VQDF_VB 64
#define VQDF_STACK 16
struct VqdfStore {
    double *theta, *sn;                /* [NCAP] a split's dequantised angle (stage 2 turns it into its cosine), its sine */
    unsigned short *nn, *off, *bitpos, *kid0, *kid1, *reg;      /* reg: size of the node's region */
    unsigned char *kind, *bits, *depth, *band;   /* kind 0 split, 1 leaf, 2 zeros */
    __device__ __forceinline__ void bind(unsigned char *p)
    {
        theta = (double *)p;
        sn = theta + VQDF_NCAP;
        nn = (unsigned short *)(p + 2 * VQDF_NCAP * 8);
        off = nn + VQDF_NCAP;
        bitpos = off + VQDF_NCAP;
        kid0 = bitpos + VQDF_NCAP;
        kid1 = kid0 + VQDF_NCAP;
        reg = kid1 + VQDF_NCAP;
        kind = (unsigned char *)(reg + VQDF_NCAP);
        bits = kind + VQDF_NCAP;
        depth = bits + VQDF_NCAP;
        band = depth + VQDF_NCAP;
    }
};
#define VQDF_STORE_BYTES (2 * VQDF_NCAP * 8 + 6 * VQDF_NCAP * 2 + 4 * VQDF_NCAP)
#define VQDF_FIXED (VQD_WORDS * 4 + 7 * VQDF_VB * 4 + 32 * 4 + VQDF_VB * VQDF_STACK * 4 + VQDF_NCAP * 2)
#define VQDF_SMEM (VQDF_FIXED + VQDF_BUF * 8 + VQDF_STORE_BYTES)

/* decode_pvq_vector for ONE lane: index b -> integer pulses y[0..L) (zeroed here) */
__device__ __forceinline__ bool vqdf_pvq_lane(const VqDecView &V, unsigned long long b, int L, int K, double *y)
{
    for (int i = 0; i < L; ++i)
        y[i] = 0.0;
    unsigned long long xb = 0;
    long long k = K;
    int l = L;
    for (int i = 0; i < L && k > 0; ++i, --l) {
        if (b == xb) {                                  /* the rest is zero, the last component takes k */
            y[L - 1] = (double)k;
            k = 0;
            break;
        }
        const unsigned long long n0 = vqd_N(V, l - 1, k);
        if (b - xb < n0)
            continue;                                   /* this component is zero */
        xb += n0;
        const unsigned long long r = b - xb;
        const unsigned long long pk1 = vqd_P(V, l - 1, k - 1);
        long long lo = 1, hi = k;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            const unsigned long long c = 2ull * (pk1 - vqd_P(V, l - 1, k - mid - 1));
            if (r < c)
                hi = mid;
            else
                lo = mid + 1;
        }
        const long long j = lo;
        if (r >= 2ull * (pk1 - vqd_P(V, l - 1, k - j - 1)))
            return false;                               /* not an index of this codebook */
        const unsigned long long base = 2ull * (pk1 - vqd_P(V, l - 1, k - j));
        const unsigned long long group = vqd_N(V, l - 1, k - j);
        const bool neg = (r - base) >= group;
        y[i] = neg ? -(double)j : (double)j;
        xb += base + (neg ? group : 0ull);
        k -= j;
    }
    return k == 0;                                      /* pulses left over: the reference raises */
}

#ifdef PACX_VQD_DEBUG
/* stage stamps of k_vq_dec_frame (thread 0 of every workgroup), a measuring aid (build.py --phase-debug) */
__device__ long long g_vqd_dbg[8];
extern "C" int pacx_debug_read_vqd(long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vqd_dbg), sizeof(long long) * n);
}
#define VQD_T(k) do { long long t_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                      if (threadIdx.x == 0 && vqd_last) atomicAdd((unsigned long long *)&g_vqd_dbg[k], (unsigned long long)(t_ - vqd_last)); \
                      vqd_last = t_; } while (0)
#else
#define VQD_T(k) do { } while (0)
#endif
__global__ __launch_bounds__(64 * VQD_WAVES, 5) void k_vq_dec_frame(PacxTables T, VqDecView V, VqDecArgs A)
{
#ifdef PACX_VQD_DEBUG
    long long vqd_last = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *words = (unsigned *)smem;                                  /* VQD_WORDS */
    int *item_pos = (int *)(smem + VQD_WORDS * 4);                       /* VB: first bit of the band's fields */
    int *item_ba = item_pos + VQDF_VB;                                   /* VB */
    int *item_end = item_ba + VQDF_VB;                                   /* VB: first bit behind the shape's fields */
    int *item_bg = item_end + VQDF_VB;                                   /* VB: gain bits before the slack */
    int *item_bs = item_bg + VQDF_VB;                                    /* VB: shape bits */
    int *item_root = item_bs + VQDF_VB;                                  /* VB: root node, -1 none */
    int *item_n = item_root + VQDF_VB;                                   /* VB: vector length */
    int *misc = item_n + VQDF_VB;                                        /* 0 nodes, 1 redo, 2 n_items, 3 short, 4 sbr, 5 flags, 6 max depth, 8..24 level fill */
    unsigned *stack = (unsigned *)(misc + 32);                           /* [VB][STACK]: parent | a_side << 16 */
    unsigned short *lst = (unsigned short *)(stack + VQDF_VB * VQDF_STACK);   /* [NCAP] a level's small nodes */
    double *buf0 = (double *)(smem + VQDF_FIXED);
    VqdfStore N;
    N.bind(smem + VQDF_FIXED + VQDF_BUF * 8);
    static_assert(VQDF_FIXED % 8 == 0, "doubles behind the fixed part");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long cf = blockIdx.x;
    if (cf >= A.n_cf)
        return;
    int nbytes = A.n_bytes[cf];
    bool bad = nbytes < 1 || nbytes > 4 * (VQD_WORDS - 2);
    if (bad)
        nbytes = 0;
    const uint8_t *src = A.payload + (A.offsets ? A.offsets[cf] : cf * (long long)A.payload_stride);
    const int n_words = (nbytes + 3) >> 2;
    for (int i = tid; i < VQD_WORDS; i += 64 * VQD_WAVES) {
        unsigned v = 0;
        if (i < n_words) {
            const int b0 = 4 * i;
            v = ((unsigned)src[b0] << 24) | ((b0 + 1 < nbytes ? (unsigned)src[b0 + 1] : 0u) << 16) |
                ((b0 + 2 < nbytes ? (unsigned)src[b0 + 2] : 0u) << 8) | (b0 + 3 < nbytes ? (unsigned)src[b0 + 3] : 0u);
        }
        words[i] = v;
    }
    double *out = A.lines + cf * PACX_M_LONG;
    for (int i = tid; i < PACX_M_LONG; i += 64 * VQD_WAVES)
        out[i] = 0.0;
    if (tid < 32)
        misc[tid] = 0;
    __syncthreads();
    VQD_T(7);
    if (tid < 64) {
        /* header: flags, then per (sub-)block the overall scale and the allocations -- what k_vq_dec's thread 0
           reads field by field, one band per lane here: the fields' positions are static inside a (sub-)block,
           a band's body starts behind the bodies of the bands before it (a scan), and the next sub-block
           behind the last body */
        const int limit = 8 * nbytes;
        const unsigned fl = bad ? 0u
                                : (unsigned)vqd_get(words, 0, 1) | ((unsigned)vqd_get(words, 1, 1) << 1) |
                                      ((unsigned)vqd_get(words, 2, 1) << 2);
        const int shrt = (fl >> 1) & 1;
        const int nb = shrt ? T.nb_short : T.nb_long;
        const int32_t *cnt = shrt ? T.band_lines_short : T.band_lines_long;
        const int head = T.n_scale_bits + T.n_mant_size_bits * nb;
        const int b = lane;
        if (3 + head > limit)
            bad = true;
        int sbr = 0;
        if (T.use_sbr && !shrt && !bad) {
            const bool coded = b < nb && b >= T.first_omitted &&
                               vqd_get(words, 3 + T.n_scale_bits + T.n_mant_size_bits * b, T.n_mant_size_bits) != 0;
            sbr = __builtin_amdgcn_ballot_w64(coded) ? 1 : 0;
        }
        int pos = 3;
        for (int s = 0; s < (shrt ? PACX_SUB : 1) && !bad; ++s) {
            if (pos + head > limit) {
                bad = true;
                break;
            }
            int a = (b < nb) ? (int)vqd_get(words, pos + T.n_scale_bits + T.n_mant_size_bits * b, T.n_mant_size_bits) : 0;
            if (a)
                a += 1;
            const int span = (b < nb) ? a * ((sbr && b >= T.first_omitted) ? 1 : cnt[b]) : 0;
            int incl = span;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o)
                    incl += t;
            }
            const int body = pos + head + incl - span;
            if (__builtin_amdgcn_ballot_w64(b < nb && (a > 16 || body + span > limit))) {
                bad = true;
                break;
            }
            if (b < nb) {
                A.bit_alloc[cf * T.band_stride + s * nb + b] = a;
                item_pos[s * nb + b] = body;
                item_ba[s * nb + b] = a;
            }
            if (lane == 0)
                A.overall[cf * PACX_SUB + s] = (int)vqd_get(words, pos, T.n_scale_bits);
            pos += head + __shfl(incl, 63, 64);
        }
        if (bad) {
            if (lane < PACX_SUB)
                A.overall[cf * PACX_SUB + lane] = 0;
            for (int i = lane; i < T.band_stride; i += 64)
                A.bit_alloc[cf * T.band_stride + i] = 0;
            if (lane == 0)
                atomicOr(&A.status[cf], 32u);                      /* PACX_ST_MALFORMED */
            sbr = 0;
        } else if (!shrt && lane >= 1 && lane < PACX_SUB) {
            A.overall[cf * PACX_SUB + lane] = 0;
        }
        if (lane == 0) {
            A.cf_flags[cf] = (uint8_t)fl;
            misc[2] = bad ? 0 : (shrt ? PACX_SUB : 1) * nb;
            misc[3] = shrt;
            misc[4] = sbr;
            A.sbr_flag[cf] = (uint8_t)sbr;
        }
    }
    __syncthreads();
    VQD_T(0);
    const int shrt = misc[3], sbr = misc[4], n_items = misc[2];
    const int nb = shrt ? T.nb_short : T.nb_long;
    const int32_t *__restrict__ lower = shrt ? T.band_lower_short : T.band_lower_long;
    const int32_t *__restrict__ count = shrt ? T.band_lines_short : T.band_lines_long;
    const double half_pi = 1.5707963267948966;
    bool undefined = false;

    /* ---- 1. parse: one band per lane */
    if (wave == 0) {
        const int vb = lane;
        const bool live = vb < n_items && item_ba[vb < n_items ? vb : 0] != 0;
        const int s = live ? vb / nb : 0, b = vb - s * nb;
        int n = live ? count[b] : 1;
        if (live && sbr && b >= T.first_omitted)
            n = 1;
        const int r_bits = live ? item_ba[vb] * n : 0;
        int bits_gain = live ? (int)floor((double)r_bits / (double)n + V.half_log2[n]) : 0;
        int bits_shape = r_bits - bits_gain;
        if (bits_shape < 0)
            bits_shape = 0;
        int pos = live ? item_pos[vb] : 0;
        unsigned *stk = stack + vb * VQDF_STACK;
        int sp = 0;
        /* the node being entered: length, bits, where it hangs (parent, 0 mid / 1 side; root: parent -1),
           its region */
        int c_n = n, c_bits = bits_shape, c_par = -1, c_side = 0, c_depth = 0;
        bool have = live && bits_shape != 0;
        int root = -1, my_depth = 0;
        bool redo = false;
        int c_reg = 0;
        if (have) {
            c_reg = 1;
            while (c_reg < n)
                c_reg <<= 1;
        }
        int c_off, room = c_reg;
        {
            int incl = room;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o)
                    incl += t;
            }
            c_off = incl - room;
            if (__shfl(incl, 63, 64) > VQDF_BUF)
                redo = true;
        }
        if (redo)
            have = false;
        while (__builtin_amdgcn_ballot_w64(have || sp > 0)) {
            if (!have && sp > 0) {                          /* the pending side half of the innermost split */
                const unsigned e = stk[--sp];
                c_par = (int)(e & 0xFFFFu);
                c_bits = (int)(e >> 16);
                c_side = 1;
                c_depth = N.depth[c_par] + 1;
                c_n = N.nn[c_par] - N.nn[c_par] / 2;
                c_reg = N.reg[c_par] >> 1;
                c_off = N.off[c_par] + c_reg;
                have = true;
            }
            if (!have)
                continue;
            /* a node of this lane's tree */
            const int id = atomicAdd(&misc[0], 1);
            if (id >= VQDF_NCAP || c_depth > VQD_DEPTH) {
                redo = true;
                have = false;
                sp = 0;
                continue;
            }
            my_depth = max(my_depth, c_depth);
            N.nn[id] = (unsigned short)c_n;
            N.off[id] = (unsigned short)c_off;
            N.reg[id] = (unsigned short)c_reg;
            N.depth[id] = (unsigned char)c_depth;
            N.band[id] = (unsigned char)vb;
            N.kid0[id] = 0xFFFF;
            N.kid1[id] = 0xFFFF;
            if (c_par < 0)
                root = id;
            else if (c_side)
                N.kid1[c_par] = (unsigned short)id;
            else
                N.kid0[c_par] = (unsigned short)id;
            if (c_bits <= 0) {
                N.kind[id] = 2;                             /* a half without bits: zeros */
                have = false;
                continue;
            }
            if (c_bits <= PACX_VQ_SPLIT_BITS) {
                const int width = V.w_of[c_n * 33 + c_bits];
                N.kind[id] = 1;
                N.bits[id] = (unsigned char)c_bits;
                N.bitpos[id] = (unsigned short)pos;
                pos += width;
                have = false;
                continue;
            }
            if (c_depth >= VQD_DEPTH) {                     /* deeper than any real tree (k_vq_dec stops here too) */
                undefined = true;
                N.kind[id] = 2;
                have = false;
                continue;
            }
            /* split: the angle, the bit split */
            const int half = c_n - c_n / 2;
            const int a_theta = (int)floor((double)c_bits / (double)half + V.half_log2[half]);
            int a_rest = c_bits - a_theta;
            if (a_rest < 0)
                a_rest = 0;
            double theta = 0.0;
            unsigned long long theta_code = 0;
            if (a_theta > 0 && a_theta <= 31) {
                /* (the same values in 32-bit integers: one conversion instruction where the 64-bit ones take a dozen) */
                const unsigned code = (unsigned)vqd_get(words, pos, a_theta);
                theta_code = code;
                const unsigned mag = code & ((1u << (a_theta - 1)) - 1u);
                double dq = (double)(2u * mag) / (double)((1u << a_theta) - 1u);
                if (code >> (a_theta - 1))
                    dq = -dq;
                theta = dq * half_pi;
            } else if (a_theta > 0 && a_theta <= 62) {
                const unsigned long long code = vqd_get(words, pos, a_theta);
                theta_code = code;
                const unsigned long long mag = code & ((1ull << (a_theta - 1)) - 1ull);
                const double den = (a_theta <= 53) ? (double)((1ull << a_theta) - 1ull) : ldexp(1.0, a_theta);
                double dq = (double)(2ull * mag) / den;
                if (code >> (a_theta - 1))
                    dq = -dq;
                theta = dq * half_pi;
            } else if (a_theta > 62) {
                undefined = true;
            }
            pos += a_theta > 0 ? a_theta : 0;
            int a_mid = 0;
            if (theta != 0.0) {
                double lt;
                if (a_theta <= PACX_VQ_THETA_TABLE_BITS && theta > 0.0)
                    lt = V.log2_tan[((1 << (a_theta - 1)) - 1) + (int)theta_code];
                else
                    lt = log2(tan(fabs(theta)) + PACX_EPS);
                const double v = ((double)a_rest - (double)(half - 1) * lt) / 2.0;
                const double f = floor(v);
                a_mid = (f < 0.0) ? 0 : ((f > (double)a_rest) ? a_rest : (int)f);
            }
            N.kind[id] = 0;
            N.theta[id] = theta;
            if (sp >= VQDF_STACK) {
                redo = true;
                have = false;
                sp = 0;
                continue;
            }
            stk[sp++] = (unsigned)id | ((unsigned)(a_rest - a_mid) << 16);
            c_par = id;
            c_side = 0;
            c_bits = a_mid;
            c_n = half;
            c_reg >>= 1;                                    /* the mid half of the region; c_off stays */
            c_depth += 1;
            have = true;
        }
        if (vb < VQDF_VB) {
            item_end[vb] = pos;
            item_bg[vb] = bits_gain;
            item_bs[vb] = bits_shape;
            item_root[vb] = root;
            item_n[vb] = n;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            my_depth = max(my_depth, __shfl_xor(my_depth, o, 64));
        if (lane == 0)
            misc[6] = my_depth;
        if (__builtin_amdgcn_ballot_w64(redo) && lane == 0)
            misc[1] = 1;
    }
    __syncthreads();
    VQD_T(1);
    if (misc[1]) {
        if (tid == 0)
            A.sbr_flag[cf] = 0x80;                          /* k_vq_dec decodes this block */
        return;
    }
    const int n_nodes = misc[0], max_depth = misc[6];
    /* ---- 2. leaves (and the halves without bits): one per lane */
    for (int j = tid; j < n_nodes; j += 64 * VQD_WAVES) {
        const int kd = N.kind[j];
        if (kd == 0) {
            /* a split has nothing to decode: its lane takes the cosine and sine of its angle off the combine
               passes, which are a chain of levels (here they run beside the index walks) */
            const double theta = N.theta[j];
            N.theta[j] = vqd_cos(theta);
            N.sn[j] = vqd_sin(theta);
            continue;
        }
        const int n = N.nn[j];
        double *y = buf0 + N.off[j];
        if (kd == 2) {
            for (int i = 0; i < n; ++i)
                y[i] = 0.0;
            continue;
        }
        const int bits = N.bits[j];
        const int K = V.k_of[n * 33 + bits];
        const int width = V.w_of[n * 33 + bits];
        if (K < 0) {
            undefined = true;
            for (int i = 0; i < n; ++i)
                y[i] = 0.0;
            continue;
        }
        const unsigned long long idx = vqd_get(words, N.bitpos[j], width);
#ifdef VQDF_STUB_LEAF          /* timing experiment only: what the index walks cost */
        for (int i = 0; i < n; ++i)
            y[i] = (double)((idx >> i) & 1ull);
#else
        if (!vqdf_pvq_lane(V, idx, n, K, y))
            undefined = true;
#endif
        /* x / ||x|| (sums of squares of integers: exact in any order) */
        double acc = 0.0;
        for (int i = 0; i < n; ++i)
            acc += y[i] * y[i];
        const double nrm = sqrt(acc);
        if (nrm != 0.0)
            for (int i = 0; i < n; ++i)
                y[i] = y[i] / nrm;
    }
    __syncthreads();
    VQD_T(2);
    /* ---- 3. combine, level by level from the deepest split; a root that is a leaf is normalised a second
       time, as the reference's non-split branch does (coder/gain_shape_quantize.py:468-472) */
    for (int d = max_depth; d >= 0; --d) {
        /* every wave walks the node store: nodes of more than 64 components are dealt round-robin and take a
           whole wave each; the others are listed (every wave writes the same list) and combined several per
           pass, one node per aligned block of Q lanes -- the xor butterfly of the normalisation then adds in the
           order of vqd_normalize's 64-lane one (plus exact zeros), and cos / sin run once per pass */
        int item = 0, n_small = 0, my_max = 0;
        for (int j0 = 0; j0 < n_nodes; j0 += 64) {
            const int j = j0 + lane;
            const bool mine = j < n_nodes && N.depth[j] == d && (N.kind[j] == 0 || (d == 0 && N.kind[j] == 1));
            const int nj = mine ? N.nn[j] : 0;
            const unsigned long long ms = __builtin_amdgcn_ballot_w64(mine && nj <= 64);
            if (mine && nj <= 64) {
                lst[n_small + __popcll(ms & ((1ull << lane) - 1ull))] = (unsigned short)j;
                my_max = max(my_max, nj);
            }
            n_small += __popcll(ms);
            unsigned long long m = __builtin_amdgcn_ballot_w64(mine && nj > 64);
            for (; m; m &= m - 1, ++item) {
                if ((item & (VQD_WAVES - 1)) != wave)
                    continue;
                const int node = j0 + __builtin_ctzll(m);
                const int n = N.nn[node];
                double *o = buf0 + N.off[node];
                if (N.kind[node] == 0) {
                    /* in place: the output overwrites the children's regions, so a lane reads everything it
                       needs (up to 8 x 64 components per half) before anybody writes */
                    const int cut = n / 2, half = n - cut;
                    const double ct = N.theta[node], st = N.sn[node];
                    const double root2 = sqrt(2.0);
                    const double *mid = buf0 + N.off[N.kid0[node]], *side = buf0 + N.off[N.kid1[node]];
                    double lft[8], rgt[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = lane + 64 * u;
                        lft[u] = rgt[u] = 0.0;
                        if (i < half) {
                            const double mm = mid[i] * ct, ss = side[i] * st;
                            lft[u] = (mm + ss) / root2;
                            rgt[u] = (mm - ss) / root2;
                        }
                    }
                    vqd_fence();
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = lane + 64 * u;
                        if (i < half) {
                            if (i < cut)
                                o[i] = lft[u];              /* an odd band drops the last left value */
                            o[cut + i] = rgt[u];
                        }
                    }
                    vqd_fence();
                }
                vqd_normalize(o, n, lane);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            my_max = max(my_max, __shfl_xor(my_max, off, 64));
        vqd_fence();
        if (n_small) {
            int lq = 1;
            while ((1 << lq) < my_max)
                ++lq;
            const int Q = 1 << lq, G = 64 >> lq;
            const double root2 = sqrt(2.0);
            for (int p0 = wave * G; p0 < n_small; p0 += VQD_WAVES * G) {
                const int g = lane >> lq, e = lane & (Q - 1);
                const bool valid = p0 + g < n_small;
                const int node = valid ? lst[p0 + g] : lst[p0];
                const int n = valid ? N.nn[node] : 0;
                double *o = buf0 + N.off[node];
                const bool split = N.kind[node] == 0;
                const int cut = n / 2;
                const double ct = split ? N.theta[node] : 0.0, st = split ? N.sn[node] : 0.0;
                double val = 0.0;
                if (e < n) {
                    if (split) {
                        const double *mid = buf0 + N.off[N.kid0[node]], *side = buf0 + N.off[N.kid1[node]];
                        const int i = e < cut ? e : e - cut;
                        const double mm = mid[i] * ct, ss = side[i] * st;
                        val = e < cut ? (mm + ss) / root2 : (mm - ss) / root2;
                    } else {
                        val = o[e];                         /* a root that is a leaf: normalised a second time */
                    }
                }
                vqd_fence();                                /* in place: everybody has read */
                double acc = fma(val, val, 0.0);
                for (int off = Q >> 1; off > 0; off >>= 1)
                    acc = acc + __shfl_xor(acc, off, 64);
                const double nrm = sqrt(acc);
                if (nrm != 0.0)
                    val = val / nrm;
                if (e < n)
                    o[e] = val;
                vqd_fence();
            }
        }
        __syncthreads();
    }
    VQD_T(3);
    /* ---- 4. gains and the lines */
    for (int vb = wave; vb < n_items; vb += VQD_WAVES) {
        const int ba = item_ba[vb];
        if (!ba)
            continue;
        const int s = vb / nb, b = vb - s * nb;
        const int n = item_n[vb];
        int at = lower[b];
        if (sbr && b >= T.first_omitted)
            at = lower[T.first_omitted] + (b - T.first_omitted);
        const int pos = item_end[vb];
        int bits_gain = item_bg[vb] + item_bs[vb] - (pos - item_pos[vb]);
        double deq = 0.0;
        if (bits_gain > 64) {
            undefined = true;
        } else if (bits_gain > 0) {
            const unsigned long long code = vqd_get(words, pos, bits_gain);
            const unsigned long long mag = (bits_gain == 64) ? (code & 0x7FFFFFFFFFFFFFFFull)
                                                              : (code & ((1ull << (bits_gain - 1)) - 1ull));
            const double den = (bits_gain <= 53) ? (double)((1ull << bits_gain) - 1ull) : ldexp(1.0, bits_gain);
            deq = (2.0 * (double)mag) / den;
            if (code >> (bits_gain - 1))
                deq = -deq;
        }
        const double sgn = (deq > 0.0) ? 1.0 : ((deq < 0.0) ? -1.0 : 0.0);
        const double gain = (sgn / 255.0 * (vqd_pow256(fabs(deq)) - 1.0)) * (double)n;
        const int root = item_root[vb];
        double *dst = out + (shrt ? s * PACX_M_SHORT : 0) + at;
        for (int i = lane; i < n; i += 64)
            dst[i] = gain * (root >= 0 ? buf0[N.off[root] + i] : 1.0);
    }
    if (__builtin_amdgcn_ballot_w64(undefined) && lane == 0)
        atomicOr(&A.status[cf], PACX_ST_VQ_UNDEFINED);
    VQD_T(4);
}

/* ------------------------------------------------------- SBR reconstruction */
#define SBR_THREADS 256

__global__ __launch_bounds__(SBR_THREADS) void k_sbr_recon(PacxTables T, VqDecView V, long long n_cf,
                                                          const uint8_t *__restrict__ sbr_flag,
                                                          double *__restrict__ lines,
                                                          uint32_t *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const long long cf = blockIdx.x;
    if (cf >= n_cf || !sbr_flag[cf])
        return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int M = PACX_M_LONG;
    const int cut = T.band_lower_long[T.first_omitted];
    const int n_omit = M - cut;
    const int r = V.gauss_r;
    if (M / (M / n_omit) > M - 1) {
        /* the interpolation needs line M/up, which does not exist when the cut lies in
           the lower half (band tables that stop at 24 kHz: sample rates above 48 kHz):
           Decode_SBR raises IndexError there (coder/codec.py:173-176) */
        if (tid == 0)
            atomicOr(&status[cf], PACX_ST_VQ_UNDEFINED);
        return;
    }
    double *ln = (double *)smem;                 /* [M] lines                          */
    double *smooth = ln + M;                     /* [n_omit]                           */
    double *old = smooth + n_omit;               /* [M/up + 2] interpolation ordinates, later a band's magnitudes */
    double *g = lines + cf * M;
    for (int i = tid; i < M; i += SBR_THREADS)
        ln[i] = g[i];
    __syncthreads();
    /* envelope = lines[cut:] ++ zeros, extended by reflection (d c b a | a b c d | d c b a): position p of
       the extended array, p = 0 .. n_env + 2 r - 1.  It is not materialised (21 KB of mostly zeros kept this
       kernel at four workgroups per CU): the few reads of the smoothing loop below compute their sample */
    const int n_env = 2 * n_omit, period = 2 * n_env;
    auto ext_at = [&](int p) {
        int m = (p - r) % period;
        if (m < 0)
            m += period;
        if (m >= n_env)
            m = period - 1 - m;
        return (m < n_omit) ? ln[cut + m] : 0.0;
    };
    /* correlate1d, symmetric kernel: centre first, then the pairs from the far end
       inwards.  The envelope holds one value per omitted band at its first nob
       positions and zeros elsewhere (its mirror image puts them at -nob..-1), so
       for output i only the distances d = i-nob+1 .. i+nob meet a non-zero sample;
       every other pair adds (0+0)*w = +0.0 and is skipped -- same sum, same order */
    const double *__restrict__ w = V.gauss;
    const int nob = T.nb_long - T.first_omitted;
    for (int i = tid; i < n_omit; i += SBR_THREADS) {
        const int c = i + r;
        double acc = ext_at(c) * w[r];
        int d_hi = i + nob, d_lo = i - nob + 1;
        if (d_hi > r)
            d_hi = r;
        if (d_lo < 1)
            d_lo = 1;
        for (int d = d_hi; d >= d_lo; --d)
            acc += (ext_at(c - d) + ext_at(c + d)) * w[r - d];
        smooth[i] = acc;
    }
    /* transposition: lines[cut + i] = spline1(freq[cut + i] / up) through the points
       (freq[k], lines[k]), k = cut/up - 1 .. M/up */
    const int up = M / n_omit;                   /* floor(len / num_omitted) */
    const int k_lo = cut / up - 1, k_hi = M / up;
    for (int k = k_lo + tid; k <= k_hi; k += SBR_THREADS)
        old[k - k_lo] = ln[k];
    __syncthreads();
    const double *__restrict__ fq = V.line_freq;
    for (int i = tid; i < n_omit; i += SBR_THREADS) {
        const double x = fq[cut + i] / (double)up;
        /* interval with fq[j] <= x < fq[j+1] (np.searchsorted side='right' - 1), clamped */
        int j = (2 * (cut + i) + 1 - up) / (2 * up);          /* floor(((cut+i)+1/2)/up - 1/2) */
        if (j < k_lo) j = k_lo;
        if (j > k_hi - 1) j = k_hi - 1;
        while (j > k_lo && x < fq[j])
            --j;
        while (j < k_hi - 1 && x >= fq[j + 1])
            ++j;
        const double xb = fq[j + 1] - x, xa = x - fq[j];
        const double ww = 1.0 / (xb + xa);
        ln[cut + i] = 0.0 + old[j - k_lo] * (ww * xb) + old[j + 1 - k_lo] * (ww * xa);
    }
    __syncthreads();
    /* per omitted band: scale to the smoothed envelope over the mean magnitude */
    if (tid < 64) {
        double *mag = old;                       /* reuse: |lines| of the band for np.mean */
        for (int b = T.first_omitted; b < T.nb_long; ++b) {
            const int lo = T.band_lower_long[b], cnt = T.band_lines_long[b];
            double mx = 0.0;
            for (int i = lane; i < cnt; i += 64) {
                const double a = fabs(ln[lo + i]);
                mag[i] = a;
                mx = fmax(mx, a);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                mx = fmax(mx, __shfl_xor(mx, off, 64));
            vqd_fence();
            if (mx > 0.0) {
                const double mean = wave_np_sum(mag, cnt, lane) / (double)cnt;
                for (int i = lane; i < cnt; i += 64)
                    ln[lo + i] = ln[lo + i] * (smooth[lo - cut + i] / mean);
            }
            vqd_fence();
        }
    }
    __syncthreads();
    for (int i = cut + tid; i < M; i += SBR_THREADS)
        g[i] = ln[i];
}

/* ---------------------------------------------------------------- launchers */
size_t pacx_vqdec_view_size(void) { return sizeof(VqDecView); }

static size_t sbr_recon_lds(void)
{
    /* LDS sized for the worst case (every line above the cut) */
    return (size_t)(PACX_M_LONG + PACX_M_LONG + PACX_M_LONG + 2) * 8;      /* lines, smoothed envelope, ordinates */
}

/* Decode_SBR's reconstruction alone (scalar-mantissa SBR streams: the view carries the Gaussian weights and the
   line frequencies only) */
void pacx_launch_sbr_recon(const PacxTables &T, const void *view, long long n_cf, const uint8_t *sbr_flag,
                           double *lines, uint32_t *status, hipStream_t st)
{
    if (n_cf > 0)
        hipLaunchKernelGGL(k_sbr_recon, dim3((unsigned)n_cf), dim3(SBR_THREADS), sbr_recon_lds(), st, T,
                           *(const VqDecView *)view, n_cf, sbr_flag, lines, status);
}

void pacx_vqdec_view_fill(void *dst, const uint64_t *n_tab, const uint64_t *p_tab, const int32_t *row_off,
                          const int32_t *k_of, const uint8_t *w_of, const double *half_log2, int l_max,
                          const double *log2_tan, const double *gauss, int gauss_r, const double *line_freq,
                          const int32_t *sizes_long, int nb_long, const int32_t *sizes_short, int nb_short)
{
    VqDecView *v = (VqDecView *)dst;
    /* ticket t codes band nb - 1 - t (mod nb; short frames: of sub-block t / nb).  Wave w owns ticket w; later
       tickets go to whichever wave is free */
    v->scr_off[0] = 0;
    for (int w = 0; w < VQD_WAVES; ++w) {
        int n = 1;
        if (w < nb_long)
            n = sizes_long[nb_long - 1 - w];
        if (w < nb_short && sizes_short[nb_short - 1 - w] > n)
            n = sizes_short[nb_short - 1 - w];
        for (int b = 0; b < nb_long - VQD_WAVES; ++b)
            n = sizes_long[b] > n ? sizes_long[b] : n;
        for (int b = 0; b < nb_short; ++b)                 /* a short frame's later tickets start over at the top band */
            n = sizes_short[b] > n ? sizes_short[b] : n;
        v->scr_off[w + 1] = v->scr_off[w] + ((3 * n + 4 * VQD_DEPTH + 1) & ~1);
    }
    v->log2_tan = log2_tan;
    v->n_tab = n_tab;
    v->p_tab = p_tab;
    v->row_off = row_off;
    v->k_of = k_of;
    v->w_of = w_of;
    v->half_log2 = half_log2;
    v->l_max = l_max;
    v->gauss = gauss;
    v->gauss_r = gauss_r;
    v->line_freq = line_freq;
}

void pacx_launch_vq_dec(const PacxTables &T, const void *view, long long n_cf, const uint8_t *payload,
                        int payload_stride, const long long *offsets, const int32_t *n_bytes,
                        uint8_t *cf_flags, int32_t *overall, int32_t *bit_alloc, double *lines,
                        uint8_t *sbr_flag, uint32_t *status, hipStream_t st)
{
    if (n_cf <= 0)
        return;
    const VqDecView &V = *(const VqDecView *)view;
    VqDecArgs A;
    A.n_cf = n_cf;
    A.payload = payload;
    A.payload_stride = payload_stride;
    A.offsets = offsets;
    A.n_bytes = n_bytes;
    A.cf_flags = cf_flags;
    A.overall = overall;
    A.bit_alloc = bit_alloc;
    A.lines = lines;
    A.sbr_flag = sbr_flag;
    A.status = status;
    A.scr_len = 0;
    const size_t fixed = VQD_WORDS * 4 + PACX_M_LONG * 8 + (2 * PACX_SUB * PACX_MAX_BANDS + 4) * 4 +
                         VQD_WAVES * VQD_DEPTH * sizeof(VqdFrame);
    const size_t smem = fixed + (size_t)V.scr_off[VQD_WAVES] * 8;
    /* the frame-level decoder first; k_vq_dec then takes the blocks it left (PACX_VQ_DEC_FRAME=0: k_vq_dec alone) */
    const char *fe = getenv("PACX_VQ_DEC_FRAME");
    A.redo = (fe && atoi(fe) == 0) ? 0 : 1;
    if (PACX_SUB * T.nb_short > VQDF_VB || T.nb_long > VQDF_VB)
        A.redo = 0;
    if (A.redo)
        hipLaunchKernelGGL(k_vq_dec_frame, dim3((unsigned)n_cf), dim3(64 * VQD_WAVES), (size_t)VQDF_SMEM, st, T, V, A);
    hipLaunchKernelGGL(k_vq_dec, dim3((unsigned)n_cf), dim3(64 * VQD_WAVES), smem, st, T, V, A);
    if (T.use_sbr)
        hipLaunchKernelGGL(k_sbr_recon, dim3((unsigned)n_cf), dim3(SBR_THREADS), sbr_recon_lds(), st, T, V, n_cf,
                           sbr_flag, lines, status);
}
