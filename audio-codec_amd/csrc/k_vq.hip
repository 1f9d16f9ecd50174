/*
 * k_vq.hip -- gain-shape (pyramid VQ) coding of the bands of a (sub-)block and
 * the VQ flavour of the .pac payload (BASELINE config 4).
 *
 *   k_vq       one workgroup per (sub-)block; its waves take bands from a
 *              shared ticket.  Per band: quantize_gain_shape
 *              (coder/gain_shape_quantize.py:476-512) = mu-law gain + the
 *              recursive mid/side split (split_band_encode :315-408) whose
 *              leaves are pyramid-VQ searches (pvq_search :30-54) turned into
 *              enumeration indices (encode_pvq_vector :105-124).  The band's
 *              bit string is written straight into the block's MSB-first bit
 *              buffer in the layout of WriiteEncodedBitsVQ
 *              (coder/pacfile.py:363-402): overall scale, all allocations,
 *              then the index lists.  A long block leaves the kernel as the
 *              finished channel-block payload.
 *   k_vq_join  short frames: concatenates the 8 sub-block strings behind the
 *              3 flag bits (coder/pacfile.py:575-592) and applies the size
 *              rule (:552-565, which still counts a scale factor per band).
 *
 * Why the band strings have static positions: a band that is coded at all
 * uses exactly bitAlloc*nLines bits -- the gain index takes whatever the shape
 * left over (:493) -- so the only data-dependent event is a band of zero gain,
 * whose allocation drops to 0 (coder/codec.py:352-353).  Gains are therefore
 * computed first (phase A), positions follow from a prefix sum, and the bands
 * are then coded independently (phase B).  The same argument shows that the
 * SBR spill rule (coder/codec.py:522-524: "sum(bits) < bitAlloc[iBand]", a
 * band total against a per-line count) cannot fire; the oracle implements it
 * literally and the tests assert it never triggers.
 *
 * Control flow inside a band is wave-uniform (one node of the split tree at a
 * time, vectors in LDS, lanes across the vector's components).
 */
#include "../../include/pacx.h"
#include "pacx_dev.h"
#include "wave_np_sum.h"

#ifndef VQ_WAVES
#define VQ_WAVES 4
#endif
#ifndef VQ_OCC
#define VQ_OCC 3                       /* waves per SIMD the register budget is set for */
#endif
#define VQ_DEPTH 16
#define VQ_WORDS 548                   /* 2192-byte payload slot, as k_pack */

struct VqView {
    const uint64_t *n_tab, *p_tab;
    const int32_t *row_off;
    const int32_t *k_of;
    const uint8_t *w_of;
    const double *half_log2;
    int l_max;
    double log_mu1;                    /* np.log(1 + 255) */
    const double *log2_tan;            /* [2^12 - 1] log2(tan(theta_q) + eps), see pacx_config */
    /* work order: bands by decreasing size (SBR-omitted long bands count as 1);
       the first VQ_WAVES entries go to waves 0.. statically, the rest by
       ticket, so wave w only needs scratch for the (w+1)-th largest band */
    uint8_t order_long[PACX_MAX_BANDS], order_short[PACX_MAX_BANDS];
    int scr_off[VQ_WAVES + 1];         /* doubles: scratch of wave w = [scr_off[w], scr_off[w+1]) */
    /* k_vq_frame2: band b's vector lives in a region of pow2ceil(size) doubles at reg_*[b] (short: of sub-block 0;
       sub-block j adds j * reg_short_total); rlog_* = log2 of the region */
    unsigned short reg_long[PACX_MAX_BANDS], reg_short[PACX_MAX_BANDS];
    unsigned char rlog_long[PACX_MAX_BANDS], rlog_short[PACX_MAX_BANDS];
    int reg_long_total, reg_short_total, max_band;
};

/* read-only table entry at a wave-uniform address: through the constant address space the
   load becomes a scalar (SMEM) one -- the compiler otherwise issues a vector load with a
   full vmcnt wait for every table read that follows one of the kernel's own stores */
template <typename T> __device__ __forceinline__ T ldc(const T *p)
{
    return *(const __attribute__((address_space(4))) T *)(unsigned long long)p;
}

/* The device library's atan / log / log2(tan) are polynomial evaluations with one or two dozen
   64-bit coefficients.  Inlined into the band loop the compiler hoists every coefficient out
   of the loops into a VGPR pair of its own -- some sixty registers held for constants, which
   is what had this kernel at 168 VGPRs with 20 spilled (the spill reloads sat inside the
   Horner chains).  As real calls (once or twice per tree node, wave-uniform arguments) the
   coefficients live only inside the callee. */
__device__ __attribute__((noinline)) double vq_atan(double x) { return atan(x); }
__device__ __attribute__((noinline)) double vq_log(double x) { return log(x); }
__device__ __attribute__((noinline)) double vq_log2_tan(double theta_q) { return log2(tan(fabs(theta_q)) + PACX_EPS); }

/* ---- table access --------------------------------------------------------- */
__device__ __forceinline__ uint64_t vq_N(const VqView &V, int l, long long k)
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

/* sum_{j=0..k} N(l,j); 0 for k < 0 */
__device__ __forceinline__ uint64_t vq_P(const VqView &V, int l, long long k)
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

/* start of row l of the two tables, copied to LDS by every workgroup: the lookups of one
   index term then cost one LDS read and ONE round trip to L2 (four independent loads)
   instead of four dependent pairs (row offset, then entry) */
#define VQ_LMAX 1024
__shared__ int vq_row_off_s[VQ_LMAX + 1];

/* index term of a component of magnitude a >= 1 with k >= a pulses left and l1 dimensions
   after it: N(l1,k) + 2 (P(l1,k-1) - P(l1,k-a)) (+ N(l1,k-a) if negative) */
/* ROWS64: only the first 64 row offsets sit in LDS (k_vq_frame: a quarter of a kilobyte instead of four --
   the workgroups per CU of that kernel are bound by LDS); longer vectors read theirs from the table itself */
#define VQ_ROWS_SMALL 64
__shared__ int vq_row_off_small[VQ_ROWS_SMALL];
template <bool ROWS64 = false>
__device__ __forceinline__ unsigned long long vq_term(const VqView &V, int l1, long long k, long long a, bool neg)
{
#ifdef VQ_STUB_TERM        /* timing experiment only (wrong indices): what the table lookups cost */
    return (unsigned long long)(k + a + (neg ? 1 : 0) + l1);
#endif
    if (l1 >= 3) {
        const long long base = ROWS64 ? (l1 < VQ_ROWS_SMALL ? vq_row_off_small[l1] : V.row_off[l1]) : vq_row_off_s[l1];
        const long long ka = k - a;
        const uint64_t nk = V.n_tab[base + k], pk1 = V.p_tab[base + k - 1];
        const uint64_t pka = V.p_tab[base + ka], nka = V.n_tab[base + ka];
        return nk + 2ull * (pk1 - pka) + (neg ? nka : 0ull);
    }
    unsigned long long term = vq_N(V, l1, k);
    term += 2ull * (vq_P(V, l1, k - 1) - vq_P(V, l1, k - a));
    if (neg)
        term += vq_N(V, l1, k - a);
    return term;
}

/* ---- wave helpers --------------------------------------------------------- */
__device__ __forceinline__ void vq_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v = v + __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v = v + (unsigned long long)__shfl_xor((long long)v, off, 64);
    return v;
}

/* MSB-first bit writer into LDS words shared by the block's waves */
__device__ __forceinline__ void vq_put32(unsigned *words, int pos, unsigned val, int width)
{
    if (width <= 0)
        return;
    val &= (width >= 32) ? 0xFFFFFFFFu : ((1u << width) - 1u);
    const int w = pos >> 5, o = pos & 31;
    const int room = 32 - o;
    if (width <= room) {
        atomicOr(&words[w], val << (room - width));
    } else {
        atomicOr(&words[w], val >> (width - room));
        atomicOr(&words[w + 1], val << (32 - (width - room)));
    }
}

__device__ __forceinline__ void vq_put64(unsigned *words, int pos, unsigned long long val, int width)
{
    if (width > 32) {
        vq_put32(words, pos, (unsigned)(val >> 32), width - 32);
        vq_put32(words, pos + width - 32, (unsigned)val, 32);
    } else {
        vq_put32(words, pos, (unsigned)val, width);
    }
}

/* Per-band emission state (uniform across the wave). */
struct VqOut {
    unsigned *words;
    int pos;                   /* next stream bit                          */
    pacx_vq_entry *log;        /* optional entry log of this band          */
    int log_cap, log_n;
    int band;
    unsigned flags;            /* PACX_ST_VQ_* raised while coding          */
#ifdef PACX_VQ_DEBUG
    long long t_last;          /* phase stamps inside the band walk (measuring aid) */
#endif
};

#ifdef PACX_VQ_DEBUG
__device__ long long g_vq_dbg[32];
/* inside the band walk: slots 8.. = split arithmetic, quad attempt, leaf pair, single leaf, climb, band set-up, gain */
#define VQ_S(o, k) do { long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                        if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&g_vq_dbg[k], (unsigned long long)(t_ - (o).t_last)); \
                        (o).t_last = t_; } while (0)
#else
#define VQ_S(o, k) do { } while (0)
#endif

__device__ __forceinline__ void vq_emit(VqOut &o, unsigned long long hi, unsigned long long lo, int width,
                                        int lane)
{
    /* The field (hi:lo < 2^width, wave-uniform) goes MSB-first to stream bit o.pos and touches at most
       five words.  Lane k < 5 cuts word w + k out of the field itself and ORs it in: one LDS atomic for
       the whole field and no branches, where a lane-0 writer went through up to eight atomics behind a
       ladder of width tests (a seventh of this kernel's instructions). */
    if (width > 0) {
        if (width < 64)
            lo &= (1ull << width) - 1ull;
        const int w = o.pos >> 5;
        const int S = 160 - (o.pos & 31) - width;         /* left shift of the field in the window of words w..w+4 */
        const int d = S - (128 - 32 * lane);              /* word w + lane = window bits 159-32 lane .. 128-32 lane */
        const int r = -d;
        const unsigned up = (unsigned)lo << (d & 31);                                         /* 0 <= d < 32 */
        const unsigned mid = (unsigned)((lo >> (r & 63)) | ((hi << ((64 - r) & 63))));          /* 1 <= r < 64 */
        const unsigned top = (unsigned)(hi >> ((r - 64) & 63));                                /* 64 <= r     */
        unsigned word = 0u;
        if (d >= 0)
            word = d < 32 ? up : 0u;
        else
            word = r < 64 ? mid : top;
        if (lane < 5 && word)
            atomicOr(&o.words[w + lane], word);
    }
    if (o.log && lane == 0 && o.log_n < o.log_cap) {
        pacx_vq_entry e;
        e.value = lo;
        e.width = width;
        e.band = o.band;
        o.log[o.log_n] = e;
    }
    o.log_n += 1;
    o.pos += width;
}

/* QuantizeUniform(x, n_bits) for x >= 0 (coder/quantize.py:14-36) evaluated the
 * way Python does: the factor 2^n - 1 becomes a float64 (exact up to 53 bits,
 * 2^n beyond), one multiply, +1, floor-divide by 2. */
__device__ __forceinline__ void vq_quantize_emit(VqOut &o, double x, int n_bits, int lane)
{
    if (n_bits <= 0) {
        vq_emit(o, 0, 0, 0, lane);
        return;
    }
    if (n_bits > 128) {                       /* beyond what the entry format carries */
        o.flags |= PACX_ST_VQ_UNDEFINED;
        vq_emit(o, 0, 0, 0, lane);
        return;
    }
    unsigned long long hi = 0, lo = 0;
    if (x >= 1.0) {                           /* code = 2^(n-1) - 1 */
        const int ones = n_bits - 1;
        if (ones >= 64) {
            lo = ~0ull;
            hi = (ones - 64 >= 64) ? ~0ull : ((1ull << (ones - 64)) - 1ull);
        } else {
            lo = (ones == 0) ? 0ull : ((~0ull) >> (64 - ones));
        }
    } else {
        const double factor = (n_bits <= 53) ? (double)((1ull << n_bits) - 1ull) : ldexp(1.0, n_bits);
        const double code = floor((factor * x + 1.0) * 0.5);
        if (n_bits <= 64) {
            lo = (unsigned long long)code;
        } else {
            const double top = floor(ldexp(code, -64));
            hi = (unsigned long long)top;
            lo = (unsigned long long)(code - ldexp(top, 64));
        }
    }
    vq_emit(o, hi, lo, n_bits, lane);
}

/* ---- one PVQ leaf --------------------------------------------------------- */
/* xs[0..n): unit vector in LDS; t1, t2: n doubles of LDS scratch each.  Returns the enumeration
   index of the K-pulse vector (ok = false for an all-zero vector: NaN pulses in the reference). */
template <bool ROWS64 = false>
__device__ __forceinline__ unsigned long long vq_leaf_idx(const VqView &V, const double *xs, int n, int K,
                                                          double *t1, double *t2, int lane, bool &ok)
{
    /* L1 norm in np.sum order */
    for (int i = lane; i < n; i += 64)
        t1[i] = fabs(xs[i]);
    vq_fence();
    const double l1 = wave_np_sum(t1, n, lane);
    vq_fence();
    ok = l1 > 0.0;
    if (!ok)
        return 0ull;
    /* target = |K x / l1|, y = floor(target) */
    const double kd = (double)K;
    double part = 0.0;
    for (int i = lane; i < n; i += 64) {
        const double t = fabs(kd * xs[i] / l1);
        const double y = floor(t);
        t1[i] = t;
        t2[i] = y;
        part += y;                            /* integers: exact in any order */
    }
    const int missing = K - (int)wave_sum_f64(part);
    vq_fence();
    if (missing > 0) {
        if (n <= 64) {
            /* the greedy loop hands one pulse each to the `missing` largest
               remainders, earliest index first among equals */
            const double r = (lane < n) ? (t1[lane] - t2[lane]) : -1.0;
            int rank = 0;
            for (int j = 0; j < n; ++j) {
                const double rj = readlane_f64(r, j);
                rank += (rj > r) || (rj == r && j < lane);
            }
            if (lane < n && rank < missing)
                t2[lane] += 1.0;
        } else {
            for (int it = 0; it < missing; ++it) {
                double best = -INFINITY;
                int best_i = 0x7fffffff;
                for (int i = lane; i < n; i += 64) {
                    const double r = t1[i] - t2[i];
                    if (r > best) {
                        best = r;
                        best_i = i;
                    }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const double ob = __shfl_xor(best, off, 64);
                    const int oi = __shfl_xor(best_i, off, 64);
                    if (ob > best || (ob == best && oi < best_i)) {
                        best = ob;
                        best_i = oi;
                    }
                }
                if (lane == 0)
                    t2[best_i] += 1.0;
                vq_fence();
            }
        }
        vq_fence();
    }
    /* enumeration index: component i (l = n-i dimensions left, k pulses left)
       of magnitude a >= 1 adds N(l-1,k) + 2 sum_{j=1}^{a-1} N(l-1,k-j)
       (+ N(l-1,k-a) if negative); np.sign(x) = 0 erases a pulse */
    unsigned long long acc = 0;
    long long k_left = K;
    for (int base = 0; base < n && k_left > 0; base += 64) {
        const int i = base + lane;
        long long a = 0;
        bool neg = false;
        if (i < n) {
            const double x = xs[i];
            if (x != 0.0)
                a = (long long)t2[i];
            neg = x < 0.0;
        }
        long long incl = a;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const long long t = __shfl_up(incl, off, 64);
            if (lane >= off)
                incl += t;
        }
        const long long k = k_left - (incl - a);
        if (a >= 1)
            acc += vq_term<ROWS64>(V, n - i - 1, k, a, neg);
        k_left -= __shfl(incl, 63, 64);
    }
    return wave_sum_u64(acc);
}

__device__ __forceinline__ void vq_leaf(const VqView &V, VqOut &o, const double *xs, int n, int bits,
                                        double *t1, double *t2, int lane)
{
    const int K = ldc(&V.k_of[n * 33 + bits]);
    const int width = ldc(&V.w_of[n * 33 + bits]);
    if (K < 0) {                              /* a 1-dimensional leaf: the reference never returns */
        o.flags |= PACX_ST_VQ_UNDEFINED;
        return;
    }
    bool ok;
    const unsigned long long idx = vq_leaf_idx(V, xs, n, K, t1, t2, lane, ok);
    if (!ok)                                  /* all-zero half: NaN pulses in the reference */
        o.flags |= PACX_ST_VQ_UNDEFINED;
    vq_emit(o, 0, ok ? idx : 0ull, width, lane);
}

/* ---- several small leaves at once ------------------------------------------ */
/* 93 % of the leaves are the two children of a bottom split (5..20 components
 * each).  A group of W lanes (W = 32: two leaves per wave, W = 16: four) codes
 * one leaf: component l of the vector in lane l of its group, the group's own
 * pulse count K.  Same arithmetic and the same summation orders as vq_leaf (an
 * xor butterfly over W lanes with zeros in the unused ones adds in the order of
 * the 64-lane one).  Returns the enumeration index in every lane of the group;
 * ok = false for an all-zero vector (NaN pulses in the reference). */
/* PACX_ST_GUARD for a leaf (handles with pacx_config.guard): the pulse search has two kinds of decisions that the last
   bits of the unit vector can turn -- floor(K |x_i| / l1) where the quotient sits at an integer, and which components get
   the `missing` pulses where the last remainder to get one and the first to go without are (nearly) equal.  The unit
   vectors agree with the reference's to a few ulps (another summation order in the norms), so the quotients to
   ~K * 4e-15 absolute: within that margin the leaf is flagged. */
#define VQ_GUARD_REL 4e-15
template <int W, bool ROWS64 = false>
__device__ __forceinline__ unsigned long long vq_leaf_group(const VqView &V, double x, int n, int K, int l,
                                                            bool &ok, bool want_guard = false, bool *near = nullptr)
{
    /* L1 norm in np.sum order (n <= 32: sequential below 8, else eight interleaved
       accumulators, fixed tree, scalar tail) */
    const double ax = fabs(x);
    double l1;
    if (n < 8) {
        l1 = -0.0;
        for (int i = 0; i < n; ++i)
            l1 = l1 + __shfl(ax, i, W);
    } else {
        const int n8 = n - (n & 7);
        double r = ax;                                   /* lanes 0..7: a[j] + a[8+j] + ... */
        for (int i = 8; i < n8; i += 8) {
            const double t = __shfl_down(ax, i, W);
            r = r + t;
        }
        const double t = r + __shfl_down(r, 1, W);
        const double u = t + __shfl_down(t, 2, W);
        l1 = u + __shfl_down(u, 4, W);
        l1 = __shfl(l1, 0, W);
        for (int i = n8; i < n; ++i)
            l1 = l1 + __shfl(ax, i, W);
    }
    ok = l1 > 0.0;
    const double kd = (double)K;
    const double tgt = ok ? fabs(kd * x / l1) : 0.0;
    double y = floor(tgt);
    double ysum = (l < n) ? y : 0.0;
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1)
        ysum = ysum + __shfl_xor(ysum, off, W);
    const int missing = K - (int)ysum;
#ifndef VQ_STUB_RANK       /* timing experiment only: what the pulse ranking costs */
    {
        const double r = (l < n) ? (tgt - y) : -1.0;
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double rj = __shfl(r, j, W);
            rank += (rj > r) || (rj == r && j < l);
        }
        if (l < n && rank < missing)
            y += 1.0;
        if (want_guard) {
            const double thr = (kd < 1.0 ? 1.0 : kd) * VQ_GUARD_REL;
            const double fr = tgt - floor(tgt);
            bool nr = l < n && ok && (fr < thr || fr > 1.0 - thr);
            double ra = (l < n && rank == missing - 1) ? r : -1.0;     /* the last remainder to get a pulse ... */
            double rb = (l < n && rank == missing) ? r : -1.0;         /* ... and the first to go without */
#pragma unroll
            for (int off = W / 2; off > 0; off >>= 1) {
                ra = fmax(ra, __shfl_xor(ra, off, W));
                rb = fmax(rb, __shfl_xor(rb, off, W));
            }
            nr = nr || (missing > 0 && ra >= 0.0 && rb >= 0.0 && ra - rb < 2.0 * thr);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(nr);
            const int g0 = (threadIdx.x & 63) & ~(W - 1);
            *near = ((m >> g0) & ((W == 64) ? ~0ull : ((1ull << W) - 1ull))) != 0ull;
        }
    }
#else
    if (l < missing)
        y += 1.0;
#endif
    const int a = (l < n && x != 0.0 && ok) ? (int)y : 0;
    int incl = a;
#pragma unroll
    for (int off = 1; off < W; off <<= 1) {
        const int t = __shfl_up(incl, off, W);
        if (l >= off)
            incl += t;
    }
    const long long k = (long long)K - (incl - a);
    unsigned long long term = 0;
    if (a >= 1)
        term = vq_term<ROWS64>(V, n - l - 1, k, a, x < 0.0);
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1)
        term = term + (unsigned long long)__shfl_xor((long long)term, off, W);
    return term;
}

/* the two children of a bottom split: mid on lanes 0-31, side on lanes 32-63 */
__device__ __forceinline__ void vq_leaf_pair(const VqView &V, VqOut &o, const double *mid, const double *side,
                                             int n, int bits_mid, int bits_side, int lane)
{
    const int h = lane >> 5, l = lane & 31;
    const int bits = h ? bits_side : bits_mid;
    const int K = ldc(&V.k_of[n * 33 + bits]);
    const int width = ldc(&V.w_of[n * 33 + bits]);
    const double x = (l < n) ? (h ? side[l] : mid[l]) : 0.0;
    bool ok;
    const unsigned long long term = vq_leaf_group<32>(V, x, n, K, l, ok);
    /* mid first, then side (coder/gain_shape_quantize.py:372-393) */
    const unsigned long long idx_mid = (unsigned long long)__shfl((long long)term, 0, 64);
    const unsigned long long idx_side = (unsigned long long)__shfl((long long)term, 32, 64);
    const int w_mid = __shfl(width, 0, 64), w_side = __shfl(width, 32, 64);
    const unsigned long long okm = __builtin_amdgcn_ballot_w64(ok);
    if (!(okm & 1ull) || !((okm >> 32) & 1ull))
        o.flags |= PACX_ST_VQ_UNDEFINED;                  /* an all-zero half: NaN pulses in the reference */
    vq_emit(o, 0, (okm & 1ull) ? idx_mid : 0ull, w_mid, lane);
    vq_emit(o, 0, ((okm >> 32) & 1ull) ? idx_side : 0ull, w_side, lane);
}

/* Two sibling bottom splits at once.  xa / xb: the two unit vectors (n <= 32
 * components each, bits_a / bits_b > 32 bits).  Split a runs on lanes 0-31, split
 * b on lanes 32-63 (fold, norms, angle, bit split: the arithmetic of vq_shape's
 * split step); if all four grandchildren turn out to be leaves, they are coded
 * on the four 16-lane quarters and the six fields go out in the reference's
 * order (theta a, a.mid, a.side, theta b, b.mid, b.side).  Otherwise nothing is
 * emitted and the caller walks the two subtrees the ordinary way. */
__device__ __forceinline__ bool vq_quad_try(const VqView &V, VqOut &o, const double *xa, const double *xb,
                                            int n, int bits_a, int bits_b, int lane)
{
    const double half_pi = 1.5707963267948966;
    const int h = lane >> 5, l = lane & 31;
    const double *xs = h ? xb : xa;
    const int bits = h ? bits_b : bits_a;
    const int cut = n / 2, hh = n - cut;                    /* hh <= 16 */
    const double left = (l < cut) ? xs[l] : 0.0;
    const double right = (l < hh) ? xs[cut + l] : 0.0;
    double m = (l < hh) ? (left + right) / 2.0 : 0.0;
    double sd = (l < hh) ? (left - right) / 2.0 : 0.0;
    double mm = fma(m, m, 0.0), ss = fma(sd, sd, 0.0);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
        mm = mm + __shfl_xor(mm, off, 32);
        ss = ss + __shfl_xor(ss, off, 32);
    }
    const double m_l2 = sqrt(mm), s_l2 = sqrt(ss);
    if (m_l2 != 0.0)
        m = m / m_l2;
    if (s_l2 != 0.0)
        sd = sd / s_l2;
    const double theta = (m_l2 == 0.0) ? 0.0 : vq_atan(s_l2 / m_l2);
    const int a_theta = (int)floor((double)bits / (double)hh + ldc(&V.half_log2[hh]));
    int a_rest = bits - a_theta;
    if (a_rest < 0)
        a_rest = 0;
    bool fine = a_theta > 0 && a_theta <= 62 && hh >= 2;
    unsigned long long code = 0;
    double theta_q = 0.0;
    if (fine) {
        const double tn = theta / half_pi;
        if (tn >= 1.0) {
            code = (1ull << (a_theta - 1)) - 1ull;
        } else {
            const double factor = (a_theta <= 53) ? (double)((1ull << a_theta) - 1ull) : ldexp(1.0, a_theta);
            code = (unsigned long long)floor((factor * tn + 1.0) * 0.5);
        }
        const unsigned long long mag = code & ((1ull << (a_theta - 1)) - 1ull);
        const double den = (a_theta <= 53) ? (double)((1ull << a_theta) - 1ull) : ldexp(1.0, a_theta);
        double dq = (double)(2ull * mag) / den;
        if (code >> (a_theta - 1))
            dq = -dq;
        theta_q = dq * half_pi;
    }
    int a_mid = 0;
    if (fine && theta_q != 0.0) {
        double lt;
        if (a_theta <= PACX_VQ_THETA_TABLE_BITS && theta_q > 0.0)
            lt = V.log2_tan[((1 << (a_theta - 1)) - 1) + (int)code];
        else
            lt = vq_log2_tan(theta_q);
        const double v = ((double)a_rest - (double)(hh - 1) * lt) / 2.0;
        const double f = floor(v);
        a_mid = (f < 0.0) ? 0 : ((f > (double)a_rest) ? a_rest : (int)f);
    }
    const int a_side = a_rest - a_mid;
    fine = fine && a_mid > 0 && a_mid <= PACX_VQ_SPLIT_BITS && a_side > 0 && a_side <= PACX_VQ_SPLIT_BITS;
    if (__builtin_amdgcn_ballot_w64(!fine))
        return false;                                       /* not two bottom splits: ordinary walk */
    /* the four leaves, one per 16-lane quarter: (a.mid, a.side, b.mid, b.side) */
    const int q = lane >> 4, ll = lane & 15;
    const double from_side = __shfl(sd, 32 * h + ll, 64);
    const double from_mid = __shfl(m, 32 * h + ll, 64);
    const double x = (ll < hh) ? ((q & 1) ? from_side : from_mid) : 0.0;
    const int lbits = (q & 1) ? a_side : a_mid;
    const int K = ldc(&V.k_of[hh * 33 + lbits]);
    const int width = ldc(&V.w_of[hh * 33 + lbits]);
    bool ok;
    const unsigned long long term = vq_leaf_group<16>(V, x, hh, K, ll, ok);
    const unsigned long long okm = __builtin_amdgcn_ballot_w64(ok);
    if ((okm & 0x0001000100010001ull) != 0x0001000100010001ull)
        o.flags |= PACX_ST_VQ_UNDEFINED;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const unsigned long long th = (unsigned long long)__shfl((long long)code, 32 * g, 64);
        vq_emit(o, 0, th, __shfl(a_theta, 32 * g, 64), lane);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int src = 32 * g + 16 * c;
            const unsigned long long idx = (unsigned long long)__shfl((long long)term, src, 64);
            vq_emit(o, 0, ((okm >> src) & 1ull) ? idx : 0ull, __shfl(width, src, 64), lane);
        }
    }
    return true;
}

/* ---- the split tree of one band ------------------------------------------- */
/* region: LDS doubles for the mid/side vectors of every depth (2 n0 + 4 VQ_DEPTH:
 * depth d takes 2*ceil(n_d/2)); a leaf borrows the still unused tail of it for
 * its two scratch vectors (what is left at depth d is at least 2 n_d).
 * stack: 2*VQ_DEPTH ints */
__device__ __forceinline__ void vq_shape(const VqView &V, VqOut &o, const double *x0, int n0, int bits0,
                                         double *region, int *stack, int lane)
{
    const double half_pi = 1.5707963267948966;           /* np.pi / 2 */
    const double *cur = x0;
    int n = n0, bits = bits0, depth = 0;
    double *reg = region;                                  /* region of the current depth */
    for (;;) {
        bool descend = false;
        if (bits > PACX_VQ_SPLIT_BITS && depth < VQ_DEPTH) {
            const int cut = n / 2, half = n - cut;
            double *mv = reg, *sv = reg + half;
            double mm = 0.0, ss = 0.0;
            for (int i = lane; i < half; i += 64) {
                const double left = (i < cut) ? cur[i] : 0.0;
                const double right = cur[cut + i];
                const double m = (left + right) / 2.0;
                const double s = (left - right) / 2.0;
                mv[i] = m;
                sv[i] = s;
                mm = fma(m, m, mm);
                ss = fma(s, s, ss);
            }
            const double m_l2 = sqrt(wave_sum_f64(mm));
            const double s_l2 = sqrt(wave_sum_f64(ss));
            for (int i = lane; i < half; i += 64) {
                if (m_l2 != 0.0)
                    mv[i] = mv[i] / m_l2;
                if (s_l2 != 0.0)
                    sv[i] = sv[i] / s_l2;
            }
            vq_fence();
            const double theta = (m_l2 == 0.0) ? 0.0 : vq_atan(s_l2 / m_l2);
            /* gain_shape_alloc(bits, half): floor(bits/half + 0.5 log2(half)) for the angle */
            int a_theta = (int)floor((double)bits / (double)half + ldc(&V.half_log2[half]));
            int a_rest = bits - a_theta;
            if (a_rest < 0)
                a_rest = 0;
            /* QuantizeUniform / DequantizeUniform of theta/(pi/2) */
            const double tn = theta / half_pi;
            double theta_q = 0.0;
            unsigned long long theta_code = 0;
            if (a_theta <= 0) {
                vq_emit(o, 0, 0, a_theta < 0 ? 0 : a_theta, lane);
            } else if (a_theta > 62) {
                o.flags |= PACX_ST_VQ_UNDEFINED;
                vq_emit(o, 0, 0, 0, lane);
            } else {
                unsigned long long code;
                if (tn >= 1.0) {
                    code = (1ull << (a_theta - 1)) - 1ull;
                } else {
                    const double factor = (a_theta <= 53) ? (double)((1ull << a_theta) - 1ull)
                                                          : ldexp(1.0, a_theta);
                    code = (unsigned long long)floor((factor * tn + 1.0) * 0.5);
                }
                vq_emit(o, 0, code, a_theta, lane);
                theta_code = code;
                const unsigned long long mag = code & ((1ull << (a_theta - 1)) - 1ull);
                const double den = (a_theta <= 53) ? (double)((1ull << a_theta) - 1ull) : ldexp(1.0, a_theta);
                double dq = (double)(2ull * mag) / den;
                if (code >> (a_theta - 1))
                    dq = -dq;
                theta_q = dq * half_pi;
            }
            /* bit_allocation_ms */
            int a_mid = 0;
            if (theta_q != 0.0) {
                /* log2(tan(theta_q) + eps): quantised angles of up to 12 bits come from the
                   table the caller evaluated (one scalar load instead of tan + log2) */
                double lt;
                if (a_theta <= PACX_VQ_THETA_TABLE_BITS && theta_q > 0.0)
                    lt = V.log2_tan[((1 << (a_theta - 1)) - 1) + (int)theta_code];
                else
                    lt = vq_log2_tan(theta_q);
                const double v = ((double)a_rest - (double)(half - 1) * lt) / 2.0;
                const double f = floor(v);
                a_mid = (f < 0.0) ? 0 : ((f > (double)a_rest) ? a_rest : (int)f);
            }
            const int a_side = a_rest - a_mid;
            stack[2 * depth] = half;
            stack[2 * depth + 1] = a_side;
            /* children live one level down */
            reg = reg + 2 * half;
            depth += 1;
            VQ_S(o, 8);
#ifdef VQ_WITH_QUAD      /* depth-first walk only: the level-by-level walk takes the trees with sibling bottom splits */
            if (a_mid > PACX_VQ_SPLIT_BITS && a_side > PACX_VQ_SPLIT_BITS && half >= 4 && half <= 32 &&
                depth < VQ_DEPTH && vq_quad_try(V, o, mv, sv, half, a_mid, a_side, lane)) {
                /* both children were bottom splits: their six fields are out */
                stack[2 * (depth - 1) + 1] = -1;            /* side done */
                VQ_S(o, 9);
            } else
#endif
            if (a_mid > 0 && a_mid <= PACX_VQ_SPLIT_BITS && a_side > 0 && a_side <= PACX_VQ_SPLIT_BITS &&
                       half >= 2 && half <= 32) {
                VQ_S(o, 9);                                 /* a quad attempt that fell through counts as one */
                /* both children are small leaves: code them side by side */
                vq_leaf_pair(V, o, mv, sv, half, a_mid, a_side, lane);
                stack[2 * (depth - 1) + 1] = -1;            /* side done */
                VQ_S(o, 10);
            } else if (a_mid > 0) {
                cur = mv;
                n = half;
                bits = a_mid;
                descend = true;
            }
        } else {
            if (bits > PACX_VQ_SPLIT_BITS)
                o.flags |= PACX_ST_VQ_UNDEFINED;           /* deeper than any real tree */
            vq_leaf(V, o, cur, n, bits > 32 ? 32 : bits, reg, reg + n, lane);
            VQ_S(o, 11);
        }
        if (descend)
            continue;
        /* climb to the nearest split whose side half is still to do */
        bool found = false;
        while (depth > 0) {
            const int half = stack[2 * (depth - 1)];
            const int a_side = stack[2 * (depth - 1) + 1];
            if (a_side >= 0) {
                stack[2 * (depth - 1) + 1] = -1;           /* side taken */
                if (a_side > 0) {
                    /* the split at depth-1 wrote into the region just below `reg` */
                    cur = (reg - 2 * half) + half;
                    n = half;
                    bits = a_side;
                    found = true;
                    break;
                }
            }
            reg = reg - 2 * half;
            depth -= 1;
        }
        VQ_S(o, 12);
        if (!found)
            return;
    }
}

/* ---- the split tree of one band, level by level ----------------------------- */
/* vq_shape walks the tree depth first, one node at a time: per node a whole wave runs the
 * scalar arithmetic of a split (two norms, six FP64 divisions, atan, quantise/dequantise of the
 * angle, the bit split) for a handful of useful lanes, and the kernel is bound by its instruction
 * count.  vq_shape_bfs walks the same tree breadth first:
 *   - the nodes of one depth are independent once their parents are split;
 *   - their folds / norms / normalisations run PACKED: P = power of two >= the largest half of the
 *     level, 64/P nodes per pass, one node per aligned block of P lanes.  The xor butterfly over a
 *     block adds the same numbers in the same order as the 64-lane one (plus exact zeros);
 *   - the scalar arithmetic of a level runs once, one node per LANE;
 *   - leaves are coded four or two at a time (vq_leaf_group), whoever their parents are;
 *   - nothing is written while walking: every node keeps its field (value, width); at the end
 *     the subtree widths go bottom-up, the stream positions top-down (the reference writes depth
 *     first: angle, mid subtree, side subtree) and all fields are ORed in at once, one per lane.
 * Same arithmetic per node as vq_shape; returns false without having written anything when the
 * tree does not fit the node store (then vq_shape codes the band). */
#define VQ_NC 96                       /* nodes per band tree */
#define VQ_NODE_BYTES (VQ_NC * 8 + 5 * VQ_NC * 2 + 4 * VQ_NC + 128 + 32)
struct VqNodes {
    unsigned long long *val;           /* [NC] field value (a split's |side|/|mid| while its level is open) */
    unsigned short *nn, *bb, *off, *tot, *pos;   /* [NC] length, bits, vector offset, subtree bits, stream bit */
    unsigned char *kind, *wid, *kid;   /* [NC] 0 split / 1 leaf / 3 leaf without a field; field width; [2 NC] children */
    unsigned char *sl, *ll;            /* [64] split / leaf nodes of the open level */
    unsigned char *lvl;                /* [VQ_DEPTH + 2] first node of every depth */
    __device__ __forceinline__ void bind(unsigned char *p)
    {
        val = (unsigned long long *)p;
        nn = (unsigned short *)(p + VQ_NC * 8);
        bb = nn + VQ_NC;
        off = bb + VQ_NC;
        tot = off + VQ_NC;
        pos = tot + VQ_NC;
        kind = (unsigned char *)(pos + VQ_NC);
        wid = kind + VQ_NC;
        kid = wid + VQ_NC;
        sl = kid + 2 * VQ_NC;
        ll = sl + 64;
        lvl = ll + 64;
    }
};

/* a field of up to 64 bits, by ONE lane (fields of different lanes may share words) */
__device__ __forceinline__ void vq_put_field(unsigned *words, int pos, unsigned long long val, int width)
{
    const int w = pos >> 5, t = (pos & 31) + width;
    if (t <= 32) {
        atomicOr(&words[w], (unsigned)val << (32 - t));
    } else if (t <= 64) {
        const unsigned long long v = val << (64 - t);
        atomicOr(&words[w], (unsigned)(v >> 32));
        atomicOr(&words[w + 1], (unsigned)v);
    } else {
        const int r = t - 64;
        const unsigned long long v = val >> r;
        atomicOr(&words[w], (unsigned)(v >> 32));
        atomicOr(&words[w + 1], (unsigned)v);
        atomicOr(&words[w + 2], (unsigned)val << (32 - r));
    }
}

__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v = max(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ __forceinline__ int wave_excl_scan_i32(int v, int lane, int &total)
{
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off)
            incl += t;
    }
    total = __shfl(incl, 63, 64);
    return incl - v;
}

__device__ __forceinline__ bool vq_shape_bfs(const VqView &V, VqOut &o, const double *x0, int n0, int bits0,
                                             double *region, int region_len, const VqNodes &N, int lane)
{
    const double half_pi = 1.5707963267948966;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int buf_cap = region_len / 2;
    double *bufs[2] = {region, region + buf_cap};
    if (lane == 0) {
        N.nn[0] = (unsigned short)n0;
        N.bb[0] = (unsigned short)bits0;
        N.off[0] = 0;
        N.kind[0] = 1;                                     /* the caller only comes with bits0 > 0 */
        N.wid[0] = 0;
        N.val[0] = 0ull;
        N.kid[0] = N.kid[1] = 0xFF;
        N.lvl[0] = 0;
    }
    if (bits0 > PACX_VQ_SPLIT_BITS && lane == 0)
        N.kind[0] = 0;
    int count = 1, lev_b = 0, lev_e = 1, depth = 0;
    bool undefined = false;                                /* per lane; joined at the end */
    const double *cur = x0;
    for (;;) {
        vq_fence();
        /* the open level: its splits and its leaves */
        int n_split = 0, n_leaf = 0;
        for (int base = lev_b; base < lev_e; base += 64) {
            const int j = base + lane;
            const int kd = (j < lev_e) ? N.kind[j] : 2;
            const unsigned long long ms = __builtin_amdgcn_ballot_w64(kd == 0);
            const unsigned long long ml = __builtin_amdgcn_ballot_w64(kd == 1);
            if (n_split + __popcll(ms) > 64 || n_leaf + __popcll(ml) > 64)
                return false;
            if (kd == 0)
                N.sl[n_split + __popcll(ms & below)] = (unsigned char)j;
            if (kd == 1)
                N.ll[n_leaf + __popcll(ml & below)] = (unsigned char)j;
            n_split += __popcll(ms);
            n_leaf += __popcll(ml);
        }
        vq_fence();
        double *nxt = bufs[depth & 1];                     /* free while this level's leaves are coded */
        /* ---- leaves of this level (vectors in cur) */
        if (n_leaf) {
            const int my = (lane < n_leaf) ? N.ll[lane] : 0;
            const int myn = (lane < n_leaf) ? N.nn[my] : 0;
            const unsigned long long big = __builtin_amdgcn_ballot_w64(myn > 32);
            const int nmax = wave_max_i32(myn > 32 ? 0 : myn);
            /* single leaves of more than 32 components (upper levels only) */
            for (unsigned long long m = big; m; m &= m - 1) {
                const int k = __builtin_ctzll(m);
                const int node = N.ll[k];
                const int n = N.nn[node];
                int bits = N.bb[node];
                bits = bits > 32 ? 32 : bits;
                const int K = ldc(&V.k_of[n * 33 + bits]);
                const int width = ldc(&V.w_of[n * 33 + bits]);
                bool ok = true;
                const unsigned long long idx = vq_leaf_idx(V, cur + N.off[node], n, K, nxt, nxt + n, lane, ok);
                if (!ok)
                    undefined = true;
                if (lane == 0) {
                    N.val[node] = ok ? idx : 0ull;
                    N.wid[node] = (unsigned char)width;
                }
            }
            if (nmax > 0) {
                const int lw = nmax <= 16 ? 4 : 5;           /* log2 of the group width */
                const int W = 1 << lw, G = 64 >> lw;
                /* compact list of the small leaves in lane order */
                const bool small = lane < n_leaf && myn <= 32;
                const unsigned long long msm = __builtin_amdgcn_ballot_w64(small);
                const int n_small = __popcll(msm);
                vq_fence();
                if (small)
                    N.sl[__popcll(msm & below)] = (unsigned char)my;   /* the split list is rebuilt below */
                vq_fence();
                for (int p0 = 0; p0 < n_small; p0 += G) {
                    const int g = lane >> lw, l = lane & (W - 1);
                    const bool valid = p0 + g < n_small;
                    const int node = valid ? N.sl[p0 + g] : 0;
                    const int n = valid ? N.nn[node] : 0;
                    int bits = valid ? N.bb[node] : 0;
                    bits = bits > 32 ? 32 : bits;
                    const int K = valid ? V.k_of[n * 33 + bits] : 0;
                    const int width = valid ? V.w_of[n * 33 + bits] : 0;
                    const double x = (valid && l < n) ? cur[N.off[node] + l] : 0.0;
                    bool ok = false;
                    unsigned long long term;
                    if (lw == 4)
                        term = vq_leaf_group<16>(V, x, n, K < 0 ? 0 : K, l, ok);
                    else
                        term = vq_leaf_group<32>(V, x, n, K < 0 ? 0 : K, l, ok);
                    if (valid && l == 0) {
                        if (K < 0) {                       /* a 1-dimensional leaf: the reference never returns */
                            N.kind[node] = 3;
                            N.wid[node] = 0;
                            N.val[node] = 0ull;
                        } else {
                            N.val[node] = ok ? term : 0ull;
                            N.wid[node] = (unsigned char)width;
                        }
                    }
                    if (valid && (K < 0 || !ok))
                        undefined = true;
                }
                /* the split list again (the leaf passes borrowed it) */
                vq_fence();
                int ns2 = 0;
                for (int base = lev_b; base < lev_e; base += 64) {
                    const int j = base + lane;
                    const int kd = (j < lev_e) ? N.kind[j] : 2;
                    const unsigned long long ms = __builtin_amdgcn_ballot_w64(kd == 0);
                    if (kd == 0)
                        N.sl[ns2 + __popcll(ms & below)] = (unsigned char)j;
                    ns2 += __popcll(ms);
                }
                vq_fence();
            }
        }
        if (!n_split)
            break;
        if (depth + 1 > VQ_DEPTH)
            return false;
        /* ---- splits of this level: where their children's vectors go */
        const int snode = (lane < n_split) ? N.sl[lane] : 0;
        const int sn = (lane < n_split) ? N.nn[snode] : 0;
        const int shalf = sn - sn / 2;
        int room;
        const int noff = wave_excl_scan_i32(2 * shalf, lane, room);
        if (room > buf_cap)
            return false;
        if (lane < n_split)
            N.tot[snode] = (unsigned short)noff;
        const unsigned long long wide = __builtin_amdgcn_ballot_w64(shalf > 64);
        const int hmax = wave_max_i32(shalf > 64 ? 0 : shalf);
        vq_fence();
        /* nodes of more than 128 components: one at a time, lanes strided over the half */
        for (unsigned long long m = wide; m; m &= m - 1) {
            const int k = __builtin_ctzll(m);
            const int node = N.sl[k];
            const int n = N.nn[node];
            const double *src = cur + N.off[node];
            const int cut = n / 2, half = n - cut;
            double *mv = nxt + N.tot[node], *sv = mv + half;
            double mm = 0.0, ss = 0.0;
            for (int i = lane; i < half; i += 64) {
                const double left = (i < cut) ? src[i] : 0.0;
                const double right = src[cut + i];
                const double mid = (left + right) / 2.0;
                const double sd = (left - right) / 2.0;
                mv[i] = mid;
                sv[i] = sd;
                mm = fma(mid, mid, mm);
                ss = fma(sd, sd, ss);
            }
            const double m_l2 = sqrt(wave_sum_f64(mm));
            const double s_l2 = sqrt(wave_sum_f64(ss));
            for (int i = lane; i < half; i += 64) {
                if (m_l2 != 0.0)
                    mv[i] = mv[i] / m_l2;
                if (s_l2 != 0.0)
                    sv[i] = sv[i] / s_l2;
            }
            if (lane == 0)
                N.val[node] = (unsigned long long)__double_as_longlong(m_l2 == 0.0 ? -1.0 : s_l2 / m_l2);
        }
        /* the others packed: one node per aligned block of P lanes */
        if (hmax > 0) {
            int lp = 0;
            while ((1 << lp) < hmax)
                ++lp;
            const int P = 1 << lp, G = 64 >> lp;
            const bool narrow = lane < n_split && shalf <= 64;
            const unsigned long long mn = __builtin_amdgcn_ballot_w64(narrow);
            const int n_narrow = __popcll(mn);
            if (narrow)
                N.ll[__popcll(mn & below)] = (unsigned char)snode;   /* the leaf list is done with */
            vq_fence();
            for (int p0 = 0; p0 < n_narrow; p0 += G) {
                const int g = lane >> lp, i = lane & (P - 1);
                const bool valid = p0 + g < n_narrow;
                const int node = valid ? N.ll[p0 + g] : 0;
                const int n = valid ? N.nn[node] : 0;
                const int cut = n / 2, half = n - cut;
                const double *src = cur + (valid ? N.off[node] : 0);
                const bool mine = i < half;
                const double left = (mine && i < cut) ? src[i] : 0.0;
                const double right = mine ? src[cut + i] : 0.0;
                double mid = mine ? (left + right) / 2.0 : 0.0;
                double sd = mine ? (left - right) / 2.0 : 0.0;
                double mm = fma(mid, mid, 0.0), ss = fma(sd, sd, 0.0);
                for (int off = P >> 1; off > 0; off >>= 1) {
                    mm = mm + __shfl_xor(mm, off, 64);
                    ss = ss + __shfl_xor(ss, off, 64);
                }
                const double m_l2 = sqrt(mm), s_l2 = sqrt(ss);
                if (m_l2 != 0.0)
                    mid = mid / m_l2;
                if (s_l2 != 0.0)
                    sd = sd / s_l2;
                if (mine) {
                    double *mv = nxt + N.tot[node];
                    mv[i] = mid;
                    mv[half + i] = sd;
                }
                if (valid && i == 0)
                    N.val[node] = (unsigned long long)__double_as_longlong(m_l2 == 0.0 ? -1.0 : s_l2 / m_l2);
            }
        }
        vq_fence();
        /* ---- the level's scalar arithmetic, one split per lane; its children join the store */
        {
            const bool has = lane < n_split;
            const int bits = has ? N.bb[snode] : 0;
            const int half = has ? shalf : 1;
            const double q = has ? __longlong_as_double((long long)N.val[snode]) : -1.0;
            const double theta = (q < 0.0) ? 0.0 : vq_atan(q);
            const int a_theta = (int)floor((double)bits / (double)half + V.half_log2[half]);
            int a_rest = bits - a_theta;
            if (a_rest < 0)
                a_rest = 0;
            const double tn = theta / half_pi;
            double theta_q = 0.0;
            unsigned long long code = 0ull;
            int w_theta = 0;
            if (a_theta > 62) {
                if (has)
                    undefined = true;
            } else if (a_theta > 0) {
                if (tn >= 1.0) {
                    code = (1ull << (a_theta - 1)) - 1ull;
                } else {
                    const double factor = (a_theta <= 53) ? (double)((1ull << a_theta) - 1ull) : ldexp(1.0, a_theta);
                    code = (unsigned long long)floor((factor * tn + 1.0) * 0.5);
                }
                w_theta = a_theta;
                const unsigned long long mag = code & ((1ull << (a_theta - 1)) - 1ull);
                const double den = (a_theta <= 53) ? (double)((1ull << a_theta) - 1ull) : ldexp(1.0, a_theta);
                double dq = (double)(2ull * mag) / den;
                if (code >> (a_theta - 1))
                    dq = -dq;
                theta_q = dq * half_pi;
            }
            int a_mid = 0;
            if (theta_q != 0.0) {
                double lt;
                if (a_theta <= PACX_VQ_THETA_TABLE_BITS && theta_q > 0.0)
                    lt = V.log2_tan[((1 << (a_theta - 1)) - 1) + (int)code];
                else
                    lt = vq_log2_tan(theta_q);
                const double v = ((double)a_rest - (double)(half - 1) * lt) / 2.0;
                const double f = floor(v);
                a_mid = (f < 0.0) ? 0 : ((f > (double)a_rest) ? a_rest : (int)f);
            }
            const int a_side = a_rest - a_mid;
            const int c_mid = (has && a_mid > 0) ? 1 : 0, c_side = (has && a_side > 0) ? 1 : 0;
            int born;
            const int first = count + wave_excl_scan_i32(c_mid + c_side, lane, born);
            if (count + born > VQ_NC)
                return false;
            if (has) {
                N.val[snode] = code;
                N.wid[snode] = (unsigned char)w_theta;
                N.kid[2 * snode] = c_mid ? (unsigned char)first : 0xFF;
                N.kid[2 * snode + 1] = c_side ? (unsigned char)(first + c_mid) : 0xFF;
                const int at = N.tot[snode];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int a = c ? a_side : a_mid;
                    if (a <= 0)
                        continue;
                    const int id = c ? first + c_mid : first;
                    const bool splits = a > PACX_VQ_SPLIT_BITS && depth + 1 < VQ_DEPTH;
                    if (a > PACX_VQ_SPLIT_BITS && !splits)
                        undefined = true;                  /* deeper than any real tree */
                    N.nn[id] = (unsigned short)half;
                    N.bb[id] = (unsigned short)(a > 65535 ? 65535 : a);
                    N.off[id] = (unsigned short)(at + (c ? half : 0));
                    N.kind[id] = splits ? 0 : 1;
                    N.wid[id] = 0;
                    N.val[id] = 0ull;
                    N.kid[2 * id] = N.kid[2 * id + 1] = 0xFF;
                }
            }
            count += born;
        }
        depth += 1;
        if (lane == 0)
            N.lvl[depth] = (unsigned char)lev_e;
        lev_b = lev_e;
        lev_e = count;
        cur = nxt;
    }
    /* ---- widths bottom-up, positions (and field numbers) top-down, then every field at once */
    if (lane == 0)
        N.lvl[depth + 1] = (unsigned char)count;
    vq_fence();
    /* field counts ride in nn[] (lengths are not needed any more), field numbers in off[] */
    for (int d = depth; d >= 0; --d) {
        const int b = N.lvl[d], e = N.lvl[d + 1];
        for (int j = b + lane; j < e; j += 64) {
            const int k0 = N.kid[2 * j], k1 = N.kid[2 * j + 1];
            int t = N.wid[j], f = (N.kind[j] == 3) ? 0 : 1;
            if (k0 != 0xFF) { t += N.tot[k0]; f += N.nn[k0]; }
            if (k1 != 0xFF) { t += N.tot[k1]; f += N.nn[k1]; }
            N.tot[j] = (unsigned short)t;
            N.nn[j] = (unsigned short)f;
        }
        vq_fence();
    }
    if (lane == 0) {
        N.pos[0] = (unsigned short)o.pos;
        N.off[0] = (unsigned short)o.log_n;
    }
    vq_fence();
    for (int d = 0; d <= depth; ++d) {
        const int b = N.lvl[d], e = N.lvl[d + 1];
        for (int j = b + lane; j < e; j += 64) {
            const int k0 = N.kid[2 * j], k1 = N.kid[2 * j + 1];
            int p = N.pos[j] + N.wid[j], r = N.off[j] + ((N.kind[j] == 3) ? 0 : 1);
            if (k0 != 0xFF) {
                N.pos[k0] = (unsigned short)p;
                N.off[k0] = (unsigned short)r;
                p += N.tot[k0];
                r += N.nn[k0];
            }
            if (k1 != 0xFF) {
                N.pos[k1] = (unsigned short)p;
                N.off[k1] = (unsigned short)r;
            }
        }
        vq_fence();
    }
    for (int j = lane; j < count; j += 64) {
        const int w = N.wid[j];
        unsigned long long v = N.val[j];
        if (w > 0 && w < 64)
            v &= (1ull << w) - 1ull;
        if (w > 0)
            vq_put_field(o.words, N.pos[j], v, w);
        if (o.log && N.kind[j] != 3 && N.off[j] < o.log_cap) {
            pacx_vq_entry e;
            e.value = v;
            e.width = w;
            e.band = o.band;
            o.log[N.off[j]] = e;
        }
    }
    if (__builtin_amdgcn_ballot_w64(undefined))
        o.flags |= PACX_ST_VQ_UNDEFINED;
    o.pos += N.tot[0];
    o.log_n += N.nn[0];
    vq_fence();
    return true;
}

/* ------------------------------------------------------------------ kernel */
struct VqArgs {
    const uint8_t *flags;
    int n_ch;
    long long n_cf;
    int mixed;                 /* 1: units are [cf][8] and flags decide long/short */
    const double *lines;       /* [cf][1024] unscaled                       */
    const int32_t *overall;    /* [cf][8]                                   */
    int32_t *bit_alloc;        /* [cf][band_stride] in: BitAlloc, out: final */
    const double *sbr_mean;    /* [cf][8] mean |FFT|/halfN of the omitted bands */
    const uint32_t *status_in;
    uint32_t *status;
    uint8_t *payload;
    int payload_stride;
    int32_t *n_bytes;
    unsigned *unit_words;      /* [cf*8][VQ_WORDS] short sub-block strings  */
    int32_t *unit_bits;        /* [cf*8][2]: written bits, size-rule bits   */
    pacx_vq_entry *log;
    int32_t *log_count;
    int log_cap;               /* entries per band                          */
    int redo;                  /* 1: code only the channel-frames k_vq_frame left (n_bytes = -1) */
    const int32_t *cf_list;    /* k_vq_frame: the channel-frames of this launch (NULL: all n_cf) ... */
    const int32_t *cf_count;   /* ... and how many (device side) */
    int bfs;                   /* shape bits from which a band's tree is walked level by level (vq_shape_bfs); 0: never */
    int rot;                   /* k_vq_frame: 1 = the wave that runs the serial stages rotates with the workgroup index */
};

#ifdef PACX_VQ_DEBUG
/* phase stamps (s_memtime) summed over all waves: a measuring aid (build.py --phase-debug) */
#define VQ_T(k) do { long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                     if ((threadIdx.x & 63) == 0 && vq_last) atomicAdd((unsigned long long *)&g_vq_dbg[k], (unsigned long long)(t_ - vq_last)); \
                     vq_last = t_; } while (0)
extern "C" int pacx_debug_read_vq(long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vq_dbg), sizeof(long long) * n);
}
#else
#define VQ_T(k) do { } while (0)
#endif

/* one (sub-)block: the body of k_vq (one workgroup per unit) and of k_vq_redo (the units k_vq_frame left) */
__device__ __forceinline__ void vq_unit_body(const PacxTables &T, const VqView &V, const VqArgs &A, const long long unit)
{
#ifdef PACX_VQ_DEBUG
    long long vq_last = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *words = (unsigned *)smem;                               /* VQ_WORDS        */
    double *gain_s = (double *)(smem + VQ_WORDS * 4);                 /* 32              */
    int *ba_s = (int *)(gain_s + PACX_MAX_BANDS);                     /* 32              */
    int *start_s = ba_s + PACX_MAX_BANDS;                             /* 33              */
    int *ticket = start_s + PACX_MAX_BANDS + 1;                       /* 1 (+2 pad)      */
    int *stack_all = ticket + 3;                                      /* waves * 2*DEPTH */
    double *xs = (double *)(stack_all + VQ_WAVES * 2 * VQ_DEPTH);     /* 1024: the block's unit shapes */
    double *scr_all = xs + PACX_M_LONG;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          /* uniform: the per-wave LDS pointers stay scalar */
    VqNodes N;
    N.bind((unsigned char *)(scr_all + V.scr_off[VQ_WAVES]) + wave * VQ_NODE_BYTES);
    const long long cf = A.mixed ? unit / PACX_SUB : unit;
    const int sb = A.mixed ? (int)(unit % PACX_SUB) : 0;
    if (cf >= A.n_cf)
        return;
    const long long frame = cf / A.n_ch;
    const unsigned fl = A.flags ? A.flags[frame] : 0u;
    const bool is_short = A.mixed && (fl & 2u);
    if (!is_short && sb != 0)
        return;
    if (is_short && A.status_in) {             /* hop dropped: nothing to code */
        unsigned st = 0;
        for (int c = 0; c < A.n_ch; ++c)
            st |= A.status_in[frame * A.n_ch + c];
        if (st & PACX_ST_ZERO_SUBBLOCK)
            return;
    }
    if (A.redo && A.n_bytes[cf] != -1)
        return;                                /* k_vq_frame coded this channel-frame */
    const int nb = is_short ? T.nb_short : T.nb_long;
    const int32_t *__restrict__ lower = is_short ? T.band_lower_short : T.band_lower_long;
    const int32_t *__restrict__ count = is_short ? T.band_lines_short : T.band_lines_long;
    const int first_omit = (!is_short && T.use_sbr) ? T.first_omitted : nb;
    const long long boff = cf * T.band_stride + (is_short ? sb * T.nb_short : 0);
    const double *__restrict__ lin = A.lines + cf * PACX_M_LONG + (is_short ? sb * PACX_M_SHORT : 0);
    const int ov = A.overall[cf * PACX_SUB + sb];
    const double up = (double)(1 << ov);
    const int lead = is_short ? 0 : 3;

    for (int i = tid; i < VQ_WORDS; i += 64 * VQ_WAVES)
        words[i] = 0u;
    for (int i = tid; i <= min(V.l_max, VQ_LMAX); i += 64 * VQ_WAVES)
        vq_row_off_s[i] = V.row_off[i];
    if (tid == 0)
        *ticket = VQ_WAVES;
    double *scr = scr_all + V.scr_off[wave];
    int *stack = stack_all + wave * 2 * VQ_DEPTH;
    const uint8_t *order = is_short ? V.order_short : V.order_long;

    VQ_T(15);
    /* phase A: gains (np.linalg.norm of the scaled band; an omitted band is the
       one-element vector [mean |FFT|]) */
    for (int b = wave; b < nb; b += VQ_WAVES) {
        double g;
        if (b >= first_omit) {
            const double v = A.sbr_mean[cf * PACX_SUB + (b - first_omit)] * up;
            g = sqrt(v * v);
        } else {
            const int lo = ldc(&lower[b]), cnt = ldc(&count[b]);
            double acc = 0.0;
            for (int i = lane; i < cnt; i += 64) {
                const double x = lin[lo + i] * up;
                xs[lo + i] = x;
                acc = fma(x, x, acc);
            }
            g = sqrt(wave_sum_f64(acc));
            /* the band's shape x / gain stays in LDS for phase B (whichever wave codes the band):
               the lines are read from memory once and no band starts with a round trip to L2 */
            for (int i = lane; i < cnt; i += 64)
                xs[lo + i] = xs[lo + i] / g;
        }
        if (lane == 0)
            gain_s[b] = g;
    }
    VQ_T(0);
    __syncthreads();
    VQ_T(1);
    if (tid < 64) {
        /* final allocations, band positions, header fields */
        int ba = 0, r_bits = 0;
        if (lane < nb) {
            ba = A.bit_alloc[boff + lane];
            if (ba && gain_s[lane] == 0.0)
                ba = 0;                                   /* coder/codec.py:352-353 */
            r_bits = ba * ((lane >= first_omit) ? 1 : count[lane]);
            A.bit_alloc[boff + lane] = ba;
            ba_s[lane] = ba;
        }
        int incl = r_bits;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off)
                incl += t;
        }
        const int head = lead + T.n_scale_bits + T.n_mant_size_bits * nb;
        if (lane < nb)
            start_s[lane] = head + incl - r_bits;
        if (lane == nb - 1)
            start_s[nb] = head + incl;
        if (lane == 0) {
            if (!is_short) {
                vq_put32(words, 0, fl & 1u, 1);
                vq_put32(words, 1, (fl >> 1) & 1u, 1);
                vq_put32(words, 2, (fl >> 2) & 1u, 1);
            }
            vq_put32(words, lead, (unsigned)ov, T.n_scale_bits);
        }
        if (lane < nb)
            vq_put32(words, lead + T.n_scale_bits + T.n_mant_size_bits * lane, (unsigned)(ba ? ba - 1 : 0),
                     T.n_mant_size_bits);
    }
    __syncthreads();
    VQ_T(2);

    /* phase B: bands largest first; wave w starts on the (w+1)-th largest, the
       rest go by ticket */
    unsigned raised = 0;
    int tk = wave;
    for (;; tk = -1) {
        if (tk < 0) {
            if (lane == 0)
                tk = atomicAdd(ticket, 1);
            tk = __builtin_amdgcn_readfirstlane(tk);
        }
        if (tk >= nb)
            break;
        const int b = order[tk];
        const int ba = ba_s[b];
        const long long log_slot = (cf * PACX_SUB + sb) * PACX_MAX_BANDS + b;
        if (!ba) {
            if (A.log_count && lane == 0)
                A.log_count[log_slot] = 0;
            continue;
        }
        VqOut o;
        o.words = words;
        o.pos = start_s[b];
        o.log = A.log ? A.log + log_slot * A.log_cap : nullptr;
        o.log_cap = A.log_cap;
        o.log_n = 0;
        o.band = b;
        o.flags = 0;
#ifdef PACX_VQ_DEBUG
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(o.t_last) :: "memory");
#endif
        const double gain = gain_s[b];
        if (b >= first_omit) {
            /* L = 1: every bit goes to the gain (gain_shape_alloc(R, 1)) */
            const double g = vq_log(1.0 + 255.0 * fabs(gain / 1.0)) / V.log_mu1;
            vq_quantize_emit(o, g, ba, lane);
        } else {
            const int lo = ldc(&lower[b]), cnt = ldc(&count[b]);
            const int r_bits = ba * cnt;
            int bits_gain = (int)floor((double)r_bits / (double)cnt + ldc(&V.half_log2[cnt]));
            int bits_shape = r_bits - bits_gain;
            if (bits_shape < 0)
                bits_shape = 0;
            if (bits_shape != 0) {
                const double *x0 = xs + lo;                /* [cnt] shape = x / gain, from phase A */
                const int before = o.pos;
                VQ_S(o, 13);
#if defined(VQ_ONLY_BFS)
                vq_shape_bfs(V, o, x0, cnt, bits_shape, scr, 2 * cnt + 4 * VQ_DEPTH, N, lane);
#elif defined(VQ_ONLY_DFS)
                vq_shape(V, o, x0, cnt, bits_shape, scr, stack, lane);
#else
                if (!A.bfs || bits_shape < A.bfs || !vq_shape_bfs(V, o, x0, cnt, bits_shape, scr, 2 * cnt + 4 * VQ_DEPTH, N, lane))
                    vq_shape(V, o, x0, cnt, bits_shape, scr, stack, lane);
#endif
                bits_gain += bits_shape - (o.pos - before);
            }
            if (bits_gain < 0)
                bits_gain = 0;
            const double g = vq_log(1.0 + 255.0 * fabs(gain / (double)cnt)) / V.log_mu1;
            vq_quantize_emit(o, g, bits_gain, lane);
            VQ_S(o, 14);
        }
        if (o.pos != start_s[b + 1])
            o.flags |= PACX_ST_VQ_UNDEFINED;               /* a band must fill its slot exactly */
        raised |= o.flags;
        if (A.log_count && lane == 0)
            A.log_count[log_slot] = o.log_n;
    }
    if (raised && A.status && lane == 0)
        atomicOr(&A.status[cf], raised);
    VQ_T(3);
    __syncthreads();
    VQ_T(4);

    /* hand the string over */
    const int written = start_s[nb];                       /* includes `lead` */
    int size_rule = T.n_scale_bits;                        /* getNumBytesNeeded */
    for (int b = 0; b < nb; ++b)
        size_rule += T.n_mant_size_bits + T.n_scale_bits;
    size_rule += written - (lead + T.n_scale_bits + T.n_mant_size_bits * nb);
    if (!is_short) {
        const int nbytes = (size_rule + 4 + 7) >> 3;
        unsigned *dst = (unsigned *)(A.payload + cf * (long long)A.payload_stride);
        for (int i = tid; i < (nbytes + 3) / 4; i += 64 * VQ_WAVES)
            dst[i] = __builtin_bswap32(words[i]);
        if (tid == 0)
            A.n_bytes[cf] = nbytes;
    } else {
        unsigned *dst = A.unit_words + unit * VQ_WORDS;
        for (int i = tid; i < (written + 31) / 32; i += 64 * VQ_WAVES)
            dst[i] = words[i];
        if (tid == 0) {
            A.unit_bits[unit * 2] = written;
            A.unit_bits[unit * 2 + 1] = size_rule;
        }
    }
    VQ_T(5);
}

__global__ __launch_bounds__(64 * VQ_WAVES, VQ_OCC) void k_vq(PacxTables T, VqView V, VqArgs A)
{
    vq_unit_body(T, V, A, blockIdx.x);
}

/* behind k_vq_frame: the channel-frames it flagged (n_bytes = -1), their 1 or 8 units one after the other.
   A grid-stride walk over the frames: a launch of one workgroup per unit, nearly all of which have nothing
   to do, cost 49 us on a block-switched batch (65 536 workgroups) */
__global__ __launch_bounds__(64 * VQ_WAVES, VQ_OCC) void k_vq_redo(PacxTables T, VqView V, VqArgs A)
{
    for (long long cf = blockIdx.x; cf < A.n_cf; cf += gridDim.x) {
        if (A.n_bytes[cf] != -1)
            continue;
        const unsigned fl = A.flags ? A.flags[cf / A.n_ch] : 0u;
        const int n_sub = (A.mixed && (fl & 2u)) ? PACX_SUB : 1;
        for (int sb = 0; sb < n_sub; ++sb) {
            __syncthreads();                               /* the previous unit's LDS is done with */
            vq_unit_body(T, V, A, A.mixed ? cf * PACX_SUB + sb : cf);
        }
        if (T.guard && A.status && threadIdx.x == 0)
            atomicOr(&A.status[cf], PACX_ST_GUARD);        /* the band-by-band coder takes no margins: flagged as it comes */
    }
}

/* ------------------------------------------------ frame-level walk (k_vq_frame) */
/* The trees of ALL bands of a (sub-)block, level by level, by one workgroup.
 *
 * k_vq hands bands to waves; inside a band a wave works on one tree node (or a few siblings) at a
 * time, and most bands are small: a root split and two leaves keep a whole wave busy for three
 * passes.  Here every node of every band's tree goes into one store:
 *   level 0 = the bands' roots, level d+1 = the children of level d;
 *   per level the nodes are sorted into classes of equal lane footprint (leaves of up to 16 / up to
 *   32 components, splits whose half fits 8 / 16 / 32 / 64 lanes, the few larger ones) and coded in
 *   PACKED passes -- eight small splits or four small leaves per pass whichever bands they belong
 *   to -- the passes dealt round-robin to the four waves;
 *   the scalar arithmetic of the level's splits runs one split per LANE, 64 at a time;
 *   nothing is written while walking; at the end subtree widths go bottom-up and stream positions
 *   top-down (depth-first order inside a band, bands at their static positions), every field is
 *   ORed in by its own lane, and the gains of all bands are quantised side by side.
 * Arithmetic per node is that of vq_shape / vq_leaf / vq_leaf_group.  A unit whose trees do not fit
 * the store (bit rates far above the shipped ones) is left to k_vq (and k_vq_join): n_bytes = -1.
 * A short-coded frame is ONE unit too: its 8 x nb_short bands are the unit's bands, its eight strings are
 * written back to back by the same workgroup (no per-sub-block launch geometry, no join pass). */
#ifndef VQF_NCAP
#define VQF_NCAP 272                   /* nodes per (sub-)block.  With the 1088-double level buffers and the small
                                          row-offset table that is 31 KB of LDS: five workgroups per CU (448 nodes: four,
                                          571 against 487 us on one box; a sixth, at 80 VGPRs with three spilled,
                                          gave nothing: 494 us) */
#endif
#define VQF_NLV 256                    /* nodes per level */
#ifndef VQF_BUF
#define VQF_BUF (PACX_M_LONG + 64)     /* doubles per level buffer */
#endif
#define VQF_VB 64                      /* bands of a unit: a long block's, or the 8 x nb_short of a short frame */
#define VQF_FIXED 4160                 /* words, gains, allocations, starts, ticket, gain bits, roots, counters */
#define VQF_SMEM (VQF_FIXED + 2 * VQF_BUF * 8 + VQF_NCAP * 8 + 6 * VQF_NCAP * 2 + 4 * VQF_NCAP + 2 * VQF_NLV * 2 + 128 + 64)

struct VqfStore {
    unsigned long long *val;
    unsigned short *nn, *bb, *off, *tot, *pos, *kid;
    unsigned char *kind, *wid, *band, *has;
    unsigned short *ord;               /* [2][NLV] a level's nodes grouped by class (two levels' lists alternate) */
    int *cls;                          /* [2][16] class starts in ord ([9] used) */
    unsigned short *lvl;               /* [VQ_DEPTH + 2] first node of every depth */
    __device__ __forceinline__ void bind(unsigned char *p)
    {
        val = (unsigned long long *)p;
        nn = (unsigned short *)(p + VQF_NCAP * 8);
        bb = nn + VQF_NCAP;
        off = bb + VQF_NCAP;
        tot = off + VQF_NCAP;
        pos = tot + VQF_NCAP;
        kid = pos + VQF_NCAP;
        kind = (unsigned char *)(kid + VQF_NCAP);
        wid = kind + VQF_NCAP;
        band = wid + VQF_NCAP;
        has = band + VQF_NCAP;
        ord = (unsigned short *)(has + VQF_NCAP);
        cls = (int *)(ord + 2 * VQF_NLV);
        lvl = (unsigned short *)(cls + 32);
    }
};

/* QuantizeUniform(x, n_bits) for x >= 0 as vq_quantize_emit evaluates it: the code's two words */
__device__ __forceinline__ void vq_quantize_code(double x, int n_bits, unsigned long long &hi, unsigned long long &lo)
{
    hi = 0;
    lo = 0;
    if (x >= 1.0) {
        const int ones = n_bits - 1;
        if (ones >= 64) {
            lo = ~0ull;
            hi = (ones - 64 >= 64) ? ~0ull : ((1ull << (ones - 64)) - 1ull);
        } else {
            lo = (ones == 0) ? 0ull : ((~0ull) >> (64 - ones));
        }
    } else {
        const double factor = (n_bits <= 53) ? (double)((1ull << n_bits) - 1ull) : ldexp(1.0, n_bits);
        const double code = floor((factor * x + 1.0) * 0.5);
        if (n_bits <= 64) {
            lo = (unsigned long long)code;
        } else {
            const double top = floor(ldexp(code, -64));
            hi = (unsigned long long)top;
            lo = (unsigned long long)(code - ldexp(top, 64));
        }
    }
}

#ifdef PACX_VQ_DEBUG
#define VQF_SUB(k) do { long long t3_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3_) :: "memory"); \
                        if (lane == 0) atomicAdd((unsigned long long *)&g_vq_dbg[k], (unsigned long long)(t3_ - t_sub)); t_sub = t3_; } while (0)
#else
#define VQF_SUB(k) do { } while (0)
#endif
/* -DPACX_VQ_WAITDBG (measuring aid, tools/vq_wait_probe.py): per ROLE of a wave (0 = sorts the levels and runs the
   scalar stage) the cycles it spends busy before, and waiting at, each of the two barriers of a level, plus the
   workgroup's whole time: stamps only next to barriers (where the LDS queue is drained anyway), accumulated in
   registers, one atomic per counter at the end -- the kernel runs as it does without them */
#ifdef PACX_VQ_WAITDBG
__device__ long long g_vqw_dbg[32];
extern "C" int pacx_debug_read_vqw(long long *out, int n, int clear)
{
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vqw_dbg), sizeof(long long) * n);
    if (clear) {
        long long z[32] = {0};
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_vqw_dbg), z, sizeof(z));
    }
    return rc;
}
#define VQW_NOW(v) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
#define VQW_BARRIER(busy, wait) do { long long a_, b_; VQW_NOW(a_); __syncthreads(); VQW_NOW(b_); \
                                     busy += a_ - vqw_t; wait += b_ - a_; vqw_t = b_; } while (0)
#else
#define VQW_BARRIER(busy, wait) __syncthreads()
#endif
#define VQF_HL 64                      /* 0.5 log2(L), L < 64, in LDS */
#define VQF_LT 127                     /* log2(tan) of the angle codes of 1..7 bits in LDS */
__shared__ double vqf_half_log2[VQF_HL];
__shared__ double vqf_log2_tan[VQF_LT + 1];
#ifndef VQF_OCC
#define VQF_OCC 5
#endif
/* GUARD: the instantiation handles with pacx_config.guard launch -- it also takes the margins of PACX_ST_GUARD (split angles,
   band gains, pulse-search floors and ties, rounding-noise lines); the other one is the kernel as it was (the margins cost
   2.5 % even when switched off at run time: registers and code size) */
template <bool GUARD>
__global__ __launch_bounds__(64 * VQ_WAVES, VQF_OCC) void k_vq_frame(PacxTables T, VqView V, VqArgs A)
{
#ifdef PACX_VQ_DEBUG
    long long vqf_last = 0;
#define VQF_T(k) do { long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                      if (threadIdx.x == 0 && vqf_last) atomicAdd((unsigned long long *)&g_vq_dbg[k], (unsigned long long)(t_ - vqf_last)); \
                      vqf_last = t_; } while (0)
#else
#define VQF_T(k) do { } while (0)
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *words = (unsigned *)smem;                               /* VQ_WORDS        */
    /* a short-coded frame is ONE unit: its 8 x nb_short bands are the unit's (virtual) bands, band
       vb = 8 j + ... = j nb + b of sub-block j -- at most VQF_VB of them */
    double *gain_s = (double *)(smem + VQ_WORDS * 4);                 /* VB              */
    int *ba_s = (int *)(gain_s + VQF_VB);                             /* VB              */
    int *start_s = ba_s + VQF_VB;                                     /* VB: first stream bit of the band's fields */
    int *end_s = start_s + VQF_VB;                                    /* VB: where they must end */
    int *misc = end_s + VQF_VB;                                       /* node count, overflow, total bits, 5 spare */
    int *bg_s = misc + 8;                                             /* VB gain bits before the shape's slack */
    int *bs_s = bg_s + VQF_VB;                                        /* VB shape bits */
    unsigned short *root_s = (unsigned short *)(bs_s + VQF_VB);       /* VB root node of a band, 0xFFFF none */
    double *buf0 = (double *)(smem + VQF_FIXED);                      /* two level buffers of VQF_BUF doubles */
    VqfStore N;
    N.bind(smem + VQF_FIXED + 2 * VQF_BUF * 8);
    static_assert(VQ_WORDS * 4 + VQF_VB * 8 + 5 * VQF_VB * 4 + 8 * 4 + VQF_VB * 2 <= VQF_FIXED && VQF_FIXED % 16 == 0,
                  "fixed part of k_vq_frame's LDS");

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef PACX_VQ_WAITDBG
    long long vqw_start, vqw_p;
    VQW_NOW(vqw_start);
    vqw_p = vqw_start;
#define VQW_P(k) do { long long n_; VQW_NOW(n_); if (tid == 0) atomicAdd((unsigned long long *)&g_vqw_dbg[21 + (k)], (unsigned long long)(n_ - vqw_p)); vqw_p = n_; } while (0)
#else
#define VQW_P(k) do { } while (0)
#endif
    /* `wave` is the wave's ROLE in the level loop, not its hardware slot: role 0 sorts the levels and runs the
       (first chunk of the) scalar stage while the others code leaves.  The four waves of a workgroup sit on the
       four SIMDs of the CU; with the role fixed to the hardware wave, the serial stages of all five resident
       workgroups land on one SIMD and the other three idle -- the role rotates with the workgroup index */
    const int hw_wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = A.rot ? ((hw_wave + (int)(blockIdx.x & 3u)) & 3) : hw_wave;
    const unsigned long long below = (1ull << lane) - 1ull;
    long long cf = blockIdx.x;
    if (A.cf_list) {                           /* a launch over one of the frame lists of a block-switched batch */
        if (cf >= (long long)*A.cf_count)
            return;
        cf = A.cf_list[cf];
    }
    if (cf >= A.n_cf)
        return;
    const long long frame = cf / A.n_ch;
    const unsigned fl = A.flags ? A.flags[frame] : 0u;
    const bool is_short = A.mixed && (fl & 2u);
    if (is_short && A.status_in) {             /* hop dropped: nothing to code (n_bytes stays 0) */
        unsigned st = 0;
        for (int c = 0; c < A.n_ch; ++c)
            st |= A.status_in[frame * A.n_ch + c];
        if (st & PACX_ST_ZERO_SUBBLOCK)
            return;
    }
    const int nb = is_short ? T.nb_short : T.nb_long;                /* bands per (sub-)block */
    const int n_sub = is_short ? PACX_SUB : 1;
    const int n_vb = n_sub * nb;                                      /* the unit's bands (<= VQF_VB, the launcher checks) */
    const int32_t *__restrict__ lower = is_short ? T.band_lower_short : T.band_lower_long;
    const int32_t *__restrict__ count = is_short ? T.band_lines_short : T.band_lines_long;
    const int first_omit = (!is_short && T.use_sbr) ? T.first_omitted : nb;
    const long long boff = cf * T.band_stride;
    const double *__restrict__ lin = A.lines + cf * PACX_M_LONG;
    const double half_pi = 1.5707963267948966;
    /* band vb = j nb + b */
    auto sub_of = [&](int vb) { return is_short ? vb / nb : 0; };

    for (int i = tid; i < VQ_WORDS; i += 64 * VQ_WAVES)
        words[i] = 0u;
    if (tid < VQ_ROWS_SMALL && tid <= V.l_max)
        vq_row_off_small[tid] = V.row_off[tid];
    /* the scalar stage of a level is one dependency chain, and it is what the kernel waits for: the two
       tables it reads sit in LDS as far as they are commonly needed (halves below 64, angles of up to 7 bits) */
    if (tid < VQF_HL && tid <= V.l_max)
        vqf_half_log2[tid] = V.half_log2[tid];
    if (tid < VQF_LT)
        vqf_log2_tan[tid] = V.log2_tan[tid];
    if (tid < 8)
        misc[tid] = 0;                                       /* [7]: guard bits raised while the lines were placed */
    VQF_T(15);
    /* phase A: gains; the unit shapes x / gain are level 0 of the walk (bufs[0], at the band's lines) */
    double *xs = buf0;
    {
        /* the unit's lines, scaled (a short frame: each sub-block by its own overall scale), in one
           coalesced sweep: the bands' norms then read LDS, not memory */
        for (int i = 2 * tid; i < PACX_M_LONG; i += 2 * 64 * VQ_WAVES) {
            const int ov = A.overall[cf * PACX_SUB + (is_short ? i / PACX_M_SHORT : 0)];
            const double up = (double)(1 << ov);
            const double2 v = *(const double2 *)(lin + i);
            xs[i] = v.x * up;
            xs[i + 1] = v.y * up;
        }
    }
    __syncthreads();
    VQW_P(0);
    for (int vb = wave; vb < n_vb; vb += VQ_WAVES) {
        const int j = sub_of(vb), b = vb - j * nb;
        double g;
        if (b >= first_omit) {
            const double up = (double)(1 << A.overall[cf * PACX_SUB]);
            const double v = A.sbr_mean[cf * PACX_SUB + (b - first_omit)] * up;
            g = sqrt(v * v);
        } else {
            const int lo = j * PACX_M_SHORT + ldc(&lower[b]), cnt = ldc(&count[b]);
            double acc = 0.0;
            for (int i = lane; i < cnt; i += 64) {
                const double x = xs[lo + i];
                acc = fma(x, x, acc);
            }
            g = sqrt(wave_sum_f64(acc));
            if (GUARD) {
                /* a line at rounding-noise level (exactly zero in exact arithmetic: 0.0 here, 1e-21 from the reference's
                   FFT) keeps or loses its pulse by the sign of that noise (np.sign(0) = 0 erases it) */
                bool tiny = false;
                for (int i = lane; i < cnt; i += 64)
                    tiny = tiny || (g > 0.0 && fabs(xs[lo + i]) < 1e-12 * g);
                if (__builtin_amdgcn_ballot_w64(tiny) && lane == 0)
                    atomicOr((unsigned *)&misc[7], 1u);
            }
            for (int i = lane; i < cnt; i += 64)
                xs[lo + i] = xs[lo + i] / g;
        }
        if (lane == 0)
            gain_s[vb] = g;
    }
    __syncthreads();
    VQW_P(1);
    if (wave == 0) {
        /* final allocations, band positions, header fields, the roots of the trees: one band per lane.
           The stream: 3 flag bits, then per (sub-)block its head (overall scale, nb allocation fields)
           and the fields of its bands back to back (coder/pacfile.py:552-592; eight of those for a short
           frame): band vb starts at 3 + (j + 1) head + the bits of all bands before it */
        const int vb = lane, j = sub_of(vb < n_vb ? vb : 0), b = vb - j * nb;
        int ba = 0, r_bits = 0, cnt = 1;
        if (vb < n_vb) {
            ba = A.bit_alloc[boff + vb];
            if (ba && gain_s[vb] == 0.0)
                ba = 0;                                   /* coder/codec.py:352-353 */
            cnt = (b >= first_omit) ? 1 : count[b];
            r_bits = ba * cnt;
            A.bit_alloc[boff + vb] = ba;
            ba_s[vb] = ba;
        }
        int total_bits;
        const int before = wave_excl_scan_i32(r_bits, lane, total_bits);
        const int head = T.n_scale_bits + T.n_mant_size_bits * nb;
        const int start = 3 + (j + 1) * head + before;
        if (vb < n_vb) {
            start_s[vb] = start;
            end_s[vb] = start + r_bits;
        }
        if (lane == 0) {
            vq_put32(words, 0, fl & 1u, 1);
            vq_put32(words, 1, (fl >> 1) & 1u, 1);
            vq_put32(words, 2, (fl >> 2) & 1u, 1);
            misc[2] = 3 + n_sub * head + total_bits;      /* bits written */
        }
        if (vb < n_vb) {
            const int sub_base = 3 + j * head + __shfl(before, vb - b, 64);
            if (b == 0)
                vq_put32(words, sub_base, (unsigned)A.overall[cf * PACX_SUB + j], T.n_scale_bits);
            vq_put32(words, sub_base + T.n_scale_bits + T.n_mant_size_bits * b, (unsigned)(ba ? ba - 1 : 0),
                     T.n_mant_size_bits);
        }
        /* gain_shape_alloc of the band; an omitted band is one number: every bit to its gain */
        int bits_gain = ba, bits_shape = 0;
        if (vb < n_vb && b < first_omit && ba) {
            bits_gain = (int)floor((double)r_bits / (double)cnt + V.half_log2[cnt]);
            bits_shape = r_bits - bits_gain;
            if (bits_shape < 0)
                bits_shape = 0;
        }
        const bool rooted = vb < n_vb && bits_shape != 0;
        const unsigned long long mr = __builtin_amdgcn_ballot_w64(rooted);
        const int id = __popcll(mr & below);
        if (vb < n_vb) {
            bg_s[vb] = bits_gain;
            bs_s[vb] = bits_shape;
            root_s[vb] = rooted ? (unsigned short)id : 0xFFFF;
        }
        if (rooted) {
            N.nn[id] = (unsigned short)cnt;
            N.bb[id] = (unsigned short)bits_shape;
            N.off[id] = (unsigned short)(j * PACX_M_SHORT + lower[b]);
            N.kind[id] = bits_shape > PACX_VQ_SPLIT_BITS ? 0 : 1;
            N.wid[id] = 0;
            N.band[id] = (unsigned char)vb;
            N.has[id] = 0;
            N.kid[id] = 0;
            N.val[id] = 0ull;
        }
        if (lane == 0) {
            misc[0] = __popcll(mr);
            N.lvl[0] = 0;
        }
    }
    __syncthreads();

    /* ---- a level's nodes grouped by class (ord / cls of that level's parity), room for what they write to the
       next buffer.  One wave.  Level d + 1 is classified while level d's leaves are still being coded from
       level d's lists: the lists alternate between two copies */
    auto classify = [&](int lev_b, int lev_e, int par) {
        unsigned short *ord_w = N.ord + par * VQF_NLV;
        int *cls_w = N.cls + par * 16;
        int cnt_c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (lane == 0) {
            misc[3] = 0;
            misc[4] = 0;
            if (lev_e - lev_b > VQF_NLV)
                misc[1] = 1;
        }
        for (int base = lev_b; base < lev_e && lev_e - lev_b <= VQF_NLV; base += 64) {
            const int j = base + lane;
            int c = 8;
            if (j < lev_e) {
                const int n = N.nn[j], half = n - n / 2;
                if (N.kind[j] == 1)
                    c = n <= 16 ? 0 : (n <= 32 ? 1 : 2);
                else
                    c = half <= 8 ? 3 : (half <= 16 ? 4 : (half <= 32 ? 5 : (half <= 64 ? 6 : 7)));
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                cnt_c[k] += __popcll(__builtin_amdgcn_ballot_w64(c == k));
        }
        int start_c[9];
        start_c[0] = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            start_c[k + 1] = start_c[k] + cnt_c[k];
        if (lane < 9) {
            int v = 0;
#pragma unroll
            for (int k = 0; k < 9; ++k)
                v = (lane == k) ? start_c[k] : v;
            cls_w[lane] = v;
        }
        int run_c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int carry = 0;
        for (int base = lev_b; base < lev_e && lev_e - lev_b <= VQF_NLV; base += 64) {
            const int j = base + lane;
            int c = 8, need = 0;
            if (j < lev_e) {
                const int n = N.nn[j], half = n - n / 2;
                if (N.kind[j] == 1) {
                    c = n <= 16 ? 0 : (n <= 32 ? 1 : 2);
                    need = c == 2 ? 2 * n : 0;           /* scratch of a single big leaf */
                } else {
                    c = half <= 8 ? 3 : (half <= 16 ? 4 : (half <= 32 ? 5 : (half <= 64 ? 6 : 7)));
                    need = 2 * half;                    /* its children's vectors */
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned long long m = __builtin_amdgcn_ballot_w64(c == k);
                if (c == k)
                    ord_w[start_c[k] + run_c[k] + __popcll(m & below)] = (unsigned short)j;
                run_c[k] += __popcll(m);
            }
            int room;
            const int at = carry + wave_excl_scan_i32(need, lane, room);
            if (j < lev_e && need)
                N.tot[j] = (unsigned short)at;
            carry += room;
        }
        if (carry > VQF_BUF && lane == 0)
            misc[1] = 1;
    };
    VQW_P(2);
    bool guarded = false;                                  /* PACX_ST_GUARD: a decision within rounding distance of its boundary */
    VQF_T(0);
#ifdef PACX_VQ_WAITDBG
    long long vqw_busy1 = 0, vqw_wait1 = 0, vqw_busy2 = 0, vqw_wait2 = 0, vqw_t, vqw_t0, vqw_levels = 0;
    VQW_NOW(vqw_t);
    vqw_t0 = vqw_t;
#endif
    bool undefined = false;
    int lev_b = 0, lev_e = misc[0], depth = 0;
    if (wave == 0)
        classify(0, lev_e, 0);
    __syncthreads();
    for (;;) {
        const double *cur = buf0 + (depth & 1) * VQF_BUF;
        double *nxt = buf0 + ((depth + 1) & 1) * VQF_BUF;
        const unsigned short *ord_c = N.ord + (depth & 1) * VQF_NLV;
        const int *cls_c = N.cls + (depth & 1) * 16;
        VQF_T(1);
        if (misc[1])
            break;
        /* ---- one pass: `p_n` nodes of class c, ord[p0 ..) */
        auto do_item = [&](int c, int p0, int p_n) {
#ifdef PACX_VQ_DEBUG
            long long t_in;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_in) :: "memory");
#endif
            if (c <= 1) {
                /* leaves, four or two per pass */
                const int lw = (c == 0) ? 4 : 5, W = 1 << lw;
                const int g = lane >> lw, l = lane & (W - 1);
                const bool valid = g < p_n;
                const int node = valid ? ord_c[p0 + g] : 0;
                const int n = valid ? N.nn[node] : 0;
                /* pulse count and index width: looked up here, beside the scalar stage, not in it */
                int bits = valid ? N.bb[node] : 0;
                bits = bits > 32 ? 32 : bits;
#ifdef VQF_STUB_KOF        /* timing experiment only (wrong pulse counts): what the dependent table read in front of a leaf pass costs */
                const int K = valid ? (bits * 3) / (n > 8 ? 2 : 1) + 1 : 0;
                const int width = valid ? bits : 0;
#else
                const int K = valid ? V.k_of[n * 33 + bits] : 0;
                const int width = valid ? V.w_of[n * 33 + bits] : 0;
#endif
                const double x = (valid && l < n) ? cur[N.off[node] + l] : 0.0;
                bool ok = false, near = false;
                unsigned long long term;
#ifdef VQF_STUB_LEAF       /* timing experiment only (wrong indices): what the small leaves cost */
                ok = true;
                term = (unsigned long long)(x != 0.0) + (unsigned long long)K;
#else
                if (GUARD) {                       /* the margins are taken by a copy of the leaf code of its own */
                    if (c == 0)
                        term = vq_leaf_group<16, true>(V, x, n, K < 0 ? 0 : K, l, ok, true, &near);
                    else
                        term = vq_leaf_group<32, true>(V, x, n, K < 0 ? 0 : K, l, ok, true, &near);
                    guarded = guarded || (valid && near);
                } else if (c == 0) {
                    term = vq_leaf_group<16, true>(V, x, n, K < 0 ? 0 : K, l, ok);
                } else {
                    term = vq_leaf_group<32, true>(V, x, n, K < 0 ? 0 : K, l, ok);
                }
#endif
                if (valid && l == 0) {
                    if (K < 0) {                   /* a 1-dimensional leaf: the reference never returns */
                        N.kind[node] = 3;
                        N.wid[node] = 0;
                        N.val[node] = 0ull;
                    } else {
                        N.val[node] = ok ? term : 0ull;
                        N.wid[node] = (unsigned char)width;
                    }
                }
                if (valid && (K < 0 || !ok))
                    undefined = true;
            } else if (c == 2) {
                const int node = ord_c[p0];
                const int n = N.nn[node];
                int bits = N.bb[node];
                bits = bits > 32 ? 32 : bits;
                const int K = ldc(&V.k_of[n * 33 + bits]);
                const int width = ldc(&V.w_of[n * 33 + bits]);
                bool ok = true;
                double *t1 = nxt + N.tot[node];
                const unsigned long long idx = vq_leaf_idx<true>(V, cur + N.off[node], n, K, t1, t1 + n, lane, ok);
                if (!ok)
                    undefined = true;
                guarded = guarded || GUARD;         /* whole-wave leaves: no margin is taken, flagged as they come */
                if (lane == 0) {
                    N.val[node] = ok ? idx : 0ull;
                    N.wid[node] = (unsigned char)width;
                }
            } else if (c < 7) {
                /* splits packed: one node per aligned block of P lanes */
                const int lp = c + 0;              /* classes 3..6 = blocks of 8, 16, 32, 64 lanes */
                const int P = 1 << lp;
                const int g = lane >> lp, i = lane & (P - 1);
                const bool valid = g < p_n;
                const int node = valid ? ord_c[p0 + g] : 0;
                const int n = valid ? N.nn[node] : 0;
                const int cut = n / 2, half = n - cut;
                const double *src = cur + (valid ? N.off[node] : 0);
                const bool mine = i < half;
                const double left = (mine && i < cut) ? src[i] : 0.0;
                const double right = mine ? src[cut + i] : 0.0;
                double mid = mine ? (left + right) / 2.0 : 0.0;
                double sd = mine ? (left - right) / 2.0 : 0.0;
                double mm = fma(mid, mid, 0.0), ss = fma(sd, sd, 0.0);
                for (int off = P >> 1; off > 0; off >>= 1) {
                    mm = mm + __shfl_xor(mm, off, 64);
                    ss = ss + __shfl_xor(ss, off, 64);
                }
                const double m_l2 = sqrt(mm), s_l2 = sqrt(ss);
                if (m_l2 != 0.0)
                    mid = mid / m_l2;
                if (s_l2 != 0.0)
                    sd = sd / s_l2;
                if (mine) {
                    double *mv = nxt + N.tot[node];
                    mv[i] = mid;
                    mv[half + i] = sd;
                }
                if (valid && i == 0)
                    N.val[node] = (unsigned long long)__double_as_longlong(m_l2 == 0.0 ? -1.0 : s_l2 / m_l2);
            } else {
                /* a split of more than 128 components: lanes strided over the half */
                const int node = ord_c[p0];
                const int n = N.nn[node];
                const double *src = cur + N.off[node];
                const int cut = n / 2, half = n - cut;
                double *mv = nxt + N.tot[node], *sv = mv + half;
                double mm = 0.0, ss = 0.0;
                for (int i = lane; i < half; i += 64) {
                    const double left = (i < cut) ? src[i] : 0.0;
                    const double right = src[cut + i];
                    const double mid = (left + right) / 2.0;
                    const double sd = (left - right) / 2.0;
                    mv[i] = mid;
                    sv[i] = sd;
                    mm = fma(mid, mid, mm);
                    ss = fma(sd, sd, ss);
                }
                const double m_l2 = sqrt(wave_sum_f64(mm));
                const double s_l2 = sqrt(wave_sum_f64(ss));
                for (int i = lane; i < half; i += 64) {
                    if (m_l2 != 0.0)
                        mv[i] = mv[i] / m_l2;
                    if (s_l2 != 0.0)
                        sv[i] = sv[i] / s_l2;
                }
                if (lane == 0)
                    N.val[node] = (unsigned long long)__double_as_longlong(m_l2 == 0.0 ? -1.0 : s_l2 / m_l2);
            }
#ifdef PACX_VQ_DEBUG
            {
                long long t_out;
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_out) :: "memory");
                if (lane == 0) {
                    atomicAdd((unsigned long long *)&g_vq_dbg[8 + c], (unsigned long long)(t_out - t_in));
                    atomicAdd((unsigned long long *)&g_vq_dbg[16 + c], 1ull);
                }
            }
#endif
        };
        /* passes are dealt round-robin, the expensive classes first (a wave's last pass is a cheap one).
           First the splits on all four waves; the leaves of the level do not depend on its scalar stage,
           so they run NEXT to it: the waves without a share of the scalar stage take them */
        /* pass t of a stage goes to wave (pattern >> 2 (t mod period)) & 3 */
        auto deal = [&](const int *order, int n_classes, unsigned pattern, int period) {
            int ph = 0;
#pragma unroll 1
            for (int k = 0; k < n_classes; ++k) {
                const int c = order[k];
                const int c_b = cls_c[c], c_n = cls_c[c + 1] - c_b;
                const int lg = (c == 0) ? 2 : (c == 1) ? 1 : (c == 2) ? 0 : (c == 3) ? 3 : (c == 4) ? 2 : (c == 5) ? 1 : 0;
                for (int p0 = c_b; p0 < c_b + c_n; p0 += 1 << lg) {
                    if ((int)((pattern >> (2 * ph)) & 3u) == wave)
                        do_item(c, p0, min(1 << lg, c_b + c_n - p0));
                    ph = (ph + 1 == period) ? 0 : ph + 1;
                }
            }
        };
        const unsigned EVEN = 0xE4u;                       /* 0, 1, 2, 3 */
        {
            const int split_order[5] = {7, 6, 5, 4, 3};
            deal(split_order, 5, EVEN, 4);
        }
        VQW_BARRIER(vqw_busy1, vqw_wait1);
        VQF_T(2);
        /* ---- the level's splits, one per lane: angle, bit split, children */
        const int s_b = cls_c[3], s_n = cls_c[8] - s_b;
        const bool early = s_n <= 64;
        for (int k0 = 64 * wave; k0 < s_n; k0 += 64 * VQ_WAVES) {
#ifdef PACX_VQ_DEBUG
            long long t_sc;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_sc) :: "memory");
#endif
#ifdef PACX_VQ_DEBUG
            long long t_sub = t_sc;
#endif
            const bool has = k0 + lane < s_n;
            const int snode = has ? ord_c[s_b + k0 + lane] : 0;
            const int sn = has ? N.nn[snode] : 2;
            const int half = sn - sn / 2;
            const int bits = has ? N.bb[snode] : 0;
            const double q = has ? __longlong_as_double((long long)N.val[snode]) : -1.0;
#ifdef VQF_STUB_ATAN       /* timing experiment only (wrong angles) */
            const double theta = (q < 0.0) ? 0.0 : q / (1.0 + q);
#else
            const double theta = (q < 0.0) ? 0.0 : vq_atan(q);
#endif
            VQF_SUB(26);
            const double hl = (half < VQF_HL) ? vqf_half_log2[half] : V.half_log2[half];
            const int a_theta = (int)floor((double)bits / (double)half + hl);
            int a_rest = bits - a_theta;
            if (a_rest < 0)
                a_rest = 0;
            const double tn = theta / half_pi;
            double theta_q = 0.0;
            unsigned long long code = 0ull;
            int w_theta = 0;
            if (!__builtin_amdgcn_ballot_w64(a_theta > 31)) {
                /* every angle of this chunk has at most 31 bits (all but freak allocations): the same values
                   in 32-bit integers -- one conversion instruction where the 64-bit ones take a dozen */
                if (a_theta > 0) {
                    const double factor = (double)((1u << a_theta) - 1u);
                    unsigned c32;
                    if (tn >= 1.0)
                        c32 = (1u << (a_theta - 1)) - 1u;
                    else
                        c32 = (unsigned)floor((factor * tn + 1.0) * 0.5);
                    code = c32;
                    w_theta = a_theta;
                    const unsigned mag = c32 & ((1u << (a_theta - 1)) - 1u);
                    double dq = (double)(2u * mag) / factor;
                    if (c32 >> (a_theta - 1))
                        dq = -dq;
                    theta_q = dq * half_pi;
                }
            } else if (a_theta > 62) {
                if (has)
                    undefined = true;
            } else if (a_theta > 0) {
                if (tn >= 1.0) {
                    code = (1ull << (a_theta - 1)) - 1ull;
                } else {
                    const double factor = (a_theta <= 53) ? (double)((1ull << a_theta) - 1ull) : ldexp(1.0, a_theta);
                    code = (unsigned long long)floor((factor * tn + 1.0) * 0.5);
                }
                w_theta = a_theta;
                const unsigned long long mag = code & ((1ull << (a_theta - 1)) - 1ull);
                const double den = (a_theta <= 53) ? (double)((1ull << a_theta) - 1ull) : ldexp(1.0, a_theta);
                double dq = (double)(2ull * mag) / den;
                if (code >> (a_theta - 1))
                    dq = -dq;
                theta_q = dq * half_pi;
            }
            if (GUARD && has && a_theta > 0 && a_theta <= 53 && pacx_quant_guard(tn, a_theta, 1e-13))
                guarded = true;                              /* the split angle sits at a boundary of its quantiser */
            VQF_SUB(27);
            int a_mid = 0;
            if (theta_q != 0.0) {
                double lt;
                if (a_theta <= PACX_VQ_THETA_TABLE_BITS && theta_q > 0.0) {
                    const int at_lt = ((1 << (a_theta - 1)) - 1) + (int)code;
                    lt = (at_lt < VQF_LT) ? vqf_log2_tan[at_lt] : V.log2_tan[at_lt];
                }
                else
                    lt = vq_log2_tan(theta_q);
                const double v = ((double)a_rest - (double)(half - 1) * lt) / 2.0;
                const double f = floor(v);
                a_mid = (f < 0.0) ? 0 : ((f > (double)a_rest) ? a_rest : (int)f);
            }
            VQF_SUB(28);
            const int a_side = a_rest - a_mid;
            const int c_mid = (has && a_mid > 0) ? 1 : 0, c_side = (has && a_side > 0) ? 1 : 0;
            int born;
            const int rel = wave_excl_scan_i32(c_mid + c_side, lane, born);
            int base = 0;
            if (lane == 0 && born)
                base = atomicAdd(&misc[0], born);
            base = __shfl(base, 0, 64);
            if (base + born > VQF_NCAP) {
                if (lane == 0)
                    misc[1] = 1;
                continue;
            }
            VQF_SUB(29);
            if (has) {
                const int first = base + rel;
                N.val[snode] = code;
                N.wid[snode] = (unsigned char)w_theta;
                N.kid[snode] = (unsigned short)first;
                N.has[snode] = (unsigned char)(c_mid | (c_side << 1));
                const int at = N.tot[snode];
                const int bd = N.band[snode];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int a = c ? a_side : a_mid;
                    if (a <= 0)
                        continue;
                    const int id = c ? first + c_mid : first;
                    const bool splits = a > PACX_VQ_SPLIT_BITS && depth + 1 < VQ_DEPTH;
                    if (a > PACX_VQ_SPLIT_BITS && !splits)
                        undefined = true;                  /* deeper than any real tree */
                    N.nn[id] = (unsigned short)half;
                    N.bb[id] = (unsigned short)(a > 65535 ? 65535 : a);
                    N.off[id] = (unsigned short)(at + (c ? half : 0));
                    N.kind[id] = splits ? 0 : 1;
                    N.wid[id] = 0;
                    N.band[id] = (unsigned char)bd;
                    N.has[id] = 0;
                    N.kid[id] = 0;
                    N.val[id] = 0ull;
                }
            }
#ifdef PACX_VQ_DEBUG
            {
                long long t2_;
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2_) :: "memory");
                if (lane == 0) {
                    atomicAdd((unsigned long long *)&g_vq_dbg[24], (unsigned long long)(t2_ - t_sc));
                    atomicAdd((unsigned long long *)&g_vq_dbg[25], 1ull);
                }
            }
#endif
        }
        {
            /* the leaves of the level, beside its scalar stage: a wave with a share of that stage (64 splits
               each) takes a seventh of what the others take */
            const int busy = (s_n + 63) >> 6;
            const int leaf_order[3] = {2, 1, 0};
            /* owners of passes 0.. : busy = 1: 1 2 3 (wave 0 sorts the next level);  2: 2 3 2 3 2 3 0 1;  3: 3 0 3 1 3 2 3 */
            const unsigned pat = busy == 1 ? 0x39u : busy == 2 ? 0x4EEEu : busy == 3 ? 0x3B73u : EVEN;
            const int per = busy == 1 ? 3 : busy == 2 ? 8 : busy == 3 ? 7 : 4;
            /* one chunk of splits: wave 0 made all the children itself and sorts the next level at once */
            if (early && wave == 0) {
                vq_fence();
                if (!misc[1])
                    classify(lev_e, misc[0], (depth + 1) & 1);
            }
            deal(leaf_order, 3, pat, per);
        }
        VQW_BARRIER(vqw_busy2, vqw_wait2);
#ifdef PACX_VQ_WAITDBG
        vqw_levels += 1;
#endif
        VQF_T(3);
        if (misc[1])
            break;
        const int n_nodes = misc[0];
        depth += 1;
        if (tid == 0)
            N.lvl[depth] = (unsigned short)lev_e;
        if (n_nodes == lev_e) {                              /* no children: the walk is over */
            depth -= 1;
            break;
        }
        lev_b = lev_e;
        lev_e = n_nodes;
        if (!early) {
            if (wave == 0)
                classify(lev_b, lev_e, depth & 1);
            __syncthreads();
        }
        if (depth > VQ_DEPTH) {                            /* cannot happen: splits stop at VQ_DEPTH - 1 */
            if (tid == 0)
                misc[1] = 1;
            __syncthreads();
            break;
        }
    }
    if (misc[1]) {
        /* does not fit the store: k_vq (and, for a short frame, k_vq_join) code this channel-frame */
        if (tid == 0)
            A.n_bytes[cf] = -1;
        return;
    }
    const int n_nodes = misc[0];
    if (tid == 0)
        N.lvl[depth + 1] = (unsigned short)n_nodes;
#ifdef PACX_VQ_WAITDBG
    {
        long long t_end;
        VQW_NOW(t_end);
        if (lane == 0) {
            atomicAdd((unsigned long long *)&g_vqw_dbg[wave * 4 + 0], (unsigned long long)vqw_busy1);
            atomicAdd((unsigned long long *)&g_vqw_dbg[wave * 4 + 1], (unsigned long long)vqw_wait1);
            atomicAdd((unsigned long long *)&g_vqw_dbg[wave * 4 + 2], (unsigned long long)vqw_busy2);
            atomicAdd((unsigned long long *)&g_vqw_dbg[wave * 4 + 3], (unsigned long long)vqw_wait2);
            if (hw_wave == 0) {
                atomicAdd((unsigned long long *)&g_vqw_dbg[16], (unsigned long long)(t_end - vqw_t0));   /* level loop */
                atomicAdd((unsigned long long *)&g_vqw_dbg[17], (unsigned long long)vqw_levels);
                atomicAdd((unsigned long long *)&g_vqw_dbg[18], 1ull);
                atomicAdd((unsigned long long *)&g_vqw_dbg[19], (unsigned long long)(vqw_t0 - vqw_start));   /* phase A */
            }
        }
    }
#endif
    __syncthreads();
    /* ---- subtree widths and field counts bottom-up (counts ride in nn[], field numbers in off[]) */
    for (int d = depth; d >= 0; --d) {
        const int b = N.lvl[d], e = N.lvl[d + 1];
        for (int j = b + tid; j < e; j += 64 * VQ_WAVES) {
            const int k0 = N.kid[j], hs = N.has[j];
            int t = N.wid[j], f = (N.kind[j] == 3) ? 0 : 1;
            if (hs & 1) { t += N.tot[k0]; f += N.nn[k0]; }
            if (hs & 2) { const int k1 = k0 + (hs & 1); t += N.tot[k1]; f += N.nn[k1]; }
            N.tot[j] = (unsigned short)t;
            N.nn[j] = (unsigned short)f;
        }
        __syncthreads();
    }
    if (hw_wave == 0 && lane < n_vb && root_s[lane] != 0xFFFF) {
        N.pos[root_s[lane]] = (unsigned short)start_s[lane];
        N.off[root_s[lane]] = 0;
    }
    __syncthreads();
    for (int d = 0; d <= depth; ++d) {
        const int b = N.lvl[d], e = N.lvl[d + 1];
        for (int j = b + tid; j < e; j += 64 * VQ_WAVES) {
            const int k0 = N.kid[j], hs = N.has[j];
            int p = N.pos[j] + N.wid[j], r = N.off[j] + ((N.kind[j] == 3) ? 0 : 1);
            if (hs & 1) {
                N.pos[k0] = (unsigned short)p;
                N.off[k0] = (unsigned short)r;
                p += N.tot[k0];
                r += N.nn[k0];
            }
            if (hs & 2) {
                const int k1 = k0 + (hs & 1);
                N.pos[k1] = (unsigned short)p;
                N.off[k1] = (unsigned short)r;
            }
        }
        __syncthreads();
    }
    VQF_T(4);
#ifdef PACX_VQ_WAITDBG
    VQW_NOW(vqw_p);
#endif
    /* ---- every field by its own lane */
    for (int j = tid; j < n_nodes; j += 64 * VQ_WAVES) {
        const int w = N.wid[j];
        unsigned long long v = N.val[j];
        if (w > 0 && w < 64)
            v &= (1ull << w) - 1ull;
        if (w > 0)
            vq_put_field(words, N.pos[j], v, w);
        if (A.log && N.kind[j] != 3 && N.off[j] < A.log_cap) {
            const int vb = N.band[j], sj = sub_of(vb);
            const long long slot = (cf * PACX_SUB + sj) * PACX_MAX_BANDS + (vb - sj * nb);
            pacx_vq_entry e;
            e.value = v;
            e.width = w;
            e.band = vb - sj * nb;
            A.log[slot * A.log_cap + N.off[j]] = e;
        }
    }
    VQW_P(3);
    /* ---- the gains of all bands side by side (mu-law, QuantizeUniform; the index soaks up the slack) */
    if (wave == 0) {
        const int vb = lane, sj = sub_of(vb < n_vb ? vb : 0), b = vb - sj * nb;
        const int ba = (vb < n_vb) ? ba_s[vb] : 0;
        const long long slot = (cf * PACX_SUB + sj) * PACX_MAX_BANDS + b;
        if (vb < n_vb && !ba && A.log_count)
            A.log_count[slot] = 0;
        const int cnt = (vb < n_vb && b < first_omit) ? count[b] : 1;
        const double gain = (vb < n_vb) ? gain_s[vb] : 0.0;
        const double g = vq_log(1.0 + 255.0 * fabs(gain / (double)cnt)) / V.log_mu1;
        if (ba) {
            const int rt = root_s[vb];
            const int used = (rt != 0xFFFF) ? N.tot[rt] : 0;
            const int fields = (rt != 0xFFFF) ? N.nn[rt] : 0;
            int bits_gain = bg_s[vb] + bs_s[vb] - used;
            if (bits_gain < 0)
                bits_gain = 0;
            int width = bits_gain;
            unsigned long long hi = 0, lo = 0;
            if (bits_gain > 128) {                         /* beyond what the entry format carries */
                undefined = true;
                width = 0;
            } else if (bits_gain > 0) {
                vq_quantize_code(g, bits_gain, hi, lo);
                if (GUARD && bits_gain <= 53 && pacx_quant_guard(g, bits_gain, 1e-13))
                    guarded = true;                          /* the band's mu-law gain sits at a boundary of its index */
            }
            const int at = start_s[vb] + used;
            if (width > 64) {
                vq_put_field(words, at, hi, width - 64);
                vq_put_field(words, at + width - 64, lo, 64);
            } else if (width > 0) {
                vq_put_field(words, at, lo, width);
            }
            if (at + width != end_s[vb])
                undefined = true;                          /* a band must fill its slot exactly */
            if (A.log && fields < A.log_cap) {
                pacx_vq_entry e;
                e.value = lo;
                e.width = width;
                e.band = b;
                A.log[slot * A.log_cap + fields] = e;
            }
            if (A.log_count)
                A.log_count[slot] = fields + 1;
        }
    }
    if (__builtin_amdgcn_ballot_w64(undefined) && lane == 0 && A.status)
        atomicOr(&A.status[cf], PACX_ST_VQ_UNDEFINED);
    if (GUARD && A.status) {
        const unsigned long long gm = __builtin_amdgcn_ballot_w64(guarded || (tid == 0 && misc[7] != 0));
        if (gm && lane == 0)
            atomicOr(&A.status[cf], PACX_ST_GUARD);
    }
    __syncthreads();
    VQW_P(4);

    VQF_T(5);
    /* hand the string over.  getNumBytesNeeded (coder/pacfile.py:342-361) also charges a scale factor per band
       that the gain-shape writer does not send */
    const int written = misc[2];                           /* with the 3 flag bits */
    const int size_rule = written - 3 + n_sub * nb * T.n_scale_bits;
    const int nbytes = (size_rule + 4 + 7) >> 3;
    unsigned *dst = (unsigned *)(A.payload + cf * (long long)A.payload_stride);
    for (int i = tid; i < (nbytes + 3) / 4; i += 64 * VQ_WAVES)
        dst[i] = __builtin_bswap32(words[i]);
    if (tid == 0)
        A.n_bytes[cf] = nbytes;
#ifdef PACX_VQ_WAITDBG
    {
        long long t_fin;
        VQW_NOW(t_fin);
        if (tid == 0)
            atomicAdd((unsigned long long *)&g_vqw_dbg[20], (unsigned long long)(t_fin - vqw_start));         /* whole unit */
    }
#endif
    VQF_T(6);
}

/* ------------------------------------------------ frame-level walk, second form (k_vq_frame2) */
/* Round 3.  k_vq_frame, measured with stamps next to its barriers (tools/vq_wait_probe.py,
 * profiles/r03_vq_wait_before.txt): a unit takes ~250 k cycles for ~8 k instructions per wave; the level loop is
 * 70 % of it, and inside a level the wave that codes the most LEAVES is the critical path, not the scalar stage
 * (leaf passes: 108 k cycles on the busiest wave against 83 k for the wave with the scalar stage and the sorting);
 * the small leaves are 45 % of the kernel's vector instructions (15 packed passes of four leaves per unit, each a
 * chain of cross-lane reads with run-time trip counts).  What changes here, bytes unchanged:
 *   - ONE buffer, in place: a node of n components owns a region of R = pow2ceil(n) doubles (a band's root the
 *     region of its band), its mid child the first half of it, its side child the second (ceil(n/2) <= R/2) -- a
 *     pass reads its node into registers before it writes, nobody else touches the region.  No room to allocate per
 *     level, and every leaf's vector stays where it is until the end of the walk;
 *   - so the level loop holds only the splits (packed passes, scalar stage), and the leaves are coded AFTER it, all
 *     of the unit's at once: ONE LEAF PER LANE for the ~60 leaves of up to 16 components (L1 norm in np.sum order,
 *     targets, the pulse ranking as 240 register compares, prefix of the pulses -- no cross-lane traffic at all),
 *     the components' (pulses, pulses left) left in place of the vector; then the enumeration terms ONE COMPONENT
 *     PER THREAD on all four waves (four independent table reads each, one round trip for the whole unit) added
 *     into the leaf's index with a 64-bit LDS atomic (integers: any order).  Leaves of 17..32 components keep the
 *     half-wave group coder, larger ones the whole-wave coder (one scratch: the first wave takes them);
 *   - node records are two words (lengths / offset / child as one 64-bit store, kind / width / band / flags as one
 *     32-bit store) instead of ten byte and short stores per child;
 *   - the bands' lines go from memory straight to their regions and the wave that placed a band also takes its
 *     norm: one barrier fewer in front of the walk.
 * Arithmetic per node is k_vq_frame's, so the bytes are (tests/test_gpu_vq.py, the path-switch matrix of
 * tests/test_gpu_round3.py: PACX_VQ_FRAME=1 selects the first form). */
#define VQ2_NCAP 272
#define VQ2_NLV 256
#define VQ2_BUF 1504                   /* doubles: sum of the bands' regions (1504 at 48 kHz, 1472 at 44.1 kHz, 1408 short);
                                          other layouts run on k_vq_frame.  With the node store and the scratch this is the
                                          LDS of five workgroups per CU to the last kilobyte */
#define VQ2_SCR 752                    /* doubles: t / y of one leaf of more than 32 components (2 n); before that, as
                                          VQ2_BUF ints, the floors of the small leaves' components (by buffer position) */
#define VQ2_VB 64
#define VQ2_FIXED 4160
#define VQ2_NODE_BYTES (VQ2_NCAP * (8 + 8 + 2 + 2 + 4 + 1) + 2 * VQ2_NLV * 2 + 128 + 64)
#define VQ2_SMEM (VQ2_FIXED + VQ2_BUF * 8 + VQ2_SCR * 8 + ((VQ2_NODE_BYTES + 15) & ~15))

struct Vq2Store {
    unsigned long long *val;
    unsigned short *r16;               /* [NCAP][4]: nn, bb, off, kid */
    unsigned short *tot, *pos;
    unsigned char *r8;                 /* [NCAP][4]: kind, wid, band, has */
    unsigned char *rl;                 /* log2 of the node's region | rotation << 4: component i sits in slot (i + rotation)
                                          & (region - 1) -- leaves are coded one per LANE, and regions are aligned to their
                                          (power of two) size: without the rotation every lane would hit the same LDS bank */
    unsigned short *ord;               /* [2][NLV] a level's splits grouped by class; after the walk: the leaf lists */
    int *cls;                          /* [2][16] class starts ([6] used) */
    unsigned short *lvl;
    __device__ __forceinline__ void bind(unsigned char *p)
    {
        val = (unsigned long long *)p;
        r16 = (unsigned short *)(p + VQ2_NCAP * 8);
        tot = r16 + 4 * VQ2_NCAP;
        pos = tot + VQ2_NCAP;
        r8 = (unsigned char *)(pos + VQ2_NCAP);
        rl = r8 + 4 * VQ2_NCAP;
        ord = (unsigned short *)(rl + VQ2_NCAP);
        cls = (int *)(ord + 2 * VQ2_NLV);
        lvl = (unsigned short *)(cls + 32);
    }
    __device__ __forceinline__ unsigned short &nn(int j) const { return r16[4 * j]; }
    __device__ __forceinline__ unsigned short &bb(int j) const { return r16[4 * j + 1]; }
    __device__ __forceinline__ unsigned short &off(int j) const { return r16[4 * j + 2]; }
    __device__ __forceinline__ unsigned short &kid(int j) const { return r16[4 * j + 3]; }
    __device__ __forceinline__ unsigned char &kind(int j) const { return r8[4 * j]; }
    __device__ __forceinline__ unsigned char &wid(int j) const { return r8[4 * j + 1]; }
    __device__ __forceinline__ unsigned char &band(int j) const { return r8[4 * j + 2]; }
    __device__ __forceinline__ unsigned char &has(int j) const { return r8[4 * j + 3]; }
    __device__ __forceinline__ int rlog(int j) const { return rl[j] & 15; }
    __device__ __forceinline__ int rot(int j) const { return rl[j] >> 4; }
    /* rotation of child c of node `parent`: any function of things both the split pass (which writes the child's
       vector) and the scalar stage (which creates its record) know; none for vectors the whole-wave coders take */
    static __device__ __forceinline__ int child_rot(int parent, int c, int half, int rlog_c)
    {
        return (half <= 32) ? ((2 * parent + c) & ((1 << rlog_c) - 1) & 15) : 0;
    }
    __device__ __forceinline__ void create(int id, int n, int bits, int offset, int knd, int bnd, int rlog_rot) const
    {
        *(unsigned long long *)&r16[4 * id] = (unsigned long long)(unsigned)n | ((unsigned long long)(unsigned)bits << 16) |
                                              ((unsigned long long)(unsigned)offset << 32);
        *(unsigned *)&r8[4 * id] = (unsigned)knd | ((unsigned)bnd << 16);
        rl[id] = (unsigned char)rlog_rot;
        val[id] = 0ull;
    }
};
static_assert(VQ2_NCAP % 4 == 0 && (VQ2_NCAP * 8) % 8 == 0, "node store alignment");
static_assert(VQ2_SCR * 8 >= VQ2_BUF * 4, "one int per buffer position in the scratch area");

/* One PVQ leaf of up to 16 components per LANE: everything pvq_search (coder/gain_shape_quantize.py:30-54) does
   up to the pulse vector, with the arithmetic and the summation orders of vq_leaf_group -- the L1 norm as
   np.sum adds it, target = |K x / l1|, floor, the `missing` pulses to the largest remainders (lowest index first
   among equals).  Leaves (pulses | sign << 31, pulses left before the component) in place of the vector: the
   enumeration terms are taken one component per thread afterwards.  Returns false for an all-zero vector.
   xs / ys are LDS pointers by type, so that every access is a ds_ instruction. */
typedef __attribute__((address_space(3))) double vq_lds_f64;
typedef __attribute__((address_space(3))) int vq_lds_i32;
__device__ __forceinline__ bool vq_leaf_lane(vq_lds_f64 *xs, int n, int K, vq_lds_i32 *ys, int n_max, int rot, int rmask)
{
    /* n_max: the longest leaf of the wave (wave-uniform loop bounds; shorter leaves ride along masked).  Nothing is
       held in register arrays but the eight accumulators of the norm: the remainders replace the vector in LDS,
       the floors sit in ys (one int per buffer position).
       Component i sits in slot (i + rot) & rmask of the leaf's region (its floor likewise in ys).  Every LDS READ
       is unconditional -- a slot index is always inside the leaf's own region -- and the value is selected
       afterwards: a predicated read would be a branch with a wait of its own, a hundred of them in a row */
#define VQ2_SLOT(i) (((i) + rot) & rmask)
    unsigned negm = 0u, nzm = 0u;
    double l1;
    {
        double seq = -0.0;                               /* fewer than 8: left to right from -0.0 */
        double r[8];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double raw = xs[VQ2_SLOT(i)];
            const double x = (i < n) ? raw : 0.0;
            const double ax = fabs(x);
            negm |= (x < 0.0) ? (1u << i) : 0u;
            nzm |= (x != 0.0) ? (1u << i) : 0u;
            if (i < 7)
                seq = (i < n) ? seq + ax : seq;
            if (i < 8)
                r[i] = ax;
            else
                r[i - 8] = (n >= 16) ? r[i - 8] + ax : r[i - 8];   /* eight accumulators over the multiple of 8 */
        }
        double tree = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
        for (int i = 8; i < 15; ++i) {
            const double raw = xs[VQ2_SLOT(i)];
            tree = (n < 16 && i < n) ? tree + fabs(raw) : tree;    /* scalar tail */
        }
        l1 = (n < 8) ? seq : tree;
    }
    const bool ok = l1 > 0.0;
    const double kd = (double)K;
    double ysum = 0.0;
#pragma unroll 8
    for (int i = 0; i < n_max; ++i) {
        const double raw = xs[VQ2_SLOT(i)];
        const double ax = (i < n) ? fabs(raw) : 0.0;
        const double tg = ok ? (kd * ax) / l1 : 0.0;     /* |K x / l1|: the sign does not change the magnitude */
        const double fl = floor(tg);
        ysum += (i < n) ? fl : 0.0;                      /* integers: exact in any order */
        if (i < n) {
            ys[VQ2_SLOT(i)] = (int)fl;
            xs[VQ2_SLOT(i)] = tg - fl;                   /* the remainder takes the component's place */
        }
    }
    const int missing = K - (int)ysum;
    if (__builtin_amdgcn_ballot_w64(missing > 0)) {
        /* the `missing` largest remainders get a pulse, the lowest index first among equals: rank of component i =
           how many beat it.  The remainders come from LDS four at a time (16 + 64 reads per leaf instead of 16 x 16) */
        unsigned rank_lo = 0u, rank_hi = 0u;             /* sixteen 4-bit counters (a rank is at most 15) */
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (4 * c < n_max) {                        /* wave-uniform */
                double tj[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double raw = xs[VQ2_SLOT(4 * c + j)];
                    tj[j] = (4 * c + j < n) ? raw : -1.0;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (i < n_max) {
                        const double raw = xs[VQ2_SLOT(i)];
                        const double ti = (i < n) ? raw : 2.0;
                        unsigned beat = 0u;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (4 * c + j < i)
                                beat += (tj[j] >= ti) ? 1u : 0u;
                            else if (4 * c + j > i)
                                beat += (tj[j] > ti) ? 1u : 0u;
                        }
                        if (i < 8)
                            rank_lo += beat << (4 * i);
                        else
                            rank_hi += beat << (4 * (i - 8));
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rank = (int)(((i < 8 ? rank_lo : rank_hi) >> (4 * (i & 7))) & 15u);
            if (i < n && rank < missing)
                ys[VQ2_SLOT(i)] += 1;
        }
    }
    int k_left = K;
#pragma unroll 8
    for (int i = 0; i < n_max; ++i) {
        const int yraw = ys[VQ2_SLOT(i)];
        int a = (i < n) ? yraw : 0;
        a = (((nzm >> i) & 1u) && ok) ? a : 0;
        if (i < n)                                       /* (pulses | sign << 31, pulses left) as the slot's 64 bits */
            xs[VQ2_SLOT(i)] = __longlong_as_double((long long)((unsigned long long)((unsigned)a | (((negm >> i) & 1u) << 31)) |
                                                               ((unsigned long long)(unsigned)k_left << 32)));
        k_left -= a;
    }
    return ok;
#undef VQ2_SLOT
}

#ifndef VQ2_OCC
#define VQ2_OCC 4                      /* registers for four workgroups per CU: at five the one-leaf-per-lane pass spills */
#endif
__global__ __launch_bounds__(64 * VQ_WAVES, VQ2_OCC) void k_vq_frame2(PacxTables T, VqView V, VqArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *words = (unsigned *)smem;                               /* VQ_WORDS        */
    double *gain_s = (double *)(smem + VQ_WORDS * 4);                 /* VB              */
    int *ba_s = (int *)(gain_s + VQ2_VB);
    int *start_s = ba_s + VQ2_VB;
    int *end_s = start_s + VQ2_VB;
    int *misc = end_s + VQ2_VB;                                       /* node count, overflow, total bits, -, -, small / mid / big leaf counts */
    int *bg_s = misc + 8;
    int *bs_s = bg_s + VQ2_VB;
    unsigned short *root_s = (unsigned short *)(bs_s + VQ2_VB);
    double *buf = (double *)(smem + VQ2_FIXED);                       /* the bands' regions */
    double *scr = buf + VQ2_BUF;                                      /* scratch of one big leaf */
    Vq2Store N;
    N.bind(smem + VQ2_FIXED + (VQ2_BUF + VQ2_SCR) * 8);
    static_assert(VQ_WORDS * 4 + VQ2_VB * 8 + 5 * VQ2_VB * 4 + 8 * 4 + VQ2_VB * 2 <= VQ2_FIXED && VQ2_FIXED % 16 == 0,
                  "fixed part of k_vq_frame2's LDS");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PACX_VQ_WAITDBG
    long long vqw_start, vqw_p;
    VQW_NOW(vqw_start);
    vqw_p = vqw_start;
    long long vqw_busy1 = 0, vqw_wait1 = 0, vqw_busy2 = 0, vqw_wait2 = 0, vqw_t = 0, vqw_levels = 0;
#define VQW2_P(k) do { long long n_; VQW_NOW(n_); if (lane == 0) atomicAdd((unsigned long long *)&g_vqw_dbg[(k)], (unsigned long long)(n_ - vqw_p)); vqw_p = n_; } while (0)
#else
#define VQW2_P(k) do { } while (0)
#endif
    const unsigned long long below = (1ull << lane) - 1ull;
    long long cf = blockIdx.x;
    if (A.cf_list) {
        if (cf >= (long long)*A.cf_count)
            return;
        cf = A.cf_list[cf];
    }
    if (cf >= A.n_cf)
        return;
    const long long frame = cf / A.n_ch;
    const unsigned fl = A.flags ? A.flags[frame] : 0u;
    const bool is_short = A.mixed && (fl & 2u);
    if (is_short && A.status_in) {             /* hop dropped: nothing to code (n_bytes stays 0) */
        unsigned st = 0;
        for (int c = 0; c < A.n_ch; ++c)
            st |= A.status_in[frame * A.n_ch + c];
        if (st & PACX_ST_ZERO_SUBBLOCK)
            return;
    }
    const int nb = is_short ? T.nb_short : T.nb_long;
    const int n_sub = is_short ? PACX_SUB : 1;
    const int n_vb = n_sub * nb;
    const int32_t *__restrict__ lower = is_short ? T.band_lower_short : T.band_lower_long;
    const int32_t *__restrict__ count = is_short ? T.band_lines_short : T.band_lines_long;
    const unsigned short *__restrict__ reg_of = is_short ? V.reg_short : V.reg_long;     /* region of band b (of sub-block 0) */
    const unsigned char *__restrict__ rlog_of = is_short ? V.rlog_short : V.rlog_long;
    const int reg_sub = is_short ? V.reg_short_total : 0;                                 /* regions of one sub-block */
    const int first_omit = (!is_short && T.use_sbr) ? T.first_omitted : nb;
    const long long boff = cf * T.band_stride;
    const double *__restrict__ lin = A.lines + cf * PACX_M_LONG;
    const double half_pi = 1.5707963267948966;
    auto sub_of = [&](int vb) { return is_short ? vb / nb : 0; };

    for (int i = tid; i < VQ_WORDS; i += 64 * VQ_WAVES)
        words[i] = 0u;
    if (tid < VQ_ROWS_SMALL && tid <= V.l_max)
        vq_row_off_small[tid] = V.row_off[tid];
    if (tid < VQF_HL && tid <= V.l_max)
        vqf_half_log2[tid] = V.half_log2[tid];
    if (tid < VQF_LT)
        vqf_log2_tan[tid] = V.log2_tan[tid];
    if (tid < 8)
        misc[tid] = 0;
    /* phase A: every wave places its bands' lines (scaled) in their regions -- all its loads in flight at once --
       and then takes their norms: the unit shapes x / gain are level 0 of the walk */
    for (int vb = wave; vb < n_vb; vb += VQ_WAVES) {
        const int j = sub_of(vb), b = vb - j * nb;
        if (b >= first_omit)
            continue;
        const int lo = j * PACX_M_SHORT + ldc(&lower[b]), cnt = ldc(&count[b]);
        const double up = (double)(1 << A.overall[cf * PACX_SUB + j]);
        double *dst = buf + j * reg_sub + reg_of[b];
        for (int i = lane; i < cnt; i += 64)
            dst[i] = lin[lo + i] * up;
    }
    vq_fence();                                            /* a wave reads back what it wrote itself */
    for (int vb = wave; vb < n_vb; vb += VQ_WAVES) {
        const int j = sub_of(vb), b = vb - j * nb;
        double g;
        if (b >= first_omit) {
            const double up = (double)(1 << A.overall[cf * PACX_SUB]);
            const double v = A.sbr_mean[cf * PACX_SUB + (b - first_omit)] * up;
            g = sqrt(v * v);
        } else {
            const int cnt = ldc(&count[b]);
            double *xs = buf + j * reg_sub + reg_of[b];
            double acc = 0.0;
            for (int i = lane; i < cnt; i += 64) {
                const double x = xs[i];
                acc = fma(x, x, acc);
            }
            g = sqrt(wave_sum_f64(acc));
            for (int i = lane; i < cnt; i += 64)
                xs[i] = xs[i] / g;
        }
        if (lane == 0)
            gain_s[vb] = g;
    }
    __syncthreads();
    if (wave == 0) {
        /* final allocations, band positions, header fields, the roots of the trees: one band per lane (as k_vq_frame) */
        const int vb = lane, j = sub_of(vb < n_vb ? vb : 0), b = vb - j * nb;
        int ba = 0, r_bits = 0, cnt = 1;
        if (vb < n_vb) {
            ba = A.bit_alloc[boff + vb];
            if (ba && gain_s[vb] == 0.0)
                ba = 0;                                   /* coder/codec.py:352-353 */
            cnt = (b >= first_omit) ? 1 : count[b];
            r_bits = ba * cnt;
            A.bit_alloc[boff + vb] = ba;
            ba_s[vb] = ba;
        }
        int total_bits;
        const int before = wave_excl_scan_i32(r_bits, lane, total_bits);
        const int head = T.n_scale_bits + T.n_mant_size_bits * nb;
        const int start = 3 + (j + 1) * head + before;
        if (vb < n_vb) {
            start_s[vb] = start;
            end_s[vb] = start + r_bits;
        }
        if (lane == 0) {
            vq_put32(words, 0, fl & 1u, 1);
            vq_put32(words, 1, (fl >> 1) & 1u, 1);
            vq_put32(words, 2, (fl >> 2) & 1u, 1);
            misc[2] = 3 + n_sub * head + total_bits;      /* bits written */
        }
        if (vb < n_vb) {
            const int sub_base = 3 + j * head + __shfl(before, vb - b, 64);
            if (b == 0)
                vq_put32(words, sub_base, (unsigned)A.overall[cf * PACX_SUB + j], T.n_scale_bits);
            vq_put32(words, sub_base + T.n_scale_bits + T.n_mant_size_bits * b, (unsigned)(ba ? ba - 1 : 0),
                     T.n_mant_size_bits);
        }
        int bits_gain = ba, bits_shape = 0;
        if (vb < n_vb && b < first_omit && ba) {
            bits_gain = (int)floor((double)r_bits / (double)cnt + V.half_log2[cnt]);
            bits_shape = r_bits - bits_gain;
            if (bits_shape < 0)
                bits_shape = 0;
        }
        const bool rooted = vb < n_vb && bits_shape != 0;
        const unsigned long long mr = __builtin_amdgcn_ballot_w64(rooted);
        const int id = __popcll(mr & below);
        if (vb < n_vb) {
            bg_s[vb] = bits_gain;
            bs_s[vb] = bits_shape;
            root_s[vb] = rooted ? (unsigned short)id : 0xFFFF;
        }
        if (rooted)
            N.create(id, cnt, bits_shape, j * reg_sub + reg_of[b], bits_shape > PACX_VQ_SPLIT_BITS ? 0 : 1, vb, rlog_of[b]);
        if (lane == 0) {
            misc[0] = __popcll(mr);
            N.lvl[0] = 0;
        }
    }
    __syncthreads();

    /* ---- the SPLITS of a level grouped by the lane footprint of their halves: up to 8 / 16 / 32 / 64 lanes, larger
       (ord / cls of that level's parity).  One wave; the leaves wait in their regions for the end of the walk */
    auto classify = [&](int lev_b, int lev_e, int par) {
        unsigned short *ord_w = N.ord + par * VQ2_NLV;
        int *cls_w = N.cls + par * 16;
        int cnt_c[5] = {0, 0, 0, 0, 0};
        if (lane == 0 && lev_e - lev_b > VQ2_NLV)
            misc[1] = 1;
        for (int base = lev_b; base < lev_e && lev_e - lev_b <= VQ2_NLV; base += 64) {
            const int j = base + lane;
            int c = 5;
            if (j < lev_e && N.kind(j) == 0) {
                const int n = N.nn(j), half = n - n / 2;
                c = half <= 8 ? 0 : (half <= 16 ? 1 : (half <= 32 ? 2 : (half <= 64 ? 3 : 4)));
            }
#pragma unroll
            for (int k = 0; k < 5; ++k)
                cnt_c[k] += __popcll(__builtin_amdgcn_ballot_w64(c == k));
        }
        int start_c[6];
        start_c[0] = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k)
            start_c[k + 1] = start_c[k] + cnt_c[k];
        if (lane < 6) {
            int v = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k)
                v = (lane == k) ? start_c[k] : v;
            cls_w[lane] = v;
        }
        int run_c[5] = {0, 0, 0, 0, 0};
        for (int base = lev_b; base < lev_e && lev_e - lev_b <= VQ2_NLV; base += 64) {
            const int j = base + lane;
            int c = 5;
            if (j < lev_e && N.kind(j) == 0) {
                const int n = N.nn(j), half = n - n / 2;
                c = half <= 8 ? 0 : (half <= 16 ? 1 : (half <= 32 ? 2 : (half <= 64 ? 3 : 4)));
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const unsigned long long m = __builtin_amdgcn_ballot_w64(c == k);
                if (c == k)
                    ord_w[start_c[k] + run_c[k] + __popcll(m & below)] = (unsigned short)j;
                run_c[k] += __popcll(m);
            }
        }
    };
    bool undefined = false;
    int lev_b = 0, lev_e = misc[0], depth = 0;
    if (wave == 0)
        classify(0, lev_e, 0);
    __syncthreads();
    if (wave == 0)
        VQW2_P(21);                                    /* phase A, thread 0's view */
#ifdef PACX_VQ_WAITDBG
    VQW_NOW(vqw_t);
#endif
    for (;;) {
        const unsigned short *ord_c = N.ord + (depth & 1) * VQ2_NLV;
        const int *cls_c = N.cls + (depth & 1) * 16;
        if (misc[1])
            break;
        const int s_n = cls_c[5];
        if (s_n == 0)
            break;                                        /* no split on this level: the walk is over */
        /* ---- packed passes over the level's splits, dealt round-robin (the expensive classes first): fold, the two
           norms, the halves normalised IN PLACE -- mid over the first half of the node's region, side over the second */
        {
            int ph = 0;
#pragma unroll 1
            for (int c = 4; c >= 0; --c) {
                const int c_b = cls_c[c], c_n = cls_c[c + 1] - c_b;
                const int lg = (c == 0) ? 3 : (c == 1) ? 2 : (c == 2) ? 1 : 0;
                for (int p0 = c_b; p0 < c_b + c_n; p0 += 1 << lg) {
                    const bool my_pass = (ph & 3) == wave;
                    ph += 1;
                    if (!my_pass)
                        continue;
                    const int p_n = min(1 << lg, c_b + c_n - p0);
                    if (c < 4) {
                        const int lp = c + 3, P = 1 << lp;             /* blocks of 8, 16, 32, 64 lanes */
                        const int g = lane >> lp, i = lane & (P - 1);
                        const bool valid = g < p_n;
                        const int node = valid ? ord_c[p0 + g] : 0;
                        const int n = valid ? N.nn(node) : 0;
                        const int cut = n / 2, half = n - cut;
                        double *reg = buf + (valid ? N.off(node) : 0);
                        const int rlg = valid ? N.rlog(node) : 1, rt = valid ? N.rot(node) : 0;
                        const int r2 = 1 << (rlg - 1), rm = 2 * r2 - 1;
                        const int rot_m = Vq2Store::child_rot(node, 0, half, rlg - 1), rot_s = Vq2Store::child_rot(node, 1, half, rlg - 1);
                        const bool mine = i < half;
                        const double left = (mine && i < cut) ? reg[(i + rt) & rm] : 0.0;
                        const double right = mine ? reg[(cut + i + rt) & rm] : 0.0;
                        double mid = mine ? (left + right) / 2.0 : 0.0;
                        double sd = mine ? (left - right) / 2.0 : 0.0;
                        double mm = fma(mid, mid, 0.0), ss = fma(sd, sd, 0.0);
                        for (int off = P >> 1; off > 0; off >>= 1) {
                            mm = mm + __shfl_xor(mm, off, 64);
                            ss = ss + __shfl_xor(ss, off, 64);
                        }
                        const double m_l2 = sqrt(mm), s_l2 = sqrt(ss);
                        if (m_l2 != 0.0)
                            mid = mid / m_l2;
                        if (s_l2 != 0.0)
                            sd = sd / s_l2;
                        if (mine) {                        /* every read of the node came back before the shuffles */
                            reg[(i + rot_m) & (r2 - 1)] = mid;
                            reg[r2 + ((i + rot_s) & (r2 - 1))] = sd;
                        }
                        if (valid && i == 0)
                            N.val[node] = (unsigned long long)__double_as_longlong(m_l2 == 0.0 ? -1.0 : s_l2 / m_l2);
                    } else {
                        /* a split of more than 128 components (at most 512): the whole node into registers first */
                        const int node = ord_c[p0];
                        const int n = N.nn(node);
                        double *reg = buf + N.off(node);
                        const int r2 = 1 << (N.rlog(node) - 1);           /* more than 128 components: no rotation, here or below */
                        const int cut = n / 2, half = n - cut;
                        double mid[4], sd[4];
                        double mm = 0.0, ss = 0.0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int i = lane + 64 * k;
                            const double left = (i < cut) ? reg[i] : 0.0;
                            const double right = (i < half) ? reg[cut + i] : 0.0;
                            mid[k] = (left + right) / 2.0;
                            sd[k] = (left - right) / 2.0;
                            if (i < half) {
                                mm = fma(mid[k], mid[k], mm);
                                ss = fma(sd[k], sd[k], ss);
                            }
                        }
                        const double m_l2 = sqrt(wave_sum_f64(mm));
                        const double s_l2 = sqrt(wave_sum_f64(ss));
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int i = lane + 64 * k;
                            if (i < half) {
                                reg[i] = (m_l2 != 0.0) ? mid[k] / m_l2 : mid[k];
                                reg[r2 + i] = (s_l2 != 0.0) ? sd[k] / s_l2 : sd[k];
                            }
                        }
                        if (lane == 0)
                            N.val[node] = (unsigned long long)__double_as_longlong(m_l2 == 0.0 ? -1.0 : s_l2 / m_l2);
                    }
                }
            }
        }
        VQW_BARRIER(vqw_busy1, vqw_wait1);
        /* ---- the level's splits, one per lane: angle, bit split, children (as k_vq_frame) */
        const bool early = s_n <= 64;
        for (int k0 = 64 * wave; k0 < s_n; k0 += 64 * VQ_WAVES) {
            const bool has = k0 + lane < s_n;
            const int snode = has ? ord_c[k0 + lane] : 0;
            const int sn = has ? N.nn(snode) : 2;
            const int half = sn - sn / 2;
            const int bits = has ? N.bb(snode) : 0;
            const double q = has ? __longlong_as_double((long long)N.val[snode]) : -1.0;
            const double theta = (q < 0.0) ? 0.0 : vq_atan(q);
            const double hl = (half < VQF_HL) ? vqf_half_log2[half] : V.half_log2[half];
            const int a_theta = (int)floor((double)bits / (double)half + hl);
            int a_rest = bits - a_theta;
            if (a_rest < 0)
                a_rest = 0;
            const double tn = theta / half_pi;
            double theta_q = 0.0;
            unsigned long long code = 0ull;
            int w_theta = 0;
            if (!__builtin_amdgcn_ballot_w64(a_theta > 31)) {
                if (a_theta > 0) {
                    const double factor = (double)((1u << a_theta) - 1u);
                    unsigned c32;
                    if (tn >= 1.0)
                        c32 = (1u << (a_theta - 1)) - 1u;
                    else
                        c32 = (unsigned)floor((factor * tn + 1.0) * 0.5);
                    code = c32;
                    w_theta = a_theta;
                    const unsigned mag = c32 & ((1u << (a_theta - 1)) - 1u);
                    double dq = (double)(2u * mag) / factor;
                    if (c32 >> (a_theta - 1))
                        dq = -dq;
                    theta_q = dq * half_pi;
                }
            } else if (a_theta > 62) {
                if (has)
                    undefined = true;
            } else if (a_theta > 0) {
                if (tn >= 1.0) {
                    code = (1ull << (a_theta - 1)) - 1ull;
                } else {
                    const double factor = (a_theta <= 53) ? (double)((1ull << a_theta) - 1ull) : ldexp(1.0, a_theta);
                    code = (unsigned long long)floor((factor * tn + 1.0) * 0.5);
                }
                w_theta = a_theta;
                const unsigned long long mag = code & ((1ull << (a_theta - 1)) - 1ull);
                const double den = (a_theta <= 53) ? (double)((1ull << a_theta) - 1ull) : ldexp(1.0, a_theta);
                double dq = (double)(2ull * mag) / den;
                if (code >> (a_theta - 1))
                    dq = -dq;
                theta_q = dq * half_pi;
            }
            int a_mid = 0;
            if (theta_q != 0.0) {
                double lt;
                if (a_theta <= PACX_VQ_THETA_TABLE_BITS && theta_q > 0.0) {
                    const int at_lt = ((1 << (a_theta - 1)) - 1) + (int)code;
                    lt = (at_lt < VQF_LT) ? vqf_log2_tan[at_lt] : V.log2_tan[at_lt];
                } else {
                    lt = vq_log2_tan(theta_q);
                }
                const double v = ((double)a_rest - (double)(half - 1) * lt) / 2.0;
                const double f = floor(v);
                a_mid = (f < 0.0) ? 0 : ((f > (double)a_rest) ? a_rest : (int)f);
            }
            const int a_side = a_rest - a_mid;
            const int c_mid = (has && a_mid > 0) ? 1 : 0, c_side = (has && a_side > 0) ? 1 : 0;
            int born;
            const int rel = wave_excl_scan_i32(c_mid + c_side, lane, born);
            int base = 0;
            if (lane == 0 && born)
                base = atomicAdd(&misc[0], born);
            base = __shfl(base, 0, 64);
            if (base + born > VQ2_NCAP) {
                if (lane == 0)
                    misc[1] = 1;
                continue;
            }
            if (has) {
                const int first = base + rel;
                N.val[snode] = code;
                N.wid(snode) = (unsigned char)w_theta;
                N.kid(snode) = (unsigned short)first;
                N.has(snode) = (unsigned char)(c_mid | (c_side << 1));
                const int at = N.off(snode), rlog = N.rlog(snode) - 1;
                const int bd = N.band(snode);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int a = c ? a_side : a_mid;
                    if (a <= 0)
                        continue;
                    const int id = c ? first + c_mid : first;
                    const bool splits = a > PACX_VQ_SPLIT_BITS && depth + 1 < VQ_DEPTH;
                    if (a > PACX_VQ_SPLIT_BITS && !splits)
                        undefined = true;                  /* deeper than any real tree */
                    if (splits && half < 2)
                        misc[1] = 1;                       /* a one-component node asked to split: its halves have no
                                                              region of their own here -- left to k_vq_redo */
                    N.create(id, half, a > 65535 ? 65535 : a, at + (c ? (1 << rlog) : 0), splits ? 0 : 1, bd,
                             rlog | (Vq2Store::child_rot(snode, c, half, rlog) << 4));
                }
            }
        }
        if (early && wave == 0) {                          /* one chunk: this wave made all the children, it sorts them at once */
            vq_fence();
            if (!misc[1])
                classify(lev_e, misc[0], (depth + 1) & 1);
        }
        VQW_BARRIER(vqw_busy2, vqw_wait2);
#ifdef PACX_VQ_WAITDBG
        vqw_levels += 1;
#endif
        if (misc[1])
            break;
        const int n_now = misc[0];
        depth += 1;
        if (tid == 0)
            N.lvl[depth] = (unsigned short)lev_e;
        if (n_now == lev_e) {                              /* no children */
            depth -= 1;
            break;
        }
        lev_b = lev_e;
        lev_e = n_now;
        if (!early) {
            if (wave == 0)
                classify(lev_b, lev_e, depth & 1);
            __syncthreads();
        }
        if (depth > VQ_DEPTH) {
            if (tid == 0)
                misc[1] = 1;
            __syncthreads();
            break;
        }
    }
    if (misc[1]) {
        if (tid == 0)
            A.n_bytes[cf] = -1;                            /* does not fit the store: k_vq_redo codes this channel-frame */
        return;
    }
    const int n_nodes = misc[0];
    if (tid == 0)
        N.lvl[depth + 1] = (unsigned short)n_nodes;
#ifdef PACX_VQ_WAITDBG
    if (lane == 0) {
        atomicAdd((unsigned long long *)&g_vqw_dbg[wave * 4 + 0], (unsigned long long)vqw_busy1);
        atomicAdd((unsigned long long *)&g_vqw_dbg[wave * 4 + 1], (unsigned long long)vqw_wait1);
        atomicAdd((unsigned long long *)&g_vqw_dbg[wave * 4 + 2], (unsigned long long)vqw_busy2);
        atomicAdd((unsigned long long *)&g_vqw_dbg[wave * 4 + 3], (unsigned long long)vqw_wait2);
        if (wave == 0) {
            atomicAdd((unsigned long long *)&g_vqw_dbg[17], (unsigned long long)vqw_levels);
            atomicAdd((unsigned long long *)&g_vqw_dbg[18], 1ull);
        }
    }
    VQW_NOW(vqw_p);
#endif
    /* ---- the leaves, all of the unit's at once.  Lists by size (the ord area is free now): up to 16 components
       (one per lane), 17..32 (half-wave groups), larger (the whole-wave coder) */
    unsigned short *small_l = N.tot, *mid_l = N.ord, *big_l = N.ord + VQ2_NLV;      /* tot is free until the widths go bottom-up */
    for (int j = tid; j < n_nodes; j += 64 * VQ_WAVES) {
        if (N.kind(j) != 1)
            continue;
        const int n = N.nn(j);
        if (n <= 16)
            small_l[atomicAdd(&misc[5], 1)] = (unsigned short)j;
        else if (n <= 32)
            mid_l[atomicAdd(&misc[6], 1) & (VQ2_NLV - 1)] = (unsigned short)j;
        else
            big_l[atomicAdd(&misc[7], 1) & (VQ2_NLV - 1)] = (unsigned short)j;
    }
    __syncthreads();
    if (wave == 0)
        VQW2_P(22);                                    /* leaf lists */
    const int n_small = misc[5], n_mid = misc[6], n_big = misc[7];
    if (n_mid > VQ2_NLV || n_big > VQ2_NLV) {                      /* more than the lists hold (no real stream): k_vq_redo */
        if (tid == 0)
            A.n_bytes[cf] = -1;
        return;
    }
    for (int s0 = 64 * wave; s0 < n_small; s0 += 64 * VQ_WAVES) {
        const bool valid = s0 + lane < n_small;
        const int node = valid ? small_l[s0 + lane] : 0;
        const int n = valid ? N.nn(node) : 0;
        int bits = valid ? N.bb(node) : 0;
        bits = bits > 32 ? 32 : bits;
        const int K = valid ? V.k_of[n * 33 + bits] : 0;
        const int width = valid ? V.w_of[n * 33 + bits] : 0;
        bool ok = true;
        const int n_max = __builtin_amdgcn_readfirstlane(wave_max_i32((valid && K >= 0) ? n : 0));
        {   /* the floors of a leaf's components: scr as int [VQ2_BUF], at the component's buffer position; a lane
               without a leaf rides along with n = 0 */
            const bool run = valid && K >= 0;
            const int at = run ? N.off(node) : 0;
            const bool fine = vq_leaf_lane((vq_lds_f64 *)(buf + at), run ? n : 0, run ? K : 0, (vq_lds_i32 *)scr + at, n_max,
                                           run ? N.rot(node) : 0, run ? (1 << N.rlog(node)) - 1 : 0);
            ok = run ? fine : true;
        }
        if (valid) {
            if (K < 0) {                                   /* a 1-dimensional leaf: the reference never returns */
                N.kind(node) = 3;
                N.wid(node) = 0;
                N.has(node) = 4;                           /* no terms to add */
                undefined = true;
            } else {
                N.wid(node) = (unsigned char)width;
                if (!ok) {
                    N.has(node) = 4;
                    undefined = true;
                }
            }
        }
    }
    if (wave == 0)
        VQW2_P(23);                                    /* small leaves, one per lane (wave 0's share) */
    /* the few leaves of 17..32 components meanwhile, two per pass: the waves that had no share of the small ones first */
    for (int m0 = 0; m0 < n_mid; m0 += 2) {
        if (((m0 >> 1) & 3) != ((wave + 3) & 3))
            continue;
        const int g = lane >> 5, l = lane & 31;
        const bool valid = m0 + g < n_mid;
        const int node = valid ? mid_l[m0 + g] : 0;
        const int n = valid ? N.nn(node) : 0;
        int bits = valid ? N.bb(node) : 0;
        bits = bits > 32 ? 32 : bits;
        const int K = valid ? V.k_of[n * 33 + bits] : 0;
        const int width = valid ? V.w_of[n * 33 + bits] : 0;
        const double x = (valid && l < n) ? buf[N.off(node) + ((l + N.rot(node)) & ((1 << N.rlog(node)) - 1))] : 0.0;
        bool ok = false;
        const unsigned long long term = vq_leaf_group<32, true>(V, x, n, K < 0 ? 0 : K, l, ok);
        if (valid && l == 0) {
            N.val[node] = (K >= 0 && ok) ? term : 0ull;
            N.wid(node) = (unsigned char)(K < 0 ? 0 : width);
            if (K < 0)
                N.kind(node) = 3;
        }
        if (valid && (K < 0 || !ok))
            undefined = true;
    }
    __syncthreads();
    if (wave == 0)
        VQW2_P(24);                                    /* mid leaves + barrier */
    /* leaves of more than 32 components: the whole-wave coder, one at a time (its scratch is the area the small
       leaves' floors were in); the last wave takes them while the others start on the terms */
    if (wave == VQ_WAVES - 1) {
        for (int b0 = 0; b0 < n_big; ++b0) {
            const int node = big_l[b0];
            const int n = N.nn(node);
            int bits = N.bb(node);
            bits = bits > 32 ? 32 : bits;
            const int K = ldc(&V.k_of[n * 33 + bits]);
            const int width = ldc(&V.w_of[n * 33 + bits]);
            bool ok = true;
            unsigned long long idx = 0ull;
            if (K >= 0 && 2 * n <= VQ2_SCR)
                idx = vq_leaf_idx<true>(V, buf + N.off(node), n, K, scr, scr + n, lane, ok);
            else
                ok = false;
            if (!ok)
                undefined = true;
            if (lane == 0) {
                N.val[node] = ok ? idx : 0ull;
                N.wid(node) = (unsigned char)(K < 0 ? 0 : width);
                if (K < 0)
                    N.kind(node) = 3;
            }
            vq_fence();
        }
    }
    /* the enumeration index of the small leaves, one COMPONENT per thread: component i of a leaf (l = n - i - 1
       dimensions behind it, k pulses left, magnitude a >= 1) adds N(l,k) + 2 (P(l,k-1) - P(l,k-a)) (+ N(l,k-a) if
       negative) -- coder/gain_shape_quantize.py:105-124 -- into the leaf's index.  Two components per thread at a
       time, their eight table reads in flight together: one round trip to L2 for (nearly) the whole unit */
    for (int t0 = tid; t0 < 16 * n_small; t0 += 2 * 64 * VQ_WAVES) {
        unsigned long long nk[2], pk1[2], pka[2], nka[2];
        int node4[2], l1d[2];
        long long k4[2], a4[2];
        bool neg4[2], live[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tt = t0 + u * 64 * VQ_WAVES;
            const bool in = tt < 16 * n_small;
            const int node = in ? small_l[tt >> 4] : 0, i = tt & 15;
            const int n = in ? N.nn(node) : 0;
            live[u] = in && i < n && N.has(node) != 4;
            unsigned long long d = 0ull;
            if (live[u])
                d = (unsigned long long)__double_as_longlong(buf[N.off(node) + ((i + N.rot(node)) & ((1 << N.rlog(node)) - 1))]);
            a4[u] = (long long)(d & 0x7FFFFFFFull);
            live[u] = live[u] && a4[u] >= 1;
            neg4[u] = ((d >> 31) & 1ull) != 0ull;
            k4[u] = (long long)(d >> 32);
            node4[u] = node;
            l1d[u] = n - i - 1;
            /* table part (l >= 3): addresses clamped to entry 0 where the closed forms apply */
            const bool tab = live[u] && l1d[u] >= 3;
            const long long base = tab ? (l1d[u] < VQ_ROWS_SMALL ? vq_row_off_small[l1d[u]] : V.row_off[l1d[u]]) : 0;
            const long long kk = tab ? k4[u] : 1, ka = tab ? k4[u] - a4[u] : 0;
            nk[u] = V.n_tab[base + kk];
            pk1[u] = V.p_tab[base + kk - 1];
            pka[u] = V.p_tab[base + ka];
            nka[u] = V.n_tab[base + ka];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!live[u])
                continue;
            unsigned long long term;
            if (l1d[u] >= 3) {
                term = nk[u] + 2ull * (pk1[u] - pka[u]) + (neg4[u] ? nka[u] : 0ull);
            } else {
                term = vq_N(V, l1d[u], k4[u]);
                term += 2ull * (vq_P(V, l1d[u], k4[u] - 1) - vq_P(V, l1d[u], k4[u] - a4[u]));
                if (neg4[u])
                    term += vq_N(V, l1d[u], k4[u] - a4[u]);
            }
            atomicAdd(&N.val[node4[u]], term);
        }
    }
    __syncthreads();
    if (wave == 0)
        VQW2_P(25);                                    /* terms (+ big leaves) + barrier */
    /* ---- subtree widths and field counts bottom-up (counts ride in nn, field numbers in off) */
    for (int d = depth; d >= 0; --d) {
        const int b = N.lvl[d], e = N.lvl[d + 1];
        for (int j = b + tid; j < e; j += 64 * VQ_WAVES) {
            const int k0 = N.kid(j), hs = N.has(j);
            int t = N.wid(j), f = (N.kind(j) == 3) ? 0 : 1;
            if (hs & 1) { t += N.tot[k0]; f += N.nn(k0); }
            if (hs & 2) { const int k1 = k0 + (hs & 1); t += N.tot[k1]; f += N.nn(k1); }
            N.tot[j] = (unsigned short)t;
            N.nn(j) = (unsigned short)f;
        }
        __syncthreads();
    }
    if (tid < n_vb && root_s[tid] != 0xFFFF) {
        N.pos[root_s[tid]] = (unsigned short)start_s[tid];
        N.off(root_s[tid]) = 0;
    }
    __syncthreads();
    for (int d = 0; d <= depth; ++d) {
        const int b = N.lvl[d], e = N.lvl[d + 1];
        for (int j = b + tid; j < e; j += 64 * VQ_WAVES) {
            const int k0 = N.kid(j), hs = N.has(j);
            int p = N.pos[j] + N.wid(j), r = N.off(j) + ((N.kind(j) == 3) ? 0 : 1);
            if (hs & 1) {
                N.pos[k0] = (unsigned short)p;
                N.off(k0) = (unsigned short)r;
                p += N.tot[k0];
                r += N.nn(k0);
            }
            if (hs & 2) {
                const int k1 = k0 + (hs & 1);
                N.pos[k1] = (unsigned short)p;
                N.off(k1) = (unsigned short)r;
            }
        }
        __syncthreads();
    }
    if (wave == 0)
        VQW2_P(26);                                    /* widths, positions */
    /* ---- every field by its own lane */
    for (int j = tid; j < n_nodes; j += 64 * VQ_WAVES) {
        const int w = N.wid(j);
        unsigned long long v = N.val[j];
        if (w > 0 && w < 64)
            v &= (1ull << w) - 1ull;
        if (w > 0)
            vq_put_field(words, N.pos[j], v, w);
        if (A.log && N.kind(j) != 3 && N.off(j) < A.log_cap) {
            const int vb = N.band(j), sj = sub_of(vb);
            const long long slot = (cf * PACX_SUB + sj) * PACX_MAX_BANDS + (vb - sj * nb);
            pacx_vq_entry e;
            e.value = v;
            e.width = w;
            e.band = vb - sj * nb;
            A.log[slot * A.log_cap + N.off(j)] = e;
        }
    }
    /* ---- the gains of all bands side by side (mu-law, QuantizeUniform; the index soaks up the slack) */
    if (tid < 64) {
        const int vb = lane, sj = sub_of(vb < n_vb ? vb : 0), b = vb - sj * nb;
        const int ba = (vb < n_vb) ? ba_s[vb] : 0;
        const long long slot = (cf * PACX_SUB + sj) * PACX_MAX_BANDS + b;
        if (vb < n_vb && !ba && A.log_count)
            A.log_count[slot] = 0;
        const int cnt = (vb < n_vb && b < first_omit) ? count[b] : 1;
        const double gain = (vb < n_vb) ? gain_s[vb] : 0.0;
        const double g = vq_log(1.0 + 255.0 * fabs(gain / (double)cnt)) / V.log_mu1;
        if (ba) {
            const int rt = root_s[vb];
            const int used = (rt != 0xFFFF) ? N.tot[rt] : 0;
            const int fields = (rt != 0xFFFF) ? N.nn(rt) : 0;
            int bits_gain = bg_s[vb] + bs_s[vb] - used;
            if (bits_gain < 0)
                bits_gain = 0;
            int width = bits_gain;
            unsigned long long hi = 0, lo = 0;
            if (bits_gain > 128) {
                undefined = true;
                width = 0;
            } else if (bits_gain > 0) {
                vq_quantize_code(g, bits_gain, hi, lo);
            }
            const int at = start_s[vb] + used;
            if (width > 64) {
                vq_put_field(words, at, hi, width - 64);
                vq_put_field(words, at + width - 64, lo, 64);
            } else if (width > 0) {
                vq_put_field(words, at, lo, width);
            }
            if (at + width != end_s[vb])
                undefined = true;                          /* a band must fill its slot exactly */
            if (A.log && fields < A.log_cap) {
                pacx_vq_entry e;
                e.value = lo;
                e.width = width;
                e.band = b;
                A.log[slot * A.log_cap + fields] = e;
            }
            if (A.log_count)
                A.log_count[slot] = fields + 1;
        }
    }
    if (__builtin_amdgcn_ballot_w64(undefined) && lane == 0 && A.status)
        atomicOr(&A.status[cf], PACX_ST_VQ_UNDEFINED);
    __syncthreads();
    const int written = misc[2];
    const int size_rule = written - 3 + n_sub * nb * T.n_scale_bits;
    const int nbytes = (size_rule + 4 + 7) >> 3;
    unsigned *dst = (unsigned *)(A.payload + cf * (long long)A.payload_stride);
    for (int i = tid; i < (nbytes + 3) / 4; i += 64 * VQ_WAVES)
        dst[i] = __builtin_bswap32(words[i]);
    if (tid == 0)
        A.n_bytes[cf] = nbytes;
#ifdef PACX_VQ_WAITDBG
    if (wave == 0)
        VQW2_P(27);                                    /* fields, gains, hand-over */
    {
        long long t_fin;
        VQW_NOW(t_fin);
        if (tid == 0)
            atomicAdd((unsigned long long *)&g_vqw_dbg[20], (unsigned long long)(t_fin - vqw_start));
    }
#endif
}

/* short frames: flags + the 8 sub-block strings, back to back */
__global__ __launch_bounds__(64) void k_vq_join(PacxTables T, const uint8_t *__restrict__ flags, int n_ch,
                                               long long n_cf, const uint32_t *__restrict__ status,
                                               const unsigned *__restrict__ unit_words,
                                               const int32_t *__restrict__ unit_bits,
                                               uint8_t *__restrict__ payload, int payload_stride,
                                               int32_t *__restrict__ n_bytes, int redo)
{
    __shared__ unsigned words[VQ_WORDS];
    const int lane = threadIdx.x;
    const long long cf = blockIdx.x;
    if (cf >= n_cf)
        return;
    const long long frame = cf / n_ch;
    const unsigned fl = flags[frame];
    if (!(fl & 2u))
        return;
    if (redo && n_bytes[cf] != -1)
        return;                                /* k_vq_frame wrote this frame's string itself */
    unsigned st = 0;
    if (status)
        for (int c = 0; c < n_ch; ++c)
            st |= status[frame * n_ch + c];
    if (st & PACX_ST_ZERO_SUBBLOCK) {
        if (lane == 0)
            n_bytes[cf] = 0;
        return;
    }
    for (int i = lane; i < VQ_WORDS; i += 64)
        words[i] = 0u;
    __syncthreads();
    if (lane == 0) {
        vq_put32(words, 0, fl & 1u, 1);
        vq_put32(words, 1, (fl >> 1) & 1u, 1);
        vq_put32(words, 2, (fl >> 2) & 1u, 1);
    }
    int pos = 3, size = 0;
    for (int sb = 0; sb < PACX_SUB; ++sb) {
        const long long unit = cf * PACX_SUB + sb;
        const int nbit = unit_bits[unit * 2];
        size += unit_bits[unit * 2 + 1];
        const unsigned *src = unit_words + unit * VQ_WORDS;
        for (int w = lane; w < (nbit + 31) / 32; w += 64) {
            const int width = (nbit - 32 * w) >= 32 ? 32 : (nbit - 32 * w);
            vq_put32(words, pos + 32 * w, src[w] >> (32 - width), width);
        }
        pos += nbit;
    }
    __syncthreads();
    const int nbytes = (size + 4 + 7) >> 3;
    unsigned *dst = (unsigned *)(payload + cf * (long long)payload_stride);
    for (int i = lane; i < (nbytes + 3) / 4; i += 64)
        dst[i] = __builtin_bswap32(words[i]);
    if (lane == 0)
        n_bytes[cf] = nbytes;
}

/* ---------------------------------------------------------------- launcher */
void pacx_launch_vq(const PacxTables &T, const void *vq_view, const uint8_t *flags, int n_ch, long long n_cf,
                    const double *lines, const int32_t *overall, int32_t *bit_alloc, const double *sbr_mean,
                    uint32_t *status, uint8_t *payload, int payload_stride, int32_t *n_bytes,
                    unsigned *unit_words, int32_t *unit_bits, pacx_vq_entry *log, int32_t *log_count,
                    int log_cap, int stage, const int32_t *cf_list, const int32_t *cf_count, hipStream_t st)
{
    /* stage 0: everything.  1: only k_vq_frame, over cf_list (a block-switched batch codes its long and its
       short frames behind their own front-end chains, on two streams).  2: what follows it (the frames it
       left, the join pass of the old coder) */
    if (n_cf <= 0)
        return;
    const VqView &V = *(const VqView *)vq_view;
    VqArgs A;
    A.cf_list = cf_list;
    A.cf_count = cf_count;
    A.flags = flags;
    A.n_ch = n_ch;
    A.n_cf = n_cf;
    A.mixed = flags ? 1 : 0;
    A.lines = lines;
    A.overall = overall;
    A.bit_alloc = bit_alloc;
    A.sbr_mean = sbr_mean;
    A.status_in = status;
    A.status = status;
    A.payload = payload;
    A.payload_stride = payload_stride;
    A.n_bytes = n_bytes;
    A.unit_words = unit_words;
    A.unit_bits = unit_bits;
    A.log = log;
    A.log_count = log_count;
    A.log_cap = log_cap;
    {
        /* PACX_VQ_BFS: 0 = depth-first walk only, n = level by level from n shape bits (1: every band).
           Small trees are quicker depth first (the level walk has a fixed cost per level); the crossover
           was measured at 96-128 bits (tools/vq_bfs_sweep.sh) */
        const char *e = getenv("PACX_VQ_BFS");
        A.bfs = e ? atoi(e) : 112;
    }
    {
        const char *e = getenv("PACX_VQ_ROT");
        A.rot = e ? atoi(e) : 1;
    }
    const size_t fixed = VQ_WORDS * 4 + PACX_MAX_BANDS * 8 + (PACX_MAX_BANDS + PACX_MAX_BANDS + 1 + 3) * 4 +
                         VQ_WAVES * 2 * VQ_DEPTH * 4;
    static_assert(fixed % 8 == 0, "the shapes behind the fixed part are doubles");
    const size_t smem = fixed + PACX_M_LONG * 8 + (size_t)V.scr_off[VQ_WAVES] * 8 + (size_t)VQ_WAVES * VQ_NODE_BYTES;
    const long long units = A.mixed ? n_cf * PACX_SUB : n_cf;
    /* the frame-level walk first; k_vq then takes the units it left (PACX_VQ_FRAME=0: k_vq alone) */
    const char *fe = getenv("PACX_VQ_FRAME");            /* 0: k_vq alone; 1 / unset: k_vq_frame; 2: k_vq_frame2 */
    A.redo = (fe && atoi(fe) == 0) ? 0 : 1;
    if (PACX_SUB * T.nb_short > VQF_VB || T.nb_long > VQF_VB)
        A.redo = 0;                             /* more bands than k_vq_frame's per-band arrays hold */
    /* PACX_VQ_FRAME=2: the second form of the frame-level walk (k_vq_frame2: in place, the leaves after the walk) where
       the band layout fits its one buffer.  Not the default: same bytes, 11 % fewer vector and 41 % fewer scalar
       instructions, and SLOWER (0.668-0.688 against 0.645 ms per vq128 step): the kernel is bound by the length of a
       unit's dependency chain at five workgroups per CU, not by issue, and the one-leaf-per-lane pass is one long
       chain on one wave where k_vq_frame codes the leaves on three waves beside the scalar stage */
    const bool form2 = (fe && atoi(fe) == 2) && V.reg_long_total <= VQ2_BUF && PACX_SUB * V.reg_short_total <= VQ2_BUF &&
                       2 * V.max_band <= VQ2_SCR && V.max_band <= 512;
    if (A.redo && stage != 2) {                 /* one workgroup per channel-frame, long or short */
        if (form2)
            hipLaunchKernelGGL(k_vq_frame2, dim3((unsigned)n_cf), dim3(64 * VQ_WAVES), (size_t)VQ2_SMEM, st, T, V, A);
        else if (T.guard)
            hipLaunchKernelGGL(k_vq_frame<true>, dim3((unsigned)n_cf), dim3(64 * VQ_WAVES), (size_t)VQF_SMEM, st, T, V, A);
        else
            hipLaunchKernelGGL(k_vq_frame<false>, dim3((unsigned)n_cf), dim3(64 * VQ_WAVES), (size_t)VQF_SMEM, st, T, V, A);
    }
    if (stage == 1)
        return;
    if (A.redo)
        hipLaunchKernelGGL(k_vq_redo, dim3((unsigned)(n_cf < 1024 ? n_cf : 1024)), dim3(64 * VQ_WAVES), smem, st, T, V, A);
    else
        hipLaunchKernelGGL(k_vq, dim3((unsigned)units), dim3(64 * VQ_WAVES), smem, st, T, V, A);
    if (A.mixed)
        hipLaunchKernelGGL(k_vq_join, dim3((unsigned)n_cf), dim3(64), 0, st, T, flags, n_ch, n_cf, status,
                           unit_words, unit_bits, payload, payload_stride, n_bytes, A.redo);
}

size_t pacx_vq_view_size(void) { return sizeof(VqView); }

/* sizes_long / sizes_short: vector dimension of every band as the coder sees it */
void pacx_vq_view_fill(void *dst, const uint64_t *n_tab, const uint64_t *p_tab, const int32_t *row_off,
                       const int32_t *k_of, const uint8_t *w_of, const double *half_log2, int l_max,
                       double log_mu1, const double *log2_tan, const int32_t *sizes_long, int nb_long,
                       const int32_t *sizes_short, int nb_short)
{
    VqView *v = (VqView *)dst;
    v->log2_tan = log2_tan;
    auto sort_desc = [](const int32_t *sz, int nb, uint8_t *order) {
        for (int i = 0; i < nb; ++i)
            order[i] = (uint8_t)i;
        for (int i = 1; i < nb; ++i)                      /* stable insertion sort, largest first */
            for (int j = i; j > 0 && sz[order[j]] > sz[order[j - 1]]; --j) {
                const uint8_t t = order[j];
                order[j] = order[j - 1];
                order[j - 1] = t;
            }
    };
    sort_desc(sizes_long, nb_long, v->order_long);
    sort_desc(sizes_short, nb_short, v->order_short);
    v->scr_off[0] = 0;
    for (int w = 0; w < VQ_WAVES; ++w) {
        int n = 1;
        if (w < nb_long)
            n = sizes_long[v->order_long[w]];
        if (w < nb_short && sizes_short[v->order_short[w]] > n)
            n = sizes_short[v->order_short[w]];
        /* tickets hand any later band to any wave */
        if (VQ_WAVES < nb_long && sizes_long[v->order_long[VQ_WAVES]] > n)
            n = sizes_long[v->order_long[VQ_WAVES]];
        if (VQ_WAVES < nb_short && sizes_short[v->order_short[VQ_WAVES]] > n)
            n = sizes_short[v->order_short[VQ_WAVES]];
        /* mid/side regions [2n + 4 depth] (the shape itself lives in the block's xs) */
        v->scr_off[w + 1] = v->scr_off[w] + ((2 * n + 4 * VQ_DEPTH + 1) & ~1);
    }
    auto regions = [](const int32_t *sz, int nb, unsigned short *reg, unsigned char *rlog) {
        int at = 0;
        for (int b = 0; b < nb; ++b) {
            int lg = 0;
            while ((1 << lg) < sz[b])
                ++lg;
            reg[b] = (unsigned short)at;
            rlog[b] = (unsigned char)lg;
            at += 1 << lg;
        }
        return at;
    };
    v->reg_long_total = regions(sizes_long, nb_long, v->reg_long, v->rlog_long);
    v->reg_short_total = regions(sizes_short, nb_short, v->reg_short, v->rlog_short);
    v->max_band = 1;
    for (int b = 0; b < nb_long; ++b)
        v->max_band = sizes_long[b] > v->max_band ? sizes_long[b] : v->max_band;
    for (int b = 0; b < nb_short; ++b)
        v->max_band = sizes_short[b] > v->max_band ? sizes_short[b] : v->max_band;
    v->n_tab = n_tab;
    v->p_tab = p_tab;
    v->row_off = row_off;
    v->k_of = k_of;
    v->w_of = w_of;
    v->half_log2 = half_log2;
    v->l_max = l_max;
    v->log_mu1 = log_mu1;
}
