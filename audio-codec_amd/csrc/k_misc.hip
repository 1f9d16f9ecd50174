/*
 * k_misc.hip -- function-level entry points of the replaced modules that are
 * not batched per frame: element-wise window multiply (coder/window.py), the
 * uniform quantiser / scale factor / mantissa with free parameters
 * (coder/quantize.py:14-36, 61-78, 99-125, 229-250) and BitAlloc with an
 * explicit budget and band layout (coder/bitalloc.py:62-121).  They exist so
 * that every public function of the five replaced modules has a GPU
 * implementation behind the same signature; the encode hot path uses the fused
 * kernels in k_mdct/k_psy/k_quant instead.
 */
#include "pacx_dev.h"

__global__ void k_window(const double *__restrict__ win, long long n_rows, int len,
                         const double *__restrict__ x, double *__restrict__ y)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rows * len)
        y[i] = win[i % len] * x[i];
}

/* op 0: vQuantizeUniform(x, a)            -> sign<<(a-1) | magnitude
 * op 1: ScaleFactor(x, a, b)              (a = nScaleBits, b = nMantBits)
 * op 2: vMantissa(x, scale, a, b)
 * op 3: MantissaFP(x, scale, a, b) (coder/quantize.py:130-150) */
__global__ void k_quant_elem(int op, long long n, const double *__restrict__ x, int scale, int a, int b,
                             int64_t *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const double v = x[i];
    if (op == 0) {
        const int64_t mag = pacx_quant_mag(fabs(v), a);
        out[i] = (v < 0.0 ? ((int64_t)1 << (a - 1)) : 0) + mag;
    } else if (op == 1) {
        out[i] = pacx_scale_factor(fabs(v), a, b);
    } else if (op == 2) {
        out[i] = pacx_mantissa(v, scale, a, b);
    } else {
        out[i] = pacx_mantissa_fp(v, scale, a, b);
    }
}

/* decode-side element ops (quantize.py mirrors): op 0: vDequantizeUniform(codes, a) (coder/quantize.py:82-95);
 * op 1: vDequantize(scale, mantissas, a = nScaleBits, b = nMantBits) (coder/quantize.py:254-274);
 * op 2: DequantizeFP(scale, mantissa, a, b) (coder/quantize.py:154-175) */
__global__ void k_dequant_elem(int op, long long n, const int64_t *__restrict__ codes, int scale, int a, int b,
                               double *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    out[i] = (op == 0) ? pacx_dequant_uniform(codes[i], a)
           : (op == 1) ? pacx_dequantize(codes[i], scale, a, b) : pacx_dequantize_fp(codes[i], scale, a, b);
}

/* mdct.py for ANY block split a + b (coder/mdct.py:14-77: MDCTslow, MDCT, IMDCT -- the reference's own
 * self-test runs a = b = 4 and a = b = 6, coder/mdct.py:86-107): the defining sums, one output per thread,
 *   forward  X[k] = 2/N sum_n x[n] cos(2 pi/N (n + n0)(k + 1/2)),  n0 = (b + 1)/2,  k < N/2
 *   inverse  x[n] = 2   sum_k X[k] cos(2 pi/N (n + n0)(k + 1/2)),                    n < N
 * with the phase reduced in integers: (2n + b + 1)(2k + 1) mod 4N, cos(2 pi p / 4N) = cospi(p / 2N), so the
 * cosine's argument is exact to the last place at any size.  The codec's own sizes (a = b = 1024 / 128) take
 * the FFT kernels; this one is O(N^2) and serves the function-level mirror for the sizes they do not. */
__global__ void k_mdct_direct(long long n_rows, int a, int b, int inverse, const double *__restrict__ x,
                              double *__restrict__ y)
{
    const int N = a + b, H = N / 2;
    const int n_out = inverse ? N : H, n_in = inverse ? H : N;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * n_out)
        return;
    const long long row = i / n_out;
    const int o = (int)(i % n_out);
    const double *__restrict__ in = x + row * n_in;
    const long long four_n = 4ll * N;
    double acc = 0.0;
    for (int j = 0; j < n_in; ++j) {
        const int n = inverse ? o : j, k = inverse ? j : o;
        const long long p = ((2ll * n + b + 1) * (2ll * k + 1)) % four_n;
        acc += in[j] * cospi((double)p / (double)(2 * N));
    }
    y[i] = inverse ? 2.0 * acc : acc * 2.0 / (double)N;
}

__global__ void k_bitalloc_generic(long long n, int nb, const int32_t *__restrict__ n_lines,
                                   const double *__restrict__ budget, int max_mant,
                                   const double *__restrict__ smr, int32_t *__restrict__ bits_out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    double s[PACX_MAX_BANDS];
    int32_t nl[PACX_MAX_BANDS], bits[PACX_MAX_BANDS];
    for (int b = 0; b < nb; ++b) {
        s[b] = smr[i * nb + b];
        nl[b] = n_lines[b];
    }
    int cap = 0;
    pacx_bit_alloc(budget[i], max_mant, nb, nl, s, bits, &cap);
    for (int b = 0; b < nb; ++b)
        bits_out[i * nb + b] = bits[b];
}

/* np.mean(np.abs(np.take(block, arange(upto), axis=1))) of coder/detect_transients.py:14 in NumPy's order, by ONE
 * lane: the [nCh, upto] array is summed flattened (channel-major; columns past the hop are the look-ahead
 * zeros) by NumPy's pairwise scheme -- up to 128 elements on eight interleaved accumulators folded
 * ((0+1)+(2+3))+((4+5)+(6+7)) plus a scalar tail, longer runs cut at n/2 rounded down to a multiple of 8 --
 * and divided by the element count.  The recursion runs on an explicit stack in LDS (depth <= 9 for 8 x 2048
 * elements).  Only k_transient's exact-tie path calls this. */
template <typename At>
__device__ __forceinline__ double np_pairwise_mean(long long n_el_ll, At at)
{
    __shared__ int st_lo[12], st_n[12], st_state[12];
    __shared__ double st_left[12];
    const long long n_el = n_el_ll;
    auto block_sum = [&](int lo, int n) {
        if (n < 8) {
            double r = -0.0;
            for (int i = 0; i < n; ++i)
                r = r + at(lo + i);
            return r;
        }
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            r[j] = at(lo + j);
        const int n8 = n - (n & 7);
        for (int i = 8; i < n8; i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                r[j] = r[j] + at(lo + i + j);
        }
        double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (int i = n8; i < n; ++i)
            s = s + at(lo + i);
        return s;
    };
    int top = 0;
    st_lo[0] = 0;
    st_n[0] = (int)n_el;
    st_state[0] = 0;
    double result = 0.0;
    while (top >= 0) {
        const int lo = st_lo[top], n = st_n[top], state = st_state[top];
        int cut = n / 2;
        cut -= cut % 8;
        if (n <= 128) {
            result = block_sum(lo, n);
            --top;
        } else if (state == 0) {
            st_state[top] = 1;
            ++top;
            st_lo[top] = lo;
            st_n[top] = cut;
            st_state[top] = 0;
        } else if (state == 1) {
            st_left[top] = result;
            st_state[top] = 2;
            ++top;
            st_lo[top] = lo + cut;
            st_n[top] = n - cut;
            st_state[top] = 0;
        } else {
            result = st_left[top] + result;
            --top;
        }
    }
    return result / (double)n_el;
}

/* the int16 hop of k_transient: |x| = 2 (|c| & 32767) / 65535, zeros past the hop */
__device__ __forceinline__ double transient_np_mean(const PacxPcmView &in, const short *base, int n_ch, int upto, int hop)
{
    return np_pairwise_mean((long long)n_ch * upto, [&](int i) {
        const int ch = i / upto, col = i - ch * upto;
        if (col >= hop)
            return 0.0;
        const int c = base[(long long)ch * in.ch_stride + (long long)col * in.samp_stride];
        return pacx_pcm16_to_f64((c < 0 ? -c : c) & 32767);
    });
}

/* Block-switching caller: parTransientDetect
 * (coder/detect_transients.py:5-23) on (hop || 1024 zeros), the only input the
 * reference's driver ever gives it (coder/pacfile.py:728-732).  One wave per hop:
 * per channel the peak |x| and its first position, then the mean of |x| over the
 * first min(max_ch(argmax)+500, 2048) columns of BOTH channels (zeros past 1024),
 * transient = any(peak / avg > 4.5); avg == 0 -> not a transient.
 * |x| = 2 (|c| & 32767) / 65535; the sum is taken exactly in integers, which decides peak / avg > 4.5
 * exactly; at an exact tie the reference's rounding decides and its float mean is redone in its order. */
__global__ __launch_bounds__(64) void k_transient(PacxPcmView in, long long n_hops, int hop,
                                                  uint8_t *__restrict__ transient)
{
    const int lane = threadIdx.x;
    const long long h = blockIdx.x;
    if (h >= n_hops)
        return;
    const short *base = (const short *)in.base + h * in.frame_stride;
    const int n_ch = in.n_ch < 8 ? in.n_ch : 8;
    int upto = 0;
    int peak_c[8];
    long long total = 0;
    const bool fast = hop == 1024 && in.samp_stride == 1 && n_ch <= 2 && (in.frame_stride & 7) == 0 &&
                      (in.ch_stride & 7) == 0 && ((uintptr_t)in.base & 15) == 0;
    if (fast) {
        /* unit-stride hops of 1024 samples: 16 samples per lane and channel in two 16-byte loads,
           kept in registers for both passes (lane l owns samples 8 l + 512 j .. + 7, j = 0, 1) */
        int mag[2][16];
        for (int ch = 0; ch < n_ch; ++ch) {
            const int4 *src = (const int4 *)(base + (long long)ch * in.ch_stride);
            const int4 q[2] = {src[lane], src[lane + 64]};
            int best = -1, where = 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int w[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int c = (k & 1) ? (w[k >> 1] >> 16) : (int)(short)(w[k >> 1] & 0xFFFF);
                    const int m = (c < 0 ? -c : c) & 32767;
                    mag[ch][8 * j + k] = m;
                    if (m > best) { best = m; where = 8 * lane + 512 * j + k; }     /* ascending inside the lane */
                }
            }
            for (int off = 32; off > 0; off >>= 1) {      /* wave arg-max, lowest index wins ties */
                const int ob = __shfl_xor(best, off, 64), ow = __shfl_xor(where, off, 64);
                if (ob > best || (ob == best && ow < where)) { best = ob; where = ow; }
            }
            peak_c[ch] = best;
            upto = max(upto, where + 500);
        }
        if (upto > 2 * hop)
            upto = 2 * hop;
        const int cols = upto < hop ? upto : hop;
        for (int ch = 0; ch < n_ch; ++ch) {
            int part = 0;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    part += (8 * lane + 512 * j + k < cols) ? mag[ch][8 * j + k] : 0;
            for (int off = 32; off > 0; off >>= 1)
                part += __shfl_xor(part, off, 64);
            total += part;
        }
    } else {
    /* pass 1: per channel the peak magnitude and its first position */
    for (int ch = 0; ch < n_ch; ++ch) {
        const short *src = base + (long long)ch * in.ch_stride;
        int best = -1, where = 0;
        for (int i = lane; i < hop; i += 64) {           /* ascending i per lane: first maximum kept */
            const int c = src[(long long)i * in.samp_stride];
            const int mag = (c < 0 ? -c : c) & 32767;
            if (mag > best) { best = mag; where = i; }
        }
        for (int off = 32; off > 0; off >>= 1) {          /* wave arg-max, lowest index wins ties */
            const int ob = __shfl_xor(best, off, 64), ow = __shfl_xor(where, off, 64);
            if (ob > best || (ob == best && ow < where)) { best = ob; where = ow; }
        }
        peak_c[ch] = best;
        upto = max(upto, where + 500);
    }
    if (upto > 2 * hop)
        upto = 2 * hop;
    const int cols = upto < hop ? upto : hop;            /* columns that are not padding zeros */
    /* pass 2: sum of |c| over the first `cols` samples of every channel (exact) */
    for (int ch = 0; ch < n_ch; ++ch) {
        const short *src = base + (long long)ch * in.ch_stride;
        int part = 0;
        for (int i = lane; i < cols; i += 64) {
            const int c = src[(long long)i * in.samp_stride];
            part += (c < 0 ? -c : c) & 32767;
        }
        for (int off = 32; off > 0; off >>= 1)
            part += __shfl_xor(part, off, 64);
        total += part;
    }
    }
    if (lane == 0) {
        bool tr = false;
        if (total > 0) {
            /* peak / avg = P n / S in integers (P the peak code, S the sum of the codes, n = nCh * upto): against
               4.5 that is 2 P n against 9 S.  Unequal integers differ by at least one part in 1.2e9, far above any
               rounding of the reference's float arithmetic, so the integer comparison IS its decision -- except
               at an exact tie, where the reference's par is 4.5 up to the rounding of its pairwise float mean
               and lands on either side: then (rare: quiet passages of small codes) that mean is redone here in
               NumPy's order */
            const long long n_el = (long long)in.n_ch * upto;
            bool tie = false;
            for (int ch = 0; ch < n_ch; ++ch) {
                const long long lhs = 2ll * peak_c[ch] * n_el, rhs = 9ll * total;
                tr = tr || lhs > rhs;
                tie = tie || lhs == rhs;
            }
            if (tie && !tr) {
                const double avg = transient_np_mean(in, base, n_ch, upto, hop);
                for (int ch = 0; ch < n_ch; ++ch)
                    tr = tr || (pacx_pcm16_to_f64(peak_c[ch]) / avg > 4.5);
            }
        }
        transient[h] = tr ? 1 : 0;
    }
}

/* detect_transients.parTransientDetect(block, thresh, axis=1) for ANY float64 block (the function-level
 * mirror; the encode path uses k_transient on the int16 hops): blocks [n_blocks][n_ch][n].  One wave per block:
 * per channel the peak |x| and its first position, cols = min(max(argmax) + 500, n), the mean of |x| over the
 * first cols columns of all channels in NumPy's pairwise order (one lane), then any(peak / avg > thresh).
 * out: 2 where avg == 0 (the reference returns the int 0 there), else 1 / 0. */
__global__ __launch_bounds__(64) void k_transient_f64(long long n_blocks, int n_ch, int n, const double *__restrict__ blocks,
                                                      double thresh, uint8_t *__restrict__ out)
{
    const int lane = threadIdx.x;
    const long long blk = blockIdx.x;
    if (blk >= n_blocks)
        return;
    const double *__restrict__ x = blocks + blk * (long long)n_ch * n;
    int upto = 0;
    for (int ch = 0; ch < n_ch; ++ch) {
        double best = -1.0;
        int where = 0;
        for (int i = lane; i < n; i += 64) {
            const double m = fabs(x[(long long)ch * n + i]);
            if (m > best) { best = m; where = i; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double ob = __shfl_xor(best, off, 64);
            const int ow = __shfl_xor(where, off, 64);
            if (ob > best || (ob == best && ow < where)) { best = ob; where = ow; }
        }
        upto = max(upto, where + 500);
    }
    const int cols = upto < n ? upto : n;
    if (lane == 0) {
        const double avg = np_pairwise_mean((long long)n_ch * cols, [&](int i) {
            const int ch = i / cols, col = i - ch * cols;
            return fabs(x[(long long)ch * n + col]);
        });
        int res = 0;
        if (avg == 0.0) {
            res = 2;
        } else {
            for (int ch = 0; ch < n_ch; ++ch) {
                double best = 0.0;
                for (int i = 0; i < n; ++i)
                    best = fmax(best, fabs(x[(long long)ch * n + i]));
                if (best / avg > thresh)
                    res = 1;
            }
        }
        out[blk] = (uint8_t)res;
    }
}

/* flags of every written hop of the driver loop (coder/pacfile.py:717-741) plus
 * the Close block: frame f gets last = T[f-2], cur = T[f-1], next = T[f]
 * (T[n_hops] = 0: the pass after EOF), the final zero block (0,0,0). */
__global__ void k_stream_flags(const uint8_t *__restrict__ transient, long long n_hops,
                               uint8_t *__restrict__ flags)
{
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_hops + 2)
        return;
    unsigned v = 0;
    if (f <= n_hops) {
        if (f >= 2 && transient[f - 2]) v |= 1u;
        if (f >= 1 && transient[f - 1]) v |= 2u;
        if (f < n_hops && transient[f]) v |= 4u;
    }
    flags[f] = (uint8_t)v;
}

/* Mixed streams: channel-frame indices of the long-coded and of the short-coded
 * frames, compacted (any order) so that persistent kernels can walk only the
 * frames they own and stay balanced.  counts[0] = long cf, counts[1] = short cf;
 * both must be zero on entry. */
__global__ __launch_bounds__(256) void k_frame_lists(const uint8_t *__restrict__ flags, long long n_frames,
                                                    int n_ch, int32_t *__restrict__ list_long,
                                                    int32_t *__restrict__ list_short,
                                                    int32_t *__restrict__ counts)
{
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool in_range = f < n_frames;
    const bool is_short = in_range && (flags[f] & 2u);
    const bool is_long = in_range && !is_short;
    const unsigned long long ms = __ballot(is_short), ml = __ballot(is_long);
    int base_s = 0, base_l = 0;
    if (lane == 0) {
        if (ms)
            base_s = atomicAdd(&counts[1], __popcll(ms) * n_ch);
        if (ml)
            base_l = atomicAdd(&counts[0], __popcll(ml) * n_ch);
    }
    base_s = __shfl(base_s, 0, 64);
    base_l = __shfl(base_l, 0, 64);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (is_short) {
        const int at = base_s + __popcll(ms & below) * n_ch;
        for (int c = 0; c < n_ch; ++c)
            list_short[at + c] = (int32_t)(f * n_ch + c);
    }
    if (is_long) {
        const int at = base_l + __popcll(ml & below) * n_ch;
        for (int c = 0; c < n_ch; ++c)
            list_long[at + c] = (int32_t)(f * n_ch + c);
    }
}

/* the same lists for batches of up to a few thousand frames, by ONE workgroup: no counters to
 * zero first (a memset launch), no atomics, and the lists come out in frame order.  The kernel sits at
 * the head of the long-coded chain of a block-switched batch, so it is built around ONE round trip to
 * memory and ONE barrier: a thread owns up to 16 consecutive frames, reads their flags together, and the
 * counts are scanned over the wave (shuffles) and over the 16 waves (LDS). */
#define LISTS_SMALL_MAX 16384
#define LISTS_SMALL_THREADS 256        /* four waves: a workgroup of sixteen waited for a whole CU's worth of free wave slots
                                          beside the persistent kernels of the other step in flight (40 us for 16 us of work) */
__global__ __launch_bounds__(LISTS_SMALL_THREADS) void k_frame_lists_small(const uint8_t *__restrict__ flags, long long n_frames,
                                                            int n_ch, int32_t *__restrict__ list_long,
                                                            int32_t *__restrict__ list_short,
                                                            int32_t *__restrict__ counts)
{
    constexpr int NW = LISTS_SMALL_THREADS / 64, PER_MAX = LISTS_SMALL_MAX / LISTS_SMALL_THREADS;     /* <= 64 frames per thread */
    static_assert(PER_MAX <= 64, "one 64-bit mask per thread");
    __shared__ int ws[NW], wl[NW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (int)((n_frames + LISTS_SMALL_THREADS - 1) / LISTS_SMALL_THREADS);
    const long long f0 = (long long)tid * per;
    unsigned long long valid = 0, shorts = 0;              /* bit j: frame f0 + j */
    for (int j = 0; j < per; ++j) {
        if (f0 + j < n_frames) {
            valid |= 1ull << j;
            if (flags[f0 + j] & 2u)
                shorts |= 1ull << j;
        }
    }
    const int ns = __popcll(shorts), nl = __popcll(valid & ~shorts);
    int ss = ns, sl = nl;                                  /* inclusive scans over the wave */
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int a = __shfl_up(ss, off, 64), b = __shfl_up(sl, off, 64);
        if (lane >= off) {
            ss += a;
            sl += b;
        }
    }
    if (lane == 63) {
        ws[wv] = ss;
        wl[wv] = sl;
    }
    __syncthreads();
    int at_s = ss - ns, at_l = sl - nl, all_s = 0, all_l = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        at_s += w < wv ? ws[w] : 0;
        at_l += w < wv ? wl[w] : 0;
        all_s += ws[w];
        all_l += wl[w];
    }
    for (int j = 0; j < per; ++j) {
        if (!((valid >> j) & 1ull))
            break;
        const long long f = f0 + j;
        if ((shorts >> j) & 1ull) {
            for (int c = 0; c < n_ch; ++c)
                list_short[at_s * n_ch + c] = (int32_t)(f * n_ch + c);
            ++at_s;
        } else {
            for (int c = 0; c < n_ch; ++c)
                list_long[at_l * n_ch + c] = (int32_t)(f * n_ch + c);
            ++at_l;
        }
    }
    if (tid == 0) {
        counts[0] = all_l * n_ch;
        counts[1] = all_s * n_ch;
    }
}

void pacx_launch_frame_lists(const uint8_t *flags, long long n_frames, int n_ch, int32_t *list_long,
                             int32_t *list_short, int32_t *counts, hipStream_t st)
{
    if (n_frames <= 0)
        return;
    if (n_frames <= LISTS_SMALL_MAX) {
        hipLaunchKernelGGL(k_frame_lists_small, dim3(1), dim3(LISTS_SMALL_THREADS), 0, st, flags, n_frames, n_ch, list_long,
                           list_short, counts);
        return;
    }
    (void)hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), st);
    hipLaunchKernelGGL(k_frame_lists, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, st, flags, n_frames,
                       n_ch, list_long, list_short, counts);
}

void pacx_launch_transient_f64(long long n_blocks, int n_ch, int n, const double *blocks, double thresh, uint8_t *out,
                               hipStream_t st)
{
    if (n_blocks > 0)
        hipLaunchKernelGGL(k_transient_f64, dim3((unsigned)n_blocks), dim3(64), 0, st, n_blocks, n_ch, n, blocks, thresh,
                           out);
}

void pacx_launch_transient(const PacxPcmView &in, long long n_hops, int hop, uint8_t *transient,
                           uint8_t *flags, hipStream_t st)
{
    if (n_hops > 0)
        hipLaunchKernelGGL(k_transient, dim3((unsigned)n_hops), dim3(64), 0, st, in, n_hops, hop, transient);
    if (flags)
        hipLaunchKernelGGL(k_stream_flags, dim3((unsigned)((n_hops + 2 + 255) / 256)), dim3(256), 0, st,
                           transient, n_hops, flags);
}

void pacx_launch_window(const double *win, long long n_rows, int len, const double *x, double *y,
                        hipStream_t st)
{
    const long long n = n_rows * len;
    if (n > 0)
        hipLaunchKernelGGL(k_window, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, win, n_rows, len, x, y);
}

void pacx_launch_quant_elem(int op, long long n, const double *x, int scale, int a, int b, int64_t *out,
                            hipStream_t st)
{
    if (n > 0)
        hipLaunchKernelGGL(k_quant_elem, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, op, n, x, scale,
                           a, b, out);
}

void pacx_launch_dequant_elem(int op, long long n, const int64_t *codes, int scale, int a, int b, double *out,
                              hipStream_t st)
{
    if (n > 0)
        hipLaunchKernelGGL(k_dequant_elem, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, op, n, codes, scale,
                           a, b, out);
}

void pacx_launch_mdct_direct(long long n_rows, int a, int b, int inverse, const double *x, double *y, hipStream_t st)
{
    const long long n = n_rows * (inverse ? (a + b) : (a + b) / 2);
    if (n > 0)
        hipLaunchKernelGGL(k_mdct_direct, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n_rows, a, b, inverse,
                           x, y);
}

void pacx_launch_bitalloc_generic(long long n, int nb, const int32_t *n_lines, const double *budget,
                                  int max_mant, const double *smr, int32_t *bits, hipStream_t st)
{
    if (n > 0)
        hipLaunchKernelGGL(k_bitalloc_generic, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, n, nb, n_lines,
                           budget, max_mant, smr, bits);
}
