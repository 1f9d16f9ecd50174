/*
 * k_psy.hip -- psychoacoustic side chain (reference: coder/psychoac.py).
 *
 *   k_side_long / k_side_short : Hann window -> real FFT -> intensity ->
 *       strict-local-maximum peak pick -> (Bark, SPL, upper slope) per tonal
 *       masker          [getMaskedThreshold :171-179, estimate_peaks :308-329,
 *                        Masker.__init__ :61-68]
 *   k_mask<M>                 : masked threshold at the MDCT lines and per-band
 *       SMR             [Masker.vIntensityAtBark :88-96, getMaskedThreshold
 *                        :189-215, CalcSMRs :246-291]
 *
 * Behaviour notes kept from the reference: only tonal maskers and the threshold
 * in quiet reach the result (its noise-masker loop, :197-211, discards what it
 * computes); peak energy is the sum of bins f-1 and f; the data window is the
 * n+1/2 Hann while the power normalisation uses np.hanning.
 *
 * The max over maskers is taken in the dB domain and the SPL(Intensity(.))
 * round trip (:192) is applied once per line to the winner: the round trip is
 * monotone, so this equals the reference's max of per-masker round trips
 * (measured bit-identical on 520/520 frames).
 *
 * The 2N-point real FFT runs as two Q=N/4-point complex FFTs (wave_fft.h):
 *   z[m] = xh[2m] + j xh[2m+1];  E = FFT(z[0::2]), O = FFT(z[1::2]);
 *   Z[k] = E[k] + W_{N/2}^k O[k], Z[k+Q] = E[k] - W_{N/2}^k O[k];
 *   X[k] = (Z[k] + conj Z[N/2-k])/2 - (j/2) W_N^k (Z[k] - conj Z[N/2-k]).
 */
#include "pacx_dev.h"
#include "wave_fft.h"

template <int DT> struct PcmStage;
template <> struct PcmStage<0> {
    typedef short elem;
    static __device__ __forceinline__ double get(const short *s, int i) { return pacx_pcm16_to_f64(s[i]); }
};
template <> struct PcmStage<1> {
    typedef double elem;
    static __device__ __forceinline__ double get(const double *s, int i) { return s[i]; }
};

template <int DT, bool FAST>
__device__ __forceinline__ void stage_samples(typename PcmStage<DT>::elem *dst, const PacxPcmView &in,
                                              long long cf, int first, int count, int lane)
{
    typedef typename PcmStage<DT>::elem E;
    const long long f = cf / in.n_ch;
    const int ch = (int)(cf - f * in.n_ch);
    const E *src = (const E *)in.base + f * in.frame_stride + ch * in.ch_stride;
    if constexpr (FAST) {
        const int4 *s4 = (const int4 *)(src + first);
        int4 *d4 = (int4 *)dst;
        for (int i = lane; i < count / 8; i += 64)
            d4[i] = s4[i];
    } else {
        for (int i = lane; i < count; i += 64)
            dst[i] = src[(long long)(first + i) * in.samp_stride];
    }
}

/* intensity of real-FFT bin i from the packed complex spectrum Z (period P) */
__device__ __forceinline__ double bin_intensity(const cplx *Z, int i, int P, cplx w, double norm)
{
    const cplx a = Z[i & (P - 1)];
    const cplx bz = Z[(P - i) & (P - 1)];
    const cplx s = make_double2(a.x + bz.x, a.y - bz.y);      /* a + conj(b) */
    const cplx d = make_double2(a.x - bz.x, a.y + bz.y);      /* a - conj(b) */
    const cplx wd = c_mul(w, d);
    const double xr = 0.5 * (s.x + wd.y), xi = 0.5 * (s.y - wd.x);
    const double mag = hypot(xr, xi);                         /* abs(x_fft)  */
    return norm * (mag * mag);
}

/* one tonal masker from bins f-1, f (coder/psychoac.py:321-328, :61-68) */
__device__ __forceinline__ PacxPeak make_peak(double left, double centre, int f, double fstep)
{
    const double e = left + centre;
    const double spl = pacx_spl_scalar(e);
    const double avg = (((double)(f - 1) * fstep) * left + ((double)f * fstep) * centre) / e;
    PacxPeak p;
    p.z = pacx_bark(avg);
    p.spl = spl;
    p.slope = -27.0 + 0.367 * fmax(spl - 40.0, 0.0);
    return p;
}

/* ------------------------------------------------------------------ long */
template <int DT, bool FAST>
__global__ __launch_bounds__(64) void k_side_long(PacxTables T, PacxPcmView in,
                                                 const uint8_t *__restrict__ flags, long long n_cf,
                                                 int skip_cur, PacxPeak *__restrict__ peaks,
                                                 int32_t *__restrict__ n_peaks)
{
    typedef typename PcmStage<DT>::elem E;
    /* LDS lifetimes: raw (until the FFT inputs are in registers) then inten share
       one region; the FFT exchange tile then the packed spectrum Z share another */
    constexpr int RAW_BYTES = (int)sizeof(E) * PACX_N_LONG;
    constexpr int B_BYTES = RAW_BYTES > 1032 * 8 ? RAW_BYTES : 1032 * 8;
    __shared__ __attribute__((aligned(16))) cplx Z[1024];
    __shared__ __attribute__((aligned(16))) char regB[B_BYTES];
    cplx *tile = Z;
    double *inten = (double *)regB;
    E *raw = (E *)regB;
    const int lane = threadIdx.x;
    const long long cf = blockIdx.x;
    if (cf >= n_cf)
        return;
    const unsigned fl = flags ? flags[cf / in.n_ch] : 0u;
    if (skip_cur && (fl & 2u))
        return;

    stage_samples<DT, FAST>(raw, in, cf, 0, PACX_N_LONG, lane);
    __syncthreads();

    const double *__restrict__ hw = T.hann_long;
    cplx ev[8], od[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int i = 4 * (lane + 64 * n1);
        ev[n1] = make_double2(hw[i] * PcmStage<DT>::get(raw, i), hw[i + 1] * PcmStage<DT>::get(raw, i + 1));
        od[n1] = make_double2(hw[i + 2] * PcmStage<DT>::get(raw, i + 2), hw[i + 3] * PcmStage<DT>::get(raw, i + 3));
    }
    __syncthreads();                  /* raw fully consumed before anything reuses LDS */
    fft512(ev, tile, T.w512, lane);
    fft512(od, tile, T.w512, lane);
    __syncthreads();                  /* tile dead: Z takes its place */
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) {
        const int k = fft512_out_index(lane, k3);
        const cplx t = c_mul(T.w1024[k], od[k3]);
        Z[k] = c_add(ev[k3], t);
        Z[k + 512] = c_sub(ev[k3], t);
    }
    __syncthreads();
    for (int i = lane; i <= 1024; i += 64)
        inten[i] = bin_intensity(Z, i, 1024, T.w2048[i], T.norm_long);
    __syncthreads();

    PacxPeak *__restrict__ out = peaks + cf * PACX_MAX_PEAKS;
    int count = 0;
    for (int base = 0; base <= 1024; base += 64) {
        const int i = base + lane;
        bool pk = false;
        double c = 0.0, l = 0.0;
        if (i >= 1 && i <= 1024) {
            c = inten[i];
            l = inten[i - 1];
            pk = (c > l) && (i == 1024 || c > inten[i + 1]);
        }
        const unsigned long long m = __ballot(pk);
        if (pk) {
            const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
            out[pos] = make_peak(l, c, i, T.fstep_long);
        }
        count += __popcll(m);
    }
    if (lane == 0)
        n_peaks[cf * PACX_SUB] = count;
}

/* ----------------------------------------------------------------- short */
template <int DT, bool FAST>
__global__ __launch_bounds__(64) void k_side_short(PacxTables T, PacxPcmView in,
                                                  const uint8_t *__restrict__ flags, long long n_cf,
                                                  int only_cur, PacxPeak *__restrict__ peaks,
                                                  int32_t *__restrict__ n_peaks)
{
    typedef typename PcmStage<DT>::elem E;
    const int SPAN = PACX_N_SHORT + (PACX_SUB - 1) * PACX_M_SHORT;
    __shared__ __attribute__((aligned(16))) cplx tile[WFFT_TILE];
    __shared__ __attribute__((aligned(16))) cplx Z[PACX_SUB * 128];
    __shared__ __attribute__((aligned(16))) double inten[PACX_SUB * 130];
    __shared__ __attribute__((aligned(16))) E raw[SPAN];
    const int lane = threadIdx.x;
    const long long cf = blockIdx.x;
    if (cf >= n_cf)
        return;
    const unsigned fl = flags ? flags[cf / in.n_ch] : 2u;
    if (only_cur && !(fl & 2u))
        return;

    stage_samples<DT, FAST>(raw, in, cf, PACX_SHORT_FIRST, SPAN, lane);
    __syncthreads();

    const int g = lane >> 3, r = lane & 7;
    const E *sub = raw + g * PACX_M_SHORT;
    const double *__restrict__ hw = T.hann_short;
    cplx ev[8], od[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int i = 4 * (r + 8 * n1);
        ev[n1] = make_double2(hw[i] * PcmStage<DT>::get(sub, i), hw[i + 1] * PcmStage<DT>::get(sub, i + 1));
        od[n1] = make_double2(hw[i + 2] * PcmStage<DT>::get(sub, i + 2), hw[i + 3] * PcmStage<DT>::get(sub, i + 3));
    }
    fft64x8(ev, tile, T.w512, lane);
    fft64x8(od, tile, T.w512, lane);
    cplx *Zg = Z + g * 128;
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) {
        const int k = fft64_out_index(lane, k3);
        const cplx t = c_mul(T.w128[k], od[k3]);
        Zg[k] = c_add(ev[k3], t);
        Zg[k + 64] = c_sub(ev[k3], t);
    }
    __syncthreads();
    double *ig = inten + g * 130;
    for (int i = r; i <= 128; i += 8)
        ig[i] = bin_intensity(Zg, i, 128, T.w256[i], T.norm_short);
    __syncthreads();

    PacxPeak *__restrict__ out = peaks + cf * PACX_MAX_PEAKS + g * 64;
    int count = 0;
    for (int base = 0; base <= 128; base += 8) {
        const int i = base + r;
        bool pk = false;
        double c = 0.0, l = 0.0;
        if (i >= 1 && i <= 128) {
            c = ig[i];
            l = ig[i - 1];
            pk = (c > l) && (i == 128 || c > ig[i + 1]);
        }
        const unsigned m = (unsigned)((__ballot(pk) >> (8 * g)) & 0xFFull);
        if (pk) {
            const int pos = count + __popc(m & ((1u << r) - 1u));
            out[pos] = make_peak(l, c, i, T.fstep_short);
        }
        count += __popc(m);
    }
    if (r == 0)
        n_peaks[cf * PACX_SUB + g] = count;
}

/* ------------------------------------------------------ mask + per-band SMR */
/* One wave per unit: a long cf (M = 1024, 16 lines per lane) or one short
 * sub-block (M = 128, 2 lines per lane; unit = 8*cf + sub-block).
 *
 * Lines are walked in chunks of 64 consecutive lines (one per lane), i.e. one
 * contiguous Bark interval [zlo, zhi] per chunk.  For every chunk the maskers
 * are first screened 64 at a time (one masker per lane): a masker whose curve
 * cannot rise above (min threshold-in-quiet of the chunk - 0.01 dB) anywhere in
 * [zlo, zhi] cannot change max(quiet, SPL(Intensity(.))) there and is skipped;
 * the survivors (ballot mask) are evaluated for all 64 lines with exactly the
 * reference's operation order.  The screen is exact: spreading curves fall
 * monotonically away from the masker, and the SPL/Intensity round trip adds at
 * most 1.2e-5 dB at levels >= the lowest threshold in quiet (-5 dB).
 */
template <int M>
__global__ __launch_bounds__(64) void k_mask(PacxTables T, const uint8_t *__restrict__ flags, int n_ch,
                                            long long n_units, int mixed,
                                            const PacxPeak *__restrict__ peaks,
                                            const int32_t *__restrict__ n_peaks,
                                            const double *__restrict__ lines,
                                            double *__restrict__ smr, double *__restrict__ thr_out)
{
    constexpr bool SHORT = (M == PACX_M_SHORT);
    constexpr int PER = M / 64;
    constexpr int MAXP = SHORT ? 64 : PACX_MAX_PEAKS;
    constexpr int LDS_DOUBLES = (3 * MAXP > M) ? 3 * MAXP : M;
    __shared__ __attribute__((aligned(16))) double lds[LDS_DOUBLES];   /* maskers, later the SMR terms */
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
    const double *__restrict__ bark = SHORT ? T.bark_short : T.bark_long;
    const double *__restrict__ quiet = SHORT ? T.thresh_short : T.thresh_long;
    const PacxPeak *__restrict__ pk = peaks + cf * PACX_MAX_PEAKS + sb * 64;
    const int np = n_peaks[cf * PACX_SUB + sb];
    const double *__restrict__ x = lines + cf * PACX_M_LONG + sb * PACX_M_SHORT;
    PacxPeak *pks = (PacxPeak *)lds;
    for (int p = lane; p < np; p += 64)
        pks[p] = pk[p];
    __syncthreads();

    double dif[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int k = lane + 64 * j;
        const double zk = bark[k], qk = quiet[k];
        const double zlo = bark[64 * j], zhi = bark[64 * j + 63];
        double qmin = qk;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            qmin = fmin(qmin, __shfl_xor(qmin, off, 64));
        qmin -= 0.01;
        double best = -INFINITY;
        for (int pb = 0; pb < np; pb += 64) {
            bool live = false;
            if (pb + lane < np) {
                const PacxPeak q = pks[pb + lane];
                double ub = 0.0;                       /* best case of the spreading gain on [zlo, zhi] */
                if (q.z < zlo - 0.5)
                    ub = q.slope * ((zlo - q.z) - 0.5);
                else if (q.z > zhi + 0.5)
                    ub = -27.0 * ((q.z - zhi) - 0.5);
                live = (q.slope > 0.0) || ((q.spl - 16.0) + ub > qmin);
            }
            unsigned long long todo = __ballot(live);
            while (todo) {
                const int b = __builtin_ctzll(todo);
                todo &= todo - 1;
                const PacxPeak q = pks[pb + b];        /* wave-uniform address: LDS broadcast */
                const double dz = zk - q.z;
                const double a = fabs(dz) - 0.5;
                double gain = 0.0;
                if (dz < -0.5)
                    gain = -27.0 * a;
                else if (dz > 0.5)
                    gain = q.slope * a;
                best = fmax(best, (q.spl + gain) - 16.0);
            }
        }
        double thr = qk;
        if (best > -INFINITY) {
            const double inten = pow(10.0, (best - 96.0) / 10.0);
            thr = fmax(thr, pacx_spl_array(inten));
        }
        if (thr_out)
            thr_out[cf * PACX_M_LONG + sb * PACX_M_SHORT + k] = thr;
        const double v = x[k];
        dif[j] = pacx_spl_array((v * v) * 4.0) - thr;
    }
    __syncthreads();                                   /* maskers no longer needed */
#pragma unroll
    for (int j = 0; j < PER; ++j)
        lds[lane + 64 * j] = dif[j];
    __syncthreads();
    const int nb = SHORT ? T.nb_short : T.nb_long;
    const int32_t *__restrict__ lower = SHORT ? T.band_lower_short : T.band_lower_long;
    const int32_t *__restrict__ count = SHORT ? T.band_lines_short : T.band_lines_long;
    double *__restrict__ out = smr + cf * T.band_stride + sb * T.nb_short;
    for (int b = 0; b < nb; ++b) {
        const int lo = lower[b], hi = lo + count[b];
        double m = -INFINITY;
        for (int k = lo + lane; k < hi; k += 64)
            m = fmax(m, lds[k]);
        m = wave_max(m);
        if (lane == 0)
            out[b] = m;
    }
}

/* ------------------------------------------------------------- launchers */
template <int DT, bool FAST>
static void launch_side(const PacxTables &T, const PacxPcmView &in, const uint8_t *flags,
                        long long n_cf, int short_blocks, int mixed, PacxPeak *peaks,
                        int32_t *n_peaks, hipStream_t st)
{
    const dim3 grid((unsigned)n_cf), block(64);
    if (!short_blocks || mixed)
        hipLaunchKernelGGL((k_side_long<DT, FAST>), grid, block, 0, st, T, in, flags, n_cf, mixed, peaks, n_peaks);
    if (short_blocks || mixed)
        hipLaunchKernelGGL((k_side_short<DT, FAST>), grid, block, 0, st, T, in, flags, n_cf, mixed, peaks, n_peaks);
}

void pacx_launch_side(const PacxTables &T, const PacxPcmView &in, int dtype, int fast,
                      const uint8_t *flags, long long n_cf, int short_blocks, int mixed,
                      PacxPeak *peaks, int32_t *n_peaks, hipStream_t st)
{
    if (n_cf <= 0)
        return;
    if (dtype == 0 && fast)
        launch_side<0, true>(T, in, flags, n_cf, short_blocks, mixed, peaks, n_peaks, st);
    else if (dtype == 0)
        launch_side<0, false>(T, in, flags, n_cf, short_blocks, mixed, peaks, n_peaks, st);
    else
        launch_side<1, false>(T, in, flags, n_cf, short_blocks, mixed, peaks, n_peaks, st);
}

void pacx_launch_mask(const PacxTables &T, const uint8_t *flags, int n_ch, long long n_cf,
                      int short_blocks, int mixed, const PacxPeak *peaks, const int32_t *n_peaks,
                      const double *lines, double *smr, double *thr_out, hipStream_t st)
{
    if (n_cf <= 0)
        return;
    if (!short_blocks || mixed)
        hipLaunchKernelGGL((k_mask<PACX_M_LONG>), dim3((unsigned)n_cf), dim3(64), 0, st, T, flags, n_ch,
                           n_cf, mixed, peaks, n_peaks, lines, smr, thr_out);
    if (short_blocks || mixed)
        hipLaunchKernelGGL((k_mask<PACX_M_SHORT>), dim3((unsigned)(n_cf * PACX_SUB)), dim3(64), 0, st, T,
                           flags, n_ch, n_cf * PACX_SUB, mixed, peaks, n_peaks, lines, smr, thr_out);
}
