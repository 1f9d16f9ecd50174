/*
 * k_mdct.hip -- window + MDCT kernels (reference: coder/window.py, coder/mdct.py,
 * called at coder/codec.py:303-305; overall scale coder/codec.py:308-310).
 *
 * One channel-frame per wave64.  The MDCT of a windowed block x[0..N) with
 * M = N/2 lines is computed through ONE complex FFT of Q = N/4 points
 * (verified there against mdct.MDCT):
 *     fold:  u[n]   = -x[3Q-1-n] - x[3Q+n],  u[Q+n] = x[n] - x[M-1-n]
 *     pre :  t[n]   = (u[2n] + j u[M-1-2n]) * d[n],  d[n] = exp(-j pi (8n+1)/(8M))
 *     FFT :  T      = FFT_Q(t)
 *     post:  y[k]   = T[k] * d[k] * (2/N);  X[2k] = Re y[k], X[M-1-2k] = -Im y[k]
 * (the single table d serves both twiddles: d[n] d[k] = exp(-j pi/(4M)) W_{2M}^{n+k}).
 * The reference instead runs a full N-point FFT; results agree to ~2e-13 of the
 * block maximum, which leaves every integer code downstream unchanged
 * (SURVEY.md section 7 step 5).
 *
 * Data movement per cf (long, int16 fast path): 4 x 16-byte coalesced loads per
 * lane of PCM -> LDS (4 KB), two LDS exchanges of 8 KB inside the FFT, 1024
 * float64 lines out = 2 KB (new hop) + 8 KB algorithmic HBM bytes.
 */
#include "pacx_dev.h"
#include "wave_fft.h"
#include "pcm_stage.h"


/* ------------------------------------------------------------------ long */
template <int DT, bool FAST>
__global__ __launch_bounds__(64) void k_mdct_long(PacxTables T, PacxPcmView in,
                                                 const uint8_t *__restrict__ flags, long long n_cf,
                                                 int skip_cur, int prewin, double *__restrict__ lines,
                                                 int32_t *__restrict__ scale_out, int scale_stride)
{
    typedef typename PcmStage<DT>::elem E;
    __shared__ __attribute__((aligned(16))) cplx tile[WFFT_TILE];
    __shared__ __attribute__((aligned(16))) E raw[PACX_N_LONG];
    const int lane = threadIdx.x;
    const long long cf = blockIdx.x;
    if (cf >= n_cf)
        return;
    const unsigned fl = flags ? flags[cf / in.n_ch] : 0u;
    if ((skip_cur & 1) && (fl & 2u))
        return;
    if ((skip_cur & 2) && pacx_window_kind(fl) == 0)
        return;                              /* sine frames were done by k_mdct_long_v2 */
    /* prewin: 0 = the window the flags select, 1 = none (mdct.MDCT on windowed data),
       2 = KBDWindow (coder/window.py:45-57) */
    const double *__restrict__ w = prewin == 1 ? T.ones : prewin == 2 ? T.kbd_long
                                                : T.win_long + pacx_window_kind(fl) * PACX_N_LONG;

    stage_samples<DT, FAST>(raw, in, cf, 0, PACX_N_LONG, lane);
    __syncthreads();

    const int Q = PACX_N_LONG / 4, M = PACX_M_LONG;
    cplx v[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int n = lane + 64 * n1;
        double re, im;
        if (n1 < 4) {            /* n < Q/2 */
            const int i0 = 3 * Q - 1 - 2 * n, i1 = 3 * Q + 2 * n, i2 = Q - 1 - 2 * n, i3 = Q + 2 * n;
            re = -(w[i0] * PcmStage<DT>::get(raw, i0)) - w[i1] * PcmStage<DT>::get(raw, i1);
            im = w[i2] * PcmStage<DT>::get(raw, i2) - w[i3] * PcmStage<DT>::get(raw, i3);
        } else {
            const int m = 2 * n - Q;
            const int i0 = m, i1 = M - 1 - m, i2 = 2 * Q + m, i3 = 4 * Q - 1 - m;
            re = w[i0] * PcmStage<DT>::get(raw, i0) - w[i1] * PcmStage<DT>::get(raw, i1);
            im = -(w[i2] * PcmStage<DT>::get(raw, i2)) - w[i3] * PcmStage<DT>::get(raw, i3);
        }
        v[n1] = c_mul(make_double2(re, im), T.tw_long[n]);
    }

    fft512(v, tile, T.w512, lane);

    const double s = 2.0 / PACX_N_LONG;      /* 2^-10, exact */
    double *__restrict__ out = lines + cf * PACX_M_LONG;
    double mx = 0.0;
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) {
        const int k = fft512_out_index(lane, k3);
        const cplx y = c_mul(v[k3], T.tw_long[k]);
        const double a = y.x * s, b = -(y.y * s);
        out[2 * k] = a;
        out[M - 1 - 2 * k] = b;
        mx = fmax(mx, fmax(fabs(a), fabs(b)));
    }
    if (scale_out) {
        mx = wave_max(mx);
        if (lane == 0)
            scale_out[cf * scale_stride] = pacx_scale_factor(mx, T.n_scale_bits, 5);
    }
}

/* ----------------------------------------------------------------- short */
/* 8 sub-blocks of 256 samples at n = 448 + 128 g (coder/pacfile.py:526-527),
 * sine window, 128 lines each: lane = 8 g + r works on sub-block g. */
template <int DT, bool FAST>
__global__ __launch_bounds__(64) void k_mdct_short(PacxTables T, PacxPcmView in,
                                                  const uint8_t *__restrict__ flags, long long n_cf,
                                                  int only_cur, int prewin, double *__restrict__ lines,
                                                  int32_t *__restrict__ scale_out,
                                                  uint32_t *__restrict__ status)
{
    typedef typename PcmStage<DT>::elem E;
    const int SPAN = PACX_N_SHORT + (PACX_SUB - 1) * PACX_M_SHORT;     /* 1152 samples */
    /* the raw samples and the FFT exchange tile follow one another in time and share their LDS
       (the windowed, folded samples are in registers before the first exchange): 13.3 KB per wave
       instead of 15.6 -- this one-wave kernel is bound by how many of its waves fit a CU */
    constexpr int RT_BYTES = (int)sizeof(cplx) * WFFT_TILE > (int)sizeof(E) * SPAN ? (int)sizeof(cplx) * WFFT_TILE
                                                                                 : (int)sizeof(E) * SPAN;
    __shared__ __attribute__((aligned(16))) char raw_tile[RT_BYTES];
    cplx *tile = (cplx *)raw_tile;
    E *raw = (E *)raw_tile;
    const int lane = threadIdx.x;
    const long long cf = blockIdx.x;
    if (cf >= n_cf)
        return;
    const unsigned fl = flags ? flags[cf / in.n_ch] : 2u;
    if (only_cur && !(fl & 2u))
        return;
    const double *__restrict__ wg = prewin == 1 ? T.ones : prewin == 2 ? T.kbd_short : T.win_short;
    /* the block's tables go to LDS with the PCM, all in ONE round trip to L2: read from global
       memory where they are used (32 window values, 16 + 7 twiddles per lane, each behind a wait)
       this one-wave kernel was a chain of some fifty dependent L2 latencies -- 45 us for the
       4 000 short-coded frames of the block-switched bench batch */
    __shared__ __attribute__((aligned(16))) double w[PACX_N_SHORT];
    __shared__ __attribute__((aligned(16))) cplx tws[PACX_N_SHORT / 4];
    __shared__ __attribute__((aligned(16))) cplx w64s[8][8];          /* W64^(r k2) at [k2][r] */
    {
        const double2 wa = *(const double2 *)(wg + 4 * lane), wb = *(const double2 *)(wg + 4 * lane + 2);
        const cplx tv = T.tw_short[lane];
        const cplx wv = T.w512[(8 * (lane & 7) * (lane >> 3)) & 511];
        stage_samples<DT, FAST>(raw, in, cf, PACX_SHORT_FIRST, SPAN, lane);
        *(double2 *)(w + 4 * lane) = wa;
        *(double2 *)(w + 4 * lane + 2) = wb;
        tws[lane] = tv;
        w64s[lane >> 3][lane & 7] = wv;
    }
    __syncthreads();

    const int g = lane >> 3, r = lane & 7;
    const E *sub = raw + g * PACX_M_SHORT;
    const int Q = PACX_N_SHORT / 4, M = PACX_M_SHORT;

    /* coder/pacfile.py:530-533: an all-zero sub-block makes the writer drop the hop */
    if (status) {
        bool nz = false;
        for (int i = r; i < PACX_N_SHORT; i += 8)
            nz = nz || (PcmStage<DT>::get(sub, i) != 0.0);
        const unsigned long long any = __ballot(nz);
        bool dropped = false;
        for (int q = 0; q < PACX_SUB; ++q)
            dropped = dropped || (((any >> (8 * q)) & 0xFFull) == 0);
        if (lane == 0)
            status[cf] = 1u | (dropped ? 2u : 0u);
    }

    cplx v[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int n = r + 8 * n1;
        double re, im;
        if (n1 < 4) {
            const int i0 = 3 * Q - 1 - 2 * n, i1 = 3 * Q + 2 * n, i2 = Q - 1 - 2 * n, i3 = Q + 2 * n;
            re = -(w[i0] * PcmStage<DT>::get(sub, i0)) - w[i1] * PcmStage<DT>::get(sub, i1);
            im = w[i2] * PcmStage<DT>::get(sub, i2) - w[i3] * PcmStage<DT>::get(sub, i3);
        } else {
            const int m = 2 * n - Q;
            const int i0 = m, i1 = M - 1 - m, i2 = 2 * Q + m, i3 = 4 * Q - 1 - m;
            re = w[i0] * PcmStage<DT>::get(sub, i0) - w[i1] * PcmStage<DT>::get(sub, i1);
            im = -(w[i2] * PcmStage<DT>::get(sub, i2)) - w[i3] * PcmStage<DT>::get(sub, i3);
        }
        v[n1] = c_mul(make_double2(re, im), tws[n]);
    }

    __syncthreads();                          /* every lane has folded its samples: raw becomes the tile */
    fft64x8_lds(v, tile, &w64s[0][0], lane);

    const double s = 2.0 / PACX_N_SHORT;     /* 2^-7 */
    double *__restrict__ out = lines + cf * PACX_M_LONG + g * PACX_M_SHORT;
    double mx = 0.0;
    /* X[2k] = a_k and X[2k+1] = X[M-1-2(63-k)] = b_(63-k): with k = r + 8 k3 the partner 63-k
       sits in lane 7-r of the same 8-lane group, register 7-k3 -- one row_half_mirror DPP
       away, so every store is a contiguous 16 bytes per lane (128 bytes per group) */
    double a[8], b[8];
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) {
        const int k = fft64_out_index(lane, k3);
        const cplx y = c_mul(v[k3], tws[k]);
        a[k3] = y.x * s;
        b[k3] = -(y.y * s);
        mx = fmax(mx, fmax(fabs(a[k3]), fabs(b[k3])));
    }
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) {
        const int k = fft64_out_index(lane, k3);
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(b[7 - k3]), 0x141, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(b[7 - k3]), 0x141, 0xf, 0xf, false);
        *(double2 *)(out + 2 * k) = make_double2(a[k3], __hiloint2double(hi, lo));
    }
    if (scale_out) {
        mx = fmax(mx, __shfl_xor(mx, 1, 64));
        mx = fmax(mx, __shfl_xor(mx, 2, 64));
        mx = fmax(mx, __shfl_xor(mx, 4, 64));
        if (r == 0)
            scale_out[cf * PACX_SUB + g] = pacx_scale_factor(mx, T.n_scale_bits, 5);
        /* PACX_ST_GUARD: a sub-block maximum at a boundary of ScaleFactor (pacx_exact.h) */
        const bool guard = T.guard && status && pacx_scale_guard(mx, T.n_scale_bits, 5, PACX_GUARD_LINE_ERR * mx);
        if (__builtin_amdgcn_ballot_w64(guard) && lane == 0)
            atomicOr(&status[cf], 16u);
    }
}

/* ------------------------------------------------------------- launchers */
template <int DT, bool FAST>
static void launch_mdct(const PacxTables &T, const PacxPcmView &in, const uint8_t *flags,
                        long long n_cf, int short_blocks, int mixed, int prewin, double *lines,
                        int32_t *scale_out, int scale_stride, uint32_t *status, hipStream_t st)
{
    const dim3 grid((unsigned)n_cf), block(64);
    /* mixed = 4: every long frame was done by k_mdct_long_v2, only the CUR frames
       are left (short kernel) */
    if (mixed == 4) {
        hipLaunchKernelGGL((k_mdct_short<DT, FAST>), grid, block, 0, st, T, in, flags, n_cf, 1, prewin, lines,
                           scale_out, status);
        return;
    }
    /* mixed = 2 / 3 (kept for callers that run v2 on sine-window frames only): the
       long kernel takes the transition-window frames (3: and skips CUR frames),
       the short kernel only CUR frames (3) */
    if (mixed == 2 || mixed == 3) {
        hipLaunchKernelGGL((k_mdct_long<DT, FAST>), grid, block, 0, st, T, in, flags, n_cf,
                           mixed == 3 ? 3 : 2, prewin, lines, scale_out, scale_stride);
        if (mixed == 3)
            hipLaunchKernelGGL((k_mdct_short<DT, FAST>), grid, block, 0, st, T, in, flags, n_cf, 1,
                               prewin, lines, scale_out, status);
        return;
    }
    if (!short_blocks || mixed)
        hipLaunchKernelGGL((k_mdct_long<DT, FAST>), grid, block, 0, st, T, in, flags, n_cf, mixed,
                           prewin, lines, scale_out, scale_stride);
    if (short_blocks || mixed)
        hipLaunchKernelGGL((k_mdct_short<DT, FAST>), grid, block, 0, st, T, in, flags, n_cf, mixed ? 1 : 0,
                           prewin, lines, scale_out, status);
}

/* mixed = 1: long kernel skips CUR frames, short kernel takes only CUR frames
 * (scale_out is then [n_cf][8]); otherwise short_blocks selects one of them. */
void pacx_launch_mdct(const PacxTables &T, const PacxPcmView &in, int dtype, int fast,
                      const uint8_t *flags, long long n_cf, int short_blocks, int mixed, int prewin,
                      double *lines, int32_t *scale_out, int scale_stride, uint32_t *status,
                      hipStream_t st)
{
    if (n_cf <= 0)
        return;
    if (dtype == 0 && fast)
        launch_mdct<0, true>(T, in, flags, n_cf, short_blocks, mixed, prewin, lines, scale_out, scale_stride, status, st);
    else if (dtype == 0)
        launch_mdct<0, false>(T, in, flags, n_cf, short_blocks, mixed, prewin, lines, scale_out, scale_stride, status, st);
    else
        launch_mdct<1, false>(T, in, flags, n_cf, short_blocks, mixed, prewin, lines, scale_out, scale_stride, status, st);
}
