/*
 * k_mdct3.hip -- k_mdct_long_v2 with TWO channel-frames in flight per wave
 * (sine window, no per-frame flags: the headline batch).  Same arithmetic, frame
 * by frame, as k_mdct2.hip; what changes is the instruction stream: the fold, the
 * three radix-8 passes with their two LDS exchanges and the epilogue of frames A
 * and B are interleaved stage by stage, so one frame's LDS round trips hide behind
 * the other's arithmetic, and every table value read from LDS (window, pre/post
 * twiddle, per-lane FFT twiddles) serves both frames.  LDS: two 8 KB tiles per
 * wave (16 waves x 8 KB as 8 waves x 2) + the 24 KB of tables.
 */
#include "pacx_dev.h"
#include "wave_fft.h"

template <int WAVES, int MINW>
__global__ __launch_bounds__(64 * WAVES, MINW) void k_mdct_long_x2(PacxTables T, PacxPcmView in, long long n_cf,
                                                                  double *__restrict__ lines,
                                                                  int32_t *__restrict__ scale_out,
                                                                  int scale_stride,
                                                                  uint32_t *__restrict__ status_init)
{
    __shared__ __attribute__((aligned(16))) cplx tiles[WAVES][2][WFFT_TILE_N];
    __shared__ __attribute__((aligned(16))) cplx twl[512];
    __shared__ __attribute__((aligned(16))) double wsin[1024];
    __shared__ __attribute__((aligned(16))) cplx w64[7][8];        /* W64^(r k2), k2 = 1..7 */
    __shared__ __attribute__((aligned(16))) cplx w1s[7][64];       /* W512^(lane k1), k1 = 1..7 */
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    cplx *tile[2] = {tiles[wv][0], tiles[wv][1]};
    const unsigned n_ch = (unsigned)in.n_ch;
    const unsigned n_waves = gridDim.x * WAVES;
    const unsigned total = (unsigned)n_cf;
    const short *base = (const short *)in.base;
    auto stage = [&](unsigned c, int f) {
        const unsigned fr = c / n_ch, ch = c - fr * n_ch;
        const int4 *src = (const int4 *)(base + (long long)fr * in.frame_stride + (long long)ch * in.ch_stride);
        char *dst = (char *)tile[f];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + lane + 64 * j),
                                             (__attribute__((address_space(3))) void *)(dst + 1024 * j), 16, 0, 0);
    };
    const unsigned g = blockIdx.x * WAVES + wv;
    /* iteration it: frames g + 2 it n_waves (A) and g + (2 it + 1) n_waves (B) */
    unsigned cfa = g, cfb = g + n_waves;
    if (cfa < total)
        stage(cfa, 0);
    if (cfb < total)
        stage(cfb, 1);
    for (int i = tid; i < 512; i += 64 * WAVES)
        twl[i] = T.tw_long[i];
    const double kscale = (2.0 / 65535.0) * (2.0 / PACX_N_LONG);
    for (int i = tid; i < 1024; i += 64 * WAVES)
        wsin[i] = T.win_long[i] * kscale;
    if (tid < 56)
        w64[tid >> 3][tid & 7] = T.w512[8 * (tid & 7) * ((tid >> 3) + 1)];
    for (int i = tid; i < 7 * 64; i += 64 * WAVES)
        w1s[i >> 6][i & 63] = T.w512[(i & 63) * ((i >> 6) + 1)];
    __syncthreads();

    const int Q = PACX_N_LONG / 4, M = PACX_M_LONG;
    const int g8 = lane >> 3, r8 = lane & 7;
    for (; cfa < total; cfa += 2 * n_waves, cfb += 2 * n_waves) {
        const bool has_b = cfb < total;
        if (status_init) {
            if (lane == 0) {
                status_init[cfa] = 0u;
                if (has_b)
                    status_init[cfb] = 0u;
            }
            if (lane >= 1 && lane < PACX_SUB) {
                scale_out[(long long)cfa * PACX_SUB + lane] = 0;
                if (has_b)
                    scale_out[(long long)cfb * PACX_SUB + lane] = 0;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* both frames' PCM has landed */
        wave_lds_fence();
        cplx v[2][8];
        /* fold + window + pre-twiddle of both frames; table values shared.  -32768 is
           mapped to 0 as in k_mdct2.hip: track the minimum, rewrite and refold (rare). */
        for (int pass = 0;; ++pass) {
            int lowest[2] = {0, 0};
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                const int n = lane + 64 * n1;
                int i0, i1, i2, i3;
                double wa, wb;
                if (n1 < 4) {
                    i0 = 3 * Q - 1 - 2 * n; i1 = 3 * Q + 2 * n; i2 = Q - 1 - 2 * n; i3 = Q + 2 * n;
                    wa = wsin[i3]; wb = wsin[i2];                    /* = w[i0], w[i1] */
                } else {
                    const int m = 2 * n - Q;
                    i0 = m; i1 = M - 1 - m; i2 = 2 * Q + m; i3 = 4 * Q - 1 - m;
                    wa = wsin[i0]; wb = wsin[i1];                    /* = w[i3], w[i2] */
                }
                const cplx tw = twl[n];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const short *raw = (const short *)tile[f];
                    const int c0 = raw[i0], c1 = raw[i1], c2 = raw[i2], c3 = raw[i3];
                    lowest[f] = min(lowest[f], min(min(c0, c1), min(c2, c3)));
                    double re, im;
                    if (n1 < 4) {
                        re = -fma(wb, (double)c1, wa * (double)c0);
                        im = fma(wb, (double)c2, -(wa * (double)c3));
                    } else {
                        re = fma(wa, (double)c0, -(wb * (double)c1));
                        im = -fma(wb, (double)c2, wa * (double)c3);
                    }
                    v[f][n1] = c_mul(make_double2(re, im), tw);
                }
            }
            if (pass || !__ballot(lowest[0] == -32768 || lowest[1] == -32768))
                break;
            wave_lds_fence();
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                unsigned *rw = (unsigned *)tile[f];
                for (int j = 0; j < 16; ++j) {
                    unsigned x = rw[lane + 64 * j];
                    if ((x & 0xFFFFu) == 0x8000u) x &= 0xFFFF0000u;
                    if ((x >> 16) == 0x8000u) x &= 0x0000FFFFu;
                    rw[lane + 64 * j] = x;
                }
            }
            wave_lds_fence();
        }
        wave_lds_fence();                 /* raw samples consumed: the tiles may be overwritten */

        /* 512-point FFT of both frames, natural order in and out (wave_fft.h fft512n,
           stage by stage for the two frames) */
#pragma unroll
        for (int f = 0; f < 2; ++f)
            dft8(v[f]);
#pragma unroll
        for (int k1 = 1; k1 < 8; ++k1) {
            const cplx w = w1s[k1 - 1][lane];
            v[0][k1] = c_mul(v[0][k1], w);
            v[1][k1] = c_mul(v[1][k1], w);
        }
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1)
                tile[f][64 * k1 + (lane ^ (8 * k1))] = v[f][k1];
        wave_lds_fence();
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2)
                v[f][n2] = tile[f][64 * g8 + 8 * (n2 ^ g8) + r8];
        wave_lds_fence();
#pragma unroll
        for (int f = 0; f < 2; ++f)
            dft8(v[f]);
#pragma unroll
        for (int k2 = 1; k2 < 8; ++k2) {
            const cplx w = w64[k2 - 1][r8];
            v[0][k2] = c_mul(v[0][k2], w);
            v[1][k2] = c_mul(v[1][k2], w);
        }
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2)
                tile[f][64 * g8 + 8 * k2 + (r8 ^ g8)] = v[f][k2];
        wave_lds_fence();
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int n3 = 0; n3 < 8; ++n3)
                v[f][n3] = tile[f][64 * r8 + 8 * g8 + (n3 ^ r8)];
        wave_lds_fence();
#pragma unroll
        for (int f = 0; f < 2; ++f)
            dft8(v[f]);

        /* the tiles are free: next pair's PCM on its way (after this wave's last tile
           reads have returned: the DMA is not ordered with its ds_reads) */
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (cfa + 2 * n_waves < total)
            stage(cfa + 2 * n_waves, 0);
        if (cfb + 2 * n_waves < total)
            stage(cfb + 2 * n_waves, 1);

        /* post-twiddle, overall scale, stores */
        cplx d[8];
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3)
            d[k3] = twl[lane + 64 * k3];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            if (f == 1 && !has_b)
                break;
            const unsigned cf = f ? cfb : cfa;
            double a[8], b[8];
            double mx = 0.0;
#pragma unroll
            for (int k3 = 0; k3 < 8; ++k3) {
                a[k3] = fma(v[f][k3].x, d[k3].x, -(v[f][k3].y * d[k3].y));      /* Re y = X[2k] */
                b[k3] = -fma(v[f][k3].x, d[k3].y, v[f][k3].y * d[k3].x);        /* -Im y = X[1023 - 2k] */
                mx = fmax(mx, fmax(fabs(a[k3]), fabs(b[k3])));
            }
            double2 *__restrict__ out = (double2 *)(lines + (long long)cf * PACX_M_LONG);
#pragma unroll
            for (int k3 = 0; k3 < 8; ++k3) {
                const double odd = __shfl(b[7 - k3], 63 - lane, 64);
                out[lane + 64 * k3] = make_double2(a[k3], odd);
            }
            if (scale_out) {
                mx = wave_max(mx);
                if (lane == 0)
                    scale_out[(long long)cf * scale_stride] = pacx_scale_factor(mx, T.n_scale_bits, 5);
            }
        }
    }
}

void pacx_launch_mdct_x2(const PacxTables &T, const PacxPcmView &in, long long n_cf, double *lines,
                         int32_t *scale_out, int scale_stride, uint32_t *status_init, int n_cu, int waves,
                         hipStream_t st)
{
    if (n_cf <= 0)
        return;
    if (status_init && (!scale_out || scale_stride != PACX_SUB))
        status_init = nullptr;
    long long blocks = (n_cf + 2 * waves - 1) / (2 * waves);
    if (blocks > n_cu)
        blocks = n_cu;
    if (waves == 8)
        hipLaunchKernelGGL((k_mdct_long_x2<8, 2>), dim3((unsigned)blocks), dim3(64 * 8), 0, st, T, in, n_cf, lines,
                           scale_out, scale_stride, status_init);
    else if (waves == 6)
        hipLaunchKernelGGL((k_mdct_long_x2<6, 2>), dim3((unsigned)blocks), dim3(64 * 6), 0, st, T, in, n_cf, lines,
                           scale_out, scale_stride, status_init);
    else
        hipLaunchKernelGGL((k_mdct_long_x2<4, 1>), dim3((unsigned)blocks), dim3(64 * 4), 0, st, T, in, n_cf, lines,
                           scale_out, scale_stride, status_init);
}
