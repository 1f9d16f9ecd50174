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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     /* wave-uniform: frame index math on the SALU */
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
                    int c0 = raw[i0], c1 = raw[i1], c2 = raw[i2], c3 = raw[i3];
                    /* keep the codes opaque 32-bit values: sign-extending loads and
                       v_min3_i32, no 16-bit narrowing with its extra v_bfe per sample */
                    asm("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
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
            if (pass || !__builtin_amdgcn_ballot_w64(lowest[0] == -32768 || lowest[1] == -32768))
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

/*
 * k_mdct_long_x2p: the two frames of a wave take turns on ONE 8 KB FFT tile, exchange
 * by exchange (A writes, B computes, A reads back, B writes, ...), which frees 8 KB per
 * wave for two PCM landing buffers of their own.  The next pair's LDS-DMA is then
 * issued right after the fold -- it has the whole FFT and epilogue to arrive instead of
 * the epilogue only -- and, being older than the epilogue's stores, is waited for with
 * a counted vmcnt that leaves those stores in flight.  A frame's reads of the tile are
 * followed by the other frame's writes without a wait in between: the DS instructions
 * of one wave execute in issue order.  The DMA is issued from inline assembly, so
 * the compiler's waitcnt pass does not know of it: it would otherwise put a
 * conservative vmcnt(0) before the next LDS read (it cannot tell the landing buffers
 * from the tile), which is what kept the landing-buffer variant of k_mdct2.hip from
 * prefetching.  Arithmetic, frame by frame, is that of k_mdct_long_x2.
 */
/* M0 (the DMA's LDS base) is on the asm's clobber list: the compiler sets M0 right before
   each of its own uses and keeps nothing live in it, the list entry only says so */
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#ifdef PACX_MDCT_DEBUG
__device__ long long g_mdct_dbg[8 * 16];
#define DBG_T(k) do { long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                      if (dbg_on) dbg_acc[k] += t_ - dbg_last; dbg_last = t_; } while (0)
extern "C" int pacx_debug_read(long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mdct_dbg), sizeof(long long) * n);
}
#else
#define DBG_T(k) do { } while (0)
#endif
/* 16-byte line stores per epilogue (1024 lines = 64 lanes x 8 stores x 2 doubles).  The counted
   wait of k_mdct_long_x2p is derived from it; tests/test_build_isa.py reads this constant and
   checks the compiled code against it. */
constexpr int EPI_STORES = 8;

/* STEP: the instantiation the whole-path entry points launch (it also initialises the frames' status words and
   sub-block scales); the stand-alone pacx_mdct_batch launches STEP = false.  Two symbols, so that a kernel trace
   tells the stand-alone launches -- the ones the HBM roofline figure is quoted on -- from the in-step ones, which
   run beside the side chain and, with two steps in flight, beside another step's kernels, and are stretched by it */
template <int WAVES, int MINW, bool STEP>
__global__ __launch_bounds__(64 * WAVES, MINW) void k_mdct_long_x2p(PacxTables T, PacxPcmView in, long long n_cf,
                                                                   double *__restrict__ lines,
                                                                   int32_t *__restrict__ scale_out,
                                                                   int scale_stride,
                                                                   uint32_t *__restrict__ status_init)
{
    __shared__ __attribute__((aligned(16))) cplx tiles[WAVES][WFFT_TILE_N];
    __shared__ __attribute__((aligned(16))) short raws[WAVES][2][PACX_N_LONG];
    __shared__ __attribute__((aligned(16))) cplx twl[512];
    __shared__ __attribute__((aligned(16))) double wsin[1024];
    __shared__ __attribute__((aligned(16))) cplx w64[7][8];        /* W64^(r k2), k2 = 1..7 */
    __shared__ __attribute__((aligned(16))) cplx w1s[7][64];       /* W512^(lane k1), k1 = 1..7 */
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    cplx *tile = tiles[wv];
    const unsigned n_ch = (unsigned)in.n_ch;
    const unsigned n_waves = gridDim.x * WAVES;
    const unsigned total = (unsigned)n_cf;
    const short *base = (const short *)in.base;
    auto stage = [&](unsigned c, int f) {
        const unsigned fr = c / n_ch, ch = c - fr * n_ch;
        const int4 *src = (const int4 *)(base + (long long)fr * in.frame_stride + (long long)ch * in.ch_stride);
        const unsigned lds = __builtin_amdgcn_readfirstlane(
            (unsigned)(size_t)(__attribute__((address_space(3))) char *)(char *)raws[wv][f]);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, off\n\t"
                     "global_load_lds_dwordx4 %0, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, off offset:2048\n\t"
                     "global_load_lds_dwordx4 %0, off offset:3072"
                     :: "v"(src + lane), "s"(lds) : "memory", "m0");
    };
    /* XCD-aware frame order: workgroups are handed to the 8 XCDs round-robin, so workgroup b
       runs on XCD b % 8.  Consecutive frames share a hop of PCM; giving every XCD a CONTIGUOUS
       run of workgroups' worth of frames keeps that shared hop in one XCD's L2 instead of
       fetching it from HBM once per XCD (a quarter of the hops were read twice) */
    const unsigned vb = (gridDim.x & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const unsigned g = vb * WAVES + wv;
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
    auto wr1 = [&](const cplx *v) {
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1)
            tile[64 * k1 + (lane ^ (8 * k1))] = v[k1];
    };
    auto rd1 = [&](cplx *v) {
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2)
            v[n2] = tile[64 * g8 + 8 * (n2 ^ g8) + r8];
    };
    auto wr2 = [&](const cplx *v) {
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2)
            tile[64 * g8 + 8 * k2 + (r8 ^ g8)] = v[k2];
    };
    auto rd2 = [&](cplx *v) {
#pragma unroll
        for (int n3 = 0; n3 < 8; ++n3)
            v[n3] = tile[64 * r8 + 8 * g8 + (n3 ^ r8)];
    };
    auto tw1 = [&](cplx *v) {
#pragma unroll
        for (int k1 = 1; k1 < 8; ++k1)
            v[k1] = c_mul(v[k1], w1s[k1 - 1][lane]);
    };
    auto tw2 = [&](cplx *v) {
#pragma unroll
        for (int k2 = 1; k2 < 8; ++k2)
            v[k2] = c_mul(v[k2], w64[k2 - 1][r8]);
    };
    auto epilogue = [&](const cplx *v, unsigned cf) {
        double a[8], b[8];
        double mx = 0.0;
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3) {
            const cplx d = twl[lane + 64 * k3];
            a[k3] = fma(v[k3].x, d.x, -(v[k3].y * d.y));      /* Re y = X[2k] */
            b[k3] = -fma(v[k3].x, d.y, v[k3].y * d.x);        /* -Im y = X[1023 - 2k] */
            mx = fmax(mx, fmax(fabs(a[k3]), fabs(b[k3])));
        }
        double odd[8];
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3)
            odd[k3] = __shfl(b[7 - k3], 63 - lane, 64);
        /* overall scale: ScaleFactor is non-increasing in its argument, so the scale of
           the block maximum is the minimum of the lanes' own scales -- a few-bit integer,
           found by bisection with one ballot per bit while the lane reversal above is in
           flight (a 64-bit max over the wave would be six dependent LDS round trips) */
        int lo = 0;
        bool guard = false;
        if (scale_out) {
            const int s = pacx_scale_factor(mx, T.n_scale_bits, 5);
            for (int bit = T.n_scale_bits - 1; bit >= 0; --bit)
                if (!__builtin_amdgcn_ballot_w64(s < lo + (1 << bit)))
                    lo += 1 << bit;
            /* PACX_ST_GUARD: the lanes that decide the minimum hold a maximum within a factor
               two of the block's; theirs sitting at a boundary of ScaleFactor flags the frame
               (line error bound relative to the block maximum, pacx_exact.h) */
            guard = STEP && T.guard && status_init && s == lo && pacx_scale_guard(mx, T.n_scale_bits, 5, 2.0 * PACX_GUARD_LINE_ERR * mx);
        }
        double2 *__restrict__ out = (double2 *)(lines + (long long)cf * PACX_M_LONG);
        static_assert(EPI_STORES * 64 * 2 == PACX_M_LONG, "one epilogue = EPI_STORES 16-byte stores per lane");
#pragma unroll
        for (int k3 = 0; k3 < EPI_STORES; ++k3)
            out[lane + 64 * k3] = make_double2(a[k3], odd[k3]);
        if (scale_out && lane == 0)
            scale_out[(long long)cf * scale_stride] = lo;
        if (__builtin_amdgcn_ballot_w64(guard) && lane == 0)
            status_init[cf] = 16u;                 /* after this lane's own zero-store of the same word */
    };
#ifdef PACX_MDCT_DEBUG
    const bool dbg_on = blockIdx.x == 7;
    long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dbg_last = 0;
    DBG_T(7);
    dbg_acc[7] = 0;
#endif
    bool first = true;
    for (; cfa < total; cfa += 2 * n_waves, cfb += 2 * n_waves) {
        const bool has_b = cfb < total;
        /* this pair's PCM must have landed; what was issued after its DMA -- the previous
           pair's 16 line stores (and 2 scale stores) -- may stay in flight.  Vector memory
           operations complete in issue order, so a count that is not above the number of
           younger operations is safe, a smaller one merely waits for more */
        if (first)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * EPI_STORES) : "memory");   /* the two epilogues' line
                                       stores; the 2 scale stores, when there are any, only make the wait stricter */
        first = false;
        DBG_T(0);
        wave_lds_fence();
        cplx v[2][8];
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
                    const short *raw = raws[wv][f];
                    int c0 = raw[i0], c1 = raw[i1], c2 = raw[i2], c3 = raw[i3];
                    asm("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
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
            if (pass || !__builtin_amdgcn_ballot_w64(lowest[0] == -32768 || lowest[1] == -32768))
                break;
            wave_lds_fence();
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                unsigned *rw = (unsigned *)raws[wv][f];
                for (int j = 0; j < 16; ++j) {
                    unsigned x = rw[lane + 64 * j];
                    if ((x & 0xFFFFu) == 0x8000u) x &= 0xFFFF0000u;
                    if ((x >> 16) == 0x8000u) x &= 0x0000FFFFu;
                    rw[lane + 64 * j] = x;
                }
            }
            wave_lds_fence();
        }
        wave_lds_fence();
        /* the landing buffers are consumed (the DMA is not ordered with this wave's own
           ds_reads: they must have returned): next pair's PCM on its way.  The output
           initialisation goes first so that it is older than the DMA. */
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DBG_T(1);
        if (STEP && status_init) {
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
        if (cfa + 2 * n_waves < total)
            stage(cfa + 2 * n_waves, 0);
        if (cfb + 2 * n_waves < total)
            stage(cfb + 2 * n_waves, 1);
        DBG_T(2);

        /* 512-point FFTs (wave_fft.h fft512n), the two frames alternating on the tile */
        dft8(v[0]); tw1(v[0]); wr1(v[0]);
        dft8(v[1]); tw1(v[1]);
        wave_lds_fence(); rd1(v[0]); wr1(v[1]);
        dft8(v[0]); tw2(v[0]);
        wave_lds_fence(); rd1(v[1]); wr2(v[0]);
        dft8(v[1]); tw2(v[1]);
        wave_lds_fence(); rd2(v[0]); wr2(v[1]);
        dft8(v[0]);
        wave_lds_fence(); rd2(v[1]);
        DBG_T(3);
        epilogue(v[0], cfa);
        DBG_T(4);
        dft8(v[1]);
        if (has_b)
            epilogue(v[1], cfb);
        wave_lds_fence();
        DBG_T(5);
    }
#ifdef PACX_MDCT_DEBUG
    if (dbg_on && lane == 0)
        for (int k = 0; k < 8; ++k)
            g_mdct_dbg[wv * 16 + k] = dbg_acc[k];
#endif
}

#pragma clang diagnostic pop

void pacx_launch_mdct_x2(const PacxTables &T, const PacxPcmView &in, long long n_cf, double *lines,
                         int32_t *scale_out, int scale_stride, uint32_t *status_init, int n_cu, int waves,
                         hipStream_t st)
{
    if (waves == 0) {                        /* the pipelined kernel (default) */
        long long blocks = (n_cf + 2 * 8 - 1) / (2 * 8);
        if (blocks > n_cu)
            blocks = n_cu;
        if (status_init)
            hipLaunchKernelGGL((k_mdct_long_x2p<8, 2, true>), dim3((unsigned)blocks), dim3(64 * 8), 0, st, T, in, n_cf, lines,
                               scale_out, scale_stride, status_init);
        else
            hipLaunchKernelGGL((k_mdct_long_x2p<8, 2, false>), dim3((unsigned)blocks), dim3(64 * 8), 0, st, T, in, n_cf, lines,
                               scale_out, scale_stride, status_init);
        return;
    }
    if (n_cf <= 0)
        return;
    if (status_init && (!scale_out || scale_stride != PACX_SUB))
        status_init = nullptr;
    long long blocks = (n_cf + 2 * 8 - 1) / (2 * 8);
    if (blocks > n_cu)
        blocks = n_cu;
    hipLaunchKernelGGL((k_mdct_long_x2<8, 2>), dim3((unsigned)blocks), dim3(64 * 8), 0, st, T, in, n_cf, lines,
                       scale_out, scale_stride, status_init);
}
