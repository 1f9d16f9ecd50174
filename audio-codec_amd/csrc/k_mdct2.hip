/*
 * k_mdct2.hip -- the roofline kernel: window + MDCT of long blocks, int16 PCM in
 * (unit stride, 16-byte aligned rows), float64 lines out.  Same math as
 * k_mdct.hip (reference: coder/window.py:14-25, coder/mdct.py:43-69 at
 * coder/codec.py:303-311) laid out for throughput:
 *
 *   - persistent workgroups of 8 waves, one channel-frame per wave at a time,
 *     grid-stride over frames; twiddle table d[n] and the sine window live in
 *     LDS once per workgroup, the per-lane FFT twiddles in registers;
 *   - the next frame's 4 KB of PCM is prefetched into registers (4 x 16-byte
 *     coalesced loads per lane) while the current frame is transformed;
 *   - natural-order 512-point FFT (wave_fft.h fft512n) on an 8 KB swizzled tile
 *     that also stages the raw int16 samples, so LDS is 8 KB per wave;
 *   - lane L ends up holding y[L + 64 j]; X[2k] = Re y[k] and
 *     X[2k+1] = -Im y[511-k] sit in mirrored lanes, one 64-lane reversal
 *     (ds_bpermute) pairs them so every store is a contiguous 16 bytes per lane
 *     (1 KB per wave instruction).
 *
 * HBM traffic per channel-frame: 2 KB of new PCM (the other half of the window
 * was read by the previous frame and is an L2 hit) + 8 KB of lines = 10 240 B.
 */
#include "pacx_dev.h"
#include "wave_fft.h"

#include <stdlib.h>


/* 8 waves per workgroup (8 KB tile each + 24 KB of tables, + 48 KB of transition windows
   with ANYWIN), one workgroup per CU, two waves per SIMD */
#define MDCT2_WAVES 8
template <bool ANYWIN>
__global__ __launch_bounds__(64 * MDCT2_WAVES, 2) void k_mdct_long_v2(
    PacxTables T, PacxPcmView in, const uint8_t *__restrict__ flags, long long n_cf, int skip_cur,
    double *__restrict__ lines, int32_t *__restrict__ scale_out, int scale_stride,
    uint32_t *__restrict__ status_init, const int32_t *__restrict__ cf_list,
    const int32_t *__restrict__ cf_count)
{
    __shared__ __attribute__((aligned(16))) cplx tiles[MDCT2_WAVES][WFFT_TILE_N];
    __shared__ __attribute__((aligned(16))) cplx twl[512];
    __shared__ __attribute__((aligned(16))) double wsin[1024];
    __shared__ __attribute__((aligned(16))) cplx w64[7][8];        /* W64^(r k2), k2 = 1..7 */
    __shared__ __attribute__((aligned(16))) cplx w1s[7][64];       /* W512^(lane k1), k1 = 1..7 */
    /* ANYWIN: the three transition windows (start / stop / start-stop, coder/window.py:61-92),
       scaled like wsin, whole: a frame's window kind is wave-uniform, so a fold element reads
       its four values at plain per-lane indices -- no region tests, no global loads whose 32
       results in flight per frame had the kernel at 256 registers with 23 spilled */
    __shared__ __attribute__((aligned(16))) double wtr[ANYWIN ? 3 : 1][ANYWIN ? PACX_N_LONG : 2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     /* wave-uniform: frame index math on the SALU */
    cplx *tile = tiles[wv];
    short *raw = (short *)tile;            /* the raw int16 samples are staged in the FFT tile */
    const unsigned n_ch = (unsigned)in.n_ch;
    const unsigned stride = gridDim.x * MDCT2_WAVES;
    /* mixed streams: walk the compacted list of the long-coded frames (k_frame_lists) instead
       of every frame -- on a castanet stream half of the frames are short-coded, and staging
       their PCM only to skip them was half of this kernel's time */
    const unsigned total = cf_list ? (unsigned)*cf_count : (unsigned)n_cf;
    auto frame_of = [&](unsigned i) -> unsigned { return cf_list ? (unsigned)cf_list[i] : i; };
    const short *base = (const short *)in.base;
    /* PCM goes HBM -> LDS without touching VGPRs (global_load_lds_dwordx4:
       wave-uniform LDS base + lane*16, per-lane global address) */
    auto stage = [&](unsigned c) {
        const unsigned f = c / n_ch, ch = c - f * n_ch;
        const int4 *src = (const int4 *)(base + (long long)f * in.frame_stride + (long long)ch * in.ch_stride);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + lane + 64 * j),
                                             (__attribute__((address_space(3))) void *)((char *)raw + 1024 * j),
                                             16, 0, 0);
    };
    /* XCD-aware frame order (workgroup b runs on XCD b % 8): every XCD takes a contiguous
       run of frames, so the hop two consecutive frames share stays in one XCD's L2
       (k_mdct3.hip, k_mdct_long_x2p) */
    const unsigned vb = (gridDim.x & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    unsigned it = vb * MDCT2_WAVES + wv;              /* position in the list (or the frame itself) */
    unsigned cf_next = it < total ? frame_of(it) : 0u;
    if (it < total)
        stage(cf_next);               /* first frame's PCM flies while the tables load */
    for (int i = tid; i < 512; i += 64 * MDCT2_WAVES)
        twl[i] = T.tw_long[i];
    /* window table with the PCM scale 2/65535 (coder/pcmfile.py:89-99 mapping)
       and the MDCT's 2/N = 2^-10 folded in: the transform is linear, so the
       int16 codes go through it as exact integers and each product w'[i]*c
       carries one rounding.  Against the reference's order (round x = 2c/65535
       first, then window) this moves a line by ~1e-16 of the block maximum,
       three orders below the 2e-13 that separates the two FFT algorithms. */
    const double kscale = (2.0 / 65535.0) * (2.0 / PACX_N_LONG);
    for (int i = tid; i < 1024; i += 64 * MDCT2_WAVES)
        wsin[i] = T.win_long[i] * kscale;
    if (ANYWIN)
        for (int i = tid; i < 3 * PACX_N_LONG; i += 64 * MDCT2_WAVES)
            wtr[0][i] = T.win_long[PACX_N_LONG + i] * kscale;
    if (tid < 56)
        w64[tid >> 3][tid & 7] = T.w512[8 * (tid & 7) * ((tid >> 3) + 1)];
    for (int i = tid; i < 7 * 64; i += 64 * MDCT2_WAVES)
        w1s[i >> 6][i & 63] = T.w512[(i & 63) * ((i >> 6) + 1)];
    __syncthreads();

    const cplx *w1 = &w1s[0][lane];                                /* w1[64 (k1-1)] */
    const cplx *w2 = &w64[0][lane & 7];                            /* w2[8 (k2-1)] */
    for (; it < total; it += stride) {
        const unsigned cf = cf_next;
        if (it + stride < total)
            cf_next = frame_of(it + stride);          /* read one iteration ahead of its DMA */
        const unsigned fl = flags ? flags[cf / n_ch] : 0u;
        /* frames this kernel leaves to k_mdct_short: short-coded (CUR) ones when asked to */
        const bool mine = !(skip_cur && (fl & 2u));
        /* transition windows (start / stop / start-stop) are not symmetric: their four
           values per fold element come from the LDS tables wtr */
        const int kind = ANYWIN ? __builtin_amdgcn_readfirstlane(pacx_window_kind(fl)) : 0;   /* !ANYWIN: launched without flags */
        const double *gw = wtr[kind > 0 ? kind - 1 : 0];
        /* output initialisation the whole-path entry points would otherwise spend two
           memset launches on: status word 0 and the 7 unused overall-scale slots of a
           long frame 0, for EVERY frame (kernels that follow on the stream overwrite /
           OR into them for the frames they own) */
        if (status_init) {
            if (lane == 0)
                status_init[cf] = 0u;
            if (lane >= 1 && lane < PACX_SUB)
                scale_out[(long long)cf * PACX_SUB + lane] = 0;
        }
        /* this frame's PCM must have landed in LDS (the DMA is the youngest vector-memory
           operation but for the two initialisation stores above: wait for everything) */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_fence();
        if (!mine) {
            /* a short-coded frame (k_mdct_short's): nothing to transform, only the chain of
               PCM fetches to keep going -- on a castanet stream that is more than half of
               the frames */
            if (it + stride < total)
                stage(cf_next);
            continue;
        }
        const int Q = PACX_N_LONG / 4, M = PACX_M_LONG;
        cplx v[8];
        /* sine window: w[2047-i] = w[i], so two table values serve each n.
           The codes enter as integers (scale folded into wsin).  The reference
           maps the code -32768 to 0 (coder/pcmfile.py:93-97 masks the magnitude
           with 32767): rather than test every sample, track the minimum and,
           only for a frame that has one, rewrite them in LDS and fold again. */
        for (int pass = 0;; ++pass) {
            int lowest = 0;
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                const int n = lane + 64 * n1;
                auto code = [&](int i) -> double {
                    int c = raw[i];
                    asm("" : "+v"(c));      /* opaque 32-bit: sign-extending load, no 16-bit narrowing */
                    lowest = min(lowest, c);
                    return (double)c;
                };
                double re, im;
                if (n1 < 4) {
                    const int i0 = 3 * Q - 1 - 2 * n, i1 = 3 * Q + 2 * n, i2 = Q - 1 - 2 * n, i3 = Q + 2 * n;
                    if (!ANYWIN || kind == 0) {
                        const double wa = wsin[i3], wb = wsin[i2];       /* = w[i0], w[i1] */
                        re = -fma(wb, code(i1), wa * code(i0));
                        im = fma(wb, code(i2), -(wa * code(i3)));
                    } else {
                        re = -fma(gw[i1], code(i1), gw[i0] * code(i0));
                        im = fma(gw[i2], code(i2), -(gw[i3] * code(i3)));
                    }
                } else {
                    const int m = 2 * n - Q;
                    const int i0 = m, i1 = M - 1 - m, i2 = 2 * Q + m, i3 = 4 * Q - 1 - m;
                    if (!ANYWIN || kind == 0) {
                        const double wa = wsin[i0], wb = wsin[i1];       /* = w[i3], w[i2] */
                        re = fma(wa, code(i0), -(wb * code(i1)));
                        im = -fma(wb, code(i2), wa * code(i3));
                    } else {
                        re = fma(gw[i0], code(i0), -(gw[i1] * code(i1)));
                        im = -fma(gw[i2], code(i2), gw[i3] * code(i3));
                    }
                }
                v[n1] = c_mul(make_double2(re, im), twl[n]);
            }
            if (pass || !__builtin_amdgcn_ballot_w64(lowest == -32768))
                break;
            /* rare: rewrite the -32768 codes of this frame to 0 in LDS and fold again */
            wave_lds_fence();
            unsigned *rw = (unsigned *)raw;
            for (int j = 0; j < 16; ++j) {
                unsigned x = rw[lane + 64 * j];
                if ((x & 0xFFFFu) == 0x8000u) x &= 0xFFFF0000u;
                if ((x >> 16) == 0x8000u) x &= 0x0000FFFFu;
                rw[lane + 64 * j] = x;
            }
            wave_lds_fence();
        }
        wave_lds_fence();                 /* raw samples consumed: their LDS may be overwritten */
        fft512n(v, tile, w1, 64, w2, 8, lane);
        /* the tile is free again: start the next frame's PCM on its way now, it
           lands during the epilogue and the other waves' work.  The DMA writes
           LDS from the vector-memory side and is not ordered with this wave's
           own ds_reads, so the FFT's last tile reads must have returned first
           (a wavefront-scope fence does not wait for them; hazard table in DESIGN.md,
           checked on the compiled code by tests/test_build_isa.py). */
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (it + stride < total)
            stage(cf_next);

        double a[8], b[8];
        double mx = 0.0;
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3) {
            const cplx d = twl[lane + 64 * k3];
            a[k3] = fma(v[k3].x, d.x, -(v[k3].y * d.y));      /* Re y = X[2k], k = lane + 64 k3 */
            b[k3] = -fma(v[k3].x, d.y, v[k3].y * d.x);        /* -Im y = X[1023 - 2k]           */
            mx = fmax(mx, fmax(fabs(a[k3]), fabs(b[k3])));
        }
        /* X[2k+1] = X[1023 - 2(511-k)] is held by lane 63-lane, register 7-k3 */
        double odd[8];
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3)
            odd[k3] = __shfl(b[7 - k3], 63 - lane, 64);
        /* overall scale = minimum of the lanes' own scales (ScaleFactor is non-increasing),
           by bisection with one ballot per bit: see k_mdct3.hip */
        int lo = 0;
        bool guard = false;
        if (scale_out) {
            const int s = pacx_scale_factor(mx, T.n_scale_bits, 5);
            for (int bit = T.n_scale_bits - 1; bit >= 0; --bit)
                if (!__builtin_amdgcn_ballot_w64(s < lo + (1 << bit)))
                    lo += 1 << bit;
            guard = T.guard && status_init && s == lo && pacx_scale_guard(mx, T.n_scale_bits, 5, 2.0 * PACX_GUARD_LINE_ERR * mx);   /* PACX_ST_GUARD, as k_mdct3.hip */
        }
        double2 *__restrict__ out = (double2 *)(lines + (long long)cf * PACX_M_LONG);
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3)
            out[lane + 64 * k3] = make_double2(a[k3], odd[k3]);
        if (scale_out && lane == 0)
            scale_out[(long long)cf * scale_stride] = lo;
        if (__builtin_amdgcn_ballot_w64(guard) && lane == 0)
            status_init[cf] = 16u;
    }
}

void pacx_launch_mdct_x2(const PacxTables &T, const PacxPcmView &in, long long n_cf, double *lines,
                         int32_t *scale_out, int scale_stride, uint32_t *status_init, int n_cu, int waves,
                         hipStream_t st);

void pacx_launch_mdct_v2(const PacxTables &T, const PacxPcmView &in, const uint8_t *flags, long long n_cf,
                         int skip_cur, double *lines, int32_t *scale_out, int scale_stride,
                         uint32_t *status_init, int n_cu, const int32_t *cf_list, const int32_t *cf_count,
                         hipStream_t st)
{
    if (n_cf <= 0)
        return;
    if (status_init && (!scale_out || scale_stride != PACX_SUB))
        status_init = nullptr;                 /* the caller keeps its memsets */
    /* PACX_MDCT_VARIANT (experiments and the kernel-equivalence test): -1 / unset = the
       defaults below; 0 = this file's kernel also for batches without flags; 7 = the
       two-tile kernel k_mdct_long_x2 */
    static int variant = -2;
    if (variant == -2) {
        const char *e = getenv("PACX_MDCT_VARIANT");
        variant = e ? atoi(e) : -1;
    }
    /* default: batches without per-frame flags (all sine windows) go to the pipelined
       two-frames-per-wave kernel of k_mdct3.hip; batches with flags (transition
       windows, frames left to the short kernel) stay here */
    if (!flags && variant != 0) {
        pacx_launch_mdct_x2(T, in, n_cf, lines, scale_out, scale_stride, status_init, n_cu, variant == 7 ? 8 : 0,
                            st);
        return;
    }
    long long blocks = (n_cf + MDCT2_WAVES - 1) / MDCT2_WAVES;
    if (blocks > n_cu)
        blocks = n_cu;                         /* one persistent workgroup per CU */
    if (flags)
        hipLaunchKernelGGL((k_mdct_long_v2<true>), dim3((unsigned)blocks), dim3(64 * MDCT2_WAVES), 0, st, T, in,
                           flags, n_cf, skip_cur, lines, scale_out, scale_stride, status_init, cf_list, cf_count);
    else
        hipLaunchKernelGGL((k_mdct_long_v2<false>), dim3((unsigned)blocks), dim3(64 * MDCT2_WAVES), 0, st, T, in,
                           flags, n_cf, skip_cur, lines, scale_out, scale_stride, status_init, cf_list, cf_count);
}
