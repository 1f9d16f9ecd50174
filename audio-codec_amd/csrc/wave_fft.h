/*
 * wave_fft.h -- FP64 complex FFTs held in the registers of ONE wave64.
 *
 * Built for the gfx950 execution model: 64 lanes x 8 complex values per lane,
 * radix-8 butterflies done in registers, data moved between lanes through a
 * per-wave LDS tile (one b128 write + one b128 read per value and exchange).
 * tools/proto_wave_fft.py is the lane-level NumPy model of exactly this
 * index math.
 *
 *   fft64x8 : eight independent 64-point FFTs, one per 8-lane group
 *             (short blocks: one sub-block per group)
 *   fft512  : one 512-point FFT = radix-8 pass + cross-lane exchange + fft64x8
 *
 * LDS tile: 8 rows x 72 complex (rows padded by 8 so that the column gather of
 * exchange 1 is bank-conflict free for ds_read_b128; exchange 2 stays inside an
 * 8-lane group and uses an XOR swizzle inside the row).
 *
 * Compiled with -ffp-contract=off: fused multiply-adds are written out.
 */
#ifndef PACX_WAVE_FFT_H
#define PACX_WAVE_FFT_H

#include <hip/hip_runtime.h>

#define WFFT_ROW 72                 /* complex values per padded row          */
#define WFFT_TILE (8 * WFFT_ROW)    /* complex values per wave tile (9216 B)  */

typedef double2 cplx;

__device__ __forceinline__ cplx c_add(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx c_sub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
/* a * b with two FMAs */
__device__ __forceinline__ cplx c_mul(cplx a, cplx b)
{
    return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}
/* a * (-j) */
__device__ __forceinline__ cplx c_mul_mj(cplx a) { return make_double2(a.y, -a.x); }

/* LDS ordering inside one wave: DS instructions of a wave execute in order, so
 * only the compiler has to be kept from moving a lane's read above the store
 * that another lane's value comes from. */
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* forward 8-point DFT in place: v[k] = sum_n v[n] exp(-2 pi i n k / 8) */
__device__ __forceinline__ void dft8(cplx v[8])
{
    const double h = 0.70710678118654752440;   /* sqrt(1/2) */
    cplx b0 = c_add(v[0], v[4]), c0 = c_sub(v[0], v[4]);
    cplx b1 = c_add(v[1], v[5]), c1 = c_sub(v[1], v[5]);
    cplx b2 = c_add(v[2], v[6]), c2 = c_sub(v[2], v[6]);
    cplx b3 = c_add(v[3], v[7]), c3 = c_sub(v[3], v[7]);
    /* odd half: c_n * W8^n */
    c1 = make_double2((c1.x + c1.y) * h, (c1.y - c1.x) * h);
    c2 = c_mul_mj(c2);
    c3 = make_double2((c3.y - c3.x) * h, -((c3.x + c3.y) * h));
    /* 4-point DFTs */
    cplx e0 = c_add(b0, b2), e1 = c_sub(b0, b2);
    cplx f0 = c_add(b1, b3), f1 = c_mul_mj(c_sub(b1, b3));
    v[0] = c_add(e0, f0);
    v[2] = c_add(e1, f1);
    v[4] = c_sub(e0, f0);
    v[6] = c_sub(e1, f1);
    cplx g0 = c_add(c0, c2), g1 = c_sub(c0, c2);
    cplx h0 = c_add(c1, c3), h1 = c_mul_mj(c_sub(c1, c3));
    v[1] = c_add(g0, h0);
    v[3] = c_add(g1, h1);
    v[5] = c_sub(g0, h0);
    v[7] = c_sub(g1, h1);
}

/*
 * Eight 64-point FFTs.  lane = 8*g + r.
 *   in : v[j]  = x_g[r + 8 j]
 *   out: v[k3] = X_g[r + 8 k3]
 * w512: table exp(-2 pi i m / 512), m < 512 (W64^q = w512[8 q]).
 */
__device__ __forceinline__ void fft64x8(cplx v[8], cplx *tile, const cplx *__restrict__ w512, int lane)
{
    const int g = lane >> 3, r = lane & 7;
    dft8(v);
#pragma unroll
    for (int k2 = 1; k2 < 8; ++k2)
        v[k2] = c_mul(v[k2], w512[8 * r * k2]);
    cplx *row = tile + g * WFFT_ROW;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2)
        row[8 * k2 + (r ^ k2)] = v[k2];
    wave_lds_fence();
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3)
        v[n3] = row[8 * r + (n3 ^ r)];
    wave_lds_fence();
    dft8(v);
}

/* the same with the per-lane twiddles W64^(r k2) taken from an LDS table w64[k2][r] */
__device__ __forceinline__ void fft64x8_lds(cplx v[8], cplx *tile, const cplx *w64, int lane)
{
    const int g = lane >> 3, r = lane & 7;
    dft8(v);
#pragma unroll
    for (int k2 = 1; k2 < 8; ++k2)
        v[k2] = c_mul(v[k2], w64[8 * k2 + r]);
    cplx *row = tile + g * WFFT_ROW;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2)
        row[8 * k2 + (r ^ k2)] = v[k2];
    wave_lds_fence();
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3)
        v[n3] = row[8 * r + (n3 ^ r)];
    wave_lds_fence();
    dft8(v);
}

/*
 * One 512-point FFT.
 *   in : v[n1] = x[lane + 64 n1]
 *   out: v[k3] = X[g + 8 r + 64 k3]   with g = lane>>3, r = lane&7
 */
__device__ __forceinline__ void fft512(cplx v[8], cplx *tile, const cplx *__restrict__ w512, int lane)
{
    dft8(v);
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1)
        v[k1] = c_mul(v[k1], w512[lane * k1]);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1)
        tile[k1 * WFFT_ROW + lane] = v[k1];
    wave_lds_fence();
    const int g = lane >> 3, r = lane & 7;
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2)
        v[n2] = tile[g * WFFT_ROW + 8 * n2 + r];
    wave_lds_fence();
    fft64x8(v, tile, w512, lane);
}

/*
 * 512-point FFT, natural order in AND out, on an unpadded 512-complex tile:
 *   in : v[n1] = x[lane + 64 n1]        out: v[k3] = X[lane + 64 k3]
 * Both exchanges are bank-conflict free through XOR swizzles (found by
 * exhaustive search over the ds_write_b128 / ds_read_b128 lane groups of
 * gfx950, tools/proto_wave_fft.py:fft512n is the model):
 *   X1 write  row k1, col lane ^ 8 k1        X1 read  row g, col 8 (n2 ^ g) + r
 *   X2 write  row g,  col 8 k2 + (r ^ g)     X2 read  row lane&7, col 8 (lane>>3) + (n3 ^ (lane&7))
 * w1[(k-1) w1_stride] = W512^(lane k), w2[(k-1) w2_stride] = W64^((lane&7) k),
 * k = 1..7: per-lane twiddles, normally LDS tables shared by the workgroup.
 */
#define WFFT_TILE_N 512
__device__ __forceinline__ void fft512n(cplx v[8], cplx *tile, const cplx *w1, int w1_stride, const cplx *w2,
                                        int w2_stride, int lane)
{
    const int g = lane >> 3, r = lane & 7;
    dft8(v);
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1)
        v[k1] = c_mul(v[k1], w1[(k1 - 1) * w1_stride]);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1)
        tile[64 * k1 + (lane ^ (8 * k1))] = v[k1];
    wave_lds_fence();
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2)
        v[n2] = tile[64 * g + 8 * (n2 ^ g) + r];
    wave_lds_fence();
    dft8(v);
#pragma unroll
    for (int k2 = 1; k2 < 8; ++k2)
        v[k2] = c_mul(v[k2], w2[(k2 - 1) * w2_stride]);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2)
        tile[64 * g + 8 * k2 + (r ^ g)] = v[k2];
    wave_lds_fence();
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3)
        v[n3] = tile[64 * r + 8 * g + (n3 ^ r)];
    wave_lds_fence();
    dft8(v);
}

/* index of the value held in register k3 after fft512 */
__device__ __forceinline__ int fft512_out_index(int lane, int k3) { return (lane >> 3) + 8 * (lane & 7) + 64 * k3; }
/* index (inside the group) of the value held in register k3 after fft64x8 */
__device__ __forceinline__ int fft64_out_index(int lane, int k3) { return (lane & 7) + 8 * k3; }

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

/* Upper bound of the wave maximum of v, without an LDS round trip: v is rounded UP to
 * float and the 32-bit maximum runs on the DPP network (row shifts, then row
 * broadcasts; max is idempotent, so no bank masks are needed), result from lane 63.
 * For screens that only need "no lane exceeds this". */
__device__ __forceinline__ double wave_max_upper(double v)
{
    float f = (float)v;
    if ((double)f < v)
        f = __int_as_float(__float_as_int(f) + (f >= 0.0f ? 1 : -1));     /* next float up (v finite) */
    int x = __float_as_int(f);
#define WMU_STEP(CTRL)                                                                            \
    x = __float_as_int(fmaxf(__int_as_float(x),                                                   \
                             __int_as_float(__builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false))))
    WMU_STEP(0x111);        /* row_shr:1 */
    WMU_STEP(0x112);        /* row_shr:2 */
    WMU_STEP(0x114);        /* row_shr:4 */
    WMU_STEP(0x118);        /* row_shr:8 */
    WMU_STEP(0x142);        /* row_bcast:15 */
    WMU_STEP(0x143);        /* row_bcast:31 */
#undef WMU_STEP
    return (double)__int_as_float(__builtin_amdgcn_readlane(x, 63));
}

/* Maxima of N (8, 16 or 32) per-lane values over the wave, all at once: at every xor step a
 * lane keeps half of its values and hands the other half to its partner, so the exchange
 * volume halves per step (16+8+4+2+1+1 values = 64 ds_bpermute for 32 maxima, six
 * dependent LDS round trips in all -- against six per maximum for wave_max); once one value
 * is left the remaining steps are plain.  On return m[0] of lane l holds the wave maximum
 * of value number l / (64 / N). */
/* one halving step with compile-time n and offset: every index of m[] is a constant, so the
 * array stays in registers (a run-time n made m[n + i] a dynamic index and sent the whole
 * array to scratch: 272 B per lane in k_mask<1024>, 150 MB of writes per launch) */
template <int HALF, int OFF, int N>
__device__ __forceinline__ void wave_max_n_step(double (&m)[N], int lane)
{
    if constexpr (HALF >= 1) {
        const bool up = (lane & OFF) != 0;
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
            const double keep = up ? m[HALF + i] : m[i], send = up ? m[i] : m[HALF + i];
            m[i] = fmax(keep, __shfl_xor(send, OFF, 64));
        }
        wave_max_n_step<HALF / 2, OFF / 2, N>(m, lane);
    } else if constexpr (OFF >= 1) {
        m[0] = fmax(m[0], __shfl_xor(m[0], OFF, 64));
        wave_max_n_step<0, OFF / 2, N>(m, lane);
    }
}

template <int N>
__device__ __forceinline__ void wave_max_n(double (&m)[N], int lane)
{
    static_assert(N == 8 || N == 16 || N == 32, "wave_max_n");
    wave_max_n_step<N / 2, 32, N>(m, lane);
}

#endif
