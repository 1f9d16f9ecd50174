/*
 * wave_np_sum.h -- np.sum / np.mean of a float64 vector held in LDS, in NumPy's
 * order, computed by one wave (the result is the same in every lane).
 *
 * NumPy's pairwise sum: fewer than 8 elements are added left to right starting
 * from -0.0; up to 128 elements run eight interleaved accumulators that are
 * folded as ((0+1)+(2+3))+((4+5)+(6+7)) followed by a scalar tail; longer
 * vectors are cut at n/2 rounded down to a multiple of 8 and the two halves
 * summed the same way.  (Model checked against np.sum for lengths 5..1000.)
 */
#ifndef WAVE_NP_SUM_H
#define WAVE_NP_SUM_H

#include <hip/hip_runtime.h>

__device__ __forceinline__ double readlane_f64(double v, int src_lane)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), src_lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

/* n <= 128.  `ga` always points into LDS: the cast lets the (deliberately not
 * inlined) body use ds_read instead of flat loads. */
typedef __attribute__((address_space(3))) const double wave_np_lds_f64;

static __device__ __noinline__ double wave_np_sum_block(const double *ga, int n, int lane)
{
    wave_np_lds_f64 *a = (wave_np_lds_f64 *)ga;
    if (n < 8) {
        double r = -0.0;
        for (int i = 0; i < n; ++i)
            r = r + a[i];
        return r;
    }
    const int n8 = n - (n & 7);
    double r = 0.0;
    if (lane < 8) {
        r = a[lane];
        for (int i = 8; i < n8; i += 8)
            r = r + a[i + lane];
    }
    const double t = r + __shfl_down(r, 1, 64);
    const double u = t + __shfl_down(t, 2, 64);
    double s = u + __shfl_down(u, 4, 64);
    s = readlane_f64(s, 0);
    for (int i = n8; i < n; ++i)
        s = s + a[i];
    return s;
}

/* any n <= 128 * 2^DEPTH: the recursion of pairwise_sum, unrolled at compile
 * time (n <= 1024 wherever this is used: DEPTH 3). */
template <int DEPTH>
__device__ inline double wave_np_sum_rec(const double *a, int n, int lane)
{
    if (n <= 128)
        return wave_np_sum_block(a, n, lane);
    if constexpr (DEPTH > 0) {
        int cut = n / 2;
        cut -= cut % 8;
        const double left = wave_np_sum_rec<DEPTH - 1>(a, cut, lane);
        const double right = wave_np_sum_rec<DEPTH - 1>(a + cut, n - cut, lane);
        return left + right;
    } else {
        return __builtin_nan("");          /* longer than the callers ever pass */
    }
}

__device__ inline double wave_np_sum(const double *a, int n, int lane)
{
    return wave_np_sum_rec<3>(a, n, lane);
}

#endif /* WAVE_NP_SUM_H */
