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
 * op 2: vMantissa(x, scale, a, b) */
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
    } else {
        out[i] = pacx_mantissa(v, scale, a, b);
    }
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

void pacx_launch_bitalloc_generic(long long n, int nb, const int32_t *n_lines, const double *budget,
                                  int max_mant, const double *smr, int32_t *bits, hipStream_t st)
{
    if (n > 0)
        hipLaunchKernelGGL(k_bitalloc_generic, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, n, nb, n_lines,
                           budget, max_mant, smr, bits);
}
