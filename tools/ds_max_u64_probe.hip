// ds_max_u64_probe -- does a contended 64-bit LDS atomic max (ds_max_u64) lose updates on
// gfx950 / ROCm 7.2?  Round 1 replaced `atomicMax((unsigned long long *)&bmax[band], bits)` in
// the band-maximum step of k_quantize (coder/codec.py:369-371: max|band| -> ScaleFactor) by two
// rounds of 32-bit atomics after a whole-file test showed a wrong scale factor about once in
// 1500 bands.  This program is that access pattern alone: 64 lanes, each owning 16 consecutive
// "lines" of a 1024-line block cut into bands (the 48 kHz band table), every lane doing one
// atomicMax per run of lines it holds of one band; the result is compared with a plain
// serial maximum.  Two variants: the 64-bit LDS atomic (8-byte aligned array, as round 1's
// `__shared__ unsigned long long bmax[32]` was) and the two-round 32-bit form that replaced it.
//   hipcc -O3 --offload-arch=gfx950 tools/ds_max_u64_probe.hip -o ds_max_u64_probe && ./ds_max_u64_probe [trials]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define NB 17                      /* slots per block in `out` (short blocks use the first 6) */
__constant__ int c_band_of[1024];

/* PER = 16: a long block (1024 lines, 17 bands); PER = 2: a short sub-block (128 lines, 6 bands),
   the case the round-1 failure was seen on */
template <int VARIANT, int PER>
__global__ __launch_bounds__(64) void k_probe(const double *__restrict__ x, unsigned long long *__restrict__ out)
{
    // VARIANT 0: 64-bit atomics (ds_max_u64), 8-byte aligned; 1: two rounds of 32-bit atomics
    __shared__ __attribute__((aligned(16))) unsigned raw[2 * 32 + 2];
    unsigned long long *bmax = (unsigned long long *)raw;
    const int lane = threadIdx.x;
    const long long blk = blockIdx.x;
    if (lane < 32) {
        raw[2 * lane] = 0u;
        raw[2 * lane + 1] = 0u;
    }
    if (lane == 0) {
        raw[64] = 0u;
        raw[65] = 0u;
    }
    __syncthreads();
    double v[PER];
    int band[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        v[j] = fabs(x[blk * 1024 + PER * lane + j]);
        band[j] = c_band_of[PER * lane + j];
    }
    if (VARIANT != 1) {
        int cur = band[0];
        double m = 0.0;
        for (int j = 0; j < PER; ++j) {
            if (band[j] != cur) {
                atomicMax(&bmax[cur], (unsigned long long)__double_as_longlong(m));
                cur = band[j];
                m = 0.0;
            }
            m = fmax(m, v[j]);
        }
        atomicMax(&bmax[cur], (unsigned long long)__double_as_longlong(m));
        __syncthreads();
        if (lane < NB)
            out[blk * NB + lane] = bmax[lane];
    } else {
        unsigned *hi_w = raw, *lo_w = raw + 32;
        int cur = band[0];
        double m = 0.0;
        for (int j = 0; j < PER; ++j) {
            if (band[j] != cur) {
                atomicMax(&hi_w[cur], (unsigned)__double2hiint(m));
                cur = band[j];
                m = 0.0;
            }
            m = fmax(m, v[j]);
        }
        atomicMax(&hi_w[cur], (unsigned)__double2hiint(m));
        __syncthreads();
        cur = band[0];
        m = 0.0;
        for (int j = 0; j < PER; ++j) {
            if (band[j] != cur) {
                if ((unsigned)__double2hiint(m) == hi_w[cur])
                    atomicMax(&lo_w[cur], (unsigned)__double2loint(m));
                cur = band[j];
                m = 0.0;
            }
            m = fmax(m, v[j]);
        }
        if ((unsigned)__double2hiint(m) == hi_w[cur])
            atomicMax(&lo_w[cur], (unsigned)__double2loint(m));
        __syncthreads();
        if (lane < NB)
            out[blk * NB + lane] = ((unsigned long long)hi_w[lane] << 32) | lo_w[lane];
    }
}

static long long run_pattern(int blocks, int per, const int *counts, int nb)
{
    /* per = 16: 1024 lines per block; per = 2: 128 lines per block (x keeps a 1024 stride) */
    const int m_lines = 64 * per;
    int band_of[1024], at = 0;
    for (int b = 0; b < nb; ++b)
        for (int k = 0; k < counts[b]; ++k)
            band_of[at++] = b;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_band_of), band_of, sizeof(band_of));
    std::vector<double> x((size_t)blocks * 1024);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < x.size(); ++i) {           // xorshift: magnitudes over many binades, many near-ties
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const int e = (int)((s >> 52) % 24);
        x[i] = ((double)(s & 0xFFFFFFFFFFFFFull) / 4503599627370496.0 - 0.5) / (double)(1ll << e);
        if ((s >> 60) == 0 && i)
            x[i] = x[i - 1];                          // exact ties
    }
    std::vector<unsigned long long> want((size_t)blocks * NB, 0ull);
    for (long long b = 0; b < blocks; ++b)
        for (int k = 0; k < m_lines; ++k) {
            double a = x[b * 1024 + k] < 0 ? -x[b * 1024 + k] : x[b * 1024 + k];
            unsigned long long bits;
            memcpy(&bits, &a, 8);
            unsigned long long &w = want[b * NB + band_of[k]];
            if (bits > w)
                w = bits;
        }
    double *dx;
    unsigned long long *dout;
    (void)hipMalloc((void **)&dx, x.size() * 8);
    (void)hipMalloc((void **)&dout, want.size() * 8);
    (void)hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice);
    std::vector<unsigned long long> got(want.size());
    const char *names[2] = {"ds_max_u64, 8-byte aligned", "two rounds of ds_max_u32"};
    long long total_bad = 0;
    for (int variant = 0; variant < 2; ++variant) {
        long long bad = 0;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipMemset(dout, 0, want.size() * 8);
            if (variant == 0 && per == 16) hipLaunchKernelGGL((k_probe<0, 16>), dim3(blocks), dim3(64), 0, 0, dx, dout);
            if (variant == 1 && per == 16) hipLaunchKernelGGL((k_probe<1, 16>), dim3(blocks), dim3(64), 0, 0, dx, dout);
            if (variant == 0 && per == 2) hipLaunchKernelGGL((k_probe<0, 2>), dim3(blocks), dim3(64), 0, 0, dx, dout);
            if (variant == 1 && per == 2) hipLaunchKernelGGL((k_probe<1, 2>), dim3(blocks), dim3(64), 0, 0, dx, dout);
            if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", names[variant]); break; }
            (void)hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost);
            for (long long b = 0; b < blocks; ++b)
                for (int k = 0; k < nb; ++k)
                    bad += got[b * NB + k] != want[b * NB + k];
        }
        printf("%-6s %-30s %lld wrong band maxima in %lld\n", per == 16 ? "long" : "short", names[variant], bad,
               5ll * blocks * nb);
        total_bad += bad;
    }
    (void)hipFree(dx);
    (void)hipFree(dout);
    return total_bad;
}

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 40000;
    const int long_counts[17] = {13, 14, 19, 17, 22, 14, 16, 19, 24, 30, 38, 47, 56, 76, 107, 149, 363};
    const int short_counts[6] = {14, 14, 13, 23, 19, 45};
    long long bad = run_pattern(blocks, 16, long_counts, 17);
    bad += run_pattern(blocks, 2, short_counts, 6);
    printf("%s\n", bad ? "LOST UPDATES SEEN" : "no lost update: ds_max_u64 under contention is exact here");
    return 0;
}
