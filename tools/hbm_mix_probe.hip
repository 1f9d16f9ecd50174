// Ceiling probe for the MDCT kernel's traffic mix: per unit read 2 KB (int16 PCM hop) and
// write 8 KB (1024 float64 lines), no arithmetic to speak of.  One unit per wave at a
// time, persistent 8-wave workgroups, 16-byte accesses as in the kernel.
// Build: hipcc -O3 --offload-arch=gfx950 tools/hbm_mix_probe.hip -o audio-codec_amd/variants/hbm_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(512) void k_mix(const int4 *__restrict__ in, double2 *__restrict__ out, unsigned n_units,
                                             int reads_per_unit)
{
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned n_waves = gridDim.x * 8;
    for (unsigned u = blockIdx.x * 8 + wv; u < n_units; u += n_waves) {
        int4 acc = make_int4(0, 0, 0, 0);
        for (int j = 0; j < reads_per_unit; ++j) {          /* 2 x 1 KB = the new hop */
            const int4 x = in[(size_t)u * 128 + 64 * j + lane];
            acc.x ^= x.x; acc.y ^= x.y; acc.z ^= x.z; acc.w ^= x.w;
        }
        const double a = (double)acc.x + (double)acc.z, b = (double)acc.y + (double)acc.w;
        double2 *o = out + (size_t)u * 512;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            o[lane + 64 * k] = make_double2(a + k, b - k);
    }
}

int main(int argc, char **argv)
{
    const unsigned n_units = argc > 1 ? (unsigned)atoi(argv[1]) : 8192;
    const int reads = argc > 2 ? atoi(argv[2]) : 2;
    int4 *in; double2 *out;
    hipMalloc(&in, (size_t)n_units * 2048 + 4096);
    hipMalloc(&out, (size_t)n_units * 8192);
    hipMemset(in, 1, (size_t)n_units * 2048 + 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned blocks = std::min(256u, (n_units + 7) / 8);
    for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(512), 0, 0, in, out, n_units, reads);
    hipDeviceSynchronize();
    std::vector<float> ts;
    for (int rep = 0; rep < 20; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i)
            hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(512), 0, 0, in, out, n_units, reads);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        ts.push_back(ms / 10);
    }
    std::sort(ts.begin(), ts.end());
    const double us = ts[ts.size() / 2] * 1e3;
    const double bytes = (double)n_units * (1024.0 * reads + 8192.0);
    printf("{\"n_units\": %u, \"reads_per_unit_KB\": %d, \"us_per_launch\": %.2f, \"GBps\": %.1f, \"frac_8TBs\": %.3f}\n",
           n_units, reads, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0);
    return 0;
}
