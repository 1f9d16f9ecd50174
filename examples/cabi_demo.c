/*
 * cabi_demo.c -- the C ABI of include/pacx.h from plain C, no Python, no PyTorch:
 * encodes a synthetic 48 kHz stereo stream (long blocks, 128 kb/s/ch) into the
 * body of a .pac file and prints what it got.
 *
 *   gcc -O2 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ examples/cabi_demo.c \
 *       -L audio-codec_amd -lpacx -L /opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/audio-codec_amd -Wl,-rpath,/opt/rocm/lib -o cabi_demo
 *
 * Tables are left NULL, so the library evaluates windows / Bark / thresholds with
 * the C math library (last-place differences from NumPy's are possible; the
 * Python host passes NumPy-evaluated tables instead).
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pacx.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_PACX(h, x) do { int rc_ = (x); if (rc_ != PACX_OK) { \
    fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, pacx_last_error(h)); return 3; } } while (0)

int main(int argc, char **argv)
{
    const int n_frames = argc > 1 ? atoi(argv[1]) : 64, n_ch = 2, hop = 1024;
    /* critical-band layout of 48 kHz / 1024 lines and of the 128-line short blocks
       (coder/psychoac.py:106-160) */
    const int32_t long_bands[17] = {13, 14, 19, 17, 22, 14, 16, 19, 24, 30, 38, 47, 56, 76, 107, 149, 363};
    const int32_t short_bands[6] = {14, 14, 13, 23, 19, 45};

    pacx_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.abi_version = PACX_ABI_VERSION;
    cfg.device = 0;
    cfg.sample_rate = 48000;
    cfg.n_lines_long = 1024;
    cfg.n_lines_short = 128;
    cfg.n_scale_bits = 4;
    cfg.n_mant_size_bits = 12;
    cfg.n_bands_long = 17;
    cfg.n_bands_short = 6;
    cfg.target_bits_per_sample = 128.0 / 48.0;
    cfg.band_lines_long = long_bands;
    cfg.band_lines_short = short_bands;
    pacx_handle *h = NULL;
    if (pacx_create(&cfg, &h) != PACX_OK) {
        fprintf(stderr, "pacx_create: %s\n", pacx_last_error(NULL));
        return 1;
    }

    /* planar int16 stream with one leading hop of zeros: frame f = hops f, f+1 */
    const size_t per_ch = (size_t)(n_frames + 1) * hop;
    int16_t *pcm = (int16_t *)calloc(per_ch * n_ch, sizeof(int16_t));
    for (int c = 0; c < n_ch; ++c)
        for (size_t i = hop; i < per_ch; ++i) {
            const double t = (double)(i - hop) / 48000.0;
            const double x = 0.4 * cos(2 * M_PI * 440.0 * t + c) + 0.2 * cos(2 * M_PI * 4400.0 * t) +
                             0.01 * ((double)rand() / RAND_MAX - 0.5);
            pcm[c * per_ch + i] = (int16_t)lrint(32767.0 * x);
        }
    const long long n_cf = (long long)n_frames * n_ch;
    const int band_stride = pacx_band_stride(h), slot = pacx_payload_stride(h);
    int16_t *d_pcm;
    int32_t *d_overall, *d_sf, *d_ba, *d_nbytes;
    uint32_t *d_status;
    uint8_t *d_payload, *d_body;
    int64_t *d_total;
    const int64_t cap = n_cf * 512;
    CHECK_HIP(hipMalloc((void **)&d_pcm, per_ch * n_ch * sizeof(int16_t)));
    CHECK_HIP(hipMemcpy(d_pcm, pcm, per_ch * n_ch * sizeof(int16_t), hipMemcpyHostToDevice));
    CHECK_HIP(hipMalloc((void **)&d_overall, n_cf * 8 * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void **)&d_sf, n_cf * band_stride * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void **)&d_ba, n_cf * band_stride * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void **)&d_nbytes, n_cf * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void **)&d_status, n_cf * sizeof(uint32_t)));
    CHECK_HIP(hipMalloc((void **)&d_payload, (size_t)n_cf * slot));
    CHECK_HIP(hipMalloc((void **)&d_body, (size_t)cap));
    CHECK_HIP(hipMalloc((void **)&d_total, sizeof(int64_t)));

    pacx_pcm view;
    view.data = d_pcm;
    view.dtype = PACX_PCM_I16;
    view.n_channels = n_ch;
    view.n_frames = n_frames;
    view.frame_stride = hop;                 /* 50 % overlap */
    view.channel_stride = (int64_t)per_ch;
    view.sample_stride = 1;
    CHECK_PACX(h, pacx_reserve(h, n_cf));
    CHECK_PACX(h, pacx_encode_pack_batch(h, &view, NULL, d_overall, d_sf, d_ba, NULL, d_status, d_payload,
                                         d_nbytes, NULL));
    CHECK_PACX(h, pacx_gather_body(h, n_cf, d_payload, d_nbytes, d_body, cap, d_total, NULL));
    CHECK_HIP(hipDeviceSynchronize());

    int64_t total = 0;
    CHECK_HIP(hipMemcpy(&total, d_total, sizeof(total), hipMemcpyDeviceToHost));
    uint8_t *body = (uint8_t *)malloc((size_t)total);
    CHECK_HIP(hipMemcpy(body, d_body, (size_t)total, hipMemcpyDeviceToHost));
    /* walk the '<L nBytes' chain */
    long long blocks = 0;
    int64_t pos = 0;
    while (pos + 4 <= total) {
        const uint32_t n = body[pos] | body[pos + 1] << 8 | body[pos + 2] << 16 | (uint32_t)body[pos + 3] << 24;
        pos += 4 + n;
        ++blocks;
    }
    printf("cabi_demo: %d stereo frames -> %lld channel-blocks, %lld body bytes (%.1f kb/s/ch), chain %s\n",
           n_frames, blocks, (long long)total, 8.0 * total / n_cf / 1024.0 * 48.0,
           (blocks == n_cf && pos == total) ? "consistent" : "BROKEN");
    pacx_destroy(h);
    return (blocks == n_cf && pos == total) ? 0 : 4;
}
