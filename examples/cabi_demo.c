/*
 * cabi_demo.c -- the C ABI of include/pacx.h from plain C, no Python, no PyTorch:
 * encodes a 48 kHz stereo stream (long blocks, 128 kb/s/ch) into the body of a .pac
 * file.
 *
 *   gcc -O2 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ examples/cabi_demo.c \
 *       -L audio-codec_amd -lpacx -L /opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/audio-codec_amd -Wl,-rpath,/opt/rocm/lib -o cabi_demo
 *
 *   cabi_demo [n_frames]                       synthetic tones + noise
 *   cabi_demo --pcm in.raw n_ch per_ch --out body.bin [--rate 44100] [--kbps 128]
 *        in.raw: planar int16, n_ch rows of per_ch samples, the first hop of every row being
 *        the prior block (zeros at the start of a file): frame f = hops f, f+1
 *
 * Every table pointer of pacx_config is left NULL: the library then uses its built-in,
 * NumPy-evaluated copies (pacx_tables_exact() == 1 at 44.1 and 48 kHz), and the band layout
 * comes from pacx_default_bands -- so this program writes the same bytes as the Python host
 * and as the reference (tests/test_gpu_cabi.py compares them with the oracle's).
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
    int n_frames = 64, n_ch = 2, rate = 48000;
    const int hop = 1024;
    double kbps = 128.0;
    const char *pcm_path = NULL, *out_path = NULL;
    size_t per_ch = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--pcm") && i + 3 < argc) {
            pcm_path = argv[i + 1];
            n_ch = atoi(argv[i + 2]);
            per_ch = (size_t)atoll(argv[i + 3]);
            i += 3;
        } else if (!strcmp(argv[i], "--out") && i + 1 < argc) {
            out_path = argv[++i];
        } else if (!strcmp(argv[i], "--rate") && i + 1 < argc) {
            rate = atoi(argv[++i]);
        } else if (!strcmp(argv[i], "--kbps") && i + 1 < argc) {
            kbps = atof(argv[++i]);
        } else {
            n_frames = atoi(argv[i]);
        }
    }
    if (pcm_path) {
        if (n_ch < 1 || per_ch < 2 * (size_t)hop || per_ch % hop) {
            fprintf(stderr, "--pcm: per_ch must be a multiple of %d and hold at least two hops\n", hop);
            return 1;
        }
        n_frames = (int)(per_ch / hop) - 1;
    } else {
        per_ch = (size_t)(n_frames + 1) * hop;
    }

    /* critical-band layout (coder/psychoac.py:106-160), evaluated by the library */
    int32_t long_bands[25], short_bands[25], nb_long = 0, nb_short = 0;
    CHECK_PACX(NULL, pacx_default_bands(rate, 1024, long_bands, &nb_long));
    CHECK_PACX(NULL, pacx_default_bands(rate, 128, short_bands, &nb_short));

    pacx_config cfg;
    memset(&cfg, 0, sizeof(cfg));                 /* every optional table: built-in */
    cfg.abi_version = PACX_ABI_VERSION;
    cfg.device = 0;
    cfg.sample_rate = rate;
    cfg.n_lines_long = 1024;
    cfg.n_lines_short = 128;
    cfg.n_scale_bits = 4;
    cfg.n_mant_size_bits = 12;
    cfg.n_bands_long = nb_long;
    cfg.n_bands_short = nb_short;
    cfg.target_bits_per_sample = kbps / (rate / 1000.0);      /* coder/pacfile.py:702 */
    cfg.band_lines_long = long_bands;
    cfg.band_lines_short = short_bands;
    pacx_handle *h = NULL;
    if (pacx_create(&cfg, &h) != PACX_OK) {
        fprintf(stderr, "pacx_create: %s\n", pacx_last_error(NULL));
        return 1;
    }

    int16_t *pcm = (int16_t *)calloc(per_ch * n_ch, sizeof(int16_t));
    if (pcm_path) {
        FILE *f = fopen(pcm_path, "rb");
        if (!f || fread(pcm, sizeof(int16_t), per_ch * n_ch, f) != per_ch * n_ch) {
            fprintf(stderr, "cannot read %zu samples from %s\n", per_ch * n_ch, pcm_path);
            return 1;
        }
        fclose(f);
    } else {
        /* planar int16 stream with one leading hop of zeros */
        for (int c = 0; c < n_ch; ++c)
            for (size_t i = hop; i < per_ch; ++i) {
                const double t = (double)(i - hop) / rate;
                const double x = 0.4 * cos(2 * M_PI * 440.0 * t + c) + 0.2 * cos(2 * M_PI * 4400.0 * t) +
                                 0.01 * ((double)rand() / RAND_MAX - 0.5);
                pcm[c * per_ch + i] = (int16_t)lrint(32767.0 * x);
            }
    }
    const long long n_cf = (long long)n_frames * n_ch;
    const int band_stride = pacx_band_stride(h), slot = pacx_payload_stride(h);
    int16_t *d_pcm;
    int32_t *d_overall, *d_sf, *d_ba, *d_nbytes;
    uint32_t *d_status;
    uint8_t *d_payload, *d_body;
    int64_t *d_total;
    const int64_t cap = n_cf * (slot + 4);
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
    if (out_path) {
        FILE *f = fopen(out_path, "wb");
        if (!f || fwrite(body, 1, (size_t)total, f) != (size_t)total) {
            fprintf(stderr, "cannot write %s\n", out_path);
            return 1;
        }
        fclose(f);
    }
    printf("cabi_demo: %d frames x %d channels -> %lld channel-blocks, %lld body bytes (%.1f kb/s/ch), chain %s, "
           "tables %s\n",
           n_frames, n_ch, blocks, (long long)total, 8.0 * total / n_cf / 1024.0 * (rate / 1000.0),
           (blocks == n_cf && pos == total) ? "consistent" : "BROKEN",
           pacx_tables_exact(h) == 1 ? "exact" : "libm");
    pacx_destroy(h);
    return (blocks == n_cf && pos == total) ? 0 : 4;
}
