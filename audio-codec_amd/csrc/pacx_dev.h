/*
 * pacx_dev.h -- device-side views shared by the kernels and the C-ABI layer.
 */
#ifndef PACX_DEV_H
#define PACX_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pacx_exact.h"

#define PACX_N_LONG 2048      /* window length of a long block               */
#define PACX_M_LONG 1024      /* MDCT lines of a long block (nMDCTLines)      */
#define PACX_N_SHORT 256
#define PACX_M_SHORT 128
#define PACX_SUB 8            /* short sub-blocks per frame                   */
#define PACX_SHORT_FIRST 448  /* first sub-block starts here (long/2-short/2) */
#define PACX_MAX_PEAKS 512    /* strict local maxima of 1025 bins             */

/* Tables resident in HBM for the life of a handle (all float64 / complex128). */
struct PacxTables {
    const double *win_long;     /* [4][2048]                                  */
    const double *win_short;    /* [256]                                      */
    const double *hann_long;    /* [2048]                                     */
    const double *hann_short;   /* [256]                                      */
    const double *ones;         /* [2048] of 1.0: MDCT of pre-windowed data     */
    const double *kbd_long;     /* [2048] KBDWindow(alpha=4), coder/window.py:53-57 */
    const double *kbd_short;    /* [256]                                      */
    const double *hann_long_pcm;   /* hann_long  * 2/65535 (int16 side chain)   */
    const double *hann_short_pcm;  /* hann_short * 2/65535                      */
    const double2 *tw_long;     /* exp(-j pi (8n+1)/8192), n < 512            */
    const double2 *tw_short;    /* exp(-j pi (8n+1)/1024), n < 64             */
    const double2 *w512;        /* exp(-2 pi j m/512),  m < 512               */
    const double2 *w1024;       /* exp(-2 pi j k/1024), k < 512               */
    const double2 *w2048;       /* exp(-2 pi j k/2048), k <= 1024             */
    const double2 *w128;        /* exp(-2 pi j k/128),  k < 64                */
    const double2 *w256;        /* exp(-2 pi j k/256),  k <= 128              */
    const double *bark_long, *thresh_long;      /* [1024]                     */
    const double *bark_short, *thresh_short;    /* [128]                      */
    const double *bark_bin_long;     /* [1025] Bark of the long FFT bin frequencies i*fstep_long: BOUNDS for the
                                        side chain's masker screen only (C math library, never in a result) */
    const int32_t *band_lower_long, *band_lines_long;     /* [nb_long]        */
    const int32_t *band_lower_short, *band_lines_short;   /* [nb_short]       */
    const uint8_t *line_band_long;   /* [1024] band of each line              */
    const uint8_t *line_band_short;  /* [128]                                 */
    double norm_long, norm_short;    /* 4/(N^2 mean(hanning^2))               */
    double fstep_long, fstep_short;  /* rfftfreq step                         */
    double target_bps;
    int nb_long, nb_short;
    int n_scale_bits, n_mant_size_bits;
    int band_stride;
    /* gain-shape / SBR variants (coder/pacfile.py:703-705, 326-327) */
    int use_vq, use_sbr;
    int first_omitted;                /* first band of sbr.omitted_bands (they are the tail) */
    const int32_t *band_lines_long_alloc;   /* nLines with omitted bands counted as 1 */
    int guard;                        /* compute PACX_ST_GUARD (pacx_config.guard) */
};

struct PacxPcmView {
    const void *base;
    long long frame_stride, ch_stride, samp_stride;
    int n_ch;
};

/* one peak (tonal masker) as the mask kernel consumes it */
struct __attribute__((aligned(8))) PacxPeak {
    double z;       /* Bark of the energy-weighted frequency                  */
    double spl;     /* SPL of the two-bin energy                              */
    double slope;   /* -27 + 0.367*max(spl-40,0): upper-side slope, dB/Bark   */
};

/* a mixed (block-switched) batch splits into two independent chains -- long-coded frames and
   short-coded frames -- that the entry points put on different streams: the launchers take
   the part in the upper bits of their `mixed` argument (0 = both parts) */
#define PACX_PART_LONG  (1 << 4)
#define PACX_PART_SHORT (2 << 4)

/* outputs of the tail fused into the long mask kernel (k_psy.hip k_mask<1024, true>) */
struct MaskTail {
    const int32_t *overall;     /* [cf][8], from the MDCT kernel              */
    int32_t *bit_alloc;         /* [cf][band_stride]                          */
    int32_t *scale_factor;      /* [cf][band_stride]                          */
    int32_t *mantissa;          /* [cf][1024] line-indexed, optional          */
    uint32_t *status;           /* [cf]                                       */
    uint8_t *payload;           /* [cf][payload_stride], optional             */
    int32_t *n_bytes;           /* [cf]                                       */
    int payload_stride;
};

#endif
