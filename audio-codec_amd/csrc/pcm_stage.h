/*
 * pcm_stage.h -- staging of one frame's PCM into LDS for the generic
 * (any stride / float64 / transition-window) kernels, and the exact
 * int16 -> signed-fraction mapping applied when reading it back.
 */
#ifndef PACX_PCM_STAGE_H
#define PACX_PCM_STAGE_H

#include "pacx_dev.h"

template <int DT> struct PcmStage;
template <> struct PcmStage<0> {              /* int16 codes */
    typedef short elem;
    static __device__ __forceinline__ double get(const short *s, int i) { return pacx_pcm16_to_f64(s[i]); }
};
template <> struct PcmStage<1> {              /* float64 signed fractions */
    typedef double elem;
    static __device__ __forceinline__ double get(const double *s, int i) { return s[i]; }
};

/* copy `count` samples starting at sample `first` of frame cf into LDS */
template <int DT, bool FAST>
__device__ __forceinline__ void stage_samples(typename PcmStage<DT>::elem *dst, const PacxPcmView &in,
                                              long long cf, int first, int count, int lane)
{
    typedef typename PcmStage<DT>::elem E;
    const long long f = cf / in.n_ch;
    const int ch = (int)(cf - f * in.n_ch);
    const E *src = (const E *)in.base + f * in.frame_stride + ch * in.ch_stride;
    if constexpr (FAST) {
        /* int16, unit stride, 16-byte aligned rows: 8 samples per lane and load */
        const int4 *s4 = (const int4 *)(src + first);
        int4 *d4 = (int4 *)dst;
        for (int i = lane; i < count / 8; i += 64)
            d4[i] = s4[i];
    } else {
        for (int i = lane; i < count; i += 64)
            dst[i] = src[(long long)(first + i) * in.samp_stride];
    }
}

#endif
