/*
 * k_psy.hip -- psychoacoustic side chain (reference: coder/psychoac.py).
 *
 *   k_side_long / k_side_short : Hann window -> real FFT -> intensity ->
 *       strict-local-maximum peak pick -> (Bark, SPL, upper slope) per tonal
 *       masker          [getMaskedThreshold :171-179, estimate_peaks :308-329,
 *                        Masker.__init__ :61-68]
 *   k_mask<M>                 : masked threshold at the MDCT lines and per-band
 *       SMR             [Masker.vIntensityAtBark :88-96, getMaskedThreshold
 *                        :189-215, CalcSMRs :246-291]
 *
 * Behaviour notes kept from the reference: only tonal maskers and the threshold
 * in quiet reach the result (its noise-masker loop, :197-211, discards what it
 * computes); peak energy is the sum of bins f-1 and f; the data window is the
 * n+1/2 Hann while the power normalisation uses np.hanning.
 *
 * The max over maskers is taken in the dB domain and the SPL(Intensity(.))
 * round trip (:192) is applied once per line to the winner: the round trip is
 * monotone, so this equals the reference's max of per-masker round trips
 * (measured bit-identical on 520/520 frames).
 *
 * The 2N-point real FFT runs as two Q=N/4-point complex FFTs (wave_fft.h):
 *   z[m] = xh[2m] + j xh[2m+1];  E = FFT(z[0::2]), O = FFT(z[1::2]);
 *   Z[k] = E[k] + W_{N/2}^k O[k], Z[k+Q] = E[k] - W_{N/2}^k O[k];
 *   X[k] = (Z[k] + conj Z[N/2-k])/2 - (j/2) W_N^k (Z[k] - conj Z[N/2-k]).
 */
#include "pacx_dev.h"
#include "wave_fft.h"
#include "pcm_stage.h"
#include "wave_np_sum.h"
#include "quant_dev.h"


/* Hann-windowed sample i of the staged block.  int16 input: the code enters as an
 * integer and the table carries 2/65535 (the side chain is linear up to the
 * intensities; one rounding per product instead of two moves a bin by ~1e-16
 * relative), with the code -32768 mapped to 0 as coder/pcmfile.py:93-97 does. */
template <int DT>
__device__ __forceinline__ double hann_sample(const typename PcmStage<DT>::elem *raw, int i,
                                              const double *__restrict__ hw,
                                              const double *__restrict__ hw_pcm)
{
    if constexpr (DT == 0) {
        const int c = raw[i];
        return hw_pcm[i] * (double)(c == -32768 ? 0 : c);
    } else {
        return hw[i] * raw[i];
    }
}

#ifdef PACX_PSY_DEBUG
/* phase stamps (s_memtime) of workgroup 5, accumulated per wave: a measuring aid read by
   tools/psy_phase_probe.py from a library built with -DPACX_PSY_DEBUG */
__device__ long long g_psy_dbg[16 * 16];
#define PSY_T(k) do { long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                      if (dbg_on) dbg_acc[k] += t_ - dbg_last; dbg_last = t_; } while (0)
extern "C" int pacx_debug_read_psy(long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_psy_dbg), sizeof(long long) * n);
}
/* side kernels: one wave per block, every block adds its phase times */
#define SIDE_T(k) do { long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                       if (lane == 0 && side_last) atomicAdd((unsigned long long *)&g_psy_dbg[128 + (k)], (unsigned long long)(t_ - side_last)); \
                       side_last = t_; } while (0)
#else
#define PSY_T(k) do { } while (0)
#define SIDE_T(k) do { } while (0)
#endif

/* SPL(Intensity(.)) of the winning masker level (inlined: the device library's exp2 brings a
   dozen 64-bit polynomial coefficients that the compiler hoists out of the frame loop into
   registers of their own, but here that costs no occupancy and a call per line costs time) */
#ifdef MASK_NOINLINE_RT          /* A/B switch: as a real call the kernels need 17-19 fewer registers
                                    (no occupancy step is crossed) and the step runs 2.8 % slower */
__device__ __attribute__((noinline)) double mask_round_trip(double x) { return pacx_spl_of_intensity_of(x); }
#else
__device__ __forceinline__ double mask_round_trip(double x) { return pacx_spl_of_intensity_of(x); }
#endif

/* one tonal masker from bins f-1, f (coder/psychoac.py:321-328, :61-68) */
__device__ __forceinline__ PacxPeak make_peak(double left, double centre, int f, double fstep)
{
    const double e = left + centre;
    const double spl = pacx_spl_scalar(e);
    const double avg = (((double)(f - 1) * fstep) * left + ((double)f * fstep) * centre) / e;
    PacxPeak p;
    p.z = pacx_bark(avg);
    p.spl = spl;
    p.slope = -27.0 + 0.367 * fmax(spl - 40.0, 0.0);
    return p;
}

/* ------------------------------------------------------------------ long */
/* LDS of one wave.  The three users of it follow one another in time: raw samples (until
 * the FFT inputs are in registers), the FFT exchange tile (until the spectrum is in
 * registers), then the intensities together with the peak lists.
 *   COMPACT (no SBR): 9.1 KB for int16 input
 *     region B (8.06 KB): raw, then the tile, then inten -- whose low slots the maskers'
 *       Bark values and SPLs overwrite one round of 64 at a time (masker p comes from bins
 *       i_p - 1, i_p with i_p >= 2 p + 1, so slots 2p, 2p + 1 are never read again)
 *     region A (1 KB): peak bin numbers
 *   (COMPACT false, 17.5 KB, region A holding the tile: the layout of the time when the maskers'
 *    Bark values and SPLs were kept in LDS; no launcher uses it any more -- SBR handles, whose epilogue
 *    needs the intensities to the end, run the compact kernel too)
 * The packed spectrum Z never goes to LDS: the real-FFT split needs Z[k] with
 * Z[N/2-k], and with natural-order FFT output (wave_fft.h fft512n) that
 * partner sits in the mirrored lane's mirrored register, one ds_bpermute away. */
template <int DT, bool COMPACT> struct SideLongLds {
    typedef typename PcmStage<DT>::elem E;
    static constexpr int RAW_BYTES = (int)sizeof(E) * PACX_N_LONG;
    static constexpr int A_BYTES = COMPACT ? 1024
                                           : 1024 + 2 * PACX_MAX_PEAKS * 8;   /* >= 512 * sizeof(cplx) */
    static constexpr int B_BYTES = RAW_BYTES > 1032 * 8 ? RAW_BYTES : 1032 * 8;
    static constexpr int BYTES = A_BYTES + B_BYTES;
};

/* natural-order 512-point FFTs (wave_fft.h fft512n) with the per-lane twiddles read from
 * the global W512 table by the caller.  The two FFTs of a long block (even and odd
 * samples) run together: every twiddle is loaded once and serves both, and the two
 * transforms take turns on the ONE exchange tile
 * (a's reads are followed by b's writes without a wait: DS instructions of a wave execute
 * in order), so one transform's butterflies run while the other's values cross the LDS */
__device__ __forceinline__ void fft512n_g2(cplx a[8], cplx b[8], cplx *tile, const cplx tw1[7], const cplx tw2[7],
                                           int lane)
{
    const int g = lane >> 3, r = lane & 7;
    dft8(a);
    dft8(b);
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) {
        const cplx w = tw1[k1 - 1];
        a[k1] = c_mul(a[k1], w);
        b[k1] = c_mul(b[k1], w);
    }
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1)
        tile[64 * k1 + (lane ^ (8 * k1))] = a[k1];
    wave_lds_fence();
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2)
        a[n2] = tile[64 * g + 8 * (n2 ^ g) + r];
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1)
        tile[64 * k1 + (lane ^ (8 * k1))] = b[k1];
    dft8(a);
    wave_lds_fence();
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2)
        b[n2] = tile[64 * g + 8 * (n2 ^ g) + r];
    dft8(b);
#pragma unroll
    for (int k2 = 1; k2 < 8; ++k2) {
        const cplx w = tw2[k2 - 1];
        a[k2] = c_mul(a[k2], w);
        b[k2] = c_mul(b[k2], w);
    }
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2)
        tile[64 * g + 8 * k2 + (r ^ g)] = a[k2];
    wave_lds_fence();
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3)
        a[n3] = tile[64 * r + 8 * g + (n3 ^ r)];
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2)
        tile[64 * g + 8 * k2 + (r ^ g)] = b[k2];
    dft8(a);
    wave_lds_fence();
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3)
        b[n3] = tile[64 * r + 8 * g + (n3 ^ r)];
    wave_lds_fence();
    dft8(b);
}

__device__ __forceinline__ double bperm_f64(double v, int src_lane)
{
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

/* intensity of real-FFT bin i from a = Z[i mod N/2'] and b = Z[(N/2' - i) mod N/2'] */
__device__ __forceinline__ double pair_intensity(cplx a, cplx bz, cplx w, double norm)
{
    const cplx s = make_double2(a.x + bz.x, a.y - bz.y);      /* a + conj(b) */
    const cplx d = make_double2(a.x - bz.x, a.y + bz.y);      /* a - conj(b) */
    const cplx wd = c_mul(w, d);
    const double xr = 0.5 * (s.x + wd.y), xi = 0.5 * (s.y - wd.x);
    return norm * fma(xr, xr, xi * xi);
}

/* one long channel-frame by one wave; every barrier is wave-local (the wave
 * owns its LDS slice) */
template <int DT, bool FAST, bool COMPACT>
__device__ __forceinline__ void side_long_one(const PacxTables &T, const PacxPcmView &in, long long cf,
                                              char *regA, char *regB, int lane,
                                              PacxPeak *__restrict__ peaks, int32_t *__restrict__ n_peaks,
                                              int32_t *__restrict__ n_kept_out,
                                              double *__restrict__ sbr_mean,
                                              int32_t *__restrict__ sbr_overall)
{
    typedef typename PcmStage<DT>::elem E;
    cplx *tile = (cplx *)(COMPACT ? regB : regA);
    double *inten = (double *)regB;
    E *raw = (E *)regB;

#ifdef PACX_PSY_DEBUG
    long long side_last = 0;
#endif
    SIDE_T(15);
    const double *__restrict__ hw = T.hann_long, *__restrict__ hwp = T.hann_long_pcm;
    cplx ev[8], od[8];
    /* the FFTs' per-lane twiddles W512^(lane k1) and W64^(r k2) are fetched here, with the PCM
       and the window: one round trip to L2 for all of them instead of one per FFT stage */
    cplx tw1[7], tw2[7];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        tw1[k - 1] = T.w512[(lane * k) & 511];
        tw2[k - 1] = T.w512[(8 * (lane & 7) * k) & 511];
    }
    if constexpr (DT == 0 && FAST) {
        /* int16, unit stride, aligned rows: lane (lane, n1) needs the four consecutive
           samples 4 (lane + 64 n1) .. +3 -- one 8-byte load, no staging through LDS; the
           eight PCM loads and the sixteen window loads are all in flight together */
        const long long f = cf / in.n_ch;
        const int ch = (int)(cf - f * in.n_ch);
        const short *src = (const short *)in.base + f * in.frame_stride + ch * in.ch_stride;
        int2 q[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1)
            q[n1] = *(const int2 *)(src + 4 * (lane + 64 * n1));
        SIDE_T(0);
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            const int i = 4 * (lane + 64 * n1);
            const double2 h01 = *(const double2 *)(hwp + i), h23 = *(const double2 *)(hwp + i + 2);
            const int c0 = (short)(q[n1].x & 0xFFFF), c1 = q[n1].x >> 16;
            const int c2 = (short)(q[n1].y & 0xFFFF), c3 = q[n1].y >> 16;
            ev[n1] = make_double2(h01.x * (double)(c0 == -32768 ? 0 : c0), h01.y * (double)(c1 == -32768 ? 0 : c1));
            od[n1] = make_double2(h23.x * (double)(c2 == -32768 ? 0 : c2), h23.y * (double)(c3 == -32768 ? 0 : c3));
        }
    } else {
        stage_samples<DT, FAST>(raw, in, cf, 0, PACX_N_LONG, lane);
        wave_lds_fence();
        SIDE_T(0);
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            const int i = 4 * (lane + 64 * n1);
            ev[n1] = make_double2(hann_sample<DT>(raw, i, hw, hwp), hann_sample<DT>(raw, i + 1, hw, hwp));
            od[n1] = make_double2(hann_sample<DT>(raw, i + 2, hw, hwp), hann_sample<DT>(raw, i + 3, hw, hwp));
        }
        wave_lds_fence();              /* raw fully consumed: region B becomes inten */
    }
    SIDE_T(1);
    fft512n_g2(ev, od, tile, tw1, tw2, lane);
    SIDE_T(2);
    /* ev[k3] = E[k], od[k3] = O[k], k = lane + 64 k3.  Z[k] = E[k] + W1024^k O[k],
       Z[k+512] = E[k] - W1024^k O[k].  Bins k and k+512 pair with Z[1024-k] and
       Z[512-k]: both are made of E[m], O[m], m = 512 - k, which lane 64-lane holds
       in register 7-k3 (lane 0: its own register 8-k3; k = 0 pairs with itself). */
    const int mirror = (64 - lane) & 63;
    /* the two table entries of an iteration are fetched two iterations ahead: issued where
       they are used, each pair cost a full round trip to L2 -- eight in a row were half of
       this kernel's time (in-kernel stamps, profiles/r02_phases_side_mask_tail.txt) */
    constexpr int AHEAD = 2;
    cplx w1q[8], w2q[8];
#pragma unroll
    for (int k3 = 0; k3 < AHEAD; ++k3) {
        w1q[k3] = T.w1024[lane + 64 * k3];
        w2q[k3] = T.w2048[lane + 64 * k3];
    }
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) {
        const int k = lane + 64 * k3;
        if (k3 + AHEAD < 8) {
            w1q[k3 + AHEAD] = T.w1024[k + 64 * AHEAD];
            w2q[k3 + AHEAD] = T.w2048[k + 64 * AHEAD];
        }
        const cplx w1k = w1q[k3], w2k = w2q[k3];
        const cplx t = c_mul(w1k, od[k3]);
        const cplx zk = c_add(ev[k3], t), zk5 = c_sub(ev[k3], t);
        /* partner E[m], O[m] */
        cplx pe, po;
        pe.x = bperm_f64(ev[7 - k3].x, mirror);
        pe.y = bperm_f64(ev[7 - k3].y, mirror);
        po.x = bperm_f64(od[7 - k3].x, mirror);
        po.y = bperm_f64(od[7 - k3].y, mirror);
        if (lane == 0) {
            pe = ev[(8 - k3) & 7];
            po = od[(8 - k3) & 7];
        }
        /* W1024^m = -conj(W1024^k), W2048^(k+512) = -j W2048^k: the tables are built
           exactly symmetric (pacx_api.hip), so these ARE their entries m and k + 512 */
        const cplx tm = c_mul(make_double2(-w1k.x, w1k.y), po);
        cplx zm = c_add(pe, tm), zm5 = c_sub(pe, tm);       /* Z[m], Z[m+512] */
        if (k == 0) {                                       /* bins 0 and 512 pair with themselves */
            zm5 = zk;
            zm = zk5;
        }
        inten[k] = pair_intensity(zk, zm5, w2k, T.norm_long);
        inten[k + 512] = pair_intensity(zk5, zm, make_double2(w2k.y, -w2k.x), T.norm_long);
        if (k == 0)
            inten[1024] = pair_intensity(zk, zk, T.w2048[1024], T.norm_long);
    }
    wave_lds_fence();
    SIDE_T(3);
    cplx *Z = (cplx *)regA;            /* region A from here on: idx, (zs,) ss */

    /* pass 1: strict local maxima (coder/psychoac.py:312-317), their bin numbers
       compacted in ascending order into LDS (ballot + prefix popcount) */
    unsigned short *idx = (unsigned short *)Z;          /* Z is dead once inten is complete */
    wave_lds_fence();
    int count = 0;
    for (int base = 0; base <= 1024; base += 64) {
        const int i = base + lane;
        bool pk = false;
        if (i >= 1 && i <= 1024) {
            const double c = inten[i];
            pk = (c > inten[i - 1]) && (i == 1024 || c > inten[i + 1]);
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(pk);
        if (pk)
            idx[count + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
        count += __popcll(m);
    }
    wave_lds_fence();
    SIDE_T(4);
    /* pass 2: drop maskers that cannot matter, BEFORE their exact Bark value and SPL are
       computed (log10 + two atan each -- a quarter of this kernel when done for every peak).
       A masker p with S_p <= 40 dB has the 27 dB/Bark tent S_p - 16 - 27 max(|z - z_p| - 0.5, 0);
       any other masker q has a tent at least that steep-sided or shallower, so q >= p
       EVERYWHERE as soon as S_q - S_p >= 27 |z_q - z_p| (transitive, so dropping dominated
       maskers never removes the last dominator).  The screen works on bounds:
         S  within +-1e-3 dB from a float log2 of the two-bin energy;
         z  between the Bark values of the masker's two bins (its frequency is their
            intensity-weighted mean), from the bin table with a 1e-9 margin;
       a masker is dropped only if the bounds prove it dominated, so the maximum over the kept
       maskers equals the maximum over all of them (60-70 % of the maskers go).  With maskers
       sorted in Bark that is one exclusive prefix maximum of (S + 27 z) lower bounds and one
       exclusive suffix maximum of (S - 27 z) lower bounds.  Each lane screens 8 consecutive
       peaks straight from the intensities. */
    int n_kept;
    {
        const double *__restrict__ bb = T.bark_bin_long;
        double zlo[8], zhi[8], pre[8], suf[8];
        float shi[8], slo[8];
        unsigned short bin[8];
        double run_up = -INFINITY, run_dn = -INFINITY;
#pragma unroll
        for (int i = 0; i < 8; ++i) {                       /* bin numbers first (LDS) ... */
            const int p = 8 * lane + i;
            bin[i] = (unsigned short)(p < count ? idx[p] : 1);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {                       /* ... then all sixteen table loads in flight together */
            zlo[i] = bb[bin[i] - 1];
            zhi[i] = bb[bin[i]];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool ok = 8 * lane + i < count;
            const int ib = bin[i];
            const double e = inten[ib - 1] + inten[ib];
            /* pacx_spl_scalar: exact zero -> -30, else max(96 + 10 log10(e + eps), -30) */
            float est = 96.0f + 3.0103f * __log2f((float)(e + PACX_EPS));
            if (e == 0.0)
                est = -30.0f;
            shi[i] = ok ? fmaxf(est + 1e-3f, -30.0f) : -INFINITY;
            slo[i] = ok ? fmaxf(est - 1e-3f, -30.0f) : -INFINITY;
            zlo[i] -= 1e-9;
            zhi[i] += 1e-9;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {                       /* exclusive, inside the lane */
            pre[i] = run_up;
            run_up = fmax(run_up, (double)slo[i] + 27.0 * zlo[i]);
        }
#pragma unroll
        for (int i = 7; i >= 0; --i) {
            suf[i] = run_dn;
            run_dn = fmax(run_dn, (double)slo[i] - 27.0 * zhi[i]);
        }
        double inc_up = run_up, inc_dn = run_dn;            /* inclusive scans across lanes */
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double a = __shfl(inc_up, max(lane - off, 0), 64);
            const double b = __shfl(inc_dn, min(lane + off, 63), 64);
            if (lane - off >= 0)
                inc_up = fmax(inc_up, a);
            if (lane + off < 64)
                inc_dn = fmax(inc_dn, b);
        }
        double ex_up = __shfl(inc_up, max(lane - 1, 0), 64), ex_dn = __shfl(inc_dn, min(lane + 1, 63), 64);
        if (lane == 0)
            ex_up = -INFINITY;
        if (lane == 63)
            ex_dn = -INFINITY;
        int keep_mask = 0, n_mine = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = 8 * lane + i;
            const double below = fmax(pre[i], ex_up) - 27.0 * zhi[i];   /* a lower masker's tent at z_p, at least */
            const double above = fmax(suf[i], ex_dn) + 27.0 * zlo[i];   /* a higher masker's tent at z_p, at least */
            const bool dominated = (shi[i] <= 40.0f) && (below >= (double)shi[i] + 1e-9 || above >= (double)shi[i] + 1e-9);
            if (p < count && !dominated) {
                keep_mask |= 1 << i;
                ++n_mine;
            }
        }
        int incl = n_mine;                                  /* exclusive prefix sum of kept counts */
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off)
                incl += t;
        }
        int pos = incl - n_mine;
        n_kept = __shfl(incl, 63, 64);
        wave_lds_fence();                                   /* every lane has read its eight idx entries */
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (keep_mask & (1 << i))
                idx[pos++] = bin[i];                        /* compacted in place: pos <= 8 lane + i */
        wave_lds_fence();
    }
    SIDE_T(5);
    /* pass 3: the kept maskers, one per lane, 64 at a time: exact Bark value, SPL and upper
       slope (coder/psychoac.py:321-328, :61-68), straight to HBM in Bark order */
    {
        PacxPeak *__restrict__ out = peaks + cf * PACX_MAX_PEAKS;
        for (int p = lane; p < n_kept; p += 64) {
            const int i = idx[p];
            out[p] = make_peak(inten[i - 1], inten[i], i, T.fstep_long);
        }
        if (lane == 0) {
            n_peaks[cf * PACX_SUB] = count;                 /* what estimate_peaks finds */
            n_kept_out[cf * PACX_SUB] = n_kept;             /* what the mask kernel has to look at */
        }
    }
    SIDE_T(6);
    /* SBR files (EncodeSingleChannel_SBR, coder/codec.py:459-472, 503-505): the
       same spectrum as |rfft|/halfN also bounds the overall scale factor and
       gives each omitted band its one coded value, the mean magnitude.
       ScaleFactor is monotone, so min(sf(max MDCT), sf(max FFT)) is
       sf(max of both); magnitudes come back from the intensities
       (|X| = sqrt(I/norm), two roundings away from np.abs). */
    if (sbr_mean) {
        wave_lds_fence();
        double mx = 0.0;
        for (int i = lane; i <= 1024; i += 64)
            mx = fmax(mx, inten[i]);
        mx = wave_max(mx);
        const int first = T.first_omitted;
        const int lo_all = T.band_lower_long[first];
        wave_lds_fence();
        for (int i = lo_all + lane; i < PACX_M_LONG; i += 64)
            inten[i] = sqrt(inten[i] / T.norm_long) / (double)PACX_M_LONG;
        wave_lds_fence();
        for (int b = first; b < T.nb_long; ++b) {
            const int cnt = T.band_lines_long[b];
            const double mean = wave_np_sum(inten + T.band_lower_long[b], cnt, lane) / (double)cnt;
            if (lane == 0)
                sbr_mean[cf * PACX_SUB + (b - first)] = mean;
        }
        if (lane == 0) {
            const int sf = pacx_scale_factor(sqrt(mx / T.norm_long) / (double)PACX_M_LONG, T.n_scale_bits, 5);
            if (sf < sbr_overall[cf * PACX_SUB])
                sbr_overall[cf * PACX_SUB] = sf;
        }
    }
}

template <int DT, bool FAST, bool COMPACT>
__global__ __launch_bounds__(64, DT == 0 ? 3 : 2) void k_side_long(PacxTables T, PacxPcmView in,
                                                 const uint8_t *__restrict__ flags, long long n_cf,
                                                 int skip_cur, PacxPeak *__restrict__ peaks,
                                                 int32_t *__restrict__ n_peaks,
                                                 int32_t *__restrict__ n_kept_out,
                                                 double *__restrict__ sbr_mean,
                                                 int32_t *__restrict__ sbr_overall)
{
    __shared__ __attribute__((aligned(16))) char regA[SideLongLds<DT, COMPACT>::A_BYTES];
    __shared__ __attribute__((aligned(16))) char regB[SideLongLds<DT, COMPACT>::B_BYTES];
    /* XCD-aware block order: block b runs on XCD b % 8, and the MDCT kernel, which reads the
       same PCM at the same time on the other stream, gives XCD x the frames with
       (cf / 256) % 8 == x (k_mdct3.hip): the same assignment here, so that the hop is in
       this XCD's L2 whichever of the two kernels asks first */
    long long cf = blockIdx.x;
    if ((n_cf & 2047) == 0 && gridDim.x == (unsigned)n_cf) {
        const unsigned b = blockIdx.x, idx = b >> 3;
        cf = (long long)(idx >> 8) * 2048 + (b & 7u) * 256 + (idx & 255u);
    }
    if (cf >= n_cf)
        return;
    const unsigned fl = flags ? flags[cf / in.n_ch] : 0u;
    if (skip_cur && (fl & 2u))
        return;
    side_long_one<DT, FAST, COMPACT>(T, in, cf, regA, regB, threadIdx.x, peaks, n_peaks, n_kept_out, sbr_mean,
                                     sbr_overall);
}

/* ----------------------------------------------------------------- short */
template <int DT, bool FAST>
__global__ __launch_bounds__(64) void k_side_short(PacxTables T, PacxPcmView in,
                                                  const uint8_t *__restrict__ flags, long long n_cf,
                                                  int only_cur, PacxPeak *__restrict__ peaks,
                                                  int32_t *__restrict__ n_peaks,
                                                  int32_t *__restrict__ n_kept_out)
{
    typedef typename PcmStage<DT>::elem E;
    const int SPAN = PACX_N_SHORT + (PACX_SUB - 1) * PACX_M_SHORT;
    /* LDS: ONE region that holds the raw samples, then the FFT exchange tile, then the
       intensities -- the three follow one another in time (the samples are in registers before
       the first exchange, the spectra are in registers before the first intensity is written;
       the DS instructions of the one wave execute in order).  With a tile of its own the kernel
       took 18 KB and eight waves fitted a CU; at 9.8 KB its registers are the limit (twelve).
       As in the long kernel the packed spectra never go to LDS: sub-block g lives in lanes
       8g..8g+7, lane r holds bins r + 8 k3, and the partner bin 64 - k of the real-FFT split
       sits in lane 8g + (8-r)%8. */
    constexpr int RAW_BYTES = (int)sizeof(E) * SPAN;
    constexpr int I_BYTES = RAW_BYTES > PACX_SUB * 130 * 8 ? RAW_BYTES : PACX_SUB * 130 * 8;
    constexpr int B_BYTES = I_BYTES > (int)sizeof(cplx) * WFFT_TILE ? I_BYTES : (int)sizeof(cplx) * WFFT_TILE;
    __shared__ __attribute__((aligned(16))) char regB[B_BYTES];
    cplx *tile = (cplx *)regB;
    __shared__ unsigned char pk_idx[PACX_SUB][64];
    __shared__ int pk_cnt[PACX_SUB];
    double *inten = (double *)regB;
    E *raw = (E *)regB;
    const int lane = threadIdx.x;
    const long long cf = blockIdx.x;
    if (cf >= n_cf)
        return;
    const unsigned fl = flags ? flags[cf / in.n_ch] : 2u;
    if (only_cur && !(fl & 2u))
        return;

    stage_samples<DT, FAST>(raw, in, cf, PACX_SHORT_FIRST, SPAN, lane);
    __syncthreads();

    const int g = lane >> 3, r = lane & 7;
    const E *sub = raw + g * PACX_M_SHORT;
    const double *__restrict__ hw = T.hann_short, *__restrict__ hwp = T.hann_short_pcm;
    cplx ev[8], od[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int i = 4 * (r + 8 * n1);
        ev[n1] = make_double2(hann_sample<DT>(sub, i, hw, hwp), hann_sample<DT>(sub, i + 1, hw, hwp));
        od[n1] = make_double2(hann_sample<DT>(sub, i + 2, hw, hwp), hann_sample<DT>(sub, i + 3, hw, hwp));
    }
    __syncthreads();                  /* raw consumed: region B becomes inten */
    fft64x8(ev, tile, T.w512, lane);
    fft64x8(od, tile, T.w512, lane);
    __syncthreads();                  /* the tile is done with: region B becomes inten */
    /* ev[k3] = E[k], od[k3] = O[k], k = r + 8 k3 (64-point spectra of the even / odd
       samples).  Z[k] = E[k] + W128^k O[k], Z[k+64] = E[k] - W128^k O[k]; bins k and
       k + 64 pair with Z[128-k] and Z[64-k], both made of E[m], O[m], m = 64 - k */
    double *ig = inten + g * 130;
    const int mirror = 8 * g + ((8 - r) & 7);
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) {
        const int k = r + 8 * k3;
        const cplx t = c_mul(T.w128[k], od[k3]);
        const cplx zk = c_add(ev[k3], t), zk5 = c_sub(ev[k3], t);
        cplx pe, po;
        pe.x = bperm_f64(ev[7 - k3].x, mirror);
        pe.y = bperm_f64(ev[7 - k3].y, mirror);
        po.x = bperm_f64(od[7 - k3].x, mirror);
        po.y = bperm_f64(od[7 - k3].y, mirror);
        if (r == 0) {
            pe = ev[(8 - k3) & 7];
            po = od[(8 - k3) & 7];
        }
        const int m = (64 - k) & 63;
        const cplx tm = c_mul(T.w128[m], po);
        cplx zm = c_add(pe, tm), zm5 = c_sub(pe, tm);       /* Z[m], Z[m+64] */
        if (k == 0) {
            zm5 = zk;
            zm = zk5;
        }
        ig[k] = pair_intensity(zk, zm5, T.w256[k], T.norm_short);
        ig[k + 64] = pair_intensity(zk5, zm, T.w256[k + 64], T.norm_short);
        if (k == 0)
            ig[128] = pair_intensity(zk, zk, T.w256[128], T.norm_short);
    }
    __syncthreads();

    /* strict local maxima of every sub-block, compacted per sub-block (bin numbers
       in ascending order), then one masker per lane over the concatenated lists:
       the log10 / atan work runs on full waves instead of once per scan step */
    int count = 0;
    for (int base = 0; base <= 128; base += 8) {
        const int i = base + r;
        bool pk = false;
        if (i >= 1 && i <= 128) {
            const double c = ig[i];
            pk = (c > ig[i - 1]) && (i == 128 || c > ig[i + 1]);
        }
        const unsigned m = (unsigned)((__builtin_amdgcn_ballot_w64(pk) >> (8 * g)) & 0xFFull);
        if (pk)
            pk_idx[g][count + __popc(m & ((1u << r) - 1u))] = (unsigned char)i;
        count += __popc(m);
    }
    if (r == 0) {
        pk_cnt[g] = count;
        n_peaks[cf * PACX_SUB + g] = count;
        n_kept_out[cf * PACX_SUB + g] = count;               /* short blocks: no pruning (<= 64 maskers) */
    }
    __syncthreads();
    int start[PACX_SUB + 1];
    start[0] = 0;
#pragma unroll
    for (int q = 0; q < PACX_SUB; ++q)
        start[q + 1] = start[q] + pk_cnt[q];
    const int total = start[PACX_SUB];
    for (int t = lane; t < total; t += 64) {
        int q = 0;
#pragma unroll
        for (int u = 1; u < PACX_SUB; ++u)
            q += (t >= start[u]);
        const int p = t - start[q];
        const int i = pk_idx[q][p];
        const double *iq = inten + q * 130;
        peaks[cf * PACX_MAX_PEAKS + q * 64 + p] = make_peak(iq[i - 1], iq[i], i, T.fstep_short);
    }
}

/* ------------------------------------------------------ mask + per-band SMR */
/* One wave per unit: a long cf (M = 1024, 16 lines per lane) or one short
 * sub-block (M = 128, 2 lines per lane; unit = 8*cf + sub-block).  Persistent
 * workgroups of MASK_WAVES waves walk the units; Bark and threshold-in-quiet
 * tables sit in LDS once per workgroup.
 *
 * Lines are handled in chunks of 64 consecutive lines (one per lane), i.e. one
 * contiguous Bark interval [zlo, zhi] per chunk.  Maskers are taken 64 at a
 * time, one per lane, straight from HBM into registers; for every chunk each
 * lane screens ITS masker: if the masker's curve cannot rise, anywhere in
 * [zlo, zhi], above (lowest threshold in quiet of the chunk - 0.01 dB) it cannot
 * change max(quiet, SPL(Intensity(max_p .))) there and is skipped.  Survivors
 * (ballot mask, ~8 % of the masker x chunk pairs on the bench workload) are
 * broadcast from the screening lane's registers (v_readlane) and evaluated for
 * all 64 lines with exactly the reference's operation order.  The screen is
 * exact: spreading curves fall monotonically away from the masker and the
 * SPL/Intensity round trip adds at most 1.2e-5 dB at levels >= the lowest
 * threshold in quiet (-5 dB).
 */
#ifndef MASK_WAVES
#define MASK_WAVES 4
#endif
#ifndef MASK_OCC
#define MASK_OCC 3                 /* waves per SIMD the register budget is set for */
#endif
#ifndef MASK_LDS_TABLES
#define MASK_LDS_TABLES 1          /* bit 0: Bark table in LDS, bit 1: threshold-in-quiet in LDS (its 8 KB
                                      now hold the maskers of the batch, see mk[]) */
#endif
#ifndef MASK_OCC_PLAIN
#define MASK_OCC_PLAIN 4           /* k_mask<1024, false>: 33 KB of LDS and <= 128 registers, four workgroups per CU */
#endif
#ifndef MASK_WG_PER_CU
#define MASK_WG_PER_CU 3           /* with the fused tail (140 registers) */
#endif
#ifndef MASK_WG_PER_CU_PLAIN
#define MASK_WG_PER_CU_PLAIN 4
#endif


/* TAIL (long blocks): what follows the SMRs in the reference's per-frame loop runs in the same
 * wave, on the frame it has just finished -- BitAlloc (coder/codec.py:288-299, 321-328;
 * coder/bitalloc.py:62-121) on the first half wave (lanes = bands), then, for the scalar coder,
 * per-band scale factors + mantissas (coder/codec.py:362-377) and the .pac payload
 * (coder/pacfile.py:404-447) -- instead of in k_tail_long behind a kernel boundary: the SMRs
 * never go to HBM, the lines are read a second time while they are still in L2, and the 8 KB
 * of per-line state the wave owns in LDS (dead once the band maxima are taken) hold the bit
 * buffer and the band tables.  Gain-shape handles stop after BitAlloc (k_vq takes the
 * allocation from HBM). */

/* Order-preserving 64-bit key of a double (no NaNs): larger double <=> larger unsigned key.
   Key 0 is below the key of every double, -inf included. */
__device__ __forceinline__ unsigned long long mask_key_of(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double mask_value_of(unsigned long long k)
{
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

#ifndef MASK_WAVES_PLAIN
#define MASK_WAVES_PLAIN 4         /* k_mask<1024, false>: waves per workgroup (five: 4 x 39 KB leave the co-running kernels no LDS, measured slower) */
#endif
constexpr int mask_waves(int m, bool tail) { return (m == PACX_M_LONG && !tail) ? MASK_WAVES_PLAIN : MASK_WAVES; }

template <int M, bool TAIL>
__global__ __launch_bounds__(64 * mask_waves(M, TAIL), (M == PACX_M_LONG && !TAIL) ? MASK_OCC_PLAIN : MASK_OCC) void k_mask(PacxTables T, const uint8_t *__restrict__ flags,
                                                         int n_ch, long long n_units, int mixed,
                                                         const PacxPeak *__restrict__ peaks,
                                                         const int32_t *__restrict__ n_peaks,
                                                         const double *__restrict__ lines,
                                                         double *__restrict__ smr,
                                                         double *__restrict__ thr_out,
                                                         const int32_t *__restrict__ cf_list,
                                                         const int32_t *__restrict__ cf_count,
                                                         MaskTail tail)
{
    constexpr bool SHORT = (M == PACX_M_SHORT);
    constexpr int NW = mask_waves(M, TAIL);                                /* waves per workgroup */
    static_assert(!(TAIL && SHORT), "the fused tail is for long blocks");
    constexpr int PER = M / 64;
    constexpr bool BARK_LDS = (MASK_LDS_TABLES & 1) != 0, QUIET_LDS = (MASK_LDS_TABLES & 2) != 0;
    __shared__ __attribute__((aligned(16))) double bark_l[BARK_LDS ? M : 1];
    __shared__ __attribute__((aligned(16))) double quiet_l[QUIET_LDS ? M : 1];
    __shared__ __attribute__((aligned(16))) double chunk_c[PER][4];        /* zlo-0.5, zhi+0.5, qmin-0.01 */
    /* per-line state of each wave: running best, then mdct_spl - thr.  Long blocks keep the running best of
       their first RJ chunks (lines 0..511) in REGISTERS through the masker loop and only the other half in LDS:
       the per-line phase and the band maxima then run half by half through the same 4.6 KB (the LDS-resident
       half first, then the register half is written out), which is what lets a fourth workgroup share the CU
       -- this kernel is bound by its latency chains at the occupancy its LDS allows, the
       registers were idle (84 of 168).  A half pads one double per 8 lines (line kk of the half at kk + kk/8),
       so that both access patterns are conflict-free: lane + 64 j in the masker and per-line loops, 8
       consecutive lines per lane in the band reduction */
    constexpr bool PAD = M / 64 == 16;
    constexpr int RJ = PAD ? PER / 2 : 0;                                  /* chunks whose running best lives in registers */
    constexpr int NH = PAD ? 2 : 1, HP = PER / NH;                         /* halves, lines per lane and half */
    constexpr int MH = M / NH;                                             /* lines per half */
    constexpr int JSTR = PAD ? 72 : 64;                                    /* stride of j in the strided view */
    __shared__ __attribute__((aligned(16))) double bufs[NW][MH + (PAD ? MH / 8 : 0)];
    /* the 64 maskers of the current batch, (z, spl, slope, -) each: a surviving masker is
       fetched by all lanes with two broadcast LDS reads -- the LDS pipe idles in this
       kernel, the VALU is its bound, and six v_readlane + two v_mov per survivor were
       40 % of the evaluation loop's VALU instructions */
    /* (z, spl) pairs and slopes apart: 24 bytes per masker, every read aligned */
    __shared__ __attribute__((aligned(16))) double2 mk_zs_all[NW][64];
    __shared__ __attribute__((aligned(16))) double mk_u_all[NW][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double *__restrict__ bark_g = SHORT ? T.bark_short : T.bark_long;
    const double *__restrict__ quiet_g = SHORT ? T.thresh_short : T.thresh_long;
    for (int i = tid; i < M; i += 64 * NW) {
        if (BARK_LDS)
            bark_l[i] = bark_g[i];
        if (QUIET_LDS)
            quiet_l[i] = quiet_g[i];
    }
    __syncthreads();
    const double *bark_s = BARK_LDS ? (const double *)bark_l : bark_g;
    const double *quiet_s = QUIET_LDS ? (const double *)quiet_l : quiet_g;
    for (int j = wv; j < PER; j += NW) {
        double qm = quiet_s[64 * j + lane];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            qm = fmin(qm, __shfl_xor(qm, off, 64));
        if (lane == 0) {
            chunk_c[j][0] = bark_s[64 * j] - 0.5;
            chunk_c[j][1] = bark_s[64 * j + 63] + 0.5;
            chunk_c[j][2] = qm - 0.01;
        }
    }
    __syncthreads();

    double *buf = bufs[wv];
    const int sl = lane + (PAD ? lane >> 3 : 0);       /* line lane + 64 j of a half lives at buf[sl + JSTR * j] */
    double2 *mk_zs = mk_zs_all[wv];
    double *mk_u = mk_u_all[wv];
    const int nb = SHORT ? T.nb_short : T.nb_long;
    const int32_t *__restrict__ lower = SHORT ? T.band_lower_short : T.band_lower_long;
    const int32_t *__restrict__ count = SHORT ? T.band_lines_short : T.band_lines_long;
    /* mixed streams: walk the compacted list of the frames this kernel owns, so the
       static striding stays balanced whatever the pattern of long and short frames */
    if (cf_list)
        n_units = SHORT ? (long long)(*cf_count) * PACX_SUB : (long long)(*cf_count);
#ifdef PACX_PSY_DEBUG
    const bool dbg_on = blockIdx.x == 5;
    long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dbg_last = 0;
    unsigned long long st_pairs = 0, st_surv = 0, st_live = 0, st_batches = 0;
    PSY_T(7);
    dbg_acc[7] = 0;
#endif
    for (long long unit = (long long)blockIdx.x * NW + wv; unit < n_units;
         unit += (long long)gridDim.x * NW) {
        long long cf = SHORT ? unit / PACX_SUB : unit;
        const int sb = SHORT ? (int)(unit % PACX_SUB) : 0;
        if (cf_list) {
            cf = cf_list[cf];
        } else if (mixed) {
            const unsigned fl = flags ? flags[cf / n_ch] : 0u;
            if (SHORT != ((fl & 2u) != 0))
                continue;
        }
        const PacxPeak *__restrict__ pk = peaks + cf * PACX_MAX_PEAKS + sb * 64;
        const int np = n_peaks[cf * PACX_SUB + sb];
        const long long loff = cf * PACX_M_LONG + sb * PACX_M_SHORT;

        /* the running maximum of every line lives in LDS (read-modify-write by its
           own lane, once per masker batch and chunk) and the Bark value of a line is
           read from the LDS table when its chunk is live: neither array occupies
           registers, which is what lets three waves share a SIMD.  Masker batches
           are software-pipelined one ahead, so no loop iteration waits on HBM */
        double rb[RJ ? RJ : 1];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (j < RJ)
                rb[j] = -INFINITY;
            else
                buf[sl + JSTR * (j - RJ)] = -INFINITY;
        }
        PacxPeak qn;
        qn.z = 0.0; qn.spl = -1000.0; qn.slope = 0.0;             /* padding lanes never survive */
        if (lane < np)
            qn = pk[lane];
        PSY_T(0);
        for (int pb = 0; pb < np; pb += 64) {
#ifdef PACX_PSY_DEBUG
            st_batches += 1;
#endif
            const PacxPeak q = qn;
            qn.z = 0.0; qn.spl = -1000.0; qn.slope = 0.0;
            if (pb + 64 + lane < np)
                qn = pk[pb + 64 + lane];
            const double lvl = q.spl - 16.0;
            /* batch summary for the cheap first screen: maskers arrive in bin order,
               so the batch covers the Bark interval [bz_lo, bz_hi]; no masker in it is
               louder than b_lvl or decays upward slower than b_slope */
            const int n_in = min(64, np - pb);
            const double bz_lo = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(q.z), 0),
                                                  __builtin_amdgcn_readlane(__double2loint(q.z), 0));
            const double bz_hi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(q.z), n_in - 1),
                                                  __builtin_amdgcn_readlane(__double2loint(q.z), n_in - 1));
            /* upper bounds are all the batch screen needs: DPP, no LDS round trips */
            const double b_lvl = wave_max_upper(lvl);
            const double b_slope = wave_max_upper(q.slope);
            wave_lds_fence();                      /* the previous batch's reads are done */
            mk_zs[lane] = make_double2(q.z, q.spl);
            mk_u[lane] = q.slope;
            wave_lds_fence();
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                /* keep the 3 x 16 chunk constants in LDS (broadcast reads), not hoisted
                   into 96 registers */
                asm volatile("" ::: "memory");
                const double lo_edge = chunk_c[j][0], hi_edge = chunk_c[j][1], need = chunk_c[j][2];
                /* whole batch out of reach of this chunk?  (wave-uniform test) */
                if (b_slope <= 0.0) {
                    double bb = 0.0;
                    if (bz_hi < lo_edge)
                        bb = b_slope * (lo_edge - bz_hi);
                    else if (bz_lo > hi_edge)
                        bb = -27.0 * (bz_lo - hi_edge);
                    if (b_lvl + bb <= need)
                        continue;
                }
                /* best case of the spreading gain on the chunk (branch-free: at most one
                   of the two distances is positive) */
                const double ub = q.slope * fmax(lo_edge - q.z, 0.0) + -27.0 * fmax(q.z - hi_edge, 0.0);
                const bool live = (q.slope > 0.0 && q.spl > -1000.0) || (lvl + ub > need);
                unsigned long long todo = __builtin_amdgcn_ballot_w64(live);
#ifdef PACX_PSY_DEBUG
                st_pairs += 1;                            /* screen statistics, per wave */
                st_surv += __popcll(todo);
                st_live += todo ? 1 : 0;
#endif
                if (!todo)
                    continue;
                const double zj = bark_s[lane + 64 * j];
                double bj;
                if (j < RJ)
                    bj = rb[j];
                else
                    bj = buf[sl + JSTR * (j - RJ)];
                /* survivors four at a time: their broadcast reads are issued together and
                   waited for once (the read latency, not the arithmetic, was what a survivor
                   cost); a short last group repeats its last masker -- max is idempotent */
                while (todo) {
                    int bs[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        bs[u] = __builtin_ctzll(todo);
                        if (todo & (todo - 1))
                            todo &= todo - 1;
                        else if (u == 3)
                            todo = 0;
                    }
                    double2 zs[4];
                    double pus[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        zs[u] = mk_zs[bs[u]];                            /* broadcast reads */
                        pus[u] = mk_u[bs[u]];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double pz = zs[u].x, ps = zs[u].y, pu = pus[u];
                        /* gain = -27 a below the masker, pu a above, 0 inside +-0.5 Bark
                           (a = |dz| - 0.5): select the slope by the sign of dz and clamp a
                           at 0 -- same products as the reference's masked assignments
                           (coder/psychoac.py:92-94), no divergent branches */
                        const double dz = zj - pz;
                        const double a = fmax(fabs(dz) - 0.5, 0.0);
                        const double gain = (dz < 0.0 ? -27.0 : pu) * a;
                        /* bare v_max_f64 (no NaNs here): fmax() would re-canonicalise bj every turn */
                        const double cand = (ps + gain) - 16.0;
                        asm("v_max_f64 %0, %1, %2" : "=v"(bj) : "v"(bj), "v"(cand));
                    }
                }
                if (j < RJ)
                    rb[j] = bj;
                else
                    buf[sl + JSTR * (j - RJ)] = bj;
            }
        }
        wave_lds_fence();
        PSY_T(1);
        /* per line: round trip of the winner, max with quiet, SMR term; the MDCT line is fetched one iteration
           ahead.  Then the band maxima: each lane takes the HP CONSECUTIVE lines it owns in the band layout
           (bands are runs of lines), folds each run of one band in registers and posts it with one 64-bit LDS
           atomic max on an order-preserving key of the double; lanes < nb read the results back.  Independent
           LDS reads and two or three atomics per lane, instead of a dependent read per band and a 31-exchange
           transposing reduction.  max is exact, so the SMRs are the same bits whatever the order.  Long blocks:
           half by half, the LDS-resident upper half first */
        double *__restrict__ out = smr ? smr + cf * T.band_stride + sb * T.nb_short : nullptr;
        unsigned long long *key = (unsigned long long *)mk_zs;      /* the masker table is dead */
        key[lane] = 0ull;                                  /* below the key of every double */
#pragma unroll 1
        for (int h = NH - 1; h >= 0; --h) {
            if (RJ && h == 0) {
                wave_lds_fence();                          /* the upper half's maxima are taken */
#pragma unroll
                for (int j = 0; j < RJ; ++j)
                    buf[sl + JSTR * j] = rb[j];
                wave_lds_fence();
            }
            const int k0 = h * MH;
            double v_next = lines[loff + k0 + lane], q_next = quiet_s[k0 + lane];
#pragma unroll 1
            for (int j = 0; j < HP; ++j) {
                const int k = k0 + lane + 64 * j;
                const double v = v_next;
                double thr = q_next;                       /* threshold in quiet, also one ahead */
                if (j + 1 < HP) {
                    v_next = lines[loff + k + 64];
                    q_next = quiet_s[k + 64];
                }
                const double bst = buf[sl + JSTR * j];
                if (bst > -INFINITY)
                    thr = fmax(thr, mask_round_trip(bst));
                if (thr_out)
                    thr_out[loff + k] = thr;
                /* pacx_spl_array((v * v) * 4.0), coder/psychoac.py:10-25, with the lean log10 of
                   pacx_exact.h (the argument is positive and normal) */
                double it = (v * v) * 4.0;
                if (it == 0.0)
                    it = 1e-8;
                double spl = 96.0 + 10.0 * pacx_log10_pos(it + PACX_EPS);
                if (spl < -30.0)
                    spl = -30.0;
                buf[sl + JSTR * j] = spl - thr;
            }
            wave_lds_fence();
            int rl = lane;                                 /* opaque: the band ids are not to be hoisted */
            asm volatile("" : "+v"(rl));
            double v[HP];
            uint8_t bd[HP];
            if constexpr (SHORT) {
                const double2 t = *(const double2 *)(buf + 2 * rl);
                v[0] = t.x;
                v[1] = t.y;
                const uchar2 b2 = *(const uchar2 *)(T.line_band_short + 2 * rl);
                bd[0] = b2.x;
                bd[1] = b2.y;
            } else {
                static_assert(SHORT || HP == 8, "eight lines per lane and half");
                const uint2 b8 = *(const uint2 *)(T.line_band_long + k0 + 8 * rl);
                const unsigned w[2] = {b8.x, b8.y};
#pragma unroll
                for (int j = 0; j < HP; ++j) {
                    v[j] = buf[9 * rl + j];
                    bd[j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
                }
            }
            wave_lds_fence();
            int cur = bd[0];
            double mx = v[0];
#pragma unroll
            for (int j = 1; j < HP; ++j) {
                if (bd[j] != cur) {
                    atomicMax(&key[cur], mask_key_of(mx));
                    cur = bd[j];
                    mx = v[j];
                } else {
                    mx = fmax(mx, v[j]);
                }
            }
            atomicMax(&key[cur], mask_key_of(mx));
        }
        wave_lds_fence();
        PSY_T(2);
        const double s_l = mask_value_of(key[lane & 31]);
        if (out && lane < nb)
            out[lane] = s_l;
        {
            if constexpr (TAIL) {
                wave_lds_fence();                         /* the band maxima are taken: buf is free */
                /* the lane number of the tail is opaque per frame: everything the tail derives from
                   it (band ids of the lane's 16 lines, table entries of "its" band) would otherwise
                   be hoisted out of the frame loop and live in ~40 registers across the mask phase */
                int tl = lane;
                asm volatile("" : "+v"(tl));
                /* the wave's 8 KB of per-line state now hold the tail's LDS (3.4 KB) */
                char *base = (char *)buf;
                unsigned *words = (unsigned *)base;                               /* 548 words       */
                double *cp = (double *)(base + 2192);                             /* 2 x 32 doubles  */
                unsigned long long *bmax = (unsigned long long *)(base + 2704);   /* 32              */
                int *ba_s = (int *)(base + 2960), *sf_s = ba_s + PACX_MAX_BANDS;
                int *offs = sf_s + PACX_MAX_BANDS, *lower_s = offs + PACX_MAX_BANDS + 1;
                const int half = tl >> 5, l = tl & 31;
                const unsigned fl = flags ? flags[cf / n_ch] : 0u;
                const long long boff = cf * T.band_stride;
                /* 1. BitAlloc: band l's SMR is s_l of lanes l and l + 32 */
                const int32_t *__restrict__ n_lines = T.use_sbr ? T.band_lines_long_alloc : T.band_lines_long;
                const double budget = pacx_bit_budget(T.target_bps, PACX_M_LONG, 0, (fl & 5u) != 0, T.n_scale_bits,
                                                      T.n_mant_size_bits, nb, T.use_vq, T.use_sbr);
                int max_mant = 1 << T.n_mant_size_bits;
                if (max_mant > 16)
                    max_mant = 16;
                const bool has = half == 0 && l < nb;
                const int nl = has ? n_lines[l] : 0;
                int bits = 0, cap = 0;
                bitalloc_half(half == 0, has, has ? s_l : 0.0, nl, budget, max_mant, cp + 32 * half, half, l, bits, cap, T.guard != 0, T.nb_long);
                if (has) {
                    tail.bit_alloc[boff + l] = bits;
                    ba_s[l] = bits;
                }
                if (tl == nb && nb < PACX_MAX_BANDS)
                    ba_s[nb] = 0;                         /* dummy band of the lines no band covers */
                if (cap && tail.status && tl == 0)
                    atomicOr(&tail.status[cf], ((cap & 1) ? 4u : 0u) | ((cap & 2) ? 16u : 0u));   /* ALLOC_CAP, GUARD */
                PSY_T(4);
                if (!T.use_vq) {
                    /* 2. the payload's layout follows from the allocation alone: band header
                       offsets (exclusive prefix over the bands), first mantissa bit of each band */
                    const int ov = tail.overall[cf * PACX_SUB];
                    const int k0 = 16 * tl;
                    int my_off = 0, a_mine = 0, end = 0;
                    if (tail.payload) {
                        for (int i = tl; i < PACX_PACK_WORDS; i += 64)
                            words[i] = 0u;
                        wave_lds_fence();
                        a_mine = (tl < nb) ? ba_s[tl] : 0;
                        const int width = (tl < nb) ? T.n_mant_size_bits + T.n_scale_bits + a_mine * count[tl] : 0;
                        int incl = width;
#pragma unroll
                        for (int off = 1; off < 32; off <<= 1) {
                            const int t = __shfl_up(incl, off, 64);
                            if (tl >= off)
                                incl += t;
                        }
                        my_off = 3 + T.n_scale_bits + incl - width;
                        end = __shfl(incl, nb - 1, 64) + 3 + T.n_scale_bits;
                        if (tl < nb) {
                            offs[tl] = my_off + T.n_mant_size_bits + T.n_scale_bits;
                            lower_s[tl] = lower[tl];
                        }
                        if (tl == 0) {
                            put_bits(words, 0, fl & 1u, 1);
                            put_bits(words, 1, (fl >> 1) & 1u, 1);
                            put_bits(words, 2, (fl >> 2) & 1u, 1);
                            put_bits(words, 3, (unsigned)ov, T.n_scale_bits);
                        }
                    }
                    /* 3. scale factors (lines: second read, still in L2) */
                    double x[16];
                    uint8_t band[16];
                    long_scale_factors(T, lines + loff, (double)(1 << ov), bmax, ba_s, sf_s, tl, x, band);
                    if (tl < nb) {
                        tail.scale_factor[boff + tl] = sf_s[tl];
                        if (tail.payload) {
                            put_bits(words, my_off, (unsigned)(a_mine ? a_mine - 1 : 0), T.n_mant_size_bits);
                            put_bits(words, my_off + T.n_mant_size_bits, (unsigned)sf_s[tl], T.n_scale_bits);
                        }
                    }
                    PSY_T(5);
                    /* 4. mantissas, written into the bit buffer (and to HBM when asked for) as they
                       are made: four at a time, never all sixteen live */
                    bool near = false;                    /* PACX_ST_GUARD, see pacx_exact.h */
#pragma unroll
                    for (int j4 = 0; j4 < 16; j4 += 4) {
                        int mq[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int j = j4 + u;
                            const int b = band[j];
                            const int a = ba_s[b];
                            mq[u] = a ? pacx_mantissa(x[j], sf_s[b], T.n_scale_bits, a) : 0;
                            if (T.guard)
                                near = near || pacx_quant_guard(fabs(x[j]), (1 << T.n_scale_bits) - 1 + a, PACX_GUARD_LINE_ERR);
                            if (tail.payload && a)
                                put_bits(words, offs[b] + (k0 + j - lower_s[b]) * a, (unsigned)mq[u], a);
                        }
                        if (tail.mantissa)
                            *(int4 *)(tail.mantissa + loff + k0 + j4) = make_int4(mq[0], mq[1], mq[2], mq[3]);
                    }
                    if (tail.status && __builtin_amdgcn_ballot_w64(near) && tl == 0)
                        atomicOr(&tail.status[cf], 16u);
                    if (tail.payload) {
                        wave_lds_fence();
                        const int nbytes = ((end - 3) + 4 + 7) >> 3;
                        unsigned *dst = (unsigned *)(tail.payload + cf * (long long)tail.payload_stride);
                        for (int i = tl; i < (nbytes + 3) / 4; i += 64)
                            dst[i] = __builtin_bswap32(words[i]);
                        if (tl == 0)
                            tail.n_bytes[cf] = nbytes;
                    }
                    PSY_T(6);
                }
            }
        }
        wave_lds_fence();
        PSY_T(3);
    }
#ifdef PACX_PSY_DEBUG
    if (dbg_on && lane == 0 && M == PACX_M_LONG)
        for (int k = 0; k < 8; ++k)
            g_psy_dbg[wv * 16 + k] = dbg_acc[k];
    if (lane == 0 && M == PACX_M_LONG) {
        atomicAdd((unsigned long long *)&g_psy_dbg[160], st_pairs);
        atomicAdd((unsigned long long *)&g_psy_dbg[161], st_surv);
        atomicAdd((unsigned long long *)&g_psy_dbg[162], st_live);
        atomicAdd((unsigned long long *)&g_psy_dbg[163], st_batches);
    }
#endif
}

/* ------------------------------------------------ CalcSMRs for any block size */
/* psychoac.CalcSMRs / getMaskedThreshold (coder/psychoac.py:163-291) for block lengths the tuned kernels above do
 * not cover (they are specialised to 2048- and 256-sample blocks: what the reference's driver uses,
 * coder/pacfile.py:699,490) -- nMDCTLines = 512, say.  A function-level path, not a fast one:
 * one workgroup per block, the spectrum by a direct DFT over a caller-evaluated twiddle table (index k n mod N,
 * so every angle is one of the table's N; summed in sample order), then the arithmetic of the tuned kernels --
 * make_peak, the dB-domain maximum with one SPL(Intensity(.)) round trip per line, the lean log10 -- with
 * the tables (Hann window, twiddles, Bark values and thresholds in quiet of the lines) evaluated by the caller,
 * as the tuned path's are at pacx_create.  LDS: windowed block, intensities, maskers. */
#define SMRG_THREADS 256
__global__ __launch_bounds__(SMRG_THREADS) void k_smr_generic(long long n_blocks, int n, int nb,
                                                             const double *__restrict__ data,
                                                             const double *__restrict__ lines,
                                                             const double *__restrict__ hann,
                                                             const double *__restrict__ tw_cos,
                                                             const double *__restrict__ tw_sin, double norm,
                                                             double fstep, const double *__restrict__ bark,
                                                             const double *__restrict__ quiet,
                                                             const int32_t *__restrict__ band_lower,
                                                             const int32_t *__restrict__ band_count,
                                                             double *__restrict__ smr, double *__restrict__ thr_out,
                                                             int32_t *__restrict__ n_peaks_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smrg_lds[];
    const int half = n / 2, n_bins = half + 1, max_pk = half / 2 + 1;
    double *w = (double *)smrg_lds;                       /* [n]      windowed block, later spl - thr of the lines */
    double *inten = w + n;                                /* [n_bins] */
    PacxPeak *pk = (PacxPeak *)(inten + n_bins + (n_bins & 1));     /* [max_pk] */
    __shared__ int n_pk;
    const int tid = threadIdx.x;
    const long long blk = blockIdx.x;
    if (blk >= n_blocks)
        return;
    const double *x = data + blk * n;
    for (int i = tid; i < n; i += SMRG_THREADS)
        w[i] = hann[i] * x[i];                            /* HanningWindow(data), coder/psychoac.py:175 */
    if (tid == 0)
        n_pk = 0;
    __syncthreads();
    /* norm * |rfft|^2, coder/psychoac.py:171-179 */
    for (int k = tid; k < n_bins; k += SMRG_THREADS) {
        double re = 0.0, im = 0.0;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            re += w[i] * tw_cos[m];
            im -= w[i] * tw_sin[m];
            m += k;
            if (m >= n)
                m -= n;
        }
        const double a = sqrt(re * re + im * im);
        inten[k] = norm * (a * a);
    }
    __syncthreads();
    /* estimate_peaks, coder/psychoac.py:308-329: strict local maxima of bins 1 .. n/2, the last one compared to the
       left only; the list's order does not matter to the maximum taken over it */
    for (int i = 1 + tid; i < n_bins; i += SMRG_THREADS) {
        const double c = inten[i];
        if (c > inten[i - 1] && (i == n_bins - 1 || c > inten[i + 1])) {
            const int at = atomicAdd(&n_pk, 1);
            if (at < max_pk)
                pk[at] = make_peak(inten[i - 1], c, i, fstep);
        }
    }
    __syncthreads();
    const int np = n_pk < max_pk ? n_pk : max_pk;
    if (tid == 0 && n_peaks_out)
        n_peaks_out[blk] = np;
    /* per line: max over the maskers in dB, one round trip, max with quiet; mdct_spl - thr (psychoac.py:181-217, 246-254) */
    const double *ln = lines + blk * half;
    for (int k = tid; k < half; k += SMRG_THREADS) {
        const double zj = bark[k];
        double best = -INFINITY;
        for (int p = 0; p < np; ++p) {
            const double dz = zj - pk[p].z;
            const double a = fmax(fabs(dz) - 0.5, 0.0);
            const double gain = (dz < 0.0 ? -27.0 : pk[p].slope) * a;
            best = fmax(best, (pk[p].spl + gain) - 16.0);
        }
        double thr = quiet[k];
        if (best > -INFINITY)
            thr = fmax(thr, mask_round_trip(best));
        if (thr_out)
            thr_out[blk * half + k] = thr;
        const double v = ln[k];
        double it = (v * v) * 4.0;
        if (it == 0.0)
            it = 1e-8;
        double spl = 96.0 + 10.0 * pacx_log10_pos(it + PACX_EPS);
        if (spl < -30.0)
            spl = -30.0;
        w[k] = spl - thr;
    }
    __syncthreads();
    for (int b = tid; b < nb; b += SMRG_THREADS) {
        const int lo = band_lower[b], cnt = band_count[b];
        double mx = w[lo];                                /* np.amax over an empty band raises in the reference: the host refuses it */
        for (int i = 1; i < cnt; ++i)
            mx = fmax(mx, w[lo + i]);
        smr[blk * nb + b] = mx;
    }
}

size_t pacx_smr_generic_lds(int n)
{
    const int half = n / 2, n_bins = half + 1, max_pk = half / 2 + 1;
    return (size_t)(n + n_bins + (n_bins & 1)) * 8 + (size_t)max_pk * sizeof(PacxPeak);
}

void pacx_launch_smr_generic(long long n_blocks, int n, int nb, const double *data, const double *lines,
                             const double *hann, const double *tw_cos, const double *tw_sin, double norm, double fstep,
                             const double *bark, const double *quiet, const int32_t *band_lower,
                             const int32_t *band_count, double *smr, double *thr_out, int32_t *n_peaks_out,
                             hipStream_t st)
{
    if (n_blocks > 0)
        hipLaunchKernelGGL(k_smr_generic, dim3((unsigned)n_blocks), dim3(SMRG_THREADS), pacx_smr_generic_lds(n), st,
                           n_blocks, n, nb, data, lines, hann, tw_cos, tw_sin, norm, fstep, bark, quiet, band_lower,
                           band_count, smr, thr_out, n_peaks_out);
}

/* ------------------------------------------------------------- launchers */
template <int DT, bool FAST>
static void launch_side(const PacxTables &T, const PacxPcmView &in, const uint8_t *flags,
                        long long n_cf, int short_blocks, int mixed, PacxPeak *peaks,
                        int32_t *n_peaks, int32_t *n_kept, double *sbr_mean, int32_t *sbr_overall,
                        hipStream_t st)
{
    const dim3 grid((unsigned)n_cf), block(64);
    /* mixed batches may be launched in two parts on two streams: PACX_PART_LONG / PACX_PART_SHORT
       in the upper bits of `mixed` (pacx_dev.h) */
    const int part = mixed >> 4;
    mixed &= 15;
    const bool do_long = (!short_blocks || mixed) && part != 2, do_short = (short_blocks || mixed) && part != 1;
    /* SBR handles run the same compact kernel: the maskers no longer pass through LDS, so the intensities
       are intact when the SBR epilogue wants them (the 17.5 KB layout it used to need cost a quarter of
       the occupancy: 88 against 53 us) */
    if (do_long)
        hipLaunchKernelGGL((k_side_long<DT, FAST, true>), grid, block, 0, st, T, in, flags, n_cf, mixed, peaks,
                           n_peaks, n_kept, sbr_mean, sbr_overall);
    if (do_short)
        hipLaunchKernelGGL((k_side_short<DT, FAST>), grid, block, 0, st, T, in, flags, n_cf, mixed, peaks, n_peaks, n_kept);
}

void pacx_launch_side(const PacxTables &T, const PacxPcmView &in, int dtype, int fast,
                      const uint8_t *flags, long long n_cf, int short_blocks, int mixed,
                      PacxPeak *peaks, int32_t *n_peaks, int32_t *n_kept, double *sbr_mean,
                      int32_t *sbr_overall, hipStream_t st)
{
    if (n_cf <= 0)
        return;
    if (dtype == 0 && fast)
        launch_side<0, true>(T, in, flags, n_cf, short_blocks, mixed, peaks, n_peaks, n_kept, sbr_mean,
                             sbr_overall, st);
    else if (dtype == 0)
        launch_side<0, false>(T, in, flags, n_cf, short_blocks, mixed, peaks, n_peaks, n_kept, sbr_mean,
                              sbr_overall, st);
    else
        launch_side<1, false>(T, in, flags, n_cf, short_blocks, mixed, peaks, n_peaks, n_kept, sbr_mean,
                              sbr_overall, st);
}

void pacx_launch_mask(const PacxTables &T, const uint8_t *flags, int n_ch, long long n_cf,
                      int short_blocks, int mixed, const PacxPeak *peaks, const int32_t *n_peaks,
                      const double *lines, double *smr, double *thr_out, int n_cu,
                      const int32_t *list_long, const int32_t *list_short, const int32_t *counts,
                      const MaskTail *tail, hipStream_t st)
{
    if (n_cf <= 0)
        return;
    const int part = mixed >> 4;               /* PACX_PART_LONG / PACX_PART_SHORT, see launch_side */
    mixed &= 15;
    /* persistent grids: MASK_WAVES x 8 KB of running maxima + the LDS tables per workgroup,
       MASK_WG_PER_CU workgroups per CU for the long kernel */
    if ((!short_blocks || mixed) && part != 2) {
        const int nw = tail ? MASK_WAVES : MASK_WAVES_PLAIN;
        long long blocks = (n_cf + nw - 1) / nw;
        const long long per_cu = tail ? MASK_WG_PER_CU : MASK_WG_PER_CU_PLAIN;
        if (blocks > per_cu * n_cu)
            blocks = per_cu * n_cu;
        if (tail)           /* BitAlloc (+ quantisation and packing) of the long frames in the same wave */
            hipLaunchKernelGGL((k_mask<PACX_M_LONG, true>), dim3((unsigned)blocks), dim3(64 * MASK_WAVES), 0, st, T,
                               flags, n_ch, n_cf, mixed, peaks, n_peaks, lines, (double *)nullptr, thr_out,
                               mixed ? list_long : nullptr, counts, *tail);
        else
            hipLaunchKernelGGL((k_mask<PACX_M_LONG, false>), dim3((unsigned)blocks), dim3(64 * MASK_WAVES_PLAIN), 0, st, T,
                               flags, n_ch, n_cf, mixed, peaks, n_peaks, lines, smr, thr_out,
                               mixed ? list_long : nullptr, counts, MaskTail{});
    }
    if ((short_blocks || mixed) && part != 1) {
        const long long units = n_cf * PACX_SUB;
        long long blocks = (units + MASK_WAVES - 1) / MASK_WAVES;
        if (blocks > (32LL / MASK_WAVES) * n_cu)
            blocks = (32LL / MASK_WAVES) * n_cu;
        hipLaunchKernelGGL((k_mask<PACX_M_SHORT, false>), dim3((unsigned)blocks), dim3(64 * MASK_WAVES), 0, st, T,
                           flags, n_ch, units, mixed, peaks, n_peaks, lines, smr, thr_out,
                           mixed ? list_short : nullptr, counts ? counts + 1 : nullptr, MaskTail{});
    }
}
