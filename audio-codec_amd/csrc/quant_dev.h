/*
 * quant_dev.h -- device helpers shared by the bit-allocation / quantisation / packing kernels
 * (k_quant.hip) and the fused mask + tail kernel (k_psy.hip): the cooperative BitAlloc of one
 * (sub-)block on a half wave, scale factors + mantissas of a long block held 16 lines per
 * lane, and the MSB-first bit writer.  Every barrier in here is WAVE-local (the callers give
 * each wave its own LDS): the helpers run inside one-wave workgroups and inside the
 * independent waves of a persistent workgroup alike.
 */
#ifndef PACX_QUANT_DEV_H
#define PACX_QUANT_DEV_H

#include "pacx_dev.h"
#include "wave_fft.h"   /* wave_lds_fence, wave_max */

/* ---------------------------------------------------------------- bitalloc */
/* BitAlloc, cooperatively: lanes = bands, one (sub-)block per 32-lane half wave
 * (two per wave).  Same arithmetic, in the same order, as pacx_bit_alloc() in
 * pacx_exact.h (the serial statement of coder/bitalloc.py:77-121, checked on the
 * CPU against the oracle); tests compare the two on the GPU bit for bit.
 *   - np.sum(nLines[valid]*SMR[valid]): valid bands are compacted (prefix
 *     popcount of the ballot mask) into LDS and added in NumPy's pairwise order
 *     (8 running sums, fixed tree, scalar tail);
 *   - the rounding "ladder": the n_flip-th smallest positive fraction is found
 *     by an all-pairs rank count over the half wave instead of a sort;
 *   - np.round -> rint (half to even); the 200-pass guard is kept.
 */
__device__ __forceinline__ int half_sum_i(int v)
{
#pragma unroll
    for (int off = 16; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 32);
    return v;
}

/* BitAlloc of one (sub-)block on one 32-lane half wave (lane l = band l).  Both
 * halves of the wave must call this together (a half without work passes
 * alive = false). */
__device__ __forceinline__ void bitalloc_half(bool alive, bool has, double s, int nl, double budget,
                                              int max_mant, double *c, int half, int l, int &bits_out,
                                              int &cap_out, bool want_guard = false, int n_b = 32)
{
    /* n_b (wave-uniform): an upper bound on the band count of either half wave -- lanes l >= n_b never hold a band, so
       the all-pairs rank count need not visit them (6 for short blocks instead of 32) */
    const unsigned lt_mask = (1u << l) - 1u;

    int bits = 0, n_flip = 0, passes = 0, cap = 0;
    unsigned dropped = 0;
    bool done = !alive;
    /* The reference's loop can oscillate for ever and then leaves through its
       200-pass guard (coder/bitalloc.py:116-119; ~0.5 % of short blocks).  The
       loop is a deterministic map of (bits, dropped, n_flip), so once a state
       repeats with period L the state after the 201st pass is known: Brent's
       cycle detection (snapshot at passes 1, 2, 4, ...) finds L, the loop then
       runs only (201 - passes) mod L more passes.  Same result, ~10 passes. */
    int snap_bits = -1, snap_flip = -1, snap_pass = 0, snap_next = 1, stop_at = -1;
    unsigned snap_dropped = 0xFFFFFFFFu;
    /* Everything up to the rounding ladder depends on the set of dropped bands only;
       while the loop merely raises n_flip on a stable set it is reused, and the
       all-pairs rank count behind ladder[n_flip-1] is made once per set. */
    unsigned cache_key = 0xFFFFFFFEu;            /* never a value of `dropped` (bit 0 clear, rest set) */
    bool valid = false, posf = false, have_rank = false;
    unsigned pmask = 0;
    int nd = 0, lt = 0, le = 0;
    double want = 0.0, frac = 0.0;
    while (__builtin_amdgcn_ballot_w64(!done)) {
        if (__builtin_amdgcn_ballot_w64(dropped != cache_key)) {    /* wave-uniform: both halves recompute together */
            cache_key = dropped;
            have_rank = false;
            valid = has && !((dropped >> l) & 1u);
            const unsigned vmask = (unsigned)(__builtin_amdgcn_ballot_w64(valid) >> (32 * half));
            const int nv = __popc(vmask);
            const int pos = __popc(vmask & lt_mask);
            const int total_i = half_sum_i(valid ? nl : 0);
            double total = (double)total_i;
            if (total_i == 0)
                total = total + 1e-12;
            if (valid)
                c[pos] = (double)nl * s;
            wave_lds_fence();
            /* np.sum of c[0..nv) */
            double sum;
            if (nv < 8) {
                sum = -0.0;
                for (int i = 0; i < nv; ++i)
                    sum = sum + c[i];
            } else {
                const int n8 = nv - (nv & 7);
                double r = 0.0;
                if (l < 8) {
                    r = c[l];
                    for (int i = 8; i < n8; i += 8)
                        r = r + c[i + l];
                }
                double t = r + __shfl_down(r, 1, 32);           /* lanes 0,2,4,6: r0+r1, r2+r3, ... */
                double u = t + __shfl_down(t, 2, 32);           /* lanes 0,4 */
                sum = u + __shfl_down(u, 4, 32);                /* lane 0 */
                for (int i = n8; i < nv; ++i)
                    sum = sum + c[i];
                sum = __shfl(sum, 0, 32);
            }
            wave_lds_fence();
            const double mean = sum / total;
            want = budget / total + (1.0 / PACX_DB_PER_BIT) * (s - mean);
            frac = (want - floor(want)) - 0.5;
            posf = valid && frac > 0.0;
            pmask = (unsigned)(__builtin_amdgcn_ballot_w64(posf) >> (32 * half));
            nd = __popc(pmask);
        }
        int new_bits = bits;
        int new_flip = n_flip;
        if (__builtin_amdgcn_ballot_w64(!have_rank && n_flip > 0 && n_flip <= nd)) {   /* ladder ranks of this set */
            have_rank = true;
            lt = 0;
            le = 0;
            for (int k = 0; k < n_b; ++k) {
                const double fk = __shfl(frac, k, 32);
                if ((pmask >> k) & 1u) {
                    lt += fk < frac;
                    le += fk <= frac;
                }
            }
        }
        if (n_flip > nd) {
            new_flip = n_flip - 1;                           /* bits keep their previous values */
        } else {
            double level = 0.0;
            const bool sel = n_flip > 0 && posf && lt <= n_flip - 1 && n_flip - 1 < le;
            const unsigned smask = (unsigned)(__builtin_amdgcn_ballot_w64(sel) >> (32 * half));
            const double pick = __shfl(frac, smask ? __builtin_ctz(smask) : 0, 32);
            if (n_flip > 0)
                level = pick;                                /* ladder[n_flip-1] */
            if (valid)
                new_bits = (int)rint(want - level);
        }
        if (new_bits > max_mant)
            new_bits = max_mant;
        const bool drop = has && new_bits < 2;
        if (drop || !has)
            new_bits = 0;
        const unsigned now = (unsigned)(__builtin_amdgcn_ballot_w64(drop) >> (32 * half));
        const int spent = half_sum_i(new_bits * nl);
        if (!done) {
            const bool stable = (now == dropped);
            bits = new_bits;
            dropped = now;
            n_flip = new_flip;
            if (stable && (double)spent <= budget) {
                done = true;
            } else {
                if (stable && (double)spent > budget)
                    n_flip += 1;
                ++passes;
                if (passes > PACX_ALLOC_MAX_PASSES || passes == stop_at) {
                    cap = 1;
                    done = true;
                }
            }
        }
        /* cycle detection on the state after this pass (per half wave) */
        {
            const bool same_lane = (bits == snap_bits);
            const unsigned eq = (unsigned)(__builtin_amdgcn_ballot_w64(same_lane || !has) >> (32 * half));
            const bool same = (eq == 0xFFFFFFFFu) && dropped == snap_dropped && n_flip == snap_flip;
            if (!done && stop_at < 0 && same) {
                const int period = passes - snap_pass;
                stop_at = passes + ((PACX_ALLOC_MAX_PASSES + 1 - passes) % period);
                if (stop_at == passes) {            /* already at the state the guard would leave in */
                    cap = 1;
                    done = true;
                }
            }
            if (passes == snap_next) {
                snap_bits = bits;
                snap_dropped = dropped;
                snap_flip = n_flip;
                snap_pass = passes;
                snap_next *= 2;
            }
        }
    }
    /* PACX_ST_GUARD: a band whose final Ropt - level sits within PACX_GUARD_ALLOC_ERR of a
       rounding boundary k + 1/2 (np.round at coder/bitalloc.py:103).  The band the ladder level
       was taken from is at k + 1/2 by construction (want - (frac(want) - 1/2)), in the
       reference's arithmetic as in this one: it is not what the flag is about. */
    if (want_guard) {
        double level = 0.0;
        bool sel = false;
        if (n_flip > 0 && n_flip <= nd) {
            sel = posf && lt <= n_flip - 1 && n_flip - 1 < le;
            const unsigned smask = (unsigned)(__builtin_amdgcn_ballot_w64(sel) >> (32 * half));
            level = __shfl(frac, smask ? __builtin_ctz(smask) : 0, 32);
        }
        const double r = want - level;
        const double d = fabs((r - floor(r)) - 0.5);
        const bool near = alive && valid && !sel && d <= PACX_GUARD_ALLOC_ERR;
        if ((unsigned)(__builtin_amdgcn_ballot_w64(near) >> (32 * half)))
            cap |= 2;
    }
    bits_out = bits;
    cap_out = cap;
}

/* ---------------------------------------------------------------- quantize */
/* One wave per (sub-)block.  Each lane owns M/64 CONSECUTIVE lines (coalesced
 * 16-byte loads and stores); band maxima go through LDS atomic max on the bit
 * pattern of |x| (non-negative doubles order like integers), so no per-band
 * loop and no dependent global loads; lanes < nBands then turn the maxima into
 * scale factors in parallel. */
/* Long block: scale factors and mantissas of the 16 consecutive lines each lane
 * owns.  ba_s[nb] must be filled (and visible) by the caller; on return sf_s[nb]
 * holds the scale factors, x / band / mant this lane's lines. */
__device__ __forceinline__ void long_scale_factors(const PacxTables &T, const double *__restrict__ lin,
                                                   double up, unsigned long long *bmax, const int *ba_s,
                                                   int *sf_s, int lane, double (&x)[16], uint8_t (&band)[16])
{
    constexpr int PER = 16;
    const int nb = T.nb_long;
    const int k0 = PER * lane;
    if (lane < PACX_MAX_BANDS)
        bmax[lane] = 0ull;
#pragma unroll
    for (int j = 0; j < PER; j += 2) {
        const double2 v = *(const double2 *)(lin + k0 + j);
        x[j] = v.x * up;
        x[j + 1] = v.y * up;
    }
    {
        const uint4 b16 = *(const uint4 *)(T.line_band_long + k0);
        const unsigned w[4] = {b16.x, b16.y, b16.z, b16.w};
#pragma unroll
        for (int j = 0; j < PER; ++j)
            band[j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
    }
    wave_lds_fence();
    /* Each lane owns runs of consecutive lines of one band.  The band maximum of |x| is
       taken on the bit pattern (non-negative doubles order like unsigned integers) with one
       64-bit LDS atomic max (ds_max_u64) per run.  (Round 1 used two rounds of 32-bit atomics
       here, blaming lost updates on ds_max_u64; tools/ds_max_u64_probe.hip runs this very access
       pattern -- long and short band layouts, ties included -- 23 million band maxima without
       one wrong result on gfx950 / ROCm 7.2, and the whole-file byte-exactness tests pass on the
       64-bit form: the round-1 failure was not the hardware's.) */
    {
        int cur = band[0];
        double m = 0.0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (band[j] != cur) {
                atomicMax(&bmax[cur], (unsigned long long)__double_as_longlong(m));
                cur = band[j];
                m = 0.0;
            }
            m = fmax(m, fabs(x[j]));
        }
        atomicMax(&bmax[cur], (unsigned long long)__double_as_longlong(m));
    }
    wave_lds_fence();
    if (lane < nb)
        sf_s[lane] = pacx_scale_factor(__longlong_as_double((long long)bmax[lane]), T.n_scale_bits, ba_s[lane]);
    wave_lds_fence();
}

/* ... followed by the mantissas of the lane's 16 lines */
__device__ __forceinline__ void quantize_long_core(const PacxTables &T, const double *__restrict__ lin,
                                                   double up, unsigned long long *bmax, const int *ba_s,
                                                   int *sf_s, int lane, double (&x)[16], uint8_t (&band)[16],
                                                   int32_t (&mant)[16], bool &near)
{
    long_scale_factors(T, lin, up, bmax, ba_s, sf_s, lane, x, band);
    near = false;                                  /* PACX_ST_GUARD of this lane's lines (pacx_exact.h) */
    if (T.guard) {                                 /* wave-uniform: handles that asked for the flag */
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int b = band[j];
            const int ba = ba_s[b];
            mant[j] = ba ? pacx_mantissa(x[j], sf_s[b], T.n_scale_bits, ba) : 0;
            near = near || pacx_quant_guard(fabs(x[j]), (1 << T.n_scale_bits) - 1 + ba, PACX_GUARD_LINE_ERR);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int b = band[j];
            const int ba = ba_s[b];
            mant[j] = ba ? pacx_mantissa(x[j], sf_s[b], T.n_scale_bits, ba) : 0;
        }
    }
}

/* -------------------------------------------------------------------- pack */
#define PACX_PACK_WORDS 548            /* 2192 bytes >= 3 + 8*(4+8*16) + 1024*16 bits */

__device__ __forceinline__ void put_bits(unsigned *words, int pos, unsigned val, int width)
{
    /* stream bit p lives in word p>>5 at bit 31-(p&31) (MSB first) */
    if (width <= 0)
        return;
    val &= (width >= 32) ? 0xFFFFFFFFu : ((1u << width) - 1u);
    const int w = pos >> 5, o = pos & 31;
    const int room = 32 - o;
    if (width <= room) {
        atomicOr(&words[w], val << (room - width));
    } else {
        atomicOr(&words[w], val >> (width - room));
        atomicOr(&words[w + 1], val << (32 - (width - room)));
    }
}

#endif
