/*
 * pacx_vq_tables.h -- host-side construction of the pyramid-VQ tables the
 * gain-shape kernel (k_vq.hip) reads.  Plain C++ (also built by
 * tests/hostcheck with g++ and compared with the oracle).
 *
 *   N(l,k)  codebook size of dimension l with k pulses
 *           (coder/gain_shape_quantize.py:69-102)
 *   P(l,k)  = sum_{j<=k} N(l,j): turns the inner sum of encode_pvq_vector
 *           (:117-119) into two lookups
 *   K(l,b)  largest k with N(l,k) <= 2^b, and the index width
 *           ceil(log2(N(l,K) + eps)) (pvq_compute_k_for_R, :275-291)
 *
 * Rows 0..2 have closed forms (N(1,k)=2, N(2,k)=4k for k>=1) and are not
 * stored: K(2,b) reaches 2^30.  Row l >= 3 is stored up to the first entry
 * above 2^32, which is all any leaf of at most 32 bits can touch.
 */
#ifndef PACX_VQ_TABLES_H
#define PACX_VQ_TABLES_H

#include <math.h>
#include <stdint.h>

#include <vector>

#define PACX_VQ_MAX_BITS 32     /* SPLIT_BITS, coder/gain_shape_quantize.py:27 */

struct PacxVqHostTables {
    int l_max = 0;
    std::vector<int32_t> k_of;        /* [(l_max+1)*33]; -1: the reference does not terminate */
    std::vector<uint8_t> w_of;        /* [(l_max+1)*33]                                       */
    std::vector<int32_t> row_off;     /* [l_max+1] start of row l in n_tab / p_tab (l >= 3)   */
    std::vector<int32_t> row_len;     /* [l_max+1]                                            */
    std::vector<uint64_t> n_tab, p_tab;
    std::vector<double> half_log2;    /* [l_max+1] 0.5*log2(l), gain_shape_alloc :59          */
};

static inline int pacx_vq_width(uint64_t n)
{
    /* ceil(log2(n + eps)): 1 for n == 1, else bit length of n-1 */
    if (n <= 1)
        return 1;
    int w = 0;
    for (uint64_t v = n - 1; v; v >>= 1)
        ++w;
    return w;
}

static inline void pacx_vq_build(int l_max, const double *half_log2_opt, PacxVqHostTables *t)
{
    const uint64_t cap = (uint64_t)1 << PACX_VQ_MAX_BITS;
    t->l_max = l_max;
    t->k_of.assign((size_t)(l_max + 1) * 33, 0);
    t->w_of.assign((size_t)(l_max + 1) * 33, 0);
    t->row_off.assign(l_max + 1, 0);
    t->row_len.assign(l_max + 1, 0);
    t->half_log2.assign(l_max + 1, 0.0);
    t->n_tab.clear();
    t->p_tab.clear();
    for (int l = 1; l <= l_max; ++l)
        t->half_log2[l] = half_log2_opt ? half_log2_opt[l] : 0.5 * log2((double)l);
    std::vector<uint64_t> prev, cur;          /* rows l-1 and l, up to the first entry > cap */
    for (int l = 1; l <= l_max; ++l) {
        for (int b = 0; b <= 32; ++b) {
            t->k_of[(size_t)l * 33 + b] = -1;
            t->w_of[(size_t)l * 33 + b] = 0;
        }
        if (l == 1)
            continue;                          /* N(1,k) = 2 never exceeds 2^b: no K exists */
        if (l == 2) {
            for (int b = 1; b <= 32; ++b) {
                const uint64_t k = (b >= 2) ? ((uint64_t)1 << (b - 2)) : 0;
                t->k_of[2 * 33 + b] = (int32_t)k;
                t->w_of[2 * 33 + b] = (uint8_t)pacx_vq_width(k ? 4 * k : 1);
            }
            continue;
        }
        cur.clear();
        cur.push_back(1);
        if (l == 3) {                          /* N(3,k) = 4k^2 + 2 (row 2 is not stored) */
            for (uint64_t k = 1;; ++k) {
                cur.push_back(4 * k * k + 2);
                if (cur.back() > cap)
                    break;
            }
        } else {
            for (size_t k = 1; k < prev.size(); ++k) {
                const uint64_t v = prev[k] + prev[k - 1] + cur[k - 1];
                cur.push_back(v);
                if (v > cap)
                    break;
            }
        }
        t->row_off[l] = (int32_t)t->n_tab.size();
        t->row_len[l] = (int32_t)cur.size();
        uint64_t run = 0;
        for (size_t k = 0; k < cur.size(); ++k) {
            run += cur[k];
            t->n_tab.push_back(cur[k]);
            t->p_tab.push_back(run);
        }
        for (int b = 1; b <= 32; ++b) {
            const uint64_t lim = (uint64_t)1 << b;
            size_t k = 0;
            while (k + 1 < cur.size() && cur[k + 1] <= lim)
                ++k;
            t->k_of[(size_t)l * 33 + b] = (int32_t)k;
            t->w_of[(size_t)l * 33 + b] = (uint8_t)pacx_vq_width(cur[k]);
        }
        prev.swap(cur);
    }
}

#endif /* PACX_VQ_TABLES_H */
