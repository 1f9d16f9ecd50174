/*
 * pacx_api.hip -- the C ABI of include/pacx.h: handle, resident tables,
 * grow-only workspace, argument checking, kernel sequencing on a HIP stream.
 */
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/pacx.h"
#include "pacx_dev.h"
#include "pacx_vq_tables.h"
#include "pacx_tables_gen.h"

/* kernels (k_mdct.hip, k_psy.hip, k_quant.hip) */
void pacx_launch_mdct(const PacxTables &T, const PacxPcmView &in, int dtype, int fast,
                      const uint8_t *flags, long long n_cf, int short_blocks, int mixed, int prewin,
                      double *lines, int32_t *scale_out, int scale_stride, uint32_t *status,
                      hipStream_t st);
void pacx_launch_mdct_v2(const PacxTables &T, const PacxPcmView &in, const uint8_t *flags, long long n_cf,
                         int skip_cur, double *lines, int32_t *scale_out, int scale_stride,
                         uint32_t *status_init, int n_cu, const int32_t *cf_list, const int32_t *cf_count,
                         hipStream_t st);
void pacx_launch_side(const PacxTables &T, const PacxPcmView &in, int dtype, int fast,
                      const uint8_t *flags, long long n_cf, int short_blocks, int mixed,
                      PacxPeak *peaks, int32_t *n_peaks, int32_t *n_kept, double *sbr_mean,
                      int32_t *sbr_overall, hipStream_t st);
void pacx_launch_mask(const PacxTables &T, const uint8_t *flags, int n_ch, long long n_cf,
                      int short_blocks, int mixed, const PacxPeak *peaks, const int32_t *n_peaks,
                      const double *lines, double *smr, double *thr_out, int n_cu,
                      const int32_t *list_long, const int32_t *list_short, const int32_t *counts,
                      const MaskTail *tail, hipStream_t st);
void pacx_launch_frame_lists(const uint8_t *flags, long long n_frames, int n_ch, int32_t *list_long,
                             int32_t *list_short, int32_t *counts, hipStream_t st);
void pacx_launch_bitalloc(const PacxTables &T, const uint8_t *flags, int n_ch, long long n_cf,
                          int short_blocks, int mixed, int skip_long, const double *smr, int32_t *bit_alloc,
                          uint32_t *status, hipStream_t st);
void pacx_launch_quantize(const PacxTables &T, const uint8_t *flags, int n_ch, long long n_cf,
                          int short_blocks, int mixed, const double *lines, const int32_t *overall,
                          int overall_stride, const int32_t *bit_alloc, int32_t *scale_factor,
                          int32_t *mantissa, hipStream_t st);
void pacx_launch_pack(const PacxTables &T, const uint8_t *flags, int n_ch, long long n_cf,
                      const int32_t *overall, const int32_t *scale_factor, const int32_t *bit_alloc,
                      const int32_t *mantissa, const uint32_t *status, uint8_t *payload,
                      int payload_stride, int32_t *n_bytes, hipStream_t st);
void pacx_launch_tail(const PacxTables &T, const uint8_t *flags, int n_ch, long long n_cf, const double *smr,
                      const double *lines, const int32_t *overall, int32_t *bit_alloc, int32_t *scale_factor,
                      int32_t *mantissa, uint32_t *status, uint8_t *payload, int payload_stride,
                      int32_t *n_bytes, const int32_t *list_short, const int32_t *count_short, int skip_long,
                      hipStream_t st);
void pacx_launch_gather(long long n_cf, const uint8_t *payload, int payload_stride,
                        const int32_t *n_bytes, long long *chunk_buf, long long *offs_buf, uint8_t *body,
                        long long capacity, long long *total, hipStream_t st);

void pacx_launch_window(const double *win, long long n_rows, int len, const double *x, double *y,
                        hipStream_t st);
void pacx_launch_dequant_elem(int op, long long n, const int64_t *codes, int scale, int a, int b, double *out,
                              hipStream_t st);
void pacx_launch_mdct_direct(long long n_rows, int a, int b, int inverse, const double *x, double *y, hipStream_t st);
void pacx_launch_imdct_plain(const PacxTables &T, long long n_rows, int short_blocks, const double *lines,
                             double *blocks, hipStream_t st);
void pacx_launch_quant_elem(int op, long long n, const double *x, int scale, int a, int b, int64_t *out,
                            hipStream_t st);
void pacx_launch_bitalloc_generic(long long n, int nb, const int32_t *n_lines, const double *budget,
                                  int max_mant, const double *smr, int32_t *bits, hipStream_t st);

void pacx_launch_transient_f64(long long n_blocks, int n_ch, int n, const double *blocks, double thresh, uint8_t *out,
                               hipStream_t st);
void pacx_launch_transient(const PacxPcmView &in, long long n_hops, int hop, uint8_t *transient,
                           uint8_t *flags, hipStream_t st);

void pacx_launch_sbr_scalar_lines(const PacxTables &T, long long n_cf, const uint8_t *cf_flags,
                                  const int32_t *scale_factor, const int32_t *bit_alloc, const int32_t *mantissa,
                                  double *lines, uint8_t *sbr_flag, int routing, hipStream_t st);
void pacx_launch_sbr_recon(const PacxTables &T, const void *view, long long n_cf, const uint8_t *sbr_flag,
                           double *lines, uint32_t *status, hipStream_t st);
void pacx_launch_unpack(const PacxTables &T, long long n_cf, const uint8_t *payload, int payload_stride,
                        const long long *offsets, const int32_t *n_bytes, uint8_t *flags_out, int32_t *overall,
                        int32_t *scale_factor, int32_t *bit_alloc, int32_t *mantissa, uint32_t *status,
                        hipStream_t st);
void pacx_launch_decode(const PacxTables &T, long long n_blocks, int n_ch, const uint8_t *cf_flags,
                        const int32_t *overall, const int32_t *scale_factor, const int32_t *bit_alloc,
                        const int32_t *mantissa, const double *lines_in, double *blocks, int16_t *pcm,
                        hipStream_t st);
/* k_vq_dec.hip */
size_t pacx_vqdec_view_size(void);
void pacx_vqdec_view_fill(void *dst, const uint64_t *n_tab, const uint64_t *p_tab, const int32_t *row_off,
                          const int32_t *k_of, const uint8_t *w_of, const double *half_log2, int l_max,
                          const double *log2_tan, const double *gauss, int gauss_r, const double *line_freq,
                          const int32_t *sizes_long, int nb_long, const int32_t *sizes_short, int nb_short);
void pacx_launch_vq_dec(const PacxTables &T, const void *view, long long n_cf, const uint8_t *payload,
                        int payload_stride, const long long *offsets, const int32_t *n_bytes,
                        uint8_t *cf_flags, int32_t *overall, int32_t *bit_alloc, double *lines,
                        uint8_t *sbr_flag, uint32_t *status, hipStream_t st);

/* k_vq.hip */
void pacx_launch_vq(const PacxTables &T, const void *vq_view, const uint8_t *flags, int n_ch, long long n_cf,
                    const double *lines, const int32_t *overall, int32_t *bit_alloc, const double *sbr_mean,
                    uint32_t *status, uint8_t *payload, int payload_stride, int32_t *n_bytes,
                    unsigned *unit_words, int32_t *unit_bits, pacx_vq_entry *log, int32_t *log_count,
                    int log_cap, int stage, const int32_t *cf_list, const int32_t *cf_count, hipStream_t st);
size_t pacx_vq_view_size(void);
void pacx_vq_view_fill(void *dst, const uint64_t *n_tab, const uint64_t *p_tab, const int32_t *row_off,
                       const int32_t *k_of, const uint8_t *w_of, const double *half_log2, int l_max,
                       double log_mu1, const double *log2_tan, const int32_t *sizes_long, int nb_long,
                       const int32_t *sizes_short, int nb_short);

#define PACX_PAYLOAD_STRIDE 2192
#define PACX_VQ_UNIT_WORDS 548

struct pacx_handle {
    int device;
    int n_cu;                         /* compute units (persistent-kernel grid sizing) */
    PacxTables T;
    std::vector<void *> owned;        /* table allocations                      */
    /* workspace (device), sized for ws_cf channel-frames */
    long long ws_cf;
    double *ws_lines;                 /* [ws_cf][1024]                          */
    double *ws_smr;                   /* [ws_cf][band_stride]                   */
    PacxPeak *ws_peaks;               /* [ws_cf][512]                           */
    int32_t *ws_npeaks;               /* [ws_cf][8] maskers found               */
    int32_t *ws_nkept;                /* [ws_cf][8] maskers left after pruning  */
    int32_t *ws_overall;              /* [ws_cf][8]                             */
    long long *ws_chunks;             /* [ws_cf/256 + 2]                        */
    long long *ws_offs;               /* [ws_cf]                                */
    int32_t *ws_lists;                /* [2*ws_cf + 2] long cf list, short cf list, counts */
    long long ws_blocks_cf;           /* decode: capacity of ws_blocks          */
    double *ws_blocks;                /* [cf][2048] blocks before overlap-add   */
    /* gain-shape coder (use_vq) */
    /* fork-join: the side chain (VALU/latency bound) runs beside the MDCT (HBM bound) */
    int fork_side;                    /* all-long scalar batches: side chain on side_stream (pacx_set_side_fork) */
    hipStream_t side_stream;
    hipEvent_t ev_fork, ev_join;
    /* mixed batches: the short-coded frames' chain (MDCT, side chain, mask, tail) runs on
       streams of its own beside the long-coded frames' */
    hipStream_t short_stream, short_side_stream;
    hipEvent_t ev_short_side, ev_short_done, ev_lists;
    std::vector<char> vq_view;        /* VqView of k_vq.hip (device pointers)   */
    std::vector<char> vqdec_view;     /* VqDecView of k_vq_dec.hip              */
    int tables_exact;                 /* every float64 table is the NumPy-evaluated one */
    long long ws_mant_cf;             /* capacity of ws_mant                    */
    int32_t *ws_mant;                 /* [cf][1024] mantissas of short frames when the caller wants none */
    long long ws_dec_cf;              /* capacity of the VQ decode buffers      */
    double *ws_dec_lines;             /* [cf][1024]                             */
    uint8_t *ws_dec_sbr;              /* [cf]                                   */
    uint32_t *ws_dec_status;          /* [cf] (scalar SBR decode without a caller's status) */
    double *ws_sbr_mean;              /* [ws_cf][8] omitted-band means           */
    long long ws_vq_cf;               /* capacity of the short-frame buffers    */
    unsigned *ws_unit_words;          /* [ws_vq_cf*8][548]                      */
    int32_t *ws_unit_bits;            /* [ws_vq_cf*8][2]                        */
    std::string err;
};

static thread_local std::string g_create_err;

static int fail(pacx_handle *h, int code, const std::string &msg)
{
    if (h)
        h->err = msg;
    else
        g_create_err = msg;
    return code;
}

#define HIP_TRY(h, call)                                                               \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(h, PACX_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

/* Inside a fork/join region (work already queued on the handle's internal streams, which share the
   handle's workspaces with the caller's stream): an error must not leave those streams unjoined --
   the caller's next call could overwrite ws_lines / ws_smr / ws_peaks / ws_lists under kernels that
   are still running.  On the error path the internal streams are drained before returning. */
static void drain_internal(pacx_handle *h);
#define HIP_TRY_FORKED(h, call)                                                        \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            drain_internal(h);                                                         \
            return fail(h, PACX_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
        }                                                                              \
    } while (0)

template <typename Tp>
static int upload(pacx_handle *h, const Tp *host, size_t n, const Tp **dev)
{
    void *p = nullptr;
    HIP_TRY(h, hipMalloc(&p, n * sizeof(Tp)));
    h->owned.push_back(p);
    HIP_TRY(h, hipMemcpy(p, host, n * sizeof(Tp), hipMemcpyHostToDevice));
    *dev = (const Tp *)p;
    return PACX_OK;
}

static std::vector<double2> unit_circle(int count, long double num_mul, long double num_add, long double den)
{
    /* exp(-j*pi*(num_mul*i + num_add)/den), evaluated in long double */
    std::vector<double2> t(count);
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int i = 0; i < count; ++i) {
        const long double a = pi * (num_mul * i + num_add) / den;
        t[i].x = (double)cosl(a);
        t[i].y = (double)(-sinl(a));
    }
    return t;
}

/* built-in float64 tables (pacx_tables_gen.h): bit patterns of the NumPy evaluation */
static std::vector<double> gen_table(const uint64_t *bits, size_t n)
{
    std::vector<double> t(n);
    memcpy(t.data(), bits, n * sizeof(double));
    return t;
}

static int gen_rate_index(int sample_rate)
{
    for (int i = 0; i < PACX_GEN_N_RATES; ++i)
        if (PACX_GEN_RATES[i] == sample_rate)
            return i;
    return -1;
}

extern "C" int pacx_tables_exact(const pacx_handle *h) { return h ? h->tables_exact : PACX_E_ARG; }

/* coder/psychoac.py:100-103 */
static const double kCbFreqLimits[25] = {100, 200, 300, 400, 510, 630, 770, 920, 1080, 1270, 1480, 1720, 2000,
                                         2320, 2700, 3150, 3700, 4400, 5300, 6400, 7700, 9500, 12000, 15500,
                                         24000};

extern "C" int pacx_default_bands(int sample_rate, int n_mdct_lines, int32_t *band_lines, int32_t *n_bands)
{
    if (sample_rate <= 0 || n_mdct_lines <= 0 || !band_lines || !n_bands)
        return PACX_E_ARG;
    /* AssignMDCTLinesFromFreqLimits (coder/psychoac.py:106-124), same operations in the
       same order: every one is a correctly rounded IEEE operation or exact */
    const double width = (double)sample_rate / (double)(2 * n_mdct_lines);
    double centers[25], counts[25];
    for (int i = 0; i < 25; ++i)
        centers[i] = floor(kCbFreqLimits[i] / width - 0.5);
    for (int i = 0; i < 25; ++i)
        counts[i] = centers[i] - (i ? centers[i - 1] : -1.0);
    for (int i = 0; i < 25; ++i)
        if (kCbFreqLimits[i] > (double)sample_rate / 2.0) {
            double sum = 0.0;
            for (int k = 0; k < i; ++k)
                sum += counts[k];
            counts[i] = (double)n_mdct_lines - sum;
            for (int k = i + 1; k < 25; ++k)
                counts[k] = 0.0;
            break;
        }
    /* ScaleFactorBands (coder/psychoac.py:143-149): a band of <= 12 lines joins its right
       neighbour (the array is cast to int first, :141) */
    std::vector<long long> n(25);
    for (int i = 0; i < 25; ++i)
        n[i] = (long long)counts[i];
    size_t i = 1;
    while (i < n.size()) {
        if (n[i - 1] <= 12) {
            n[i] += n[i - 1];
            n.erase(n.begin() + (long)(i - 1));
        } else {
            ++i;
        }
    }
    for (size_t k = 0; k < n.size(); ++k)
        band_lines[k] = (int32_t)n[k];
    *n_bands = (int32_t)n.size();
    return PACX_OK;
}

extern "C" int pacx_abi_version(void) { return PACX_ABI_VERSION; }

extern "C" const char *pacx_last_error(const pacx_handle *h) { return h ? h->err.c_str() : g_create_err.c_str(); }

extern "C" int pacx_band_stride(const pacx_handle *h) { return h ? h->T.band_stride : PACX_E_ARG; }

extern "C" int pacx_payload_stride(const pacx_handle *h) { return h ? PACX_PAYLOAD_STRIDE : PACX_E_ARG; }

/* The bands need not reach the last MDCT line: above 48 kHz the critical-band table
 * ends at 24 kHz and the reference leaves the lines beyond it uncoded
 * (coder/psychoac.py:106-124).  Those lines map to the dummy band index nb, whose
 * allocation the kernels keep at zero. */
static int build_bands(pacx_handle *h, const int32_t *lines, int nb, int total,
                       const int32_t **d_lower, const int32_t **d_lines, const uint8_t **d_map, int *covered)
{
    std::vector<int32_t> lower(nb), cnt(lines, lines + nb);
    std::vector<uint8_t> map(total, (uint8_t)nb);
    int at = 0;
    for (int b = 0; b < nb; ++b) {
        lower[b] = at;
        if (cnt[b] <= 0)
            return fail(h, PACX_E_ARG, "band with no lines");
        for (int k = 0; k < cnt[b] && at + k < total; ++k)
            map[at + k] = (uint8_t)b;
        at += cnt[b];
    }
    if (at > total)
        return fail(h, PACX_E_ARG, "band line counts exceed the number of MDCT lines");
    if (at < total && nb >= PACX_MAX_BANDS)
        return fail(h, PACX_E_UNSUPPORTED, "at most 31 bands when the bands do not cover every line");
    *covered = at;
    int rc;
    if ((rc = upload(h, lower.data(), nb, d_lower))) return rc;
    if ((rc = upload(h, cnt.data(), nb, d_lines))) return rc;
    return upload(h, map.data(), total, d_map);
}

extern "C" int pacx_create(const pacx_config *cfg, pacx_handle **out)
{
    if (!cfg || !out)
        return fail(nullptr, PACX_E_ARG, "pacx_create: null argument");
    *out = nullptr;
    if (cfg->abi_version != PACX_ABI_VERSION)
        return fail(nullptr, PACX_E_ARG, "pacx_create: abi_version mismatch");
    if (cfg->n_lines_long != PACX_M_LONG || cfg->n_lines_short != PACX_M_SHORT)
        return fail(nullptr, PACX_E_UNSUPPORTED,
                    "pacx_create: kernels are built for nMDCTLines 1024 (long) / 128 (short)");
    if (cfg->n_bands_long < 1 || cfg->n_bands_long > PACX_MAX_BANDS || cfg->n_bands_short < 1 ||
        cfg->n_bands_short > 8 || !cfg->band_lines_long || !cfg->band_lines_short)
        return fail(nullptr, PACX_E_ARG, "pacx_create: band tables missing or too large");
    if (cfg->n_scale_bits < 1 || cfg->n_scale_bits > 4 || cfg->n_mant_size_bits < 1 ||
        cfg->n_mant_size_bits > 16 || cfg->sample_rate <= 0)
        return fail(nullptr, PACX_E_UNSUPPORTED, "pacx_create: nScaleBits must be 1..4, nMantSizeBits 1..16");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(nullptr, PACX_E_HIP, "pacx_create: no HIP device visible (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= n_dev)
        return fail(nullptr, PACX_E_ARG, "pacx_create: device ordinal out of range");

    pacx_handle *h = new pacx_handle();
    h->device = cfg->device;
    h->ws_cf = 0;
    h->ws_blocks_cf = 0;
    h->ws_blocks = nullptr;
    h->ws_sbr_mean = nullptr;
    h->fork_side = 0;
    h->side_stream = nullptr;
    h->ev_fork = nullptr;
    h->ev_join = nullptr;
    h->short_stream = nullptr;
    h->short_side_stream = nullptr;
    h->ev_short_side = nullptr;
    h->ev_short_done = nullptr;
    h->ev_lists = nullptr;
    h->ws_mant_cf = 0;
    h->ws_mant = nullptr;
    h->ws_dec_cf = 0;
    h->ws_dec_lines = nullptr;
    h->ws_dec_sbr = nullptr;
    h->ws_dec_status = nullptr;
    h->ws_vq_cf = 0;
    h->ws_unit_words = nullptr;
    h->ws_unit_bits = nullptr;
    h->ws_lines = nullptr; h->ws_smr = nullptr; h->ws_peaks = nullptr; h->ws_npeaks = nullptr;
    h->ws_overall = nullptr; h->ws_chunks = nullptr; h->ws_offs = nullptr; h->ws_nkept = nullptr;
    h->ws_lists = nullptr;
    h->tables_exact = 1;
    memset(&h->T, 0, sizeof(h->T));
    int rc = PACX_OK;
#define TRY(x) do { rc = (x); if (rc) { g_create_err = h->err; pacx_destroy(h); return rc; } } while (0)
    {
        hipError_t e = hipSetDevice(cfg->device);
        if (e != hipSuccess) {
            g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e);
            delete h;
            return PACX_E_HIP;
        }
    }
    if (hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&h->short_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->short_side_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_short_side, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_short_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_lists, hipEventDisableTiming) != hipSuccess) {
        g_create_err = "pacx_create: could not create the side stream / events";
        pacx_destroy(h);
        return PACX_E_HIP;
    }
    {
        hipDeviceProp_t prop;
        h->n_cu = (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0)
                      ? prop.multiProcessorCount : 256;
    }
    PacxTables &T = h->T;
    const int NL = PACX_N_LONG, NS = PACX_N_SHORT, ML = PACX_M_LONG, MS = PACX_M_SHORT;
    const double sr = cfg->sample_rate;

    /* windows */
    std::vector<double> wl(4 * NL), ws(NS), hl(NL), hs(NS);
    if (cfg->win_long) {
        memcpy(wl.data(), cfg->win_long, sizeof(double) * 4 * NL);
    } else {
        /* coder/window.py:61-92: np.concatenate of sine halves, ones and zeros */
        const std::vector<double> sl = gen_table(PACX_GEN_SINE_LONG, NL), ss = gen_table(PACX_GEN_SINE_SHORT, NS);
        const int pad = NL / 4 - NS / 4;
        for (int i = 0; i < NL; ++i) {
            wl[i] = sl[i];
            double st;                                    /* start window, coder/window.py:67-69 */
            if (i < NL / 2) st = sl[i];
            else if (i < NL / 2 + pad) st = 1.0;
            else if (i < NL / 2 + pad + NS / 2) st = ss[NS / 2 + (i - NL / 2 - pad)];
            else st = 0.0;
            wl[NL + i] = st;
            double ssw;                                   /* start-stop, coder/window.py:88-90 */
            if (i < pad) ssw = 0.0;
            else if (i < pad + NS / 2) ssw = ss[i - pad];
            else if (i < pad + NS / 2 + 2 * pad) ssw = 1.0;
            else if (i < pad + NS + 2 * pad) ssw = ss[NS / 2 + (i - pad - NS / 2 - 2 * pad)];
            else ssw = 0.0;
            wl[3 * NL + i] = ssw;
        }
        for (int i = 0; i < NL; ++i)
            wl[2 * NL + i] = wl[NL + (NL - 1 - i)];       /* stop = flipped start */
    }
    if (cfg->win_short) memcpy(ws.data(), cfg->win_short, sizeof(double) * NS);
    else ws = gen_table(PACX_GEN_SINE_SHORT, NS);
    if (cfg->hann_long) memcpy(hl.data(), cfg->hann_long, sizeof(double) * NL);
    else hl = gen_table(PACX_GEN_HANN_LONG, NL);
    if (cfg->hann_short) memcpy(hs.data(), cfg->hann_short, sizeof(double) * NS);
    else hs = gen_table(PACX_GEN_HANN_SHORT, NS);
    {
        std::vector<double> kl = gen_table(PACX_GEN_KBD_LONG, NL), ks = gen_table(PACX_GEN_KBD_SHORT, NS);
        if (cfg->kbd_long) memcpy(kl.data(), cfg->kbd_long, sizeof(double) * NL);
        if (cfg->kbd_short) memcpy(ks.data(), cfg->kbd_short, sizeof(double) * NS);
        TRY(upload(h, kl.data(), kl.size(), &T.kbd_long));
        TRY(upload(h, ks.data(), ks.size(), &T.kbd_short));
    }
    TRY(upload(h, wl.data(), wl.size(), &T.win_long));
    TRY(upload(h, ws.data(), ws.size(), &T.win_short));
    TRY(upload(h, hl.data(), hl.size(), &T.hann_long));
    TRY(upload(h, hs.data(), hs.size(), &T.hann_short));
    {
        std::vector<double> ones(NL, 1.0);
        TRY(upload(h, ones.data(), ones.size(), &T.ones));
        /* Hann tables with the PCM scale folded in (see k_psy.hip) */
        std::vector<double> hlp(NL), hsp(NS);
        for (int i = 0; i < NL; ++i) hlp[i] = hl[i] * (2.0 / 65535.0);
        for (int i = 0; i < NS; ++i) hsp[i] = hs[i] * (2.0 / 65535.0);
        TRY(upload(h, hlp.data(), hlp.size(), &T.hann_long_pcm));
        TRY(upload(h, hsp.data(), hsp.size(), &T.hann_short_pcm));
    }

    /* twiddles */
    {
        std::vector<double2> t;
        t = unit_circle(512, 8, 1, 8192);  TRY(upload(h, t.data(), t.size(), &T.tw_long));
        t = unit_circle(64, 8, 1, 1024);   TRY(upload(h, t.data(), t.size(), &T.tw_short));
        t = unit_circle(512, 2, 0, 512);   TRY(upload(h, t.data(), t.size(), &T.w512));
        /* the two tables of the long side chain's real-FFT split are made exactly
           symmetric (a handful of entries move by one ulp), so that the kernel derives
           W1024^(512-k) = -conj(W1024^k) and W2048^(k+512) = -j W2048^k from the entry it
           has already loaded instead of fetching them */
        t = unit_circle(512, 2, 0, 1024);
        for (int k = 1; k < 256; ++k)
            t[512 - k] = make_double2(-t[k].x, t[k].y);
        t[256].x = 0.0;
        TRY(upload(h, t.data(), t.size(), &T.w1024));
        t = unit_circle(1025, 2, 0, 2048);
        for (int k = 0; k <= 512; ++k)
            t[k + 512] = make_double2(t[k].y, -t[k].x);
        TRY(upload(h, t.data(), t.size(), &T.w2048));
        t = unit_circle(64, 2, 0, 128);    TRY(upload(h, t.data(), t.size(), &T.w128));
        t = unit_circle(129, 2, 0, 256);   TRY(upload(h, t.data(), t.size(), &T.w256));
    }

    /* psychoacoustic tables at the MDCT line frequencies (coder/psychoac.py:183-184) */
    {
        std::vector<double> bl(ML), tl(ML), bs(MS), ts(MS);
        /* NULL tables: the built-in NumPy-evaluated copies where the sample rate has them,
           else the C math library (pacx_tables_exact() then says 0) */
        const int ri = gen_rate_index(cfg->sample_rate);
        const double *g_bl = ri >= 0 ? (const double *)PACX_GEN_BARK_LONG + (size_t)ri * ML : nullptr;
        const double *g_tl = ri >= 0 ? (const double *)PACX_GEN_THRESH_LONG + (size_t)ri * ML : nullptr;
        const double *g_bs = ri >= 0 ? (const double *)PACX_GEN_BARK_SHORT + (size_t)ri * MS : nullptr;
        const double *g_ts = ri >= 0 ? (const double *)PACX_GEN_THRESH_SHORT + (size_t)ri * MS : nullptr;
        if (ri < 0 && (!cfg->bark_long || !cfg->thresh_long || !cfg->bark_short || !cfg->thresh_short))
            h->tables_exact = 0;
        for (int k = 0; k < ML; ++k) {
            const double f = (sr / (2 * ML)) * (k + 0.5);
            bl[k] = cfg->bark_long ? cfg->bark_long[k] : (g_bl ? g_bl[k] : pacx_bark(f));
            tl[k] = cfg->thresh_long ? cfg->thresh_long[k] : (g_tl ? g_tl[k] : pacx_thresh_quiet(f));
        }
        for (int k = 0; k < MS; ++k) {
            const double f = (sr / (2 * MS)) * (k + 0.5);
            bs[k] = cfg->bark_short ? cfg->bark_short[k] : (g_bs ? g_bs[k] : pacx_bark(f));
            ts[k] = cfg->thresh_short ? cfg->thresh_short[k] : (g_ts ? g_ts[k] : pacx_thresh_quiet(f));
        }
        TRY(upload(h, bl.data(), bl.size(), &T.bark_long));
        TRY(upload(h, tl.data(), tl.size(), &T.thresh_long));
        TRY(upload(h, bs.data(), bs.size(), &T.bark_short));
        TRY(upload(h, ts.data(), ts.size(), &T.thresh_short));
    }
    /* FFT power normalisation 4/(N^2 mean(np.hanning(N)^2)) and rfftfreq step */
    auto hanning_norm = [](int n) {
        long double acc = 0;
        for (int i = 0; i < n; ++i) {
            const long double w = 0.5L - 0.5L * cosl(2.0L * 3.14159265358979323846264338327950288L * i / (n - 1));
            acc += w * w;
        }
        return (double)(4.0L / ((long double)n * n * (acc / n)));
    };
    (void)hanning_norm;       /* kept as the formula; the built-in values are NumPy's */
    T.norm_long = cfg->fft_norm_long != 0.0 ? cfg->fft_norm_long : ((const double *)PACX_GEN_FFT_NORM)[0];
    T.norm_short = cfg->fft_norm_short != 0.0 ? cfg->fft_norm_short : ((const double *)PACX_GEN_FFT_NORM)[1];
    T.fstep_long = cfg->fft_freq_step_long != 0.0 ? cfg->fft_freq_step_long : 1.0 / (NL * (1.0 / sr));
    T.fstep_short = cfg->fft_freq_step_short != 0.0 ? cfg->fft_freq_step_short : 1.0 / (NS * (1.0 / sr));

    {
        /* Bark of the long side-chain FFT's bin frequencies: a masker made of bins i-1 and i has its
           Bark value between entries i-1 and i (the kernel adds a margin) */
        std::vector<double> bb(NL / 2 + 1);
        for (int i = 0; i <= NL / 2; ++i)
            bb[i] = pacx_bark((double)i * T.fstep_long);
        TRY(upload(h, bb.data(), bb.size(), &T.bark_bin_long));
    }
    T.nb_long = cfg->n_bands_long;
    T.nb_short = cfg->n_bands_short;
    int covered_long = ML, covered_short = MS;
    TRY(build_bands(h, cfg->band_lines_long, T.nb_long, ML, &T.band_lower_long, &T.band_lines_long,
                    &T.line_band_long, &covered_long));
    TRY(build_bands(h, cfg->band_lines_short, T.nb_short, MS, &T.band_lower_short, &T.band_lines_short,
                    &T.line_band_short, &covered_short));
    if (covered_short < MS && T.nb_short >= 8) {
        g_create_err = "pacx_create: at most 7 short bands when they do not cover every line";
        pacx_destroy(h);
        return PACX_E_UNSUPPORTED;
    }
    T.band_stride = T.nb_long > PACX_SUB * T.nb_short ? T.nb_long : PACX_SUB * T.nb_short;
    T.n_scale_bits = cfg->n_scale_bits;
    T.n_mant_size_bits = cfg->n_mant_size_bits;
    T.target_bps = cfg->target_bits_per_sample;

    /* coding variant */
    T.guard = cfg->guard ? 1 : 0;
    T.use_vq = cfg->use_vq ? 1 : 0;
    T.use_sbr = cfg->use_sbr ? 1 : 0;
    T.first_omitted = T.nb_long;
    T.band_lines_long_alloc = T.band_lines_long;
    if (T.use_sbr) {
        /* sbr.omitted_bands (coder/sbr.py:6-9): bands starting at or above upperLine[-1] // 2 */
        std::vector<int32_t> alloc_lines(cfg->band_lines_long, cfg->band_lines_long + T.nb_long);
        const int cut = (covered_long - 1) / 2;      /* sfBands.upperLine[-1] // 2 */
        int at = 0;
        for (int b = 0; b < T.nb_long; ++b) {
            if (at >= cut) {
                if (T.first_omitted == T.nb_long)
                    T.first_omitted = b;
                alloc_lines[b] = 1;
            }
            at += cfg->band_lines_long[b];
        }
        if (T.nb_long - T.first_omitted > PACX_SUB) {
            g_create_err = "pacx_create: more than 8 SBR-omitted bands";
            pacx_destroy(h);
            return PACX_E_UNSUPPORTED;
        }
        TRY(upload(h, alloc_lines.data(), alloc_lines.size(), &T.band_lines_long_alloc));
    }
    const uint64_t *d_n = nullptr, *d_p = nullptr;
    const int32_t *d_off = nullptr, *d_k = nullptr;
    const uint8_t *d_w = nullptr;
    const double *d_hl = nullptr, *d_lt = nullptr;
    int l_max = 1;
    if (T.use_vq) {
        for (int b = 0; b < T.nb_long; ++b)
            l_max = cfg->band_lines_long[b] > l_max ? cfg->band_lines_long[b] : l_max;
        for (int b = 0; b < T.nb_short; ++b)
            l_max = cfg->band_lines_short[b] > l_max ? cfg->band_lines_short[b] : l_max;
        PacxVqHostTables vt;
        /* NULL gain-shape tables: the built-in NumPy-evaluated copies (l_max <= 1024 always:
           a band cannot have more lines than the block) */
        pacx_vq_build(l_max, cfg->half_log2 ? cfg->half_log2 : (const double *)PACX_GEN_HALF_LOG2, &vt);
        if (vt.n_tab.empty()) {            /* l_max < 3: rows 0..2 are closed forms */
            vt.n_tab.push_back(0);
            vt.p_tab.push_back(0);
        }
        TRY(upload(h, vt.n_tab.data(), vt.n_tab.size(), &d_n));
        TRY(upload(h, vt.p_tab.data(), vt.p_tab.size(), &d_p));
        TRY(upload(h, vt.row_off.data(), vt.row_off.size(), &d_off));
        TRY(upload(h, vt.k_of.data(), vt.k_of.size(), &d_k));
        TRY(upload(h, vt.w_of.data(), vt.w_of.size(), &d_w));
        TRY(upload(h, vt.half_log2.data(), vt.half_log2.size(), &d_hl));
        h->vq_view.resize(pacx_vq_view_size());
        std::vector<int32_t> sizes_long(cfg->band_lines_long, cfg->band_lines_long + T.nb_long);
        for (int b = T.first_omitted; b < T.nb_long; ++b)
            sizes_long[b] = 1;
        /* log2(tan(theta_q) + eps) of the quantised split angles (bit_allocation_ms) */
        std::vector<double> lt((1u << PACX_VQ_THETA_TABLE_BITS) - 1, 0.0);
        memcpy(lt.data(), cfg->vq_log2_tan ? cfg->vq_log2_tan : (const double *)PACX_GEN_VQ_LOG2_TAN,
               sizeof(double) * lt.size());
        static_assert(sizeof(PACX_GEN_VQ_LOG2_TAN) / 8 == (1u << PACX_VQ_THETA_TABLE_BITS) - 1, "log2-tan table");
        TRY(upload(h, lt.data(), lt.size(), &d_lt));
        pacx_vq_view_fill(h->vq_view.data(), d_n, d_p, d_off, d_k, d_w, d_hl, l_max,
                          cfg->log_mu1 != 0.0 ? cfg->log_mu1 : ((const double *)PACX_GEN_LOG_MU1)[0], d_lt, sizes_long.data(), T.nb_long,
                          cfg->band_lines_short, T.nb_short);
    }
    if (T.use_vq || T.use_sbr) {
        /* decode side: Gaussian weights of gaussian_filter1d(sigma=200) (radius
           int(4*200 + 0.5)) and the MDCT line frequencies of Decode_SBR (scalar-mantissa SBR
           handles: these two alone, for k_sbr_recon) */
        const int gr = cfg->sbr_gauss ? cfg->sbr_gauss_radius : 800;
        if (gr < 1 || gr > 4096) {
            g_create_err = "pacx_create: sbr_gauss_radius out of range";
            pacx_destroy(h);
            return PACX_E_ARG;
        }
        std::vector<double> gw(2 * gr + 1), lf(ML);
        static_assert(sizeof(PACX_GEN_SBR_GAUSS) / 8 == 2 * 800 + 1, "gaussian weights");
        memcpy(gw.data(), cfg->sbr_gauss ? cfg->sbr_gauss : (const double *)PACX_GEN_SBR_GAUSS,
               sizeof(double) * gw.size());
        for (int k = 0; k < ML; ++k)
            lf[k] = cfg->line_freq_long ? cfg->line_freq_long[k] : (k + 0.5) * (sr / (2 * ML));
        const double *d_gw, *d_lf;
        TRY(upload(h, gw.data(), gw.size(), &d_gw));
        TRY(upload(h, lf.data(), lf.size(), &d_lf));
        h->vqdec_view.resize(pacx_vqdec_view_size());
        pacx_vqdec_view_fill(h->vqdec_view.data(), d_n, d_p, d_off, d_k, d_w, d_hl, l_max, d_lt, d_gw, gr, d_lf,
                             cfg->band_lines_long, T.nb_long, cfg->band_lines_short, T.nb_short);
    }
#undef TRY
    *out = h;
    return PACX_OK;
}

static void drain_internal(pacx_handle *h)
{
    for (hipStream_t s2 : {h->side_stream, h->short_stream, h->short_side_stream})
        if (s2)
            (void)hipStreamSynchronize(s2);
}

static int post_launch_forked(pacx_handle *h, const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        drain_internal(h);
        return fail(h, PACX_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    }
    return PACX_OK;
}

static void free_ws(pacx_handle *h)
{
    void *p[] = {h->ws_lines, h->ws_smr, h->ws_peaks, h->ws_npeaks, h->ws_overall, h->ws_chunks, h->ws_offs,
                 h->ws_nkept, h->ws_sbr_mean, h->ws_lists};
    for (void *q : p)
        if (q)
            (void)hipFree(q);
    h->ws_sbr_mean = nullptr;
    h->ws_lists = nullptr;
    h->ws_lines = nullptr; h->ws_smr = nullptr; h->ws_peaks = nullptr; h->ws_npeaks = nullptr;
    h->ws_overall = nullptr; h->ws_chunks = nullptr; h->ws_offs = nullptr; h->ws_nkept = nullptr;
    h->ws_cf = 0;
}

extern "C" void pacx_destroy(pacx_handle *h)
{
    if (!h)
        return;
    (void)hipSetDevice(h->device);
    if (h->side_stream) {
        (void)hipStreamSynchronize(h->side_stream);
        (void)hipStreamDestroy(h->side_stream);
    }
    for (hipStream_t s2 : {h->short_stream, h->short_side_stream})
        if (s2) {
            (void)hipStreamSynchronize(s2);
            (void)hipStreamDestroy(s2);
        }
    for (hipEvent_t e2 : {h->ev_short_side, h->ev_short_done, h->ev_lists})
        if (e2)
            (void)hipEventDestroy(e2);
    if (h->ev_fork)
        (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join)
        (void)hipEventDestroy(h->ev_join);
    free_ws(h);
    if (h->ws_blocks)
        (void)hipFree(h->ws_blocks);
    if (h->ws_unit_words)
        (void)hipFree(h->ws_unit_words);
    if (h->ws_unit_bits)
        (void)hipFree(h->ws_unit_bits);
    if (h->ws_mant)
        (void)hipFree(h->ws_mant);
    if (h->ws_dec_lines)
        (void)hipFree(h->ws_dec_lines);
    if (h->ws_dec_sbr)
        (void)hipFree(h->ws_dec_sbr);
    if (h->ws_dec_status)
        (void)hipFree(h->ws_dec_status);
    for (void *p : h->owned)
        (void)hipFree(p);
    delete h;
}

extern "C" int pacx_reserve(pacx_handle *h, int64_t n_cf)
{
    if (!h || n_cf < 0)
        return fail(h, PACX_E_ARG, "pacx_reserve: bad argument");
    if (n_cf <= h->ws_cf)
        return PACX_OK;
    if (n_cf > 0x7fffffffLL / PACX_SUB)
        return fail(h, PACX_E_ARG, "pacx_reserve: too many channel-frames for one call");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    free_ws(h);
    const size_t n = (size_t)n_cf;
    /* all or nothing: a failing hipMalloc (the workspace of a 262 144-frame batch is 8 GB) leaves the
       handle with NO workspace and ws_cf = 0 -- nothing leaks, and a later, smaller call reserves again */
    struct { void **p; size_t bytes; } want[] = {
        {(void **)&h->ws_lines, n * PACX_M_LONG * sizeof(double)},
        {(void **)&h->ws_smr, n * h->T.band_stride * sizeof(double)},
        {(void **)&h->ws_peaks, n * PACX_MAX_PEAKS * sizeof(PacxPeak)},
        {(void **)&h->ws_npeaks, n * PACX_SUB * sizeof(int32_t)},
        {(void **)&h->ws_nkept, n * PACX_SUB * sizeof(int32_t)},
        {(void **)&h->ws_overall, n * PACX_SUB * sizeof(int32_t)},
        {(void **)&h->ws_chunks, (n / 256 + 2) * sizeof(long long)},
        {(void **)&h->ws_offs, (n + 1) * sizeof(long long)},
        {(void **)&h->ws_lists, (2 * n + 2) * sizeof(int32_t)},
        {(void **)&h->ws_sbr_mean, h->T.use_sbr ? n * PACX_SUB * sizeof(double) : 0},
    };
    for (auto &w : want) {
        if (!w.bytes)
            continue;
        hipError_t e = hipMalloc(w.p, w.bytes);
        if (e != hipSuccess) {
            *w.p = nullptr;
            free_ws(h);
            (void)hipGetLastError();             /* the failed allocation must not poison the next launch check */
            return fail(h, PACX_E_HIP, std::string("pacx_reserve: hipMalloc of ") + std::to_string(w.bytes) +
                                           " bytes: " + hipGetErrorString(e) + " (workspace released)");
        }
    }
    h->ws_cf = n_cf;
    return PACX_OK;
}

/* validate a pcm view; fills the device-side view and the fast-path flag */
static int check_pcm(pacx_handle *h, const pacx_pcm *in, PacxPcmView *v, int *fast, long long *n_cf)
{
    if (!in || (!in->data && in->n_frames != 0))
        return fail(h, PACX_E_ARG, "pcm view or data pointer is null");
    if (in->dtype != PACX_PCM_I16 && in->dtype != PACX_PCM_F64)
        return fail(h, PACX_E_ARG, "pcm dtype must be PACX_PCM_I16 or PACX_PCM_F64");
    if (in->n_channels < 1 || in->n_frames < 0 || in->sample_stride < 1 || in->frame_stride < 0 ||
        in->channel_stride < 0)
        return fail(h, PACX_E_ARG, "pcm view: bad channel count, frame count or stride");
    if (in->n_frames * in->n_channels > 0x7fffffffLL / PACX_SUB)
        return fail(h, PACX_E_ARG, "pcm view: too many channel-frames for one call");
    v->base = in->data;
    v->frame_stride = in->frame_stride;
    v->ch_stride = in->channel_stride;
    v->samp_stride = in->sample_stride;
    v->n_ch = in->n_channels;
    *fast = (in->dtype == PACX_PCM_I16 && in->sample_stride == 1 && ((uintptr_t)in->data % 16) == 0 &&
             in->frame_stride % 8 == 0 && in->channel_stride % 8 == 0);
    *n_cf = in->n_frames * in->n_channels;
    return PACX_OK;
}

static int post_launch(pacx_handle *h, const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(h, PACX_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return PACX_OK;
}

extern "C" int pacx_mdct_batch(pacx_handle *h, const pacx_pcm *in, const uint8_t *frame_flags,
                               int mode, double *lines, int32_t *max_scale, void *stream)
{
    if (!h)
        return PACX_E_ARG;
    PacxPcmView v;
    int fast;
    long long n_cf;
    int rc = check_pcm(h, in, &v, &fast, &n_cf);
    if (rc)
        return rc;
    if (n_cf == 0)
        return PACX_OK;
    if (!lines)
        return fail(h, PACX_E_ARG, "pacx_mdct_batch: lines is null");
    if ((mode & PACX_MDCT_KBD) && (frame_flags || (mode & PACX_MDCT_PREWINDOWED)))
        return fail(h, PACX_E_ARG, "pacx_mdct_batch: PACX_MDCT_KBD takes no frame flags and no PREWINDOWED");
    HIP_TRY(h, hipSetDevice(h->device));
    const int short_blocks = (mode & PACX_MDCT_SHORT) ? 1 : 0;
    if (mode & PACX_MDCT_KBD) {              /* window override 2 = the KBD tables */
        pacx_launch_mdct(h->T, v, in->dtype, fast, nullptr, n_cf, short_blocks, 0, 2, lines, max_scale,
                         short_blocks ? PACX_SUB : 1, nullptr, (hipStream_t)stream);
        return post_launch(h, "pacx_mdct_batch");
    }
    if (fast && !short_blocks && !(mode & PACX_MDCT_PREWINDOWED)) {
        pacx_launch_mdct_v2(h->T, v, frame_flags, n_cf, 0, lines, max_scale, 1, nullptr, h->n_cu, nullptr, nullptr,
                            (hipStream_t)stream);
        return post_launch(h, "pacx_mdct_batch");     /* v2 handles all four long windows */
    }
    pacx_launch_mdct(h->T, v, in->dtype, fast, frame_flags, n_cf, short_blocks, 0,
                     (mode & PACX_MDCT_PREWINDOWED) ? 1 : 0, lines, max_scale,
                     short_blocks ? PACX_SUB : 1, nullptr, (hipStream_t)stream);
    return post_launch(h, "pacx_mdct_batch");
}

extern "C" int pacx_smr_batch(pacx_handle *h, const pacx_pcm *in, const double *lines, int short_blocks,
                              double *smr, double *threshold, int32_t *n_peaks, void *stream)
{
    if (!h)
        return PACX_E_ARG;
    PacxPcmView v;
    int fast;
    long long n_cf;
    int rc = check_pcm(h, in, &v, &fast, &n_cf);
    if (rc)
        return rc;
    if (n_cf == 0)
        return PACX_OK;
    if (!lines || !smr)
        return fail(h, PACX_E_ARG, "pacx_smr_batch: lines or smr is null");
    HIP_TRY(h, hipSetDevice(h->device));
    if ((rc = pacx_reserve(h, n_cf)))
        return rc;
    hipStream_t st = (hipStream_t)stream;
    const int sb = short_blocks ? 1 : 0;
    pacx_launch_side(h->T, v, in->dtype, fast, nullptr, n_cf, sb, 0, h->ws_peaks, h->ws_npeaks, h->ws_nkept,
                     nullptr, nullptr, st);
    pacx_launch_mask(h->T, nullptr, in->n_channels, n_cf, sb, 0, h->ws_peaks, h->ws_nkept, lines, smr,
                     threshold, h->n_cu, nullptr, nullptr, nullptr, nullptr, st);
    if (n_peaks) {
        if (sb)
            HIP_TRY(h, hipMemcpyAsync(n_peaks, h->ws_npeaks, (size_t)n_cf * PACX_SUB * sizeof(int32_t),
                                      hipMemcpyDeviceToDevice, st));
        else
            HIP_TRY(h, hipMemcpy2DAsync(n_peaks, sizeof(int32_t), h->ws_npeaks, PACX_SUB * sizeof(int32_t),
                                        sizeof(int32_t), (size_t)n_cf, hipMemcpyDeviceToDevice, st));
    }
    return post_launch(h, "pacx_smr_batch");
}

size_t pacx_smr_generic_lds(int n);
void pacx_launch_smr_generic(long long n_blocks, int n, int nb, const double *data, const double *lines,
                             const double *hann, const double *tw_cos, const double *tw_sin, double norm, double fstep,
                             const double *bark, const double *quiet, const int32_t *band_lower,
                             const int32_t *band_count, double *smr, double *thr_out, int32_t *n_peaks_out,
                             hipStream_t st);

extern "C" int pacx_smr_generic_batch(pacx_handle *h, int64_t n_blocks, int n_samples, const double *data,
                                      const double *lines, const pacx_smr_tables *t, double *smr, double *threshold,
                                      int32_t *n_peaks, void *stream)
{
    if (!h)
        return PACX_E_ARG;
    if (n_blocks == 0)
        return PACX_OK;
    if (n_blocks < 0 || !data || !lines || !smr || !t || !t->hann || !t->tw_cos || !t->tw_sin || !t->bark ||
        !t->quiet || !t->band_lower || !t->band_lines)
        return fail(h, PACX_E_ARG, "pacx_smr_generic_batch: bad argument");
    if (n_samples < 16 || n_samples > 8192 || (n_samples & 1) || t->n_bands < 1 || t->n_bands > PACX_MAX_BANDS)
        return fail(h, PACX_E_UNSUPPORTED, "pacx_smr_generic_batch: blocks of 16..8192 samples (even), 1..32 bands");
    if (pacx_smr_generic_lds(n_samples) > 150 * 1024)
        return fail(h, PACX_E_UNSUPPORTED, "pacx_smr_generic_batch: block too long for the LDS of a CU");
    HIP_TRY(h, hipSetDevice(h->device));
    pacx_launch_smr_generic(n_blocks, n_samples, t->n_bands, data, lines, t->hann, t->tw_cos, t->tw_sin, t->fft_norm,
                            t->fft_freq_step, t->bark, t->quiet, t->band_lower, t->band_lines, smr, threshold, n_peaks,
                            (hipStream_t)stream);
    return post_launch(h, "pacx_smr_generic_batch");
}

extern "C" int pacx_bitalloc_batch(pacx_handle *h, int64_t n_cf, int n_channels, const uint8_t *frame_flags,
                                   int short_blocks, const double *smr, int32_t *bit_alloc,
                                   uint32_t *status, void *stream)
{
    if (!h)
        return PACX_E_ARG;
    if (n_cf == 0)
        return PACX_OK;
    if (n_cf < 0 || n_channels < 1 || !smr || !bit_alloc)
        return fail(h, PACX_E_ARG, "pacx_bitalloc_batch: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    pacx_launch_bitalloc(h->T, frame_flags, n_channels, n_cf, short_blocks ? 1 : 0, 0, 0, smr, bit_alloc, status,
                         (hipStream_t)stream);
    return post_launch(h, "pacx_bitalloc_batch");
}

extern "C" int pacx_quantize_batch(pacx_handle *h, int64_t n_cf, const double *lines,
                                   const int32_t *overall_scale, const int32_t *bit_alloc, int short_blocks,
                                   int32_t *scale_factor, int32_t *mantissa, void *stream)
{
    if (!h)
        return PACX_E_ARG;
    if (n_cf == 0)
        return PACX_OK;
    if (n_cf < 0 || !lines || !overall_scale || !bit_alloc || !scale_factor || !mantissa)
        return fail(h, PACX_E_ARG, "pacx_quantize_batch: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    pacx_launch_quantize(h->T, nullptr, 1, n_cf, short_blocks ? 1 : 0, 0, lines, overall_scale,
                         short_blocks ? PACX_SUB : 1, bit_alloc, scale_factor, mantissa, (hipStream_t)stream);
    return post_launch(h, "pacx_quantize_batch");
}

static int encode_scalar(pacx_handle *h, const pacx_pcm *in, const uint8_t *frame_flags,
                         int32_t *overall_scale, int32_t *scale_factor, int32_t *bit_alloc,
                         int32_t *mantissa, uint32_t *status, uint8_t *payload, int32_t *n_bytes,
                         void *stream, const char *what)
{
    if (!h)
        return PACX_E_ARG;
    PacxPcmView v;
    int fast;
    long long n_cf;
    int rc = check_pcm(h, in, &v, &fast, &n_cf);
    if (rc)
        return rc;
    if (n_cf == 0)
        return PACX_OK;
    if (!overall_scale || !scale_factor || !bit_alloc || !status || (!payload && !mantissa) ||
        (payload && !n_bytes))
        return fail(h, PACX_E_ARG, std::string(what) + ": null output pointer");
    if (h->T.use_vq)
        return fail(h, PACX_E_UNSUPPORTED, std::string(what) + ": handle was created with use_vq "
                                                                "(call pacx_encode_vq_batch)");
    HIP_TRY(h, hipSetDevice(h->device));
    if ((rc = pacx_reserve(h, n_cf)))
        return rc;
    hipStream_t st = (hipStream_t)stream;
    const PacxTables &T = h->T;
    const int mixed = frame_flags ? 1 : 0;     /* without flags every frame is a long sine block */
    const int n_ch = in->n_channels;
    /* k_tail_short (and the tails of the long frames) pack from registers; only the separate-kernel fallback for
       layouts with more than 8 short bands (k_quantize<128> -> k_pack) reads the mantissas back from memory.  Round 3:
       the workspace copy is no longer written when nobody asked for mantissas (it was 4 KB per channel-frame of HBM
       writes in every block-switched step) */
    if (!mantissa && mixed && T.nb_short > 8) {
        if (n_cf > h->ws_mant_cf) {
            HIP_TRY(h, hipDeviceSynchronize());
            if (h->ws_mant)
                (void)hipFree(h->ws_mant);
            h->ws_mant = nullptr;
            h->ws_mant_cf = 0;
            HIP_TRY(h, hipMalloc((void **)&h->ws_mant, (size_t)n_cf * PACX_M_LONG * sizeof(int32_t)));
            h->ws_mant_cf = n_cf;
        }
        mantissa = h->ws_mant;
    }
    if (!fast) {                               /* fast path: k_mdct_long_v2 initialises both itself */
        HIP_TRY(h, hipMemsetAsync(status, 0, (size_t)n_cf * sizeof(uint32_t), st));
        HIP_TRY(h, hipMemsetAsync(overall_scale, 0, (size_t)n_cf * PACX_SUB * sizeof(int32_t), st));
    }
    /* mixed streams: compacted lists of the long- and of the short-coded frames -- every
       persistent kernel below walks its own list.  The four-stream schedule forks first: the side
       chains and the short-block MDCT go by the flags alone and start while the lists are made
       (+2.3 % on the block-switched bench, A/B on one box) */
    const char *split_env = getenv("PACX_SPLIT_SHORT");       /* 0: short frames on the long frames' streams */
    const bool split = mixed && fast && !h->T.use_sbr && !(split_env && atoi(split_env) == 0);
    if (split)
        HIP_TRY(h, hipEventRecord(h->ev_fork, st));
    if (mixed)
        pacx_launch_frame_lists(frame_flags, in->n_frames, n_ch, h->ws_lists, h->ws_lists + n_cf,
                                h->ws_lists + 2 * n_cf, st);
    if (split)
        HIP_TRY(h, hipEventRecord(h->ev_lists, st));
    /* long frames: masked threshold, SMRs, BitAlloc, scale factors / mantissas and the payload
       in ONE kernel (k_mask<1024, true>: the wave that has a frame's SMRs goes on with it; the
       lines are read from HBM once and the SMRs never leave the chip: 210 MB of HBM traffic per
       8192-frame step instead of 280), or the tail in k_tail_long behind a kernel boundary.
       Measured A/B on the same box: block-switched batches are 2 % faster
       fused; all-long batches 2 % faster UNFUSED (the separate tail kernel pairs two frames per
       wave for BitAlloc and runs at 20 waves per CU instead of 12) -- the default follows the
       clock, PACX_FUSE_TAIL=1 / 0 forces either (read per call: tests flip it). */
    const char *fuse_env = getenv("PACX_FUSE_TAIL");
    const int fuse = fuse_env ? (atoi(fuse_env) != 0) : mixed;
    MaskTail mt;
    mt.overall = overall_scale; mt.bit_alloc = bit_alloc; mt.scale_factor = scale_factor; mt.mantissa = mantissa;
    mt.status = status; mt.payload = payload; mt.n_bytes = n_bytes; mt.payload_stride = PACX_PAYLOAD_STRIDE;
    int32_t *const list_long = h->ws_lists, *const list_short = h->ws_lists + n_cf, *const counts = h->ws_lists + 2 * n_cf;
    if (T.use_sbr) {
        /* scalar mantissas in an SBR file (coder/codec.py:426-482, 529-555; long blocks only, short
           ones take the plain path, coder/pacfile.py:639-643): the side chain folds max|FFT| into
           the overall scale the MDCT wrote, so it runs behind the MDCT on one stream; BitAlloc counts
           the omitted bands as one line and budgets from the full block (T.use_sbr in the tail
           kernels), and a frame whose omitted band gets bits is where the reference raises:
           PACX_ST_REF_RAISES, n_bytes 0.  Not a tuned path -- the reference's driver never selects it. */
        if (fast) {
            pacx_launch_mdct_v2(T, v, frame_flags, n_cf, mixed, h->ws_lines, overall_scale, PACX_SUB, status,
                                h->n_cu, mixed ? list_long : nullptr, mixed ? counts : nullptr, st);
            if (mixed)
                pacx_launch_mdct(T, v, in->dtype, fast, frame_flags, n_cf, 0, 4, 0, h->ws_lines, overall_scale,
                                 PACX_SUB, status, st);
        } else {
            pacx_launch_mdct(T, v, in->dtype, fast, frame_flags, n_cf, 0, mixed, 0, h->ws_lines, overall_scale,
                             PACX_SUB, status, st);
        }
        pacx_launch_side(T, v, in->dtype, fast, frame_flags, n_cf, 0, mixed, h->ws_peaks, h->ws_npeaks, h->ws_nkept,
                         h->ws_sbr_mean, overall_scale, st);
        pacx_launch_mask(T, frame_flags, n_ch, n_cf, 0, mixed, h->ws_peaks, h->ws_nkept, h->ws_lines, h->ws_smr,
                         nullptr, h->n_cu, list_long, list_short, counts, nullptr, st);
        pacx_launch_tail(T, frame_flags, n_ch, n_cf, h->ws_smr, h->ws_lines, overall_scale, bit_alloc, scale_factor,
                         mantissa, status, payload, PACX_PAYLOAD_STRIDE, n_bytes, list_short, counts + 1, 0, st);
        return post_launch(h, what);
    }
    /* All-long batches: the whole step on the caller's stream, or (pacx_set_side_fork) the side chain, which only
       reads the PCM, forked to the handle's second stream next to the transform.  The fork pays only where HIP puts
       the two streams on ONE hardware queue -- with two handles in a process it does (50.6 against 49.3 M cf/s with
       two steps in flight), with one handle it does not, and a fork and a join across hardware queues (13 + 12 us)
       cost more than the 20 us of overlap: 39.1 against 42.6 M cf/s with one step in flight.
       PACX_ONE_STREAM=0/1 forces either for every handle (read per call: a test flips it) */
    const char *one_env = getenv("PACX_ONE_STREAM");
    const bool one_stream = !split && (one_env ? atoi(one_env) != 0 : !h->fork_side);
    hipStream_t side_st = one_stream ? st : h->side_stream;
    if (!split && !one_stream)
        HIP_TRY_FORKED(h, hipEventRecord(h->ev_fork, st));
    if (!one_stream)
        HIP_TRY_FORKED(h, hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
    if (split) {
        /* A block-switched batch is two independent chains that touch disjoint frames:
             long-coded :  k_mdct_long_v2 || k_side_long  ->  k_mask<1024> (+ tail)
             short-coded:  k_mdct_short  || k_side_short ->  k_mask<128> -> k_tail_short
           Every one of these kernels is latency-bound at the occupancy its registers and LDS
           allow and none fills the chip with half of the frames, so the two chains run side by
           side on four streams and meet again before the body gather. */
        /* The side chains run on their chains' own streams: two streams per handle, not four.  HIP maps a process's
           streams onto four hardware queues; with two handles (two steps in flight) the short side chain of one
           landed on the other's short-chain queue, behind its mask and tail kernels (kernel trace),
           and alone the forks and joins of four streams cost more than the overlap of a side chain with its 40 us
           transform: bs128 33.9 -> 35.4 M cf/s with two steps in flight, 25.9 -> 30.7 M with one.
           PACX_BS_TWO_STREAMS=0: the four-stream schedule of round 2 */
        const char *two_s = getenv("PACX_BS_TWO_STREAMS");             /* read per call: a test flips it */
        const int two_env = (two_s && atoi(two_s) == 0) ? 0 : 1;
        hipStream_t sl_st = two_env ? st : h->side_stream, ss_st = two_env ? h->short_stream : h->short_side_stream;
        HIP_TRY_FORKED(h, hipStreamWaitEvent(h->short_stream, h->ev_fork, 0));
        if (!two_env)
            HIP_TRY_FORKED(h, hipStreamWaitEvent(h->short_side_stream, h->ev_fork, 0));
        pacx_launch_side(T, v, in->dtype, fast, frame_flags, n_cf, 0, mixed | PACX_PART_LONG, h->ws_peaks, h->ws_npeaks,
                         h->ws_nkept, nullptr, nullptr, sl_st);
        if (!two_env)
            HIP_TRY_FORKED(h, hipEventRecord(h->ev_join, h->side_stream));
        pacx_launch_side(T, v, in->dtype, fast, frame_flags, n_cf, 0, mixed | PACX_PART_SHORT, h->ws_peaks, h->ws_npeaks,
                         h->ws_nkept, nullptr, nullptr, ss_st);
        if (!two_env)
            HIP_TRY_FORKED(h, hipEventRecord(h->ev_short_side, h->short_side_stream));
        /* short chain */
        pacx_launch_mdct(T, v, in->dtype, fast, frame_flags, n_cf, 0, 4, 0, h->ws_lines, overall_scale, PACX_SUB, status,
                         h->short_stream);
        if (!two_env)
            HIP_TRY_FORKED(h, hipStreamWaitEvent(h->short_stream, h->ev_short_side, 0));
        HIP_TRY_FORKED(h, hipStreamWaitEvent(h->short_stream, h->ev_lists, 0));
        pacx_launch_mask(T, frame_flags, n_ch, n_cf, 0, mixed | PACX_PART_SHORT, h->ws_peaks, h->ws_nkept, h->ws_lines,
                         h->ws_smr, nullptr, h->n_cu, list_long, list_short, counts, nullptr, h->short_stream);
        pacx_launch_tail(T, frame_flags, n_ch, n_cf, h->ws_smr, h->ws_lines, overall_scale, bit_alloc, scale_factor,
                         mantissa, status, payload, PACX_PAYLOAD_STRIDE, n_bytes, list_short, counts + 1, 1,
                         h->short_stream);
        HIP_TRY_FORKED(h, hipEventRecord(h->ev_short_done, h->short_stream));
        /* long chain */
        pacx_launch_mdct_v2(T, v, frame_flags, n_cf, mixed, h->ws_lines, overall_scale, PACX_SUB, status, h->n_cu,
                            list_long, counts, st);
        if (!two_env)
            HIP_TRY_FORKED(h, hipStreamWaitEvent(st, h->ev_join, 0));
        pacx_launch_mask(T, frame_flags, n_ch, n_cf, 0, mixed | PACX_PART_LONG, h->ws_peaks, h->ws_nkept, h->ws_lines,
                         h->ws_smr, nullptr, h->n_cu, list_long, list_short, counts, fuse ? &mt : nullptr, st);
        if (!fuse)
            pacx_launch_tail(T, frame_flags, n_ch, n_cf, h->ws_smr, h->ws_lines, overall_scale, bit_alloc, scale_factor,
                             mantissa, status, payload, PACX_PAYLOAD_STRIDE, n_bytes, nullptr, nullptr, 2, st);
        HIP_TRY_FORKED(h, hipStreamWaitEvent(st, h->ev_short_done, 0));       /* both chains done */
        return post_launch_forked(h, what);
    }
    pacx_launch_side(T, v, in->dtype, fast, frame_flags, n_cf, 0, mixed, h->ws_peaks, h->ws_npeaks, h->ws_nkept,
                     nullptr, nullptr, side_st);
    if (!one_stream)
        HIP_TRY_FORKED(h, hipEventRecord(h->ev_join, h->side_stream));
    if (fast) {
        /* long frames: persistent roofline kernel; short (CUR) frames: k_mdct_short */
        pacx_launch_mdct_v2(T, v, frame_flags, n_cf, mixed, h->ws_lines, overall_scale, PACX_SUB, status,
                            h->n_cu, mixed ? list_long : nullptr, mixed ? counts : nullptr, st);
        if (mixed)
            pacx_launch_mdct(T, v, in->dtype, fast, frame_flags, n_cf, 0, 4, 0, h->ws_lines, overall_scale,
                             PACX_SUB, status, st);
    } else {
        pacx_launch_mdct(T, v, in->dtype, fast, frame_flags, n_cf, 0, mixed, 0, h->ws_lines, overall_scale,
                         PACX_SUB, status, st);
    }
    if (!one_stream)
        HIP_TRY_FORKED(h, hipStreamWaitEvent(st, h->ev_join, 0));       /* join */
    pacx_launch_mask(T, frame_flags, n_ch, n_cf, 0, mixed, h->ws_peaks, h->ws_nkept, h->ws_lines, h->ws_smr,
                     nullptr, h->n_cu, list_long, list_short, counts, fuse ? &mt : nullptr, st);
    /* what is left: the long frames when not fused, the short-coded frames of a mixed batch */
    if (!fuse || mixed)
        pacx_launch_tail(T, frame_flags, n_ch, n_cf, h->ws_smr, h->ws_lines, overall_scale, bit_alloc, scale_factor,
                         mantissa, status, payload, PACX_PAYLOAD_STRIDE, n_bytes, list_short, counts + 1, fuse, st);
    return post_launch_forked(h, what);
}

extern "C" int pacx_set_side_fork(pacx_handle *h, int enable)
{
    if (!h)
        return PACX_E_ARG;
    h->fork_side = enable ? 1 : 0;
    return PACX_OK;
}

extern "C" int pacx_encode_batch(pacx_handle *h, const pacx_pcm *in, const uint8_t *frame_flags,
                                 int32_t *overall_scale, int32_t *scale_factor, int32_t *bit_alloc,
                                 int32_t *mantissa, uint32_t *status, void *stream)
{
    return encode_scalar(h, in, frame_flags, overall_scale, scale_factor, bit_alloc, mantissa, status, nullptr,
                         nullptr, stream, "pacx_encode_batch");
}

extern "C" int pacx_encode_pack_batch(pacx_handle *h, const pacx_pcm *in, const uint8_t *frame_flags,
                                      int32_t *overall_scale, int32_t *scale_factor, int32_t *bit_alloc,
                                      int32_t *mantissa, uint32_t *status, uint8_t *payload, int32_t *n_bytes,
                                      void *stream)
{
    if (h && in && in->n_frames > 0 && (!payload || !n_bytes))
        return fail(h, PACX_E_ARG, "pacx_encode_pack_batch: payload and n_bytes are required");
    return encode_scalar(h, in, frame_flags, overall_scale, scale_factor, bit_alloc, mantissa, status, payload,
                         n_bytes, stream, "pacx_encode_pack_batch");
}

extern "C" int pacx_encode_vq_batch(pacx_handle *h, const pacx_pcm *in, const uint8_t *frame_flags,
                                    int32_t *overall_scale, int32_t *bit_alloc, uint8_t *payload,
                                    int32_t *n_bytes, uint32_t *status, pacx_vq_entry *entries,
                                    int32_t *entry_count, int32_t entries_per_band, void *stream)
{
    if (!h)
        return PACX_E_ARG;
    PacxPcmView v;
    int fast;
    long long n_cf;
    int rc = check_pcm(h, in, &v, &fast, &n_cf);
    if (rc)
        return rc;
    if (n_cf == 0)
        return PACX_OK;
    if (!overall_scale || !bit_alloc || !payload || !n_bytes || !status)
        return fail(h, PACX_E_ARG, "pacx_encode_vq_batch: null output pointer");
    if ((entries && (!entry_count || entries_per_band < 1)) || (!entries && entry_count && entries_per_band != 0))
        return fail(h, PACX_E_ARG, "pacx_encode_vq_batch: entries need entry_count and entries_per_band >= 1");
    if (!h->T.use_vq)
        return fail(h, PACX_E_UNSUPPORTED, "pacx_encode_vq_batch: handle was created without use_vq");
    HIP_TRY(h, hipSetDevice(h->device));
    if ((rc = pacx_reserve(h, n_cf)))
        return rc;
    hipStream_t st = (hipStream_t)stream;
    const PacxTables &T = h->T;
    const int mixed = frame_flags ? 1 : 0;
    const int n_ch = in->n_channels;
    if (mixed && n_cf > h->ws_vq_cf) {
        HIP_TRY(h, hipDeviceSynchronize());
        if (h->ws_unit_words) (void)hipFree(h->ws_unit_words);
        if (h->ws_unit_bits) (void)hipFree(h->ws_unit_bits);
        h->ws_unit_words = nullptr;
        h->ws_unit_bits = nullptr;
        h->ws_vq_cf = 0;
        HIP_TRY(h, hipMalloc((void **)&h->ws_unit_words,
                             (size_t)n_cf * PACX_SUB * PACX_VQ_UNIT_WORDS * sizeof(unsigned)));
        HIP_TRY(h, hipMalloc((void **)&h->ws_unit_bits, (size_t)n_cf * PACX_SUB * 2 * sizeof(int32_t)));
        h->ws_vq_cf = n_cf;
    }
    if (!fast) {
        HIP_TRY(h, hipMemsetAsync(status, 0, (size_t)n_cf * sizeof(uint32_t), st));
        HIP_TRY(h, hipMemsetAsync(overall_scale, 0, (size_t)n_cf * PACX_SUB * sizeof(int32_t), st));
    }
    /* every long-coded frame gets its n_bytes from the coder; only dropped short hops keep the zero */
    if (mixed)
        HIP_TRY(h, hipMemsetAsync(n_bytes, 0, (size_t)n_cf * sizeof(int32_t), st));
    MaskTail mt;
    memset(&mt, 0, sizeof(mt));
    mt.bit_alloc = bit_alloc;
    mt.status = status;
    int vq_stage = 0;
    const char *split_env = getenv("PACX_SPLIT_SHORT");       /* 0: short frames on the long frames' stream */
    if (mixed && fast && !(split_env && atoi(split_env) == 0)) {
        /* a block-switched batch: the long-coded and the short-coded frames are two independent chains up
           to the gain-shape coder (which takes all frames), as in the scalar entry point:
             long :  k_mdct_long_v2 -> k_side_long (folds max|FFT| into the overall scale of SBR blocks) ->
                     k_mask<1024> with BitAlloc
             short:  k_mdct_short -> k_side_short -> k_mask<128> -> k_bitalloc
           side by side on two streams; the side chains and the short MDCT go by the flags alone and start
           while the frame lists are made */
        int32_t *const list_long = h->ws_lists, *const list_short = h->ws_lists + n_cf, *const counts = h->ws_lists + 2 * n_cf;
        const char *vfs_env = getenv("PACX_VQ_FUSE_ALLOC");     /* 0: k_bitalloc behind the mask kernel here too */
        const int vq_fuse_split = vfs_env ? (atoi(vfs_env) != 0) : 1;
        const char *ol_env = getenv("PACX_VQ_ONE_LAUNCH");
        const bool one_launch = ol_env && atoi(ol_env) != 0;
        HIP_TRY(h, hipEventRecord(h->ev_fork, st));
        HIP_TRY_FORKED(h, hipStreamWaitEvent(h->short_stream, h->ev_fork, 0));
        pacx_launch_frame_lists(frame_flags, in->n_frames, n_ch, list_long, list_short, counts, st);
        HIP_TRY_FORKED(h, hipEventRecord(h->ev_lists, st));
        /* short chain */
        pacx_launch_mdct(T, v, in->dtype, fast, frame_flags, n_cf, 0, 4, 0, h->ws_lines, overall_scale, PACX_SUB, status,
                         h->short_stream);
        pacx_launch_side(T, v, in->dtype, fast, frame_flags, n_cf, 0, mixed | PACX_PART_SHORT, h->ws_peaks, h->ws_npeaks,
                         h->ws_nkept, nullptr, nullptr, h->short_stream);
        HIP_TRY_FORKED(h, hipStreamWaitEvent(h->short_stream, h->ev_lists, 0));
        pacx_launch_mask(T, frame_flags, n_ch, n_cf, 0, mixed | PACX_PART_SHORT, h->ws_peaks, h->ws_nkept, h->ws_lines,
                         h->ws_smr, nullptr, h->n_cu, list_long, list_short, counts, nullptr, h->short_stream);
        pacx_launch_bitalloc(T, frame_flags, n_ch, n_cf, 0, mixed, 1, h->ws_smr, bit_alloc, status, h->short_stream);
        /* Each chain goes on into the gain-shape coder with its own frames: two k_vq_frame launches side by side on
           the two streams (0.685 ms per shipped128 step with direct launches).  PACX_VQ_ONE_LAUNCH=1: ONE launch
           over all frames behind the join of the two chains (0.74-0.77 ms) -- which is what to use when the step is
           replayed from a hipGraph, where the two launches do not overlap (0.822 ms); bench.py therefore does not
           capture gain-shape steps */
        if (!one_launch)
            pacx_launch_vq(T, h->vq_view.data(), frame_flags, n_ch, n_cf, h->ws_lines, overall_scale, bit_alloc,
                           h->ws_sbr_mean, status, payload, PACX_PAYLOAD_STRIDE, n_bytes, h->ws_unit_words,
                           h->ws_unit_bits, entries, entry_count, entries ? entries_per_band : 0, 1, list_short, counts + 1,
                           h->short_stream);
        HIP_TRY_FORKED(h, hipEventRecord(h->ev_short_done, h->short_stream));
        /* long chain */
        pacx_launch_mdct_v2(T, v, frame_flags, n_cf, mixed, h->ws_lines, overall_scale, PACX_SUB, status, h->n_cu,
                            list_long, counts, st);
        pacx_launch_side(T, v, in->dtype, fast, frame_flags, n_cf, 0, mixed | PACX_PART_LONG, h->ws_peaks, h->ws_npeaks,
                         h->ws_nkept, T.use_sbr ? h->ws_sbr_mean : nullptr, T.use_sbr ? overall_scale : nullptr, st);
        pacx_launch_mask(T, frame_flags, n_ch, n_cf, 0, mixed | PACX_PART_LONG, h->ws_peaks, h->ws_nkept, h->ws_lines,
                         h->ws_smr, nullptr, h->n_cu, list_long, list_short, counts, vq_fuse_split ? &mt : nullptr, st);
        if (!vq_fuse_split)       /* BitAlloc of the long frames in k_bitalloc behind the mask kernel (part 2 = long only) */
            pacx_launch_bitalloc(T, frame_flags, n_ch, n_cf, 0, mixed, 2, h->ws_smr, bit_alloc, status, st);
        if (!one_launch)
            pacx_launch_vq(T, h->vq_view.data(), frame_flags, n_ch, n_cf, h->ws_lines, overall_scale, bit_alloc,
                           h->ws_sbr_mean, status, payload, PACX_PAYLOAD_STRIDE, n_bytes, h->ws_unit_words,
                           h->ws_unit_bits, entries, entry_count, entries ? entries_per_band : 0, 1, list_long, counts, st);
        HIP_TRY_FORKED(h, hipStreamWaitEvent(st, h->ev_short_done, 0));       /* both chains done */
        vq_stage = one_launch ? 0 : 2;
    } else {
        if (mixed)
            pacx_launch_frame_lists(frame_flags, in->n_frames, n_ch, h->ws_lists, h->ws_lists + n_cf,
                                    h->ws_lists + 2 * n_cf, st);
        if (fast) {
            pacx_launch_mdct_v2(T, v, frame_flags, n_cf, mixed, h->ws_lines, overall_scale, PACX_SUB, status,
                                h->n_cu, mixed ? h->ws_lists : nullptr, mixed ? h->ws_lists + 2 * n_cf : nullptr, st);
            if (mixed)
                pacx_launch_mdct(T, v, in->dtype, fast, frame_flags, n_cf, 0, 4, 0, h->ws_lines, overall_scale,
                                 PACX_SUB, status, st);
        } else {
            pacx_launch_mdct(T, v, in->dtype, fast, frame_flags, n_cf, 0, mixed, 0, h->ws_lines, overall_scale,
                             PACX_SUB, status, st);
        }
        /* the side chain follows the MDCT on the same stream: with SBR it folds max|FFT| into the overall scale
           the MDCT wrote, and without SBR a fork to a second stream costs more than it hides here (0.677 against
           0.661 ms per step, A/B on one box) */
        pacx_launch_side(T, v, in->dtype, fast, frame_flags, n_cf, 0, mixed, h->ws_peaks, h->ws_npeaks, h->ws_nkept,
                         T.use_sbr ? h->ws_sbr_mean : nullptr, T.use_sbr ? overall_scale : nullptr, st);
        /* BitAlloc of the long frames inside the mask kernel or in k_bitalloc behind it: as with the scalar
           coder's tail, all-long batches are a little faster unfused (0.637 against 0.642 ms per step, A/B on one
           box); PACX_VQ_FUSE_ALLOC=0/1 forces either */
        const char *vf_env = getenv("PACX_VQ_FUSE_ALLOC");
        const int vq_fuse = vf_env ? (atoi(vf_env) != 0) : mixed;
        pacx_launch_mask(T, frame_flags, n_ch, n_cf, 0, mixed, h->ws_peaks, h->ws_nkept, h->ws_lines, h->ws_smr,
                         nullptr, h->n_cu, h->ws_lists, h->ws_lists + n_cf, h->ws_lists + 2 * n_cf, vq_fuse ? &mt : nullptr, st);
        pacx_launch_bitalloc(T, frame_flags, n_ch, n_cf, 0, mixed, vq_fuse, h->ws_smr, bit_alloc, status, st);
    }
    pacx_launch_vq(T, h->vq_view.data(), frame_flags, n_ch, n_cf, h->ws_lines, overall_scale, bit_alloc,
                   h->ws_sbr_mean, status, payload, PACX_PAYLOAD_STRIDE, n_bytes, h->ws_unit_words,
                   h->ws_unit_bits, entries, entry_count, entries ? entries_per_band : 0, vq_stage, nullptr, nullptr, st);
    return post_launch(h, "pacx_encode_vq_batch");
}

extern "C" int pacx_pack_batch(pacx_handle *h, int64_t n_cf, int n_channels, const uint8_t *frame_flags,
                               const int32_t *overall_scale, const int32_t *scale_factor,
                               const int32_t *bit_alloc, const int32_t *mantissa, const uint32_t *status,
                               uint8_t *payload, int32_t *n_bytes, void *stream)
{
    if (!h)
        return PACX_E_ARG;
    if (n_cf == 0)
        return PACX_OK;
    if (n_cf < 0 || n_channels < 1 || !overall_scale || !scale_factor || !bit_alloc || !mantissa || !payload ||
        !n_bytes)
        return fail(h, PACX_E_ARG, "pacx_pack_batch: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    pacx_launch_pack(h->T, frame_flags, n_channels, n_cf, overall_scale, scale_factor, bit_alloc, mantissa,
                     status, payload, PACX_PAYLOAD_STRIDE, n_bytes, (hipStream_t)stream);
    return post_launch(h, "pacx_pack_batch");
}

extern "C" int pacx_gather_body(pacx_handle *h, int64_t n_cf, const uint8_t *payload, const int32_t *n_bytes,
                                uint8_t *body, int64_t body_capacity, int64_t *total_bytes, void *stream)
{
    if (!h)
        return PACX_E_ARG;
    if (n_cf == 0) {
        if (total_bytes)
            HIP_TRY(h, hipMemsetAsync(total_bytes, 0, sizeof(int64_t), (hipStream_t)stream));
        return PACX_OK;
    }
    if (n_cf < 0 || !payload || !n_bytes || !body || body_capacity < 0)
        return fail(h, PACX_E_ARG, "pacx_gather_body: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = pacx_reserve(h, n_cf);
    if (rc)
        return rc;
    pacx_launch_gather(n_cf, payload, PACX_PAYLOAD_STRIDE, n_bytes, h->ws_chunks, h->ws_offs, body, body_capacity,
                       (long long *)total_bytes, (hipStream_t)stream);
    return post_launch(h, "pacx_gather_body");
}

/* ---- function-level entry points (k_misc.hip) --------------------------- */
extern "C" int pacx_window_batch(pacx_handle *h, int window, int64_t n_rows, const double *x, double *y,
                                 void *stream)
{
    if (!h)
        return PACX_E_ARG;
    if (n_rows < 0 || !x || !y)
        return fail(h, PACX_E_ARG, "pacx_window_batch: bad argument");
    const double *w;
    int len;
    switch (window) {
    case PACX_WIN_SINE: case PACX_WIN_START: case PACX_WIN_STOP: case PACX_WIN_STARTSTOP:
        w = h->T.win_long + window * PACX_N_LONG; len = PACX_N_LONG; break;
    case PACX_WIN_SINE_SHORT: w = h->T.win_short; len = PACX_N_SHORT; break;
    case PACX_WIN_HANN: w = h->T.hann_long; len = PACX_N_LONG; break;
    case PACX_WIN_HANN_SHORT: w = h->T.hann_short; len = PACX_N_SHORT; break;
    case PACX_WIN_KBD: w = h->T.kbd_long; len = PACX_N_LONG; break;
    case PACX_WIN_KBD_SHORT: w = h->T.kbd_short; len = PACX_N_SHORT; break;
    default: return fail(h, PACX_E_ARG, "pacx_window_batch: unknown window");
    }
    HIP_TRY(h, hipSetDevice(h->device));
    pacx_launch_window(w, n_rows, len, x, y, (hipStream_t)stream);
    return post_launch(h, "pacx_window_batch");
}

extern "C" int pacx_window_table_batch(pacx_handle *h, const double *table, int len, int64_t n_rows,
                                       const double *x, double *y, void *stream)
{
    if (!h)
        return PACX_E_ARG;
    if (n_rows < 0 || len < 1 || !table || !x || !y)
        return fail(h, PACX_E_ARG, "pacx_window_table_batch: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    pacx_launch_window(table, n_rows, len, x, y, (hipStream_t)stream);
    return post_launch(h, "pacx_window_table_batch");
}

static int quant_elem(pacx_handle *h, int op, int64_t n, const double *x, int scale, int a, int b,
                      int64_t *out, void *stream, const char *what)
{
    if (!h)
        return PACX_E_ARG;
    if (n < 0 || !x || !out)
        return fail(h, PACX_E_ARG, std::string(what) + ": bad argument");
    const int r_bits = (op == 0) ? a : ((1 << a) - 1 + b);
    if (a < 1 || b < 0 || r_bits < 1 || r_bits > 62 || (op >= 2 && (b < 1 || scale < 0 || scale > (1 << a) - 1)))
        return fail(h, PACX_E_UNSUPPORTED, std::string(what) + ": bit widths out of range");
    HIP_TRY(h, hipSetDevice(h->device));
    pacx_launch_quant_elem(op, n, x, scale, a, b, out, (hipStream_t)stream);
    return post_launch(h, what);
}

extern "C" int pacx_quantize_uniform(pacx_handle *h, int64_t n, const double *x, int n_bits, int64_t *codes,
                                     void *stream)
{
    return quant_elem(h, 0, n, x, 0, n_bits, 0, codes, stream, "pacx_quantize_uniform");
}

extern "C" int pacx_scale_factor(pacx_handle *h, int64_t n, const double *x, int n_scale_bits,
                                 int n_mant_bits, int64_t *scale, void *stream)
{
    return quant_elem(h, 1, n, x, 0, n_scale_bits, n_mant_bits, scale, stream, "pacx_scale_factor");
}

extern "C" int pacx_mantissa(pacx_handle *h, int64_t n, const double *x, int scale, int n_scale_bits,
                             int n_mant_bits, int64_t *mantissa, void *stream)
{
    return quant_elem(h, 2, n, x, scale, n_scale_bits, n_mant_bits, mantissa, stream, "pacx_mantissa");
}

static int dequant_elem(pacx_handle *h, int op, int64_t n, const int64_t *codes, int scale, int a, int b,
                        double *out, void *stream, const char *what)
{
    if (!h)
        return PACX_E_ARG;
    if (n < 0 || !codes || !out)
        return fail(h, PACX_E_ARG, std::string(what) + ": bad argument");
    const int r_bits = (op == 0) ? a : ((1 << a) - 1 + b);
    /* 2 * code and 2^R - 1 must be exact doubles for the one division to be the reference's value */
    if (a < 1 || b < 0 || r_bits < 1 || r_bits > 53 || (op >= 1 && (b < 1 || scale < 0 || scale > (1 << a) - 1)))
        return fail(h, PACX_E_UNSUPPORTED, std::string(what) + ": bit widths out of range");
    HIP_TRY(h, hipSetDevice(h->device));
    pacx_launch_dequant_elem(op, n, codes, scale, a, b, out, (hipStream_t)stream);
    return post_launch(h, what);
}

extern "C" int pacx_dequantize_uniform(pacx_handle *h, int64_t n, const int64_t *codes, int n_bits, double *x,
                                       void *stream)
{
    return dequant_elem(h, 0, n, codes, 0, n_bits, 0, x, stream, "pacx_dequantize_uniform");
}

extern "C" int pacx_dequantize(pacx_handle *h, int64_t n, const int64_t *mantissa, int scale, int n_scale_bits,
                               int n_mant_bits, double *x, void *stream)
{
    return dequant_elem(h, 1, n, mantissa, scale, n_scale_bits, n_mant_bits, x, stream, "pacx_dequantize");
}

extern "C" int pacx_mantissa_fp(pacx_handle *h, int64_t n, const double *x, int scale, int n_scale_bits,
                                int n_mant_bits, int64_t *mantissa, void *stream)
{
    return quant_elem(h, 3, n, x, scale, n_scale_bits, n_mant_bits, mantissa, stream, "pacx_mantissa_fp");
}

extern "C" int pacx_dequantize_fp(pacx_handle *h, int64_t n, const int64_t *mantissa, int scale, int n_scale_bits,
                                  int n_mant_bits, double *x, void *stream)
{
    return dequant_elem(h, 2, n, mantissa, scale, n_scale_bits, n_mant_bits, x, stream, "pacx_dequantize_fp");
}

extern "C" int pacx_imdct_batch(pacx_handle *h, int64_t n_rows, int mode, const double *lines, double *blocks,
                                void *stream)
{
    if (!h)
        return PACX_E_ARG;
    if (n_rows == 0)
        return PACX_OK;
    if (n_rows < 0 || n_rows > 0x7fffffffLL || !lines || !blocks || (mode & ~PACX_MDCT_SHORT))
        return fail(h, PACX_E_ARG, "pacx_imdct_batch: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    pacx_launch_imdct_plain(h->T, n_rows, (mode & PACX_MDCT_SHORT) ? 1 : 0, lines, blocks, (hipStream_t)stream);
    return post_launch(h, "pacx_imdct_batch");
}

extern "C" int pacx_mdct_direct_batch(pacx_handle *h, int64_t n_rows, int a, int b, int inverse, const double *x,
                                      double *y, void *stream)
{
    if (!h)
        return PACX_E_ARG;
    if (n_rows == 0)
        return PACX_OK;
    if (n_rows < 0 || !x || !y || a < 1 || b < 1 || ((a + b) & 1))
        return fail(h, PACX_E_ARG, "pacx_mdct_direct_batch: bad argument (a, b >= 1, a + b even)");
    if (a + b > 16384 || n_rows * (long long)(a + b) > 0x7fffffffLL * 128)
        return fail(h, PACX_E_UNSUPPORTED, "pacx_mdct_direct_batch: block longer than 16384 samples");
    HIP_TRY(h, hipSetDevice(h->device));
    pacx_launch_mdct_direct(n_rows, a, b, inverse ? 1 : 0, x, y, (hipStream_t)stream);
    return post_launch(h, "pacx_mdct_direct_batch");
}

extern "C" int pacx_bitalloc_generic(pacx_handle *h, int64_t n, int n_bands, const int32_t *band_lines,
                                     const double *budget, int max_mant_bits, const double *smr,
                                     int32_t *bit_alloc, void *stream)
{
    if (!h)
        return PACX_E_ARG;
    if (n < 0 || n_bands < 1 || n_bands > PACX_MAX_BANDS || !band_lines || !budget || !smr || !bit_alloc)
        return fail(h, PACX_E_ARG, "pacx_bitalloc_generic: bad argument (at most 32 bands)");
    HIP_TRY(h, hipSetDevice(h->device));
    pacx_launch_bitalloc_generic(n, n_bands, band_lines, budget, max_mant_bits, smr, bit_alloc,
                                 (hipStream_t)stream);
    return post_launch(h, "pacx_bitalloc_generic");
}

extern "C" int pacx_transient_flags(pacx_handle *h, const pacx_pcm *hops, uint8_t *transient,
                                    uint8_t *frame_flags, void *stream)
{
    if (!h)
        return PACX_E_ARG;
    PacxPcmView v;
    int fast;
    long long n_cf;
    int rc = check_pcm(h, hops, &v, &fast, &n_cf);
    if (rc)
        return rc;
    if (hops->dtype != PACX_PCM_I16 || !transient)
        return fail(h, PACX_E_ARG, "pacx_transient_flags: int16 hops and a transient buffer are required");
    if (hops->n_channels > 8)
        return fail(h, PACX_E_UNSUPPORTED, "pacx_transient_flags: at most 8 channels");
    HIP_TRY(h, hipSetDevice(h->device));
    pacx_launch_transient(v, hops->n_frames, PACX_M_LONG, transient, frame_flags, (hipStream_t)stream);
    return post_launch(h, "pacx_transient_flags");
}

/* ---- decode side (k_decode.hip) ------------------------------------------ */
extern "C" int pacx_transient_detect_f64(pacx_handle *h, int64_t n_blocks, int n_channels, int n_samples,
                                         const double *blocks, double thresh, uint8_t *result, void *stream)
{
    if (!h)
        return PACX_E_ARG;
    if (n_blocks == 0)
        return PACX_OK;
    if (n_blocks < 0 || n_blocks > 0x7fffffffLL || n_channels < 1 || n_samples < 1 || !blocks || !result)
        return fail(h, PACX_E_ARG, "pacx_transient_detect_f64: bad argument");
    if ((long long)n_channels * n_samples > 0x7fffffffLL)
        return fail(h, PACX_E_UNSUPPORTED, "pacx_transient_detect_f64: block too large");
    HIP_TRY(h, hipSetDevice(h->device));
    pacx_launch_transient_f64(n_blocks, n_channels, n_samples, blocks, thresh, result, (hipStream_t)stream);
    return post_launch(h, "pacx_transient_detect_f64");
}

extern "C" int pacx_unpack_batch(pacx_handle *h, int64_t n_cf, const uint8_t *payload, int payload_stride,
                                 const int64_t *offsets, const int32_t *n_bytes, uint8_t *cf_flags,
                                 int32_t *overall_scale, int32_t *scale_factor, int32_t *bit_alloc,
                                 int32_t *mantissa, uint32_t *status, void *stream)
{
    if (!h)
        return PACX_E_ARG;
    if (n_cf == 0)
        return PACX_OK;
    if (n_cf < 0 || !payload || !n_bytes || !cf_flags || !overall_scale || !scale_factor || !bit_alloc ||
        !mantissa || (!offsets && payload_stride <= 0))
        return fail(h, PACX_E_ARG, "pacx_unpack_batch: bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    pacx_launch_unpack(h->T, n_cf, payload, payload_stride, (const long long *)offsets, n_bytes, cf_flags,
                       overall_scale, scale_factor, bit_alloc, mantissa, status, (hipStream_t)stream);
    return post_launch(h, "pacx_unpack_batch");
}

/* the decoders' own workspaces: windowed blocks when the caller wants PCM only; lines + SBR flags (+ status words) */
static int reserve_dec_blocks(pacx_handle *h, long long n_cf)
{
    if (n_cf <= h->ws_blocks_cf)
        return PACX_OK;
    HIP_TRY(h, hipDeviceSynchronize());
    if (h->ws_blocks)
        (void)hipFree(h->ws_blocks);
    h->ws_blocks = nullptr;
    h->ws_blocks_cf = 0;
    HIP_TRY(h, hipMalloc((void **)&h->ws_blocks, (size_t)n_cf * PACX_N_LONG * sizeof(double)));
    h->ws_blocks_cf = n_cf;
    return PACX_OK;
}

static int reserve_dec_lines(pacx_handle *h, long long n_cf)
{
    if (n_cf <= h->ws_dec_cf)
        return PACX_OK;
    HIP_TRY(h, hipDeviceSynchronize());
    if (h->ws_dec_lines) (void)hipFree(h->ws_dec_lines);
    if (h->ws_dec_sbr) (void)hipFree(h->ws_dec_sbr);
    if (h->ws_dec_status) (void)hipFree(h->ws_dec_status);
    h->ws_dec_lines = nullptr;
    h->ws_dec_sbr = nullptr;
    h->ws_dec_status = nullptr;
    h->ws_dec_cf = 0;
    HIP_TRY(h, hipMalloc((void **)&h->ws_dec_lines, (size_t)n_cf * PACX_M_LONG * sizeof(double)));
    HIP_TRY(h, hipMalloc((void **)&h->ws_dec_sbr, (size_t)n_cf));
    HIP_TRY(h, hipMalloc((void **)&h->ws_dec_status, (size_t)n_cf * sizeof(uint32_t)));
    h->ws_dec_cf = n_cf;
    return PACX_OK;
}

static int decode_scalar(pacx_handle *h, const char *what, int64_t n_blocks, int n_channels, const uint8_t *cf_flags,
                         const int32_t *overall_scale, const int32_t *scale_factor, const int32_t *bit_alloc,
                         const int32_t *mantissa, double *lines, double *blocks, int16_t *pcm, uint32_t *status,
                         int routing, void *stream)
{
    if (!h)
        return PACX_E_ARG;
    if (routing < 0 || routing > 2 || n_blocks < 0 || n_channels < 1 || (n_blocks > 0 && (!cf_flags || !overall_scale || !scale_factor ||
                                                              !bit_alloc || !mantissa)) ||
        (!blocks && !pcm && !lines))
        return fail(h, PACX_E_ARG, std::string(what) + ": bad argument");
    if (h->T.use_vq)
        return fail(h, PACX_E_UNSUPPORTED, std::string(what) + ": handle was created with use_vq "
                                           "(pacx_decode_vq_batch reads gain-shape streams)");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const long long n_cf = n_blocks * n_channels;
    double *work = blocks;
    if (!work && pcm && n_cf > 0) {
        const int rc = reserve_dec_blocks(h, n_cf);
        if (rc != PACX_OK)
            return rc;
        work = h->ws_blocks;
    }
    const double *lines_in = nullptr;
    if (!h->T.use_sbr)
        routing = 0;
    if (n_cf > 0 && (routing || lines)) {
        /* an SBR file (PACFile.Decode, coder/pacfile.py:645-668: long blocks with a coded omitted band are
           Decode_SBR's, coder/codec.py:95-222 scalar branch), or a caller who wants the dequantised lines */
        const int rc = reserve_dec_lines(h, n_cf);
        if (rc != PACX_OK)
            return rc;
        double *ln = lines ? lines : h->ws_dec_lines;
        uint32_t *stw = status ? status : h->ws_dec_status;
        HIP_TRY(h, hipMemsetAsync(stw, 0, (size_t)n_cf * sizeof(uint32_t), st));
        pacx_launch_sbr_scalar_lines(h->T, n_cf, cf_flags, scale_factor, bit_alloc, mantissa, ln, h->ws_dec_sbr, routing,
                                     st);
        if (routing)
            pacx_launch_sbr_recon(h->T, h->vqdec_view.data(), n_cf, h->ws_dec_sbr, ln, stw, st);
        lines_in = ln;
    } else if (status && n_cf > 0) {
        HIP_TRY(h, hipMemsetAsync(status, 0, (size_t)n_cf * sizeof(uint32_t), st));
    }
    if (work || pcm)
        pacx_launch_decode(h->T, n_blocks, n_channels, cf_flags, overall_scale, scale_factor, bit_alloc, mantissa,
                           lines_in, work, pcm, st);
    return post_launch(h, what);
}

extern "C" int pacx_decode_batch(pacx_handle *h, int64_t n_blocks, int n_channels, const uint8_t *cf_flags,
                                 const int32_t *overall_scale, const int32_t *scale_factor,
                                 const int32_t *bit_alloc, const int32_t *mantissa, double *blocks,
                                 int16_t *pcm, void *stream)
{
    if (h && !blocks && !pcm)
        return fail(h, PACX_E_ARG, "pacx_decode_batch: bad argument");
    return decode_scalar(h, "pacx_decode_batch", n_blocks, n_channels, cf_flags, overall_scale, scale_factor, bit_alloc,
                         mantissa, nullptr, blocks, pcm, nullptr, 0, stream);
}

extern "C" int pacx_decode_sbr_batch(pacx_handle *h, int64_t n_blocks, int n_channels, const uint8_t *cf_flags,
                                     const int32_t *overall_scale, const int32_t *scale_factor,
                                     const int32_t *bit_alloc, const int32_t *mantissa, int routing,
                                     double *lines, double *blocks, int16_t *pcm, uint32_t *status, void *stream)
{
    return decode_scalar(h, "pacx_decode_sbr_batch", n_blocks, n_channels, cf_flags, overall_scale, scale_factor,
                         bit_alloc, mantissa, lines, blocks, pcm, status, routing ? 2 : 1, stream);
}

extern "C" int pacx_decode_vq_batch(pacx_handle *h, int64_t n_blocks, int n_channels, const uint8_t *payload,
                                    int payload_stride, const int64_t *offsets, const int32_t *n_bytes,
                                    uint8_t *cf_flags, int32_t *overall_scale, int32_t *bit_alloc,
                                    double *lines, double *blocks, int16_t *pcm, uint32_t *status,
                                    void *stream)
{
    if (!h)
        return PACX_E_ARG;
    if (n_blocks < 0 || n_channels < 1 ||
        (n_blocks > 0 && (!payload || !n_bytes || !cf_flags || !overall_scale || !bit_alloc || !status ||
                          (!offsets && payload_stride <= 0))))
        return fail(h, PACX_E_ARG, "pacx_decode_vq_batch: bad argument");
    if (!h->T.use_vq)
        return fail(h, PACX_E_UNSUPPORTED, "pacx_decode_vq_batch: handle was created without use_vq");
    HIP_TRY(h, hipSetDevice(h->device));
    const long long n_cf = n_blocks * n_channels;
    hipStream_t st = (hipStream_t)stream;
    {
        const int rc = reserve_dec_lines(h, n_cf);
        if (rc != PACX_OK)
            return rc;
    }
    double *ln = lines ? lines : h->ws_dec_lines;
    double *work = blocks;
    if (!work && pcm && n_cf > 0) {
        const int rc = reserve_dec_blocks(h, n_cf);
        if (rc != PACX_OK)
            return rc;
        work = h->ws_blocks;
    }
    if (n_cf > 0) {
        HIP_TRY(h, hipMemsetAsync(status, 0, (size_t)n_cf * sizeof(uint32_t), st));
        pacx_launch_vq_dec(h->T, h->vqdec_view.data(), n_cf, payload, payload_stride, (const long long *)offsets,
                           n_bytes, cf_flags, overall_scale, bit_alloc, ln, h->ws_dec_sbr, status, st);
    }
    if (work || pcm)
        pacx_launch_decode(h->T, n_blocks, n_channels, cf_flags, overall_scale, nullptr, nullptr, nullptr, ln,
                           work, pcm, st);
    return post_launch(h, "pacx_decode_vq_batch");
}
