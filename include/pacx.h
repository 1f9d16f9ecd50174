/*
 * pacx.h -- C ABI of the MI355X (gfx950) batched audio-frame encode path.
 *
 * Drop-in boundary for the per-frame hot loop of Abhipray/audio-codec.  The
 * reference has no FFI layer: the path sits behind plain Python calls,
 *     PACFile.Encode        coder/pacfile.py:627-643
 *     codec.Encode          coder/codec.py:225-263
 *     codec.EncodeSingleChannel  coder/codec.py:266-380
 * and the five modules those call (window.py, mdct.py, psychoac.py,
 * bitalloc.py, quantize.py).  Each entry point below names the reference
 * function(s) it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative PACX_E_* code and
 *     never throws; pacx_last_error() gives the text;
 *   - the CALLER owns every data buffer.  All `const void*` / `T*` data
 *     arguments are DEVICE pointers (hipMalloc / torch tensor .data_ptr())
 *     unless the parameter is documented as host;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*, NULL =
 *     default stream) and is asynchronous; the library keeps a grow-only
 *     device workspace per handle (pacx_reserve() sizes it up front so that
 *     no allocation happens inside a timed or graph-captured region);
 *   - one handle per device; handles are independent (no global state).
 *
 * Units: a "channel-frame" (cf) is one EncodeSingleChannel call: 2*nMDCTLines
 * samples of one channel in, nMDCTLines MDCT lines out.  cf index =
 * frame * n_channels + channel (the order blocks appear in a .pac file).
 */
#ifndef PACX_H
#define PACX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PACX_ABI_VERSION 7

/* error codes */
#define PACX_OK            0
#define PACX_E_ARG        -1   /* bad argument                               */
#define PACX_E_UNSUPPORTED -2  /* configuration outside what the kernels do  */
#define PACX_E_HIP        -3   /* a HIP runtime call failed                  */
#define PACX_E_NOMEM      -4

/* sample formats of pacx_pcm.dtype */
#define PACX_PCM_I16 0         /* 16-bit PCM codes (coder/pcmfile.py:89-99 contract) */
#define PACX_PCM_F64 1         /* signed fractions, as codec.Encode receives them    */

/* per-frame flag bits (coder/pacfile.py:575-577 order) */
#define PACX_FLAG_LAST 1u
#define PACX_FLAG_CUR  2u
#define PACX_FLAG_NEXT 4u

/* window kinds chosen by codec.getCorrectWindow (coder/codec.py:30-44) */
#define PACX_WIN_SINE 0
#define PACX_WIN_START 1
#define PACX_WIN_STOP 2
#define PACX_WIN_STARTSTOP 3
/* further tables addressable through pacx_window_batch */
#define PACX_WIN_SINE_SHORT 4  /* SineWindow on a 256-sample short block      */
#define PACX_WIN_HANN 5        /* HanningWindow, 2048 (coder/window.py:29-41) */
#define PACX_WIN_HANN_SHORT 6  /* HanningWindow, 256                          */
#define PACX_WIN_KBD 7         /* KBDWindow(alpha = 4), 2048 (coder/window.py:45-57) */
#define PACX_WIN_KBD_SHORT 8   /* KBDWindow(alpha = 4), 256                   */

/* mode bits of pacx_mdct_batch */
#define PACX_MDCT_SHORT 1        /* 8 short sub-blocks per frame              */
#define PACX_MDCT_PREWINDOWED 2  /* input already windowed: plain mdct.MDCT   */
#define PACX_MDCT_KBD 4          /* KBDWindow instead of the sine window, long or
                                    short blocks: MDCT(KBDWindow(x), halfN, halfN), the
                                    expression at coder/bitalloc.py:161 (frame_flags
                                    must be NULL)                              */

/* per-cf status word bits written by pacx_encode_batch */
#define PACX_ST_SHORT        1u   /* coded as 8 short sub-blocks               */
#define PACX_ST_ZERO_SUBBLOCK 2u  /* a short sub-block was all zeros: the
                                     reference drops the whole hop
                                     (coder/pacfile.py:530-533)                */
#define PACX_ST_ALLOC_CAP    4u   /* BitAlloc left through its 200-pass guard
                                     (coder/bitalloc.py:116-119)               */

#define PACX_ST_VQ_UNDEFINED  8u   /* gain-shape coder reached a case where the
                                     reference itself fails (a 1-dimensional
                                     PVQ leaf never returns, an all-zero half
                                     gives NaN pulses, a gain index wider than
                                     128 bits): the band's bits are zeros     */

#define PACX_ST_GUARD        16u   /* (handles created with pacx_config.guard = 1)
                                     a rounding decision of this cf sat within a few
                                     ulps of its boundary: a mantissa / scale-factor
                                     quantiser input (2^R-1)|x|+1 next to an even
                                     integer (coder/quantize.py:73), or a BitAlloc
                                     value Ropt - level next to k + 1/2
                                     (coder/bitalloc.py:103); with use_vq also a split
                                     angle or a band's mu-law gain at a boundary of its
                                     quantiser, a pulse search whose floor(K|x|/l1) or
                                     whose last pulse hangs on the last bits of the unit
                                     vector (coder/gain_shape_quantize.py:30-54, 315-408),
                                     and a coded band with a line at rounding-noise level
                                     (np.sign(0) erases a pulse).  The codes are still the
                                     ones this arithmetic gives; a harness that needs
                                     certainty recomputes flagged frames on the CPU  */
#define PACX_ST_MALFORMED    32u   /* decode: the payload of this channel-block is
                                     truncated or carries an impossible field (the
                                     reference raises "Only read a partial block of
                                     coded PACFile data", coder/pacfile.py:203-205):
                                     its outputs are zeros                          */
#define PACX_ST_REF_RAISES   64u   /* scalar mantissas + SBR (use_sbr without use_vq), long frame:
                                      an SBR-omitted band received bits.  The reference quantises the
                                      band's one value with vMantissa(np.mean(..)) -- a NumPy scalar its
                                      vQuantizeUniform assigns into -- and raises TypeError there
                                      (coder/codec.py:541-546, coder/quantize.py:73-74; recorded in
                                      tests/golden/sbr_scalar.json), so no output is defined: the
                                      frame's n_bytes is 0, its other outputs are unspecified, and the
                                      host mirror raises the reference's error                      */

#define PACX_SHORT_PER_FRAME 8    /* sub-blocks of a short frame (coder/pacfile.py:527) */

typedef struct pacx_handle pacx_handle;

/*
 * Static configuration = the CodingParams attributes the path reads
 * (coder/pacfile.py:699-707, 323-330).  Table pointers are HOST pointers,
 * copied at create time.  Tables marked "optional" may be NULL: the library
 * then uses its built-in copies (csrc/pacx_tables_gen.h: the reference's NumPy
 * expressions evaluated once and stored as bit patterns, so a C host gets the
 * same bits as the Python host) -- windows and gain-shape tables always, the
 * Bark / threshold-in-quiet tables for 44.1 and 48 kHz.  At other sample rates
 * those two are evaluated with the C math library (last-place differences from
 * NumPy's are possible); pacx_tables_exact() tells which case a handle is in.
 */
typedef struct pacx_config {
    int32_t abi_version;            /* PACX_ABI_VERSION                           */
    int32_t device;                 /* HIP device ordinal                         */
    int32_t sample_rate;            /* Hz                                         */
    int32_t n_lines_long;           /* nMDCTLines of a long block: 1024           */
    int32_t n_lines_short;          /* 128 (coder/pacfile.py:490)                 */
    int32_t n_scale_bits;           /* 4                                          */
    int32_t n_mant_size_bits;       /* 12                                         */
    int32_t n_bands_long;           /* sfBands.nBands                             */
    int32_t n_bands_short;          /* sfBandsShort.nBands                        */
    double  target_bits_per_sample; /* kb/s per channel / (sampleRate/1000)       */
    const int32_t *band_lines_long;   /* [n_bands_long]  sfBands.nLines            */
    const int32_t *band_lines_short;  /* [n_bands_short] sfBandsShort.nLines       */
    /* optional float64 tables, evaluated by the caller with NumPy so that they
       are bit-identical to the reference's: */
    const double *win_long;         /* [4][2*n_lines_long] kinds PACX_WIN_*       */
    const double *win_short;        /* [2*n_lines_short] sine                     */
    const double *hann_long;        /* [2*n_lines_long]  coder/window.py:37-39    */
    const double *hann_short;       /* [2*n_lines_short]                          */
    const double *bark_long;        /* [n_lines_long]  Bark(mdct line freq)       */
    const double *thresh_long;      /* [n_lines_long]  Thresh(mdct line freq)     */
    const double *bark_short;       /* [n_lines_short]                            */
    const double *thresh_short;     /* [n_lines_short]                            */
    double fft_norm_long;           /* 4/(N^2 mean(np.hanning(N)^2)); 0 = compute */
    double fft_norm_short;
    double fft_freq_step_long;      /* np.fft.rfftfreq step 1/(N*(1/sr)); 0 = compute */
    double fft_freq_step_short;
    /* coding variant of the file (coder/pacfile.py:703-705):
       use_vq  -- gain-shape pyramid VQ of every band instead of scale factor +
                  mantissas (coder/gain_shape_quantize.py); the handle then
                  serves pacx_encode_vq_batch instead of pacx_encode_batch;
       use_sbr -- long blocks go through EncodeSingleChannel_SBR: the bands of
                  sbr.omitted_bands (coder/sbr.py:6-9) carry one value each.
                  With use_vq this is the configuration the reference's driver
                  selects below 128 kb/s.  Without use_vq (scalar mantissas,
                  coder/codec.py:529-555) the reference is defined only while the
                  omitted bands get no bits -- pacx_encode_batch / _pack_batch then
                  follow it bit for bit (budget from the full block, max|FFT| in the
                  overall scale, BitAlloc_SBR's one-line bands) and flag the frames
                  on which it raises with PACX_ST_REF_RAISES. */
    int32_t use_vq;
    int32_t use_sbr;
    const double *half_log2;        /* optional [max band lines + 1]: 0.5*np.log2(L) */
    const double *vq_log2_tan;      /* optional [2^12 - 1]: log2(tan(theta_q) + eps) of every
                                       quantised split angle of <= 12 bits; the codes of width a
                                       start at 2^(a-1) - 1 (bit_allocation_ms, :302-309)       */
    double log_mu1;                 /* np.log(256.0) of mu_law_fn; 0 = compute       */
    /* decode side of an SBR file (coder/codec.py:147, 163-164), optional: */
    const double *sbr_gauss;        /* [2r+1] normalised weights of gaussian_filter1d(sigma=200) */
    int32_t sbr_gauss_radius;       /* r = int(4*200 + 0.5) = 800                    */
    const double *line_freq_long;   /* [n_lines_long] (k + 1/2) * sampleRate/(2*n_lines_long)   */
    /* KBDWindow(alpha = 4) tables (coder/window.py:53-57), optional: */
    const double *kbd_long;         /* [2*n_lines_long]                              */
    const double *kbd_short;        /* [2*n_lines_short]                             */
    /* 1: compute PACX_ST_GUARD (rounding decisions near their boundaries) in the whole-path
       entry points, the gain-shape one included; costs about 3 % of the encode throughput, off by default */
    int32_t guard;
} pacx_config;

/* one written field of a gain-shape coded band (see pacx_encode_vq_batch) */
typedef struct pacx_vq_entry {
    uint64_t value;                 /* low 64 bits of the index                    */
    int32_t width;                  /* bits written                                */
    int32_t band;
} pacx_vq_entry;

/*
 * Strided view of PCM input.  Sample s of frame f, channel c is element
 *     data[f*frame_stride + c*channel_stride + s*sample_stride]   (strides in elements)
 * for s in [0, 2*n_lines_long).  A stream with 50 % overlap
 * (coder/pacfile.py:460-464: frame = prior hop || new hop) has
 * frame_stride = n_lines_long * sample_stride and holds n_frames+1 hops, the
 * first being the prior block (zeros at the start of a file,
 * coder/pacfile.py:335-339).  Independent frames use frame_stride >= 2*n_lines_long.
 * Fast path: dtype I16, sample_stride 1, data 16-byte aligned, strides
 * multiples of 8.
 */
typedef struct pacx_pcm {
    const void *data;
    int32_t dtype;                  /* PACX_PCM_*                                 */
    int32_t n_channels;
    int64_t n_frames;
    int64_t frame_stride;
    int64_t channel_stride;
    int64_t sample_stride;
} pacx_pcm;

/* ---- lifetime ---------------------------------------------------------- */
int  pacx_create(const pacx_config *cfg, pacx_handle **out);
void pacx_destroy(pacx_handle *h);
/* text of the last error on this handle (h may be NULL: last create error) */
const char *pacx_last_error(const pacx_handle *h);
int  pacx_abi_version(void);
/* ints per cf in the scale_factor / bit_alloc outputs:
   max(n_bands_long, 8*n_bands_short) */
int  pacx_band_stride(const pacx_handle *h);
/* bytes per cf slot in the packed-payload output of pacx_pack_batch */
int  pacx_payload_stride(const pacx_handle *h);
/* pre-size the device workspace for batches of up to n_cf channel-frames */
int  pacx_reserve(pacx_handle *h, int64_t n_cf);
/* 1 if every float64 table of the handle is bit-identical to the reference's NumPy
   evaluation (supplied by the caller or built in), 0 if some came from the C math
   library (NULL Bark / threshold tables at a sample rate without built-in copies) */
int  pacx_tables_exact(const pacx_handle *h);
/*
 * psychoac.py band layout for hosts without NumPy: AssignMDCTLinesFromFreqLimits
 * (coder/psychoac.py:106-124) followed by the merge rule of ScaleFactorBands
 * (:143-149: a band of <= 12 lines joins its right neighbour).  Plain IEEE double
 * arithmetic, bit-for-bit the reference's.  band_lines (HOST, room for 25 ints)
 * receives sfBands.nLines, *n_bands their number.  No handle needed.
 */
int  pacx_default_bands(int sample_rate, int n_mdct_lines, int32_t *band_lines, int32_t *n_bands);

/* ---- stage entry points (one per replaced reference module) ------------ */

/*
 * window.py + mdct.py: MDCT(window(data), halfN, halfN)[:halfN]
 * (coder/codec.py:303-305; window choice coder/codec.py:30-44; int16 input is
 * first mapped as coder/pcmfile.py:89-99 does).
 *   mode 0: every frame is one long block; the window kind comes
 *       from frame_flags (NULL = all sine).  lines: [n_cf][n_lines_long].
 *   mode & PACX_MDCT_PREWINDOWED: no window is applied (mdct.MDCT alone,
 *       coder/mdct.py:43-69 with a = b = halfN).
 *   mode & PACX_MDCT_SHORT: every frame is cut into 8 short sub-blocks at
 *       n = 448 + 128 j (coder/pacfile.py:526-527), sine-windowed.
 *       lines: [n_cf][8][n_lines_short].
 *   mode & PACX_MDCT_KBD: as mode 0 / PACX_MDCT_SHORT with the KBD window
 *       (alpha = 4) in place of the sine window; frame_flags must be NULL.
 * frame_flags: device uint8 [n_frames] of PACX_FLAG_* or NULL.
 * max_scale (optional, device int32 [n_cf] or [n_cf][8]): the overall scale
 * factor ScaleFactor(max|line|, nScaleBits) (coder/codec.py:308-310).
 * The lines written are NOT multiplied by 2^scale.
 */
int pacx_mdct_batch(pacx_handle *h, const pacx_pcm *in, const uint8_t *frame_flags,
                    int mode, double *lines, int32_t *max_scale, void *stream);

/*
 * psychoac.py: CalcSMRs(data, mdctLines*2^scale, scale, sampleRate, sfBands)
 * (coder/psychoac.py:220-291, with getMaskedThreshold :163-217 and
 * estimate_peaks :308-329 inside).  `lines` are the UNSCALED MDCT lines
 * (X / 2^scale is what the reference feeds to SPL, :250-254).
 *   smr:       [n_cf][band_stride]  (short: sub-block j at [j*n_bands_short ...])
 *   threshold: optional [n_cf][n_lines_long] masked threshold in dB SPL
 *   n_peaks:   optional int32 [n_cf] (short: [n_cf][8]) tonal maskers found
 */
int pacx_smr_batch(pacx_handle *h, const pacx_pcm *in, const double *lines,
                   int short_blocks, double *smr, double *threshold,
                   int32_t *n_peaks, void *stream);

/*
 * bitalloc.py: BitAlloc(bitBudget, maxMantBits, nBands, nLines, SMRs)
 * (coder/bitalloc.py:62-121) with the budget rule of coder/codec.py:288-299
 * evaluated from the frame flags (scalar-mantissa variant).
 *   bit_alloc: int32 [n_cf][band_stride]; status (optional) uint32 [n_cf].
 */
int pacx_bitalloc_batch(pacx_handle *h, int64_t n_cf, int n_channels,
                        const uint8_t *frame_flags, int short_blocks,
                        const double *smr, int32_t *bit_alloc, uint32_t *status,
                        void *stream);

/*
 * quantize.py: per band ScaleFactor(max|line|, nScaleBits, bitAlloc) and
 * vMantissa(lines, scale, nScaleBits, bitAlloc) (coder/codec.py:362-377) on
 * lines * 2^overall_scale.
 *   scale_factor: int32 [n_cf][band_stride]
 *   mantissa:     int32 [n_cf][n_lines_long], LINE-indexed (0 where the band
 *                 got no bits); the reference's dense layout is the
 *                 concatenation of the allocated bands.
 */
int pacx_quantize_batch(pacx_handle *h, int64_t n_cf, const double *lines,
                        const int32_t *overall_scale, const int32_t *bit_alloc,
                        int short_blocks, int32_t *scale_factor, int32_t *mantissa,
                        void *stream);

/* ---- the whole path ---------------------------------------------------- */

/*
 * codec.Encode for a batch (coder/codec.py:225-380, scalar mantissas):
 * window -> MDCT -> overall scale -> SMR -> BitAlloc -> scale factors and
 * mantissas, for every channel of every frame.  Frames whose PACX_FLAG_CUR
 * bit is set are coded as 8 short sub-blocks (coder/pacfile.py:489-547).
 *   overall_scale: int32 [n_cf][8]      (long frames use [0])
 *   scale_factor, bit_alloc: int32 [n_cf][band_stride]
 *   mantissa: int32 [n_cf][n_lines_long] line-indexed
 *   status:   uint32 [n_cf] PACX_ST_*
 */
int pacx_encode_batch(pacx_handle *h, const pacx_pcm *in, const uint8_t *frame_flags,
                      int32_t *overall_scale, int32_t *scale_factor,
                      int32_t *bit_alloc, int32_t *mantissa, uint32_t *status,
                      void *stream);

/*
 * pacx_encode_batch followed by pacx_pack_batch in one call (what a file writer
 * wants: codec.Encode + PACFile.writeEncodedBits for every block,
 * coder/pacfile.py:466-482, 552-608).  Long frames run BitAlloc, quantisation
 * and packing in one kernel.  `mantissa` may be NULL when the line-indexed
 * mantissas are not wanted; the other outputs are as in the two calls above.
 */
int pacx_encode_pack_batch(pacx_handle *h, const pacx_pcm *in, const uint8_t *frame_flags,
                           int32_t *overall_scale, int32_t *scale_factor, int32_t *bit_alloc,
                           int32_t *mantissa, uint32_t *status, uint8_t *payload, int32_t *n_bytes,
                           void *stream);

/*
 * The shipped configuration of the reference (coder/pacfile.py:699-707: useVQ
 * always, useSBR below 128 kb/s) for a batch, from PCM to finished payloads:
 *   codec.Encode / EncodeSingleChannel with useVQ   coder/codec.py:239-246, 292-294, 330-360
 *   codec.Encode_SBR / EncodeSingleChannel_SBR      coder/codec.py:383-531 (long blocks of an SBR file,
 *                                                   routing coder/pacfile.py:639-643)
 *   BitAlloc_SBR                                    coder/bitalloc.py:123-145
 *   quantize_gain_shape, split_band_encode, quantize_pvq, pvq_search,
 *   encode_pvq_vector, pvq_compute_k_for_R, gain_shape_alloc,
 *   bit_allocation_ms, mu_law_fn                    coder/gain_shape_quantize.py:30-124, 244-408, 476-512
 *   getNumBytesNeeded + WriiteEncodedBitsVQ         coder/pacfile.py:342-402, 552-592
 * The handle must have been created with use_vq.  Field lists are variable
 * length, so the natural output is the bit string itself:
 *   overall_scale: int32 [n_cf][8];  bit_alloc: int32 [n_cf][band_stride], the
 *   FINAL allocation (a band of zero gain drops to 0, coder/codec.py:352-353);
 *   payload: uint8 [n_cf][payload_stride], n_bytes: int32 [n_cf] as pacx_pack_batch;
 *   status: uint32 [n_cf] PACX_ST_*.
 * entries / entry_count (optional, for callers that want the reference's
 * (indices, idx_bits) lists): entries [n_cf][8][32][entries_per_band] receives
 * the fields of sub-block j, band b in writing order, entry_count [n_cf][8][32]
 * how many there were (may exceed entries_per_band: the excess is not stored).
 */
int pacx_encode_vq_batch(pacx_handle *h, const pacx_pcm *in, const uint8_t *frame_flags,
                         int32_t *overall_scale, int32_t *bit_alloc, uint8_t *payload,
                         int32_t *n_bytes, uint32_t *status, pacx_vq_entry *entries,
                         int32_t *entry_count, int32_t entries_per_band, void *stream);

/*
 * pacfile.py bit layout (coder/pacfile.py:404-447, 552-577; bitpack.py:37-102):
 * per cf the MSB-first payload  last|cur|next | overallScale(4) |
 * per band: (alloc-1 or 0)(nMantSizeBits) scaleFactor(nScaleBits) mantissas,
 * eight such bodies for a short frame.
 *   payload: uint8 [n_cf][payload_stride]; n_bytes: int32 [n_cf]
 *   (0 for a hop the reference drops).
 */
int pacx_pack_batch(pacx_handle *h, int64_t n_cf, int n_channels,
                    const uint8_t *frame_flags, const int32_t *overall_scale,
                    const int32_t *scale_factor, const int32_t *bit_alloc,
                    const int32_t *mantissa, const uint32_t *status,
                    uint8_t *payload, int32_t *n_bytes, void *stream);

/*
 * Concatenate "<L nBytes" + payload of every cf in order (the body of a .pac
 * file after its header, coder/pacfile.py:566-568,608).
 *   body: uint8 [>= sum(4 + n_bytes)]; total_bytes: device int64 [1].
 */
int pacx_gather_body(pacx_handle *h, int64_t n_cf, const uint8_t *payload,
                     const int32_t *n_bytes, uint8_t *body, int64_t body_capacity,
                     int64_t *total_bytes, void *stream);

/* ---- function-level entry points --------------------------------------
 * One per remaining public function of the replaced modules, so that each has
 * a GPU implementation behind the reference's signature.  The batched path
 * above does not call these. */

/* window.py: y = window * x for n_rows rows (SineWindow / StartWindow /
 * StopWindow / StartStopWindow / HanningWindow / KBDWindow; coder/window.py:14-92).
 * `window` is a PACX_WIN_* id; rows are 2048 (256 for *_SHORT) float64. */
int pacx_window_batch(pacx_handle *h, int window, int64_t n_rows, const double *x, double *y,
                      void *stream);
/* the same with a caller-supplied DEVICE table of `len` float64 (any block length,
 * e.g. KBDWindow with another alpha or N): y[r][i] = table[i] * x[r][i] */
int pacx_window_table_batch(pacx_handle *h, const double *table, int len, int64_t n_rows,
                            const double *x, double *y, void *stream);
/* quantize.py: vQuantizeUniform(x, n_bits) (coder/quantize.py:61-78) */
int pacx_quantize_uniform(pacx_handle *h, int64_t n, const double *x, int n_bits, int64_t *codes,
                          void *stream);
/* quantize.py: ScaleFactor(x[i], n_scale_bits, n_mant_bits) (coder/quantize.py:99-125) */
int pacx_scale_factor(pacx_handle *h, int64_t n, const double *x, int n_scale_bits, int n_mant_bits,
                      int64_t *scale, void *stream);
/* quantize.py: vMantissa(x, scale, n_scale_bits, n_mant_bits) (coder/quantize.py:229-250) */
int pacx_mantissa(pacx_handle *h, int64_t n, const double *x, int scale, int n_scale_bits,
                  int n_mant_bits, int64_t *mantissa, void *stream);
/* quantize.py, decode side: vDequantizeUniform(codes, n_bits) (coder/quantize.py:82-95; the scalar
 * DequantizeUniform, :40-57, gives the same values) -- float64 signed fractions; n_bits <= 53 */
int pacx_dequantize_uniform(pacx_handle *h, int64_t n, const int64_t *codes, int n_bits, double *x,
                            void *stream);
/* quantize.py: vDequantize(scale, mantissa, n_scale_bits, n_mant_bits) (coder/quantize.py:254-274; the
 * scalar Dequantize, :200-225, likewise) */
int pacx_dequantize(pacx_handle *h, int64_t n, const int64_t *mantissa, int scale, int n_scale_bits,
                    int n_mant_bits, double *x, void *stream);
/* quantize.py: MantissaFP / DequantizeFP (coder/quantize.py:130-175), the floating-point pair of the module
 * (not used by the codec; for a complete module swap and the module's self-test, :283-319) */
int pacx_mantissa_fp(pacx_handle *h, int64_t n, const double *x, int scale, int n_scale_bits,
                     int n_mant_bits, int64_t *mantissa, void *stream);
int pacx_dequantize_fp(pacx_handle *h, int64_t n, const int64_t *mantissa, int scale, int n_scale_bits,
                       int n_mant_bits, double *x, void *stream);
/* mdct.py: IMDCT(lines, a, a) (coder/mdct.py:56-62, 73-77) for the codec's block sizes, on the FFT kernels
 * of the decode path.  mode 0: rows of n_lines_long lines -> rows of 2*n_lines_long samples; PACX_MDCT_SHORT:
 * rows of 8 x n_lines_short lines -> the eight 256-sample outputs overlap-added at 448 + 128 s inside a
 * 2048-sample row (a lone sub-block 0 comes out at samples 448..703).  No window, no overall scale. */
int pacx_imdct_batch(pacx_handle *h, int64_t n_rows, int mode, const double *lines, double *blocks,
                     void *stream);
/* mdct.py for ANY split a + b (coder/mdct.py:14-77: MDCTslow / MDCT / IMDCT at the sizes the FFT kernels do
 * not cover -- the reference's own self-test uses a = b = 4 and 6): the defining cosine sums with an exact
 * integer phase reduction, O(N^2) per row.  forward: x [n_rows][a+b] -> y [n_rows][(a+b)/2]; inverse: the
 * other way round (the reference's scaling: 2/N on the forward transform, 2 on the inverse). */
int pacx_mdct_direct_batch(pacx_handle *h, int64_t n_rows, int a, int b, int inverse, const double *x,
                           double *y, void *stream);
/* bitalloc.py: BitAlloc(budget[i], max_mant_bits, n_bands, band_lines, smr[i])
 * for n independent problems (coder/bitalloc.py:62-121); all pointers device. */
int pacx_bitalloc_generic(pacx_handle *h, int64_t n, int n_bands, const int32_t *band_lines,
                          const double *budget, int max_mant_bits, const double *smr,
                          int32_t *bit_alloc, void *stream);

/* detect_transients.py + the flag shifting of the driver loop
 * (coder/detect_transients.py:5-23, coder/pacfile.py:717-741): `hops` views the
 * stream hop by hop (frame_stride = one hop, n_frames = number of hops read from
 * the file); transient[h] = parTransientDetect(hop h || zeros).  If frame_flags
 * is not NULL it receives n_frames + 2 PACX_FLAG_* bytes: one per block the
 * driver writes (every hop, the last hop a second time, the Close block). */
int pacx_transient_flags(pacx_handle *h, const pacx_pcm *hops, uint8_t *transient,
                         uint8_t *frame_flags, void *stream);

/* detect_transients.parTransientDetect(block, thresh, axis=1) (coder/detect_transients.py:5-23) for any
 * float64 blocks [n_blocks][n_channels][n_samples] (device): result[i] = 2 where the mean is exactly zero (the
 * reference returns the int 0 there), else 1 / 0 = any(peak / avg > thresh).  The mean is taken in NumPy's
 * pairwise order.  The encode path itself uses pacx_transient_flags on the int16 hops. */
int pacx_transient_detect_f64(pacx_handle *h, int64_t n_blocks, int n_channels, int n_samples,
                              const double *blocks, double thresh, uint8_t *result, void *stream);

/*
 * All-long scalar batches run on the caller's stream alone (the default), or with the side chain (FFT, peaks: it
 * only reads the PCM) forked to a second stream of the handle beside the transform.  The fork pays where the
 * runtime puts both streams on one hardware queue -- a process that keeps several steps in flight on several
 * handles (engine.EncoderPool switches it on) -- and costs where it does not: a fork and a join across hardware
 * queues take longer than the transform they hide (one handle, one step in flight: 39.1 against 42.6 M cf/s).
 */
int pacx_set_side_fork(pacx_handle *h, int enable);

/*
 * psychoac.CalcSMRs / getMaskedThreshold (coder/psychoac.py:163-291) for block lengths other than the two the
 * handle's tuned kernels are built for (2 * n_lines_long and 2 * n_lines_short samples: 2048 and 256, the
 * reference driver's, coder/pacfile.py:699,490) -- nMDCTLines = 512, for instance.  A function-level path: one
 * workgroup per block, the spectrum by a direct DFT, then the tuned kernels' arithmetic.  All tables are the
 * CALLER's, evaluated the way the reference evaluates them (the Python mirror does it with NumPy), on the device:
 */
typedef struct pacx_smr_tables {
    const double *hann;             /* [n_samples] 0.5 (1 - cos(2 pi (n + 1/2) / N)), window.HanningWindow      */
    const double *tw_cos, *tw_sin;  /* [n_samples] cos / sin (2 pi m / N)                                       */
    double fft_norm;                /* 4 / (N^2 mean(np.hanning(N)^2)), coder/psychoac.py:172-173                */
    double fft_freq_step;           /* rfftfreq(N, 1 / sampleRate)[1]                                            */
    const double *bark, *quiet;     /* [n_samples / 2] Bark value and threshold in quiet of the MDCT lines       */
    const int32_t *band_lower;      /* [n_bands] first line of a band                                            */
    const int32_t *band_lines;      /* [n_bands] its line count (> 0)                                            */
    int32_t n_bands;                /* 1 .. 32                                                                   */
} pacx_smr_tables;
/*
 *   data:   float64 [n_blocks][n_samples], the time block CalcSMRs gets (unwindowed);
 *   lines:  float64 [n_blocks][n_samples / 2], MDCTdata / 2^MDCTscale;
 *   smr:    float64 [n_blocks][n_bands]; threshold (optional): [n_blocks][n_samples / 2] dB SPL;
 *   n_peaks (optional): int32 [n_blocks].
 */
int pacx_smr_generic_batch(pacx_handle *h, int64_t n_blocks, int n_samples, const double *data,
                           const double *lines, const pacx_smr_tables *t, double *smr, double *threshold,
                           int32_t *n_peaks, void *stream);

/* ---- decode side (scalar-mantissa streams) ---------- */

/*
 * Inverse of pacx_pack_batch: parse the payload of every channel-block
 * (coder/pacfile.py:185-213, 264-266).  Payload i starts at
 * payload + offsets[i] (offsets != NULL, e.g. a .pac body) or at
 * payload + i*payload_stride.  Outputs use the layouts of pacx_encode_batch;
 * cf_flags: uint8 [n_cf] PACX_FLAG_* read from each payload.
 * status (optional): uint32 [n_cf], PACX_ST_MALFORMED for a channel-block whose
 * fields run past its n_bytes or carry an allocation above maxMantBits = 16 (its
 * other outputs are then zeros; nothing is read beyond n_bytes either way).
 */
int pacx_unpack_batch(pacx_handle *h, int64_t n_cf, const uint8_t *payload, int payload_stride,
                      const int64_t *offsets, const int32_t *n_bytes, uint8_t *cf_flags,
                      int32_t *overall_scale, int32_t *scale_factor, int32_t *bit_alloc,
                      int32_t *mantissa, uint32_t *status, void *stream);

/*
 * codec.Decode for a batch (coder/codec.py:47-92: vDequantize, / 2^overall, IMDCT,
 * window) and the overlap-and-add + 16-bit PCM mapping around it
 * (coder/pacfile.py:272-295, coder/pcmfile.py:127-134).  Blocks are in stream
 * order, n_blocks hops of n_channels channels.
 *   blocks: optional float64 [n_blocks*n_channels][2*n_lines_long], the windowed
 *           IMDCT output before overlap-and-add (what codec.Decode returns);
 *   pcm:    optional int16 [(n_blocks+1)*n_lines_long][n_channels] (interleaved):
 *           every hop plus the final half-block the reference flushes at EOF.
 */
int pacx_decode_batch(pacx_handle *h, int64_t n_blocks, int n_channels, const uint8_t *cf_flags,
                      const int32_t *overall_scale, const int32_t *scale_factor,
                      const int32_t *bit_alloc, const int32_t *mantissa, double *blocks,
                      int16_t *pcm, void *stream);

/*
 * The scalar-mantissa blocks of an SBR file (handle created with use_sbr and without use_vq).
 * pacx_decode_batch is codec.Decode for every block, whatever the handle.  This entry routes:
 *   every_long_block == 0: as PACFile.Decode does (coder/pacfile.py:645-668) -- a long block with
 *           bits in an omitted band is codec.Decode_SBR's, every other block codec.Decode's;
 *   every_long_block != 0: codec.Decode_SBR on every long block (the function itself).
 * Decode_SBR's scalar branch (coder/codec.py:95-222 with useVQ off, :117-134): one line per omitted
 * band, dequantised from the mantissa at THAT line of the line-indexed array -- pacx_unpack_batch
 * leaves a coded omitted band's single mantissa (coder/pacfile.py:203-205) on every line of the
 * band, as the reader does -- then the reconstruction of :136-198.  Optional outputs:
 *   lines:  float64 [n_cf][n_lines_long], the dequantised lines after the reconstruction and
 *           BEFORE the division by 2^overallScale (short frames: 8 x n_lines_short);
 *   status: uint32 [n_cf], PACX_ST_VQ_UNDEFINED where Decode_SBR raises IndexError (the cut lies
 *           in the lower half of the spectrum: band tables of rates above 48 kHz); that block
 *           decodes from its lines as they stood before the reconstruction.
 * No encoder of the reference writes a scalar block with a coded omitted band (it raises there:
 * PACX_ST_REF_RAISES); its reader and decoder take one, so this one does.  On a handle without
 * use_sbr this is pacx_decode_batch with the lines as an extra output.
 */
int pacx_decode_sbr_batch(pacx_handle *h, int64_t n_blocks, int n_channels, const uint8_t *cf_flags,
                          const int32_t *overall_scale, const int32_t *scale_factor,
                          const int32_t *bit_alloc, const int32_t *mantissa, int every_long_block,
                          double *lines, double *blocks, int16_t *pcm, uint32_t *status, void *stream);

/*
 * Decode of gain-shape coded channel-blocks (handle created with use_vq), from
 * payload to PCM:
 *   PACFile.ReadDataBlock / getDecodedBlock   coder/pacfile.py:177-298 (useVQ branch)
 *   PACFile.Decode routing                    coder/pacfile.py:645-668
 *   codec.Decode with useVQ                   coder/codec.py:47-92
 *   codec.Decode_SBR                          coder/codec.py:95-222 (incl. the SciPy
 *                                             gaussian_filter1d / interp1d calls)
 *   dequantize_gain_shape, split_band_decode, dequantize_pvq,
 *   decode_pvq_vector, inv_mu_law_fn          coder/gain_shape_quantize.py:127-176, 259-272,
 *                                             298-299, 411-473, 515-541
 * Payload addressing as pacx_unpack_batch.  Outputs: cf_flags uint8 [n_cf],
 * overall_scale int32 [n_cf][8], bit_alloc int32 [n_cf][band_stride], status
 * uint32 [n_cf]; optional lines float64 [n_cf][n_lines_long] (the MDCT lines
 * after SBR reconstruction, BEFORE the division by 2^overallScale), blocks and
 * pcm as pacx_decode_batch.
 */
int pacx_decode_vq_batch(pacx_handle *h, int64_t n_blocks, int n_channels, const uint8_t *payload,
                         int payload_stride, const int64_t *offsets, const int32_t *n_bytes,
                         uint8_t *cf_flags, int32_t *overall_scale, int32_t *bit_alloc, double *lines,
                         double *blocks, int16_t *pcm, uint32_t *status, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PACX_H */
