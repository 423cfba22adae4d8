/*
 * vdf_oracle.c -- CPU restatement of the vid_dup_finder_lib hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the timed
 * "port" CPU baseline.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product path (vid_dup_finder_lib_amd)
 * never links, imports or calls anything in oracle/.
 *
 * PARITY STATUS
 *   search / search_with_references / hamming_distance: pinned structurally
 *     by the reference's own tests (vid_dup_finder_lib/tests/test_find_all.rs:
 *     134-315, src/video_hashing/video_hash.rs:325-371,
 *     src/video_hashing/search_algorithm.rs:203-208), restated in
 *     tests/test_oracle_reference_scenarios.py.
 *   letterbox crop detection: pinned by the reference's known-answer tests
 *     (vid_dup_finder_common/src/video_frames_gray.rs:216-459), restated in
 *     tests/test_oracle_letterbox.py.
 *   hash bits (resize + 3-D DCT): PARITY UNPINNED.  The reference is Rust
 *     (no cargo/rustc here), its tests hold no known-answer hash vector, and
 *     the arithmetic lives in un-vendored crates (rustdct "0.7",
 *     fast_image_resize "5.1", no Cargo.lock).  Their published algorithms are
 *     restated below; the 16x16-input path (resize == copy) depends only on
 *     the DCT-II definition and is the exact contract.
 *
 * All file:line citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DCT_SIZE 16   /* vid_dup_finder_lib/src/definitions.rs:34 */
#define HASH_SIZE 10  /* definitions.rs:36 */
#define HASH_BITS 1000 /* definitions.rs:42 */
#define HASH_WORDS 16  /* definitions.rs:43 (usize = 64 bit) */

#define ORACLE_OK 0
#define ORACLE_E_NOT_ENOUGH_FRAMES (-1) /* Error::NotEnoughFrames, video_hashing/mod.rs:27 */
#define ORACLE_E_BAD_DIMS (-2)

/* Rust `x as u32` for f64: truncating, saturating, NaN -> 0. */
static uint32_t f64_as_u32(double x)
{
    if (!(x == x)) return 0;
    if (x <= 0.0) return 0;
    if (x >= 4294967295.0) return 4294967295u;
    return (uint32_t)x;
}

/* search_algorithm.rs:64,82  `(tolerance * TOLERANCE_SCALING_FACTOR) as u32`
 * with TOLERANCE_SCALING_FACTOR = 10^3 (definitions.rs:40). */
uint32_t vdf_oracle_tolerance_int(double tolerance)
{
    return f64_as_u32(tolerance * 1000.0);
}

/* video_hash.rs:311-317: sum of popcount(x^y) over ALL 16 words (padding bits
 * 1000..1023 included). */
uint32_t vdf_oracle_hamming(const uint64_t *x, const uint64_t *y)
{
    uint32_t acc = 0;
    for (int i = 0; i < HASH_WORDS; i++) acc += (uint32_t)__builtin_popcountll(x[i] ^ y[i]);
    return acc;
}

/* ------------------------------------------------------------------------
 * Resize: vid_dup_finder_common/src/resize_gray.rs:11-54 calls
 * fast_image_resize 5.1 `Resizer::new().resize(.., ResizeOptions::new().crop(..))`
 * = ResizeAlg::Convolution(FilterType::Lanczos3) on PixelType::U8.  The crate
 * source is not under /root/reference; this restates its published algorithm:
 *   - coefficient windows per output pixel (f64, normalised to sum 1),
 *   - i16 quantisation with the largest precision p such that
 *     round(max_w * 2^(p+1)) < 2^15,
 *   - horizontal pass into a u8 temporary (only the rows the vertical pass
 *     needs), then vertical pass,
 *   - out = clamp((2^(p-1) + sum pix*w) >> p, 0, 255),
 *   - equal source/destination size: plain copy.
 * ---------------------------------------------------------------------- */
static double sinc_filter(double x)
{
    if (x == 0.0) return 1.0;
    x *= M_PI;
    return sin(x) / x;
}

static double lanczos3(double x)
{
    if (x >= -3.0 && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3.0);
    return 0.0;
}

typedef struct {
    int out_size;
    int window;      /* taps allotted per output pixel */
    int precision;   /* fixed-point bits */
    int32_t *start;  /* [out_size] first source index */
    int32_t *size;   /* [out_size] number of taps */
    int16_t *w;      /* [out_size * window] */
} oracle_coeffs;

static void coeffs_free(oracle_coeffs *c)
{
    free(c->start); free(c->size); free(c->w);
    memset(c, 0, sizeof *c);
}

/* in0/in1: crop box along this axis (resize_gray.rs:37-46 passes the whole
 * image: Crop::from_edge_offsets(..,0,0,0,0), video_hash.rs:57). */
static int coeffs_build(oracle_coeffs *c, uint32_t in_size, double in0, double in1, uint32_t out_size)
{
    memset(c, 0, sizeof *c);
    double scale = (in1 - in0) / (double)out_size;
    if (in_size == 0 || out_size == 0 || scale <= 0.0) return -1;
    double filter_scale = scale > 1.0 ? scale : 1.0;
    double radius = 3.0 * filter_scale;
    int window = (int)ceil(radius) * 2 + 1;
    double recip = 1.0 / filter_scale;

    double *vals = (double *)calloc((size_t)window * out_size, sizeof(double));
    c->start = (int32_t *)calloc(out_size, sizeof(int32_t));
    c->size = (int32_t *)calloc(out_size, sizeof(int32_t));
    c->w = (int16_t *)calloc((size_t)window * out_size, sizeof(int16_t));
    c->out_size = (int)out_size;
    c->window = window;
    if (!vals || !c->start || !c->size || !c->w) { free(vals); coeffs_free(c); return -1; }

    double max_w = 0.0;
    int have_max = 0;
    for (uint32_t o = 0; o < out_size; o++) {
        double in_center = in0 + ((double)o + 0.5) * scale;
        double lo = floor(in_center - radius);
        if (lo < 0.0) lo = 0.0;
        double hi = ceil(in_center + radius);
        if (hi > (double)in_size) hi = (double)in_size;
        uint32_t x_min = (uint32_t)lo, x_max = (uint32_t)hi;
        double center = in_center - 0.5;
        double *k = vals + (size_t)o * window;
        int n = 0;
        double ww = 0.0;
        uint32_t bound_start = x_min, bound_end = x_max;
        for (uint32_t x = x_min; x < x_max; x++) {
            double w = lanczos3(((double)x - center) * recip);
            if (x == bound_start && w == 0.0) {
                bound_start++; /* drop leading zero taps */
            } else {
                k[n++] = w;
                ww += w;
            }
        }
        for (int i = n - 1; i >= 0; i--) { /* drop trailing zero taps */
            if (bound_end <= bound_start || k[i] != 0.0) break;
            bound_end--;
        }
        if (ww != 0.0)
            for (int i = 0; i < n; i++) k[i] /= ww;
        c->start[o] = (int32_t)bound_start;
        c->size[o] = (int32_t)(bound_end - bound_start);
    }
    /* max over every stored value, the zero padding of short windows included */
    for (size_t i = 0; i < (size_t)window * out_size; i++) {
        if (!have_max || vals[i] > max_w) { max_w = vals[i]; have_max = 1; }
    }
    int precision = 0;
    for (int p = 0; p < 16; p++) {
        precision = p;
        int32_t next = (int32_t)round(max_w * (double)(1 << (p + 1)));
        if (next >= (1 << 15)) break;
    }
    c->precision = precision;
    double q = (double)(1 << precision);
    for (size_t i = 0; i < (size_t)window * out_size; i++) c->w[i] = (int16_t)round(vals[i] * q);
    free(vals);
    return 0;
}

static uint8_t clip8(int32_t v, int precision)
{
    int32_t s = v >> precision; /* arithmetic shift */
    if (s < 0) s = 0;
    if (s > 255) s = 255;
    return (uint8_t)s;
}

/* One frame W x H (row-major u8) -> 16 x 16 (row-major u8). */
int vdf_oracle_resize_frame_u8(const uint8_t *src, uint32_t w, uint32_t h, uint8_t *dst)
{
    const uint32_t D = DCT_SIZE;
    if (w == 0 || h == 0) return ORACLE_E_BAD_DIMS;
    if (w == D && h == D) { memcpy(dst, src, D * D); return ORACLE_OK; }
    oracle_coeffs ch, cv;
    int need_h = (w != D), need_v = (h != D);
    if (need_h && coeffs_build(&ch, w, 0.0, (double)w, D)) return ORACLE_E_BAD_DIMS;
    if (need_v && coeffs_build(&cv, h, 0.0, (double)h, D)) { if (need_h) coeffs_free(&ch); return ORACLE_E_BAD_DIMS; }

    if (need_h && need_v) {
        int32_t y_first = cv.start[0];
        int32_t y_last = cv.start[D - 1] + cv.size[D - 1];
        int32_t th = y_last - y_first;
        uint8_t *tmp = (uint8_t *)malloc((size_t)th * D);
        int32_t init_h = 1 << (ch.precision - 1), init_v = 1 << (cv.precision - 1);
        for (int32_t y = 0; y < th; y++) {
            const uint8_t *row = src + (size_t)(y + y_first) * w;
            for (uint32_t o = 0; o < D; o++) {
                int32_t ss = init_h;
                const int16_t *k = ch.w + (size_t)o * ch.window;
                for (int32_t t = 0; t < ch.size[o]; t++) ss += (int32_t)row[ch.start[o] + t] * (int32_t)k[t];
                tmp[(size_t)y * D + o] = clip8(ss, ch.precision);
            }
        }
        for (uint32_t oy = 0; oy < D; oy++) {
            const int16_t *k = cv.w + (size_t)oy * cv.window;
            int32_t s0 = cv.start[oy] - y_first;
            for (uint32_t x = 0; x < D; x++) {
                int32_t ss = init_v;
                for (int32_t t = 0; t < cv.size[oy]; t++) ss += (int32_t)tmp[(size_t)(s0 + t) * D + x] * (int32_t)k[t];
                dst[oy * D + x] = clip8(ss, cv.precision);
            }
        }
        free(tmp);
    } else if (need_h) {
        int32_t init_h = 1 << (ch.precision - 1);
        for (uint32_t y = 0; y < D; y++)
            for (uint32_t o = 0; o < D; o++) {
                int32_t ss = init_h;
                const int16_t *k = ch.w + (size_t)o * ch.window;
                for (int32_t t = 0; t < ch.size[o]; t++) ss += (int32_t)src[(size_t)y * w + ch.start[o] + t] * (int32_t)k[t];
                dst[y * D + o] = clip8(ss, ch.precision);
            }
    } else {
        int32_t init_v = 1 << (cv.precision - 1);
        for (uint32_t oy = 0; oy < D; oy++) {
            const int16_t *k = cv.w + (size_t)oy * cv.window;
            for (uint32_t x = 0; x < D; x++) {
                int32_t ss = init_v;
                for (int32_t t = 0; t < cv.size[oy]; t++) ss += (int32_t)src[(size_t)(cv.start[oy] + t) * w + x] * (int32_t)k[t];
                dst[oy * D + x] = clip8(ss, cv.precision);
            }
        }
    }
    if (need_h) coeffs_free(&ch);
    if (need_v) coeffs_free(&cv);
    return ORACLE_OK;
}

/* Exposes the quantised coefficient table of one axis (for cross-checks against
 * the numpy twin and the device tables).  Returns the precision, or <0. */
int vdf_oracle_resize_coeffs(uint32_t in_size, uint32_t out_size, int32_t *start, int32_t *size,
                             int16_t *w, int32_t w_capacity, int32_t *window)
{
    oracle_coeffs c;
    if (coeffs_build(&c, in_size, 0.0, (double)in_size, out_size)) return -1;
    *window = c.window;
    if ((int64_t)c.window * out_size > w_capacity) { coeffs_free(&c); return -2; }
    memcpy(start, c.start, out_size * sizeof(int32_t));
    memcpy(size, c.size, out_size * sizeof(int32_t));
    memcpy(w, c.w, (size_t)c.window * out_size * sizeof(int16_t));
    int p = c.precision;
    coeffs_free(&c);
    return p;
}

/* ------------------------------------------------------------------------
 * 3-D DCT: video_hashing/dct_3d.rs:15-53 (cube fill, [frame][x][y], pix-128),
 * video_hashing/raw_dct_ops.rs:107-142 (pass along y, x, then t; result back
 * in [t][x][y] order).  rustdct "0.7" process_dct2 = unnormalised DCT-II:
 *   X[k] = sum_n x[n] * cos(pi * k * (n + 1/2) / N).
 * Only the sign of each coefficient is consumed (dct_3d.rs:55-62), so the
 * ORDER of the floating-point operations matters exactly where a coefficient
 * is mathematically zero: lines that are constant (static clips, black
 * frames) or mirror-symmetric.  DctPlanner::plan_dct2(16) returns
 * Type2And3Butterfly16, which - like Butterfly8 and Butterfly4 inside it - is
 * one step of the crate's split-radix algorithm (Type2And3SplitRadix): the
 * even outputs are the half-size DCT-II of the sums x[n] + x[N-1-n], the odd
 * outputs come from the differences x[n] - x[N-1-n], rotated by the twiddles
 * e^{i pi (2n+1) / 2N} into two quarter-size DCT-IIs.  Under that structure
 * every AC output of a constant line and every odd output of a symmetric
 * line is an exact +-0.0 (bit 0), which a direct cosine sum does not give.
 * The crate's source is not under /root/reference (un-vendored, no lockfile):
 * this restates its published algorithm; products and sums are rounded
 * separately as Rust does (build with -ffp-contract=off).  Twiddles as
 * rustdct::twiddles::single_twiddle(i, len).conj(): angle = (-2 pi / len) * i.
 * ---------------------------------------------------------------------- */
static double g_tw4[2], g_tw8[2][2], g_tw16[4][2]; /* (cos, sin) of pi (2 i + 1) / (2 N) */
static int g_tw_ready = 0;

static void twiddle(int i, int fft_len, double *out)
{
    const double angle_constant = M_PI * -2.0 / (double)fft_len;
    const double angle = angle_constant * (double)i;
    out[0] = cos(angle);
    out[1] = -sin(angle); /* .conj() */
}

static void cos_init(void)
{
    if (g_tw_ready) return;
    twiddle(1, 16, g_tw4);
    for (int i = 0; i < 2; i++) twiddle(2 * i + 1, 32, g_tw8[i]);
    for (int i = 0; i < 4; i++) twiddle(2 * i + 1, 64, g_tw16[i]);
    g_tw_ready = 1;
}

static void dct2_len2(double *b)
{
    const double sum = b[0] + b[1];
    b[1] = (b[0] - b[1]) * M_SQRT1_2; /* f64::consts::FRAC_1_SQRT_2 */
    b[0] = sum;
}

static void dct2_len4(double *b)
{
    double in2[2] = {b[3] + b[0], b[1] + b[2]};
    const double lower = b[0] - b[3], upper = b[1] - b[2];
    const double cos_in = lower * g_tw4[0] + upper * g_tw4[1];
    const double sin_in = upper * g_tw4[0] - lower * g_tw4[1];
    dct2_len2(in2);
    b[0] = in2[0];
    b[1] = cos_in; /* the quarter-size transforms have length 1 */
    b[2] = in2[1];
    b[3] = -sin_in;
}

static void dct2_len8(double *b)
{
    double in2[4], ev[2], od[2];
    for (int i = 0; i < 2; i++) {
        const double bottom = b[i], top = b[7 - i], hb = b[3 - i], ht = b[4 + i];
        in2[i] = top + bottom;
        in2[3 - i] = hb + ht;
        const double lower = bottom - top, upper = hb - ht;
        const double cos_in = lower * g_tw8[i][0] + upper * g_tw8[i][1];
        const double sin_in = upper * g_tw8[i][0] - lower * g_tw8[i][1];
        ev[i] = cos_in;
        od[1 - i] = (i % 2 == 0) ? sin_in : -sin_in;
    }
    dct2_len4(in2);
    dct2_len2(ev);
    dct2_len2(od);
    b[0] = in2[0];
    b[1] = ev[0];
    b[2] = in2[1];
    { /* i = 1, quarter_len = 2: (i + quarter_len) odd */
        const double c = ev[1], sv = od[1];
        b[3] = c + sv;
        b[4] = in2[2];
        b[5] = c - sv;
        b[6] = in2[3];
    }
    b[7] = -od[0];
}

static void dct2_len16(double *b)
{
    double in2[8], ev[4], od[4];
    for (int i = 0; i < 4; i++) {
        const double bottom = b[i], top = b[15 - i], hb = b[7 - i], ht = b[8 + i];
        in2[i] = top + bottom;
        in2[7 - i] = hb + ht;
        const double lower = bottom - top, upper = hb - ht;
        const double cos_in = lower * g_tw16[i][0] + upper * g_tw16[i][1];
        const double sin_in = upper * g_tw16[i][0] - lower * g_tw16[i][1];
        ev[i] = cos_in;
        od[3 - i] = (i % 2 == 0) ? sin_in : -sin_in;
    }
    dct2_len8(in2);
    dct2_len4(ev);
    dct2_len4(od);
    b[0] = in2[0];
    b[1] = ev[0];
    b[2] = in2[1];
    for (int i = 1; i < 4; i++) {
        const double c = ev[i];
        const double sv = ((i + 4) % 2 == 0) ? -od[4 - i] : od[4 - i];
        b[i * 4 - 1] = c + sv;
        b[i * 4] = in2[i * 2];
        b[i * 4 + 1] = c - sv;
        b[i * 4 + 2] = in2[i * 2 + 1];
    }
    b[15] = -od[0];
}

static void dct16_line(double *x, int stride)
{
    double b[DCT_SIZE];
    for (int n = 0; n < DCT_SIZE; n++) b[n] = x[n * stride];
    dct2_len16(b);
    for (int k = 0; k < DCT_SIZE; k++) x[k * stride] = b[k];
}

/* one line, exposed for the tests */
void vdf_oracle_dct16(double *line)
{
    cos_init();
    dct16_line(line, 1);
}

/* cube[t][x][y], in place. */
void vdf_oracle_dct3d(double *cube)
{
    cos_init();
    const int N = DCT_SIZE;
    for (int t = 0; t < N; t++) /* raw_dct_ops.rs:118-120: last axis (y) */
        for (int x = 0; x < N; x++) dct16_line(cube + (t * N + x) * N, 1);
    for (int t = 0; t < N; t++) /* :123-126: after swap(2,1) -> along x */
        for (int y = 0; y < N; y++) dct16_line(cube + t * N * N + y, N);
    for (int x = 0; x < N; x++) /* :129-132: after swap(2,0) -> along t */
        for (int y = 0; y < N; y++) dct16_line(cube + x * N + y, N * N);
}

/* frames16: n_frames x 16(rows y) x 16(cols x) u8, already 16x16.
 * out_hash: 16 words.  out_coefs (nullable): 1000 f64 in bit order.
 * dct_3d.rs:25 take(16); :47-52 fewer than 16 -> None -> NotEnoughFrames. */
int vdf_oracle_hash_frames16(const uint8_t *frames16, uint32_t n_frames, uint64_t *out_hash, double *out_coefs)
{
    if (n_frames < DCT_SIZE) return ORACLE_E_NOT_ENOUGH_FRAMES;
    const int N = DCT_SIZE;
    static __thread double cube[DCT_SIZE * DCT_SIZE * DCT_SIZE];
    for (int t = 0; t < N; t++)
        for (int y = 0; y < N; y++)
            for (int x = 0; x < N; x++) /* dct_3d.rs:40-44: m[frame][col][row] = pix - 128 */
                cube[(t * N + x) * N + y] = (double)frames16[(t * N + y) * N + x] - 128.0;
    vdf_oracle_dct3d(cube);
    memset(out_hash, 0, HASH_WORDS * sizeof(uint64_t));
    int bit = 0;
    for (int kt = 0; kt < HASH_SIZE; kt++) /* dct_3d.rs:55-66: [..10,..10,..10] logical order */
        for (int kx = 0; kx < HASH_SIZE; kx++)
            for (int ky = 0; ky < HASH_SIZE; ky++, bit++) {
                double c = cube[(kt * N + kx) * N + ky];
                if (out_coefs) out_coefs[bit] = c;
                if (c > 0.0) out_hash[bit >> 6] |= (uint64_t)1 << (bit & 63); /* Lsb0, video_hash.rs:64-68 */
            }
    return ORACLE_OK;
}

/* video_hash.rs:45-73 from_frames for one clip: frames = n_frames x H x W u8
 * with given strides (bytes).  First 16 frames are used. */
int vdf_oracle_hash_clip(const uint8_t *frames, uint32_t n_frames, uint32_t w, uint32_t h, size_t frame_stride,
                         uint64_t *out_hash, double *out_coefs)
{
    if (n_frames == 0) return ORACLE_E_NOT_ENOUGH_FRAMES; /* video_hash.rs:53 */
    if (w == 0 || h == 0) return ORACLE_E_BAD_DIMS;
    if (n_frames < DCT_SIZE) return ORACLE_E_NOT_ENOUGH_FRAMES;
    uint8_t small[DCT_SIZE * DCT_SIZE * DCT_SIZE];
    for (int t = 0; t < DCT_SIZE; t++) {
        int rc = vdf_oracle_resize_frame_u8(frames + (size_t)t * frame_stride, w, h, small + t * DCT_SIZE * DCT_SIZE);
        if (rc) return rc;
    }
    return vdf_oracle_hash_frames16(small, DCT_SIZE, out_hash, out_coefs);
}

/* Batch form used by the CPU baseline (single thread per call; bench.py fans
 * calls out over a thread pool the way the app's rayon par_bridge does,
 * vid_dup_finder_app/src/video_hash_filesystem_cache/video_hash_filesystem_cache.rs:246). */
int vdf_oracle_hash_clips(const uint8_t *frames, size_t n_clips, uint32_t frames_per_clip, uint32_t w, uint32_t h,
                          uint64_t *out_hashes)
{
    size_t fs = (size_t)w * h, cs = fs * frames_per_clip;
    for (size_t c = 0; c < n_clips; c++) {
        int rc = vdf_oracle_hash_clip(frames + c * cs, frames_per_clip, w, h, fs, out_hashes + c * HASH_WORDS, NULL);
        if (rc) return rc;
    }
    return ORACLE_OK;
}

/* ------------------------------------------------------------------------
 * Search.  Inputs are already in the order Search::sort leaves them
 * (search_algorithm.rs:55-61: stable by (duration, src_path)); sorting needs
 * paths and is restated in oracle/vdf_oracle.py.
 * Output CSR: offsets[n_groups+1], members[]; both caller-allocated:
 *   self-search: offsets capacity n/2+2, members capacity n.
 * ---------------------------------------------------------------------- */

/* search_algorithm.rs:81-171, literal: monotone rhs, matched flags, consumption. */
int64_t vdf_oracle_search_self(const uint64_t *hashes, const uint32_t *dur, size_t n, uint32_t tol_int,
                               uint64_t *offsets, uint64_t *members)
{
    offsets[0] = 0;
    if (n == 0) return 0; /* :89-91 */
    uint8_t *matched = (uint8_t *)calloc(n, 1);
    size_t lhs = 0, rhs = 0, n_groups = 0, n_members = 0;
    for (;;) {
        /* advance_rhs :93-117 */
        uint32_t thresh = f64_as_u32((double)dur[lhs] * 1.1);
        while (rhs < n) {
            if (matched[rhs]) rhs++;
            else if (dur[rhs] > thresh) break;
            else rhs++;
        }
        if (lhs < rhs) { /* :138-162 */
            matched[lhs] = 1;
            size_t first = n_members;
            const uint64_t *target = hashes + lhs * HASH_WORDS;
            for (size_t cand = lhs + 1; cand < rhs; cand++) {
                if (!matched[cand] && vdf_oracle_hamming(target, hashes + cand * HASH_WORDS) <= tol_int) {
                    members[n_members++] = cand;
                    matched[cand] = 1;
                }
            }
            if (n_members != first) {
                members[n_members++] = lhs; /* target pushed last, :159 */
                offsets[++n_groups] = n_members;
            }
        }
        /* advance_lhs :119-129 */
        do { lhs++; } while (lhs < n && matched[lhs]);
        if (lhs >= n) break;
    }
    free(matched);
    /* ret.reverse() :167 -- reverse group order, keep member order */
    uint64_t *tmp_m = (uint64_t *)malloc((n_members + 1) * sizeof(uint64_t));
    uint64_t *tmp_o = (uint64_t *)malloc((n_groups + 1) * sizeof(uint64_t));
    size_t pos = 0;
    tmp_o[0] = 0;
    for (size_t g = 0; g < n_groups; g++) {
        size_t src = n_groups - 1 - g;
        size_t len = offsets[src + 1] - offsets[src];
        memcpy(tmp_m + pos, members + offsets[src], len * sizeof(uint64_t));
        pos += len;
        tmp_o[g + 1] = pos;
    }
    memcpy(members, tmp_m, n_members * sizeof(uint64_t));
    memcpy(offsets, tmp_o, (n_groups + 1) * sizeof(uint64_t));
    free(tmp_m); free(tmp_o);
    return (int64_t)n_groups;
}

static size_t partition_point_lt(const uint32_t *dur, size_t n, uint32_t v)
{ /* first index with !(dur < v) */
    size_t lo = 0, hi = n;
    while (lo < hi) { size_t mid = lo + (hi - lo) / 2; if (dur[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}
static size_t partition_point_le(const uint32_t *dur, size_t n, uint32_t v)
{ /* first index with !(dur <= v) */
    size_t lo = 0, hi = n;
    while (lo < hi) { size_t mid = lo + (hi - lo) / 2; if (dur[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo;
}

/* video_dup_finder.rs:19-46 + search_algorithm.rs:63-77,173-185.
 * Refs in input order, consume=false.  Groups only for refs with >=1 hit;
 * ref_index[g] = position of the reference in the input.  If members == NULL
 * only counts (returns n_groups, *total_members set).  */
int64_t vdf_oracle_search_refs(const uint64_t *cand_hashes, const uint32_t *cand_dur, size_t n_cand,
                               const uint64_t *ref_hashes, const uint32_t *ref_dur, size_t n_ref, uint32_t tol_int,
                               uint64_t *offsets, uint64_t *members, int64_t *ref_index, uint64_t *total_members)
{
    size_t n_groups = 0, n_members = 0;
    if (offsets) offsets[0] = 0;
    for (size_t r = 0; r < n_ref; r++) {
        uint32_t lo_d = f64_as_u32((double)ref_dur[r] * 0.95); /* :174 */
        uint32_t hi_d = f64_as_u32((double)ref_dur[r] * 1.05); /* :179 */
        size_t lhs = partition_point_lt(cand_dur, n_cand, lo_d);
        size_t rhs = partition_point_le(cand_dur, n_cand, hi_d);
        size_t first = n_members;
        const uint64_t *target = ref_hashes + r * HASH_WORDS;
        for (size_t e = lhs; e < rhs; e++) {
            if (vdf_oracle_hamming(target, cand_hashes + e * HASH_WORDS) <= tol_int) {
                if (members) members[n_members] = e;
                n_members++;
            }
        }
        if (n_members != first) {
            if (ref_index) ref_index[n_groups] = (int64_t)r;
            n_groups++;
            if (offsets) offsets[n_groups] = n_members;
        }
    }
    if (total_members) *total_members = n_members;
    return (int64_t)n_groups;
}

/* Number of (target, candidate) comparisons the reference's windows admit when
 * nothing is consumed (SURVEY.md section 8d "pairs"). */
uint64_t vdf_oracle_pairs_self(const uint32_t *dur, size_t n)
{
    uint64_t pairs = 0;
    size_t rhs = 0;
    for (size_t i = 0; i < n; i++) {
        uint32_t thresh = f64_as_u32((double)dur[i] * 1.1);
        if (rhs < i + 1) rhs = i + 1;
        while (rhs < n && dur[rhs] <= thresh) rhs++;
        pairs += rhs - (i + 1);
    }
    return pairs;
}

/* ------------------------------------------------------------------------
 * Letterbox crop detection (the step right before the path).
 * vid_dup_finder_common/src/video_frames_gray.rs:38-128 (letterbox_crop with
 * LetterboxColour::AnyColour(tol)) and :201-210 (cropdetect_letterbox: frames
 * 0, 8, 16, .. at most 8 of them, crops united = per-edge minimum,
 * crop.rs:53-68).  Pinned by the reference's own known-answer tests
 * (video_frames_gray.rs:216-459), restated in tests/test_oracle_letterbox.py.
 * ---------------------------------------------------------------------- */
static int strip_is_letterbox(const uint8_t *p, size_t step, uint32_t len, uint32_t tol)
{
    uint32_t hist[256];
    memset(hist, 0, sizeof hist);
    for (uint32_t i = 0; i < len; i++) hist[p[(size_t)i * step]]++;
    uint32_t mode = 0, best = 0;
    for (uint32_t v = 0; v < 256; v++) /* Iterator::max_by_key keeps the LAST maximum */
        if (hist[v] >= best) { best = hist[v]; mode = v; }
    uint32_t lo = mode > tol ? mode - tol : 0, hi = mode + tol > 255 ? 255 : mode + tol, count = 0;
    for (uint32_t v = lo; v <= hi; v++) count += hist[v];
    return (double)count / (double)len > 0.9; /* min_proportion, :66,95-98 */
}

/* out = {left, right, top, bottom}; pitch in bytes. */
void vdf_oracle_letterbox_crop(const uint8_t *frame, uint32_t w, uint32_t h, size_t pitch, uint32_t tol, uint32_t *out)
{
    uint32_t l = 0, r = 0, t = 0, b = 0;
    while (l < w && strip_is_letterbox(frame + l, pitch, h, tol)) l++;
    while (r < w && strip_is_letterbox(frame + (w - r - 1), pitch, h, tol)) r++;
    while (t < h && strip_is_letterbox(frame + (size_t)t * pitch, 1, w, tol)) t++;
    while (b < h && strip_is_letterbox(frame + (size_t)(h - b - 1) * pitch, 1, w, tol)) b++;
    if ((int64_t)w - l - r >= 1 && (int64_t)h - t - b >= 1) { /* :119-127 */
        out[0] = l; out[1] = r; out[2] = t; out[3] = b;
    } else {
        out[0] = out[1] = out[2] = out[3] = 0;
    }
}

/* cropdetect_letterbox over a clip: frames 0, 8, ... (step_by(8).take(8)), tol 16.  Returns 0, or
 * ORACLE_E_NOT_ENOUGH_FRAMES for an empty clip (detect_crop -> None, video_hash_builder.rs:195). */
int vdf_oracle_cropdetect_letterbox(const uint8_t *frames, uint32_t n_frames, uint32_t w, uint32_t h,
                                    size_t frame_stride, uint32_t *out)
{
    if (n_frames == 0) return ORACLE_E_NOT_ENOUGH_FRAMES;
    uint32_t acc[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    uint32_t used = 0;
    for (uint32_t f = 0; f < n_frames && used < 8; f += 8, used++) {
        uint32_t c[4];
        vdf_oracle_letterbox_crop(frames + (size_t)f * frame_stride, w, h, w, 16, c);
        for (int k = 0; k < 4; k++) acc[k] = c[k] < acc[k] ? c[k] : acc[k];
    }
    memcpy(out, acc, sizeof acc);
    return ORACLE_OK;
}

/* crop_video_frames + from_frames (video_hash_builder.rs:188-204,222): detect, copy the crop box out of every
 * frame, hash the cropped frames. */
int vdf_oracle_hash_clip_letterbox(const uint8_t *frames, uint32_t n_frames, uint32_t w, uint32_t h, size_t frame_stride,
                                   uint64_t *out_hash, double *out_coefs, uint32_t *out_crop)
{
    uint32_t c[4];
    int rc = vdf_oracle_cropdetect_letterbox(frames, n_frames, w, h, frame_stride, c);
    if (rc) return rc;
    if (out_crop) memcpy(out_crop, c, sizeof c);
    const uint32_t cw = w - c[0] - c[1], ch = h - c[2] - c[3];
    const uint32_t nf = n_frames < DCT_SIZE ? n_frames : DCT_SIZE;
    uint8_t *buf = (uint8_t *)malloc((size_t)cw * ch * (nf ? nf : 1));
    for (uint32_t f = 0; f < nf; f++)
        for (uint32_t y = 0; y < ch; y++)
            memcpy(buf + ((size_t)f * ch + y) * cw, frames + (size_t)f * frame_stride + (size_t)(y + c[2]) * w + c[0], cw);
    rc = vdf_oracle_hash_clip(buf, nf, cw, ch, (size_t)cw * ch, out_hash, out_coefs);
    free(buf);
    return rc;
}
