/*
 * spectro_oracle.c -- CPU restatement of the spectrogram-rs hot path.  TEST INFRASTRUCTURE ONLY.
 * See spectro_oracle.h for scope, citations and the "PARITY UNPINNED" statement.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction, so every f32
 * operation rounds exactly once, as rustc's default code generation does).
 *
 * Citations are file:line in /root/reference (the reference is NOT needed at build or run time).
 */
#define _GNU_SOURCE
#include "spectro_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* sizes                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* Rust `f as usize`: truncation toward zero, saturating, NaN -> 0 */
static size_t f32_as_usize(float v)
{
    if (!(v > 0.0f)) return 0; /* also NaN */
    if (v >= 18446744073709551616.0f) return (size_t)-1;
    return (size_t)v;
}
static size_t f64_as_usize(double v)
{
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551616.0) return (size_t)-1;
    return (size_t)v;
}

size_t orc_window_samples(float sample_rate, float period)
{
    /* fft.rs:19, :41 -- (period * sample_rate) as usize */
    volatile float prod = period * sample_rate;
    return f32_as_usize(prod);
}

size_t orc_hop_samples(float sample_rate, float stride)
{
    /* audio_transform.rs:35 -- (self.stride * self.transform.sample_rate()) as usize */
    volatile float prod = stride * sample_rate;
    return f32_as_usize(prod);
}

size_t orc_num_frames(size_t n, size_t W, size_t H)
{
    /* audio_transform.rs:37-41: a frame is produced while at least W samples can be peeked,
     * then H are skipped.  Batch form (quirk Q1 not reproduced). */
    if (W == 0 || H == 0 || n < W) return 0;
    return (n - W) / H + 1;
}

/* ------------------------------------------------------------------------------------------ */
/* FFT: forward, unnormalised, F[k] = sum_n z[n] e^{-2 pi i n k / P}                          */
/* [third-party] fftw 0.8.0 -> FFTW3 fftwf_plan_dft_1d(P, FORWARD, MEASURE) (fft.rs:20-24,77). */
/* FFTW's plan is chosen by timing, so its rounding is not reproducible; what is restated is  */
/* the DFT it computes, with a plain mixed-radix decimation-in-time recursion in the same     */
/* working precision (float, twiddles rounded from double) -- and in double for ORC_F64.      */
/* ------------------------------------------------------------------------------------------ */

typedef struct { float re, im; } cf;
typedef struct { double re, im; } cd;

typedef struct {
    size_t P;
    cf *twf; /* twf[j] = e^{-2 pi i j / P}, rounded from double */
    cd *twd;
} plan_t;

static size_t smallest_factor(size_t n)
{
    if ((n & 1) == 0) return 2;
    for (size_t p = 3; p * p <= n; p += 2)
        if (n % p == 0) return p;
    return n;
}

static plan_t *plan_new(size_t P)
{
    plan_t *pl = (plan_t *)malloc(sizeof(plan_t));
    pl->P = P;
    pl->twf = (cf *)malloc(sizeof(cf) * P);
    pl->twd = (cd *)malloc(sizeof(cd) * P);
    for (size_t j = 0; j < P; ++j) {
        /* exact octant reduction is not needed at these sizes: cos/sin of a double angle */
        double ang = -2.0 * M_PI * (double)j / (double)P;
        double c = cos(ang), s = sin(ang);
        /* make the axis values exact */
        if (4 * j == P) { c = 0.0; s = -1.0; }
        if (2 * j == P) { c = -1.0; s = 0.0; }
        if (4 * j == 3 * P) { c = 0.0; s = 1.0; }
        pl->twd[j].re = c; pl->twd[j].im = s;
        pl->twf[j].re = (float)c; pl->twf[j].im = (float)s;
    }
    return pl;
}

#define PLAN_CACHE 8
static plan_t *g_plans[PLAN_CACHE];
static pthread_mutex_t g_plan_lock = PTHREAD_MUTEX_INITIALIZER;

static const plan_t *plan_get(size_t P)
{
    pthread_mutex_lock(&g_plan_lock);
    int slot = -1;
    for (int i = 0; i < PLAN_CACHE; ++i) {
        if (g_plans[i] && g_plans[i]->P == P) { plan_t *p = g_plans[i]; pthread_mutex_unlock(&g_plan_lock); return p; }
        if (!g_plans[i] && slot < 0) slot = i;
    }
    plan_t *pl = plan_new(P);
    if (slot < 0) slot = 0; /* leak the evicted plan deliberately: another thread may hold it */
    g_plans[slot] = pl;
    pthread_mutex_unlock(&g_plan_lock);
    return pl;
}

static inline cf cf_mul(cf a, cf b)
{
    cf r;
    r.re = a.re * b.re - a.im * b.im;
    r.im = a.re * b.im + a.im * b.re;
    return r;
}
static inline cd cd_mul(cd a, cd b)
{
    cd r;
    r.re = a.re * b.re - a.im * b.im;
    r.im = a.re * b.im + a.im * b.re;
    return r;
}

/* out[0..n) = DFT_n(in[0], in[istride], ...); tw[j*ts] = w_n^j */
static void fft_rec_f(const cf *in, size_t istride, cf *out, size_t n, const cf *tw, size_t ts)
{
    if (n == 1) { out[0] = in[0]; return; }
    size_t p = smallest_factor(n), m = n / p;
    for (size_t q = 0; q < p; ++q) fft_rec_f(in + q * istride, istride * p, out + q * m, m, tw, ts * p);
    if (p == 2) {
        for (size_t k = 0; k < m; ++k) {
            cf t = cf_mul(out[m + k], tw[k * ts]);
            cf a = out[k];
            out[k].re = a.re + t.re; out[k].im = a.im + t.im;
            out[k + m].re = a.re - t.re; out[k + m].im = a.im - t.im;
        }
        return;
    }
    cf ybuf[64];
    cf *y = p <= 64 ? ybuf : (cf *)malloc(sizeof(cf) * p);
    for (size_t k = 0; k < m; ++k) {
        for (size_t q = 0; q < p; ++q) y[q] = q ? cf_mul(out[q * m + k], tw[(q * k) * ts]) : out[k];
        for (size_t r = 0; r < p; ++r) {
            cf acc = y[0];
            for (size_t q = 1; q < p; ++q) {
                cf t = cf_mul(y[q], tw[((q * r * m) % n) * ts]);
                acc.re += t.re; acc.im += t.im;
            }
            out[k + r * m] = acc;
        }
    }
    if (y != ybuf) free(y);
}

static void fft_rec_d(const cd *in, size_t istride, cd *out, size_t n, const cd *tw, size_t ts)
{
    if (n == 1) { out[0] = in[0]; return; }
    size_t p = smallest_factor(n), m = n / p;
    for (size_t q = 0; q < p; ++q) fft_rec_d(in + q * istride, istride * p, out + q * m, m, tw, ts * p);
    if (p == 2) {
        for (size_t k = 0; k < m; ++k) {
            cd t = cd_mul(out[m + k], tw[k * ts]);
            cd a = out[k];
            out[k].re = a.re + t.re; out[k].im = a.im + t.im;
            out[k + m].re = a.re - t.re; out[k + m].im = a.im - t.im;
        }
        return;
    }
    cd ybuf[64];
    cd *y = p <= 64 ? ybuf : (cd *)malloc(sizeof(cd) * p);
    for (size_t k = 0; k < m; ++k) {
        for (size_t q = 0; q < p; ++q) y[q] = q ? cd_mul(out[q * m + k], tw[(q * k) * ts]) : out[k];
        for (size_t r = 0; r < p; ++r) {
            cd acc = y[0];
            for (size_t q = 1; q < p; ++q) {
                cd t = cd_mul(y[q], tw[((q * r * m) % n) * ts]);
                acc.re += t.re; acc.im += t.im;
            }
            out[k + r * m] = acc;
        }
    }
    if (y != ybuf) free(y);
}

/* ------------------------------------------------------------------------------------------ */
/* FastFourierTransform::process  (fft.rs:43-99)                                              */
/* ------------------------------------------------------------------------------------------ */

void orc_hann_window(size_t W, float *out)
{
    /* fft.rs:61: 0.5 * (1.0 - ((f32::TAU() * i as f32) / (W as f32)).cos())
     * TAU_f32 = 6.2831855; product rounded to f32 BEFORE the divide (quirk Q6). */
    const float tau = 6.28318530717958647692528676655900577f;
    const float wf = (float)W;
    for (size_t i = 0; i < W; ++i) {
        volatile float prod = tau * (float)i;
        volatile float q = prod / wf;
        float c = cosf(q);
        volatile float om = 1.0f - c;
        out[i] = 0.5f * om;
    }
}

/* work buffers for one frame */
typedef struct {
    cf *zf, *Ff;
    cd *zd, *Fd;
    float *win;
} frame_ws;

static frame_ws *ws_new(size_t W)
{
    frame_ws *ws = (frame_ws *)malloc(sizeof(frame_ws));
    size_t P = 2 * W;
    ws->zf = (cf *)malloc(sizeof(cf) * P);
    ws->Ff = (cf *)malloc(sizeof(cf) * P);
    ws->zd = (cd *)malloc(sizeof(cd) * P);
    ws->Fd = (cd *)malloc(sizeof(cd) * P);
    ws->win = (float *)malloc(sizeof(float) * (W ? W : 1));
    orc_hann_window(W, ws->win);
    return ws;
}
static void ws_free(frame_ws *ws)
{
    free(ws->zf); free(ws->Ff); free(ws->zd); free(ws->Fd); free(ws->win); free(ws);
}

/* l, r: pointers to the first left / right sample, with the given element strides */
static void process_frame(const plan_t *pl, frame_ws *ws, const float *l, size_t lstride, const float *r,
                          size_t rstride, size_t W, int precision, float *out, double *out64)
{
    const size_t P = 2 * W, M = W - 1;
    /* fft.rs:48-69: pack (l, r) -> l + i r; multiply by the Hann factor (complex * real);
     * pad with zeros to 2W; window occupies [0, W). */
    for (size_t i = 0; i < W; ++i) {
        float s = ws->win[i];
        ws->zf[i].re = l[i * lstride] * s;
        ws->zf[i].im = r[i * rstride] * s;
    }
    for (size_t i = W; i < P; ++i) { ws->zf[i].re = 0.0f; ws->zf[i].im = 0.0f; }

    if (precision == ORC_F32) {
        /* fft.rs:76-77 */
        fft_rec_f(ws->zf, 1, ws->Ff, P, pl->twf, 1);
        /* fft.rs:81-98: a = F[k], b = F[P-k], k = 1..M
         *   left  = norm(a + conj(b)) / 2.0 ; right = norm(a - conj(b)) / 2.0   (norm = hypotf)
         *   then * (2.0 / W as f32) */
        const float scale = 2.0f / (float)W;
        for (size_t j = 0; j < M; ++j) {
            size_t k = j + 1;
            cf a = ws->Ff[k], b = ws->Ff[P - k];
            volatile float sre = a.re + b.re, sim = a.im - b.im; /* a + conj(b) */
            volatile float dre = a.re - b.re, dim = a.im + b.im; /* a - conj(b) */
            volatile float left = hypotf(sre, sim) / 2.0f;
            volatile float right = hypotf(dre, dim) / 2.0f;
            out[2 * j + 0] = left * scale;
            out[2 * j + 1] = right * scale;
        }
    } else {
        for (size_t i = 0; i < P; ++i) { ws->zd[i].re = ws->zf[i].re; ws->zd[i].im = ws->zf[i].im; }
        fft_rec_d(ws->zd, 1, ws->Fd, P, pl->twd, 1);
        const double scale = 2.0 / (double)W;
        for (size_t j = 0; j < M; ++j) {
            size_t k = j + 1;
            cd a = ws->Fd[k], b = ws->Fd[P - k];
            double left = hypot(a.re + b.re, a.im - b.im) / 2.0 * scale;
            double right = hypot(a.re - b.re, a.im + b.im) / 2.0 * scale;
            if (out) { out[2 * j + 0] = (float)left; out[2 * j + 1] = (float)right; }
            if (out64) { out64[2 * j + 0] = left; out64[2 * j + 1] = right; }
        }
    }
}

int orc_fft_process(const float *lr, size_t n_avail, size_t W, int precision, float *out, double *out64)
{
    /* fft.rs:72: fewer than W samples -> None */
    if (W < 2 || n_avail < W) return 0;
    const plan_t *pl = plan_get(2 * W);
    frame_ws *ws = ws_new(W);
    process_frame(pl, ws, lr, 2, lr + 1, 2, W, precision, out, out64);
    ws_free(ws);
    return 1;
}

typedef struct {
    const float *pcm; size_t n; int channels; size_t W, H, first, count; int precision; float *out;
    size_t begin, end; const plan_t *pl;
} stream_job;

static void *stream_worker(void *arg)
{
    stream_job *jb = (stream_job *)arg;
    const size_t W = jb->W, H = jb->H, M = W - 1;
    const int C = jb->channels;
    const int pairs = C == 1 ? 1 : C / 2;
    frame_ws *ws = ws_new(W);
    for (size_t i = jb->begin; i < jb->end; ++i) {
        size_t t = jb->first + i;
        const float *base = jb->pcm + (t * H) * (size_t)C;
        for (int p = 0; p < pairs; ++p) {
            float *o = jb->out + ((i * (size_t)pairs + (size_t)p) * M) * 2;
            if (C == 1) /* audio_input_list_model.rs:67-69: mono -> (s, s) */
                process_frame(jb->pl, ws, base, 1, base, 1, W, jb->precision, o, NULL);
            else
                process_frame(jb->pl, ws, base + 2 * p, (size_t)C, base + 2 * p + 1, (size_t)C, W, jb->precision, o, NULL);
        }
    }
    ws_free(ws);
    return NULL;
}

size_t orc_stream_process(const float *pcm, size_t n, int channels, size_t W, size_t H, size_t first,
                          size_t count, int precision, int threads, float *out)
{
    if (channels < 1 || (channels > 1 && (channels & 1)) || W < 2 || H < 1) return 0;
    size_t total = orc_num_frames(n, W, H);
    if (first >= total) return 0;
    if (count > total - first) count = total - first;
    if (threads < 1) threads = 1;
    if ((size_t)threads > count) threads = (int)(count ? count : 1);
    const plan_t *pl = plan_get(2 * W);
    stream_job *jobs = (stream_job *)calloc((size_t)threads, sizeof(stream_job));
    pthread_t *tids = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    for (int i = 0; i < threads; ++i) {
        stream_job *jb = &jobs[i];
        jb->pcm = pcm; jb->n = n; jb->channels = channels; jb->W = W; jb->H = H; jb->first = first; jb->count = count;
        jb->precision = precision; jb->out = out; jb->pl = pl;
        jb->begin = count * (size_t)i / (size_t)threads;
        jb->end = count * (size_t)(i + 1) / (size_t)threads;
        if (threads == 1) stream_worker(jb);
        else pthread_create(&tids[i], NULL, stream_worker, jb);
    }
    if (threads > 1) for (int i = 0; i < threads; ++i) pthread_join(tids[i], NULL);
    free(jobs); free(tids);
    return count;
}

/* ------------------------------------------------------------------------------------------ */
/* The same frame loop through the REAL FFTW, if the host has one (bench.py's cpu_baseline leg)  */
/* ------------------------------------------------------------------------------------------ */
/* fft.rs:20-24,68,76-77 call FFTW3 (single precision) through the fftw crate.  No libfftw3f ships in this image, so
 * nothing here is linked: the library is looked up at run time (dlopen "libfftw3f.so.3") and the function says so when
 * it is absent.  as_written != 0 keeps what fft.rs does per frame -- two fftwf_malloc'ed zero-filled buffers
 * (AlignedVec::new, :68,76), the Hann factor through cosf for every sample (:61) -- and 0 hoists both out of the loop
 * ("as written" and "best case").  Plan: fftwf_plan_dft_1d(2W, FORWARD, FFTW_MEASURE), made once (:20-24). */
#include <dlfcn.h>

typedef void *(*fftwf_plan_dft_1d_t)(int, void *, void *, int, unsigned);
typedef void (*fftwf_execute_dft_t)(void *, void *, void *);
typedef void *(*fftwf_malloc_t)(size_t);
typedef void (*fftwf_free_t)(void *);
typedef void (*fftwf_destroy_plan_t)(void *);
static struct {
    int tried, ok;
    fftwf_plan_dft_1d_t plan_dft_1d; fftwf_execute_dft_t execute_dft; fftwf_malloc_t malloc_; fftwf_free_t free_;
    fftwf_destroy_plan_t destroy_plan;
} g_fftw;

int orc_fftw_available(void)
{
    if (!g_fftw.tried) {
        g_fftw.tried = 1;
        void *h = dlopen("libfftw3f.so.3", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libfftw3f.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            g_fftw.plan_dft_1d = (fftwf_plan_dft_1d_t)dlsym(h, "fftwf_plan_dft_1d");
            g_fftw.execute_dft = (fftwf_execute_dft_t)dlsym(h, "fftwf_execute_dft");
            g_fftw.malloc_ = (fftwf_malloc_t)dlsym(h, "fftwf_malloc");
            g_fftw.free_ = (fftwf_free_t)dlsym(h, "fftwf_free");
            g_fftw.destroy_plan = (fftwf_destroy_plan_t)dlsym(h, "fftwf_destroy_plan");
            g_fftw.ok = g_fftw.plan_dft_1d && g_fftw.execute_dft && g_fftw.malloc_ && g_fftw.free_ && g_fftw.destroy_plan;
        }
    }
    return g_fftw.ok;
}

typedef struct { stream_job jb; void *plan; int as_written; } fftw_job;

static void *fftw_worker(void *arg)
{
    fftw_job *fj = (fftw_job *)arg;
    const stream_job *jb = &fj->jb;
    const size_t W = jb->W, H = jb->H, M = W - 1, P = 2 * W;
    const int C = jb->channels, pairs = C == 1 ? 1 : C / 2;
    cf *z = NULL, *F = NULL;
    float *win = NULL;
    if (!fj->as_written) {
        z = (cf *)g_fftw.malloc_(sizeof(cf) * P); F = (cf *)g_fftw.malloc_(sizeof(cf) * P);
        win = (float *)malloc(sizeof(float) * W);
        orc_hann_window(W, win);
        memset(z, 0, sizeof(cf) * P);
    }
    const float tau = 6.28318530717958647692f;
    for (size_t i = jb->begin; i < jb->end; ++i) {
        const float *base = jb->pcm + ((jb->first + i) * H) * (size_t)C;
        for (int p = 0; p < pairs; ++p) {
            const float *l = C == 1 ? base : base + 2 * p, *r = C == 1 ? base : base + 2 * p + 1;
            if (fj->as_written) {   /* fft.rs:68,76: two fresh zero-initialised aligned buffers per frame */
                z = (cf *)g_fftw.malloc_(sizeof(cf) * P); F = (cf *)g_fftw.malloc_(sizeof(cf) * P);
                memset(z, 0, sizeof(cf) * P); memset(F, 0, sizeof(cf) * P);
            }
            for (size_t n = 0; n < W; ++n) {
                const float sfac = fj->as_written ? 0.5f * (1.0f - cosf((tau * (float)n) / (float)W)) : win[n];   /* :61 */
                z[n].re = l[n * (size_t)C] * sfac; z[n].im = r[n * (size_t)C] * sfac;
            }
            g_fftw.execute_dft(fj->plan, z, F);   /* :77 */
            float *o = jb->out + ((i * (size_t)pairs + (size_t)p) * M) * 2;
            const float scale = 2.0f / (float)W;
            for (size_t j = 0; j < M; ++j) {       /* :81-98 */
                const size_t k = j + 1;
                const cf a = F[k], b = F[P - k];
                o[2 * j + 0] = hypotf(a.re + b.re, a.im - b.im) / 2.0f * scale;
                o[2 * j + 1] = hypotf(a.re - b.re, a.im + b.im) / 2.0f * scale;
            }
            if (fj->as_written) { g_fftw.free_(z); g_fftw.free_(F); }
        }
    }
    if (!fj->as_written) { g_fftw.free_(z); g_fftw.free_(F); free(win); }
    return NULL;
}

/* returns the number of frames processed, or (size_t)-1 when the host has no libfftw3f */
size_t orc_fftw_stream_process(const float *pcm, size_t n, int channels, size_t W, size_t H, size_t first, size_t count,
                               int threads, int as_written, float *out)
{
    if (!orc_fftw_available()) return (size_t)-1;
    if (channels < 1 || (channels > 1 && (channels & 1)) || W < 2 || H < 1) return 0;
    size_t total = orc_num_frames(n, W, H);
    if (first >= total) return 0;
    if (count > total - first) count = total - first;
    if (threads < 1) threads = 1;
    if ((size_t)threads > count) threads = (int)(count ? count : 1);
    /* the planner is not thread-safe and MEASURE overwrites its arrays: plan once, here, on scratch buffers */
    cf *pa = (cf *)g_fftw.malloc_(sizeof(cf) * 2 * W), *pb = (cf *)g_fftw.malloc_(sizeof(cf) * 2 * W);
    void *plan = g_fftw.plan_dft_1d((int)(2 * W), pa, pb, -1 /* FFTW_FORWARD */, 0u /* FFTW_MEASURE */);
    if (!plan) { g_fftw.free_(pa); g_fftw.free_(pb); return 0; }
    fftw_job *jobs = (fftw_job *)calloc((size_t)threads, sizeof(fftw_job));
    pthread_t *tids = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    for (int i = 0; i < threads; ++i) {
        stream_job *jb = &jobs[i].jb;
        jb->pcm = pcm; jb->n = n; jb->channels = channels; jb->W = W; jb->H = H; jb->first = first; jb->count = count; jb->out = out;
        jb->begin = count * (size_t)i / (size_t)threads;
        jb->end = count * (size_t)(i + 1) / (size_t)threads;
        jobs[i].plan = plan; jobs[i].as_written = as_written;
        if (threads == 1) fftw_worker(&jobs[i]);
        else pthread_create(&tids[i], NULL, fftw_worker, &jobs[i]);
    }
    if (threads > 1) for (int i = 0; i < threads; ++i) pthread_join(tids[i], NULL);
    g_fftw.destroy_plan(plan);
    g_fftw.free_(pa); g_fftw.free_(pb);
    free(jobs); free(tids);
    return count;
}

/* ------------------------------------------------------------------------------------------ */
/* InterpolatedFrequencySample  (interpolated_frequency_sample.rs)                            */
/* ------------------------------------------------------------------------------------------ */

float orc_period(size_t M, uint32_t sample_rate)
{
    /* :52-54  2.0 * self.magnitudes.len() as f32 / self.sample_rate.0 as f32 */
    volatile float a = 2.0f * (float)M;
    return a / (float)sample_rate;
}

float orc_index_of(float frequency, size_t M, uint32_t sample_rate)
{
    /* :24-31  (frequency * period).clamp(0.0, (len - 1) as f32) */
    volatile float idx = frequency * orc_period(M, sample_rate);
    float hi = (float)(M - 1);
    float v = idx;
    if (v < 0.0f) v = 0.0f; /* f32::clamp: NaN stays NaN */
    if (v > hi) v = hi;
    return v;
}

void orc_cubic_interpolate(const float *data, size_t M, float index, float out[2])
{
    /* :89-105, Paul Bourke cubic.  x0 = (floor as usize - 1).max(0) underflows in the reference
     * when floor == 0 (quirk Q4: panic / wrap); restated with saturation. */
    float fl = floorf(index);
    volatile float mu = index - fl;
    size_t x1 = f32_as_usize(fl);
    size_t x0 = x1 > 0 ? x1 - 1 : 0;
    size_t x2 = x1 + 1 < M - 1 ? x1 + 1 : M - 1;
    size_t x3 = x1 + 2 < M - 1 ? x1 + 2 : M - 1;
    /* num_traits::pow(mu, 2) = mu*mu ; pow(mu, 3) = mu * (mu*mu)  (exponentiation by squaring) */
    volatile float mu2 = mu * mu;
    volatile float mu3 = mu * mu2;
    for (int c = 0; c < 2; ++c) {
        float y0 = data[2 * x0 + c], y1 = data[2 * x1 + c], y2 = data[2 * x2 + c], y3 = data[2 * x3 + c];
        volatile float t0 = y3 - y2;
        volatile float t1 = t0 - y0;
        volatile float a0 = t1 + y1;          /* y3 - y2 - y0 + y1 */
        volatile float t2 = y0 - y1;
        volatile float a1 = t2 - a0;          /* y0 - y1 - a0 */
        volatile float a2 = y2 - y0;
        float a3 = y1;
        volatile float p0 = a0 * mu3;
        volatile float p1 = a1 * mu2;
        volatile float p2 = a2 * mu;
        volatile float s0 = p0 + p1;
        volatile float s1 = p2 + a3;
        out[c] = s0 + s1;                     /* (a0*mu^3) + (a1*mu^2) + (a2*mu + a3) */
    }
}

void orc_cosine_interpolate(const float *data, size_t M, float index, float out[2])
{
    /* :79-86.  clamp(low+1, len-1) panics in the reference when low+1 > len-1; saturated here. */
    const float pi = 3.14159265358979323846264338327950288f;
    size_t low = f32_as_usize(floorf(index));
    size_t high = f32_as_usize(ceilf(index));
    if (high < low + 1) high = low + 1;
    if (high > M - 1) high = M - 1;
    volatile float off = index - (float)low;
    volatile float ang = off * pi;
    volatile float c = cosf(ang);
    volatile float om = 1.0f - c;
    volatile float o2 = om / 2.0f;
    volatile float w0 = 1.0f - o2;
    for (int ch = 0; ch < 2; ++ch) {
        volatile float a = data[2 * low + ch] * w0;
        volatile float b = data[2 * high + ch] * o2;
        out[ch] = a + b;
    }
}

size_t orc_num_samples_in(size_t M, uint32_t sample_rate, float f0, float f1)
{
    /* :63-64 */
    float i0 = orc_index_of(f0, M, sample_rate), i1 = orc_index_of(f1, M, sample_rate);
    volatile float d = i1 - i0;
    size_t n = f32_as_usize(floorf(d));
    return n < 1 ? 1 : n;
}

void orc_magnitude_in(const float *data, size_t M, uint32_t sample_rate, float f0, float f1, int interp,
                      float out[2])
{
    /* :60-75 */
    size_t n = orc_num_samples_in(M, sample_rate, f0, f1);
    /* [third-party] iter_num_tools 0.7.1 lin_space over a half-open Range:
     * step = (end - start) / n ; x_i = start + i * step, i = 0..n-1 (end excluded) */
    volatile float span = f1 - f0;
    volatile float step = span / (float)n;
    float sum[2] = {0.0f, 0.0f}; /* Complex::sum starts from zero */
    for (size_t i = 0; i < n; ++i) {
        volatile float off = (float)i * step;
        volatile float f = f0 + off;
        float idx = orc_index_of(f, M, sample_rate);
        float v[2];
        if (interp == ORC_INTERP_COSINE) orc_cosine_interpolate(data, M, idx, v);
        else orc_cubic_interpolate(data, M, idx, v);
        volatile float s0 = sum[0] + v[0], s1 = sum[1] + v[1];
        sum[0] = s0; sum[1] = s1;
    }
    out[0] = sum[0] / (float)n;
    out[1] = sum[1] / (float)n;
}

/* ------------------------------------------------------------------------------------------ */
/* LogCoordf64  (log_scaling.rs)                                                              */
/* ------------------------------------------------------------------------------------------ */

double orc_log_unmap(double f_min, double f_max, double zero_point, int p, int pmin, int pmax)
{
    /* log_scaling.rs:160-191: linear = ln(start)..ln(end), start/end = range - zero_point
     * (negative ranges and the zero-start fix-up are not reachable from simple_spectrogram.rs:107).
     * [third-party] plotters 0.3.5 RangedCoordf64::unmap:
     *   logical_offset = (p - min) as f64 / (max - min) as f64 ; (hi - lo) * logical_offset + lo
     * log_scaling.rs:116-118: exp(), then + zero_point. */
    double lo = log(f_min - zero_point), hi = log(f_max - zero_point);
    volatile double off = (double)(p - pmin) / (double)(pmax - pmin);
    volatile double lin = (hi - lo) * off + lo;
    return exp(lin) + zero_point;
}

/* ------------------------------------------------------------------------------------------ */
/* ColorScheme  (colorscheme.rs)                                                              */
/* ------------------------------------------------------------------------------------------ */

static orc_gradient_fn g_gradient_fn = NULL;
static void *g_gradient_user = NULL;
void orc_set_gradient_fn(orc_gradient_fn fn, void *user) { g_gradient_fn = fn; g_gradient_user = user; }

/* colour of the gradient at t: the table + index rule, or the continuous callback when gradient == NULL */
static void gradient_eval(const uint8_t *gradient, int n_lut, int lut_mode, double t, uint8_t rgb[3]);

int orc_lut_index(double t, int n_lut, int lut_mode)
{
    /* [third-party] colorous 1.0.12 Gradient::eval_continuous for the 256-entry ramps
     * (Viridis, Magma, Inferno, Plasma).  The table and this rule are inputs, not constants. */
    double x;
    if (lut_mode == ORC_LUT_ROUND_NM1) x = floor(t * (double)(n_lut - 1) + 0.5);
    else x = floor(t * (double)n_lut);
    size_t i = f64_as_usize(x); /* NaN, negatives -> 0 */
    if (i > (size_t)(n_lut - 1)) i = (size_t)(n_lut - 1);
    return (int)i;
}

uint8_t orc_alpha_u8(float alpha)
{
    /* simple_spectrogram.rs:159: (alpha * 255.0) as u8 -- saturating, NaN -> 0 */
    volatile float v = alpha * 255.0f;
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

static void gradient_eval(const uint8_t *gradient, int n_lut, int lut_mode, double t, uint8_t rgb[3])
{
    if (!gradient) { g_gradient_fn(t, rgb, g_gradient_user); return; }
    int idx = orc_lut_index(t, n_lut, lut_mode);
    rgb[0] = gradient[3 * idx + 0]; rgb[1] = gradient[3 * idx + 1]; rgb[2] = gradient[3 * idx + 2];
}

static float bounded_db(float min_db, float max_db, float l, float r)
{
    /* colorscheme.rs:59-61 */
    volatile float ll = l * l, rr = r * r;
    volatile float power = ll + rr;              /* norm_sqr */
    volatile float arg = power + 1e-7f;
    volatile float db = 10.0f * log10f(arg);
    volatile float num = db - min_db;
    volatile float den = max_db - min_db;
    return num / den;
}

void orc_color_for(const uint8_t *gradient, int n_lut, int lut_mode, int stereo, float min_db, float max_db,
                   float l, float r, uint8_t rgb[3], float *alpha)
{
    float bounded = bounded_db(min_db, max_db, l, r);
    if (stereo) {
        /* :63-66: t = l as f64 / l1_norm(l, r) as f64 ; alpha = magnitude_bounded */
        volatile float l1 = fabsf(l) + fabsf(r);
        double t = (double)l / (double)l1;
        gradient_eval(gradient, n_lut, lut_mode, t, rgb);
        *alpha = bounded;
    } else {
        /* :67-70 */
        gradient_eval(gradient, n_lut, lut_mode, (double)bounded, rgb);
        *alpha = 1.0f;
    }
}

void orc_render_column(const float *mags, size_t M, uint32_t sample_rate, int R, double f_min, double f_max,
                       int interp, const uint8_t *gradient, int n_lut, int lut_mode, int stereo, float min_db,
                       float max_db, uint8_t *rgba)
{
    /* simple_spectrogram.rs:141-161 */
    for (int py = 0; py < R; ++py) {
        float f0 = (float)orc_log_unmap(f_min, f_max, 0.0, py, 0, R);     /* :142, :145 */
        float f1 = (float)orc_log_unmap(f_min, f_max, 0.0, py + 1, 0, R); /* :143, :145 */
        float m[2];
        orc_magnitude_in(mags, M, sample_rate, f0, f1, interp, m);         /* :147 */
        uint8_t rgb[3]; float alpha;
        orc_color_for(gradient, n_lut, lut_mode, stereo, min_db, max_db, m[0], m[1], rgb, &alpha); /* :152 */
        int y = R - py - 1;                                                 /* :150 */
        rgba[4 * y + 0] = rgb[0]; rgba[4 * y + 1] = rgb[1]; rgba[4 * y + 2] = rgb[2];
        rgba[4 * y + 3] = orc_alpha_u8(alpha);                              /* :159 */
    }
}

void orc_lookup_table(const uint8_t *gradient, int n_lut, int lut_mode, int stereo, int res, float *table)
{
    /* colorscheme.rs:73-92 */
    for (int i = 0; i < res; ++i)
        for (int j = 0; j < res; ++j) {
            volatile float magnitude = (float)i / (float)(res - 1);
            volatile float jf = (float)j / (float)(res - 1);
            volatile float pan = 1.0f - jf;
            float *o = table + 4 * ((size_t)i * (size_t)res + (size_t)j);
            uint8_t c[3];
            gradient_eval(gradient, n_lut, lut_mode, stereo ? (double)pan : (double)magnitude, c);
            o[0] = (float)c[0] / 256.0f;
            o[1] = (float)c[1] / 256.0f;
            o[2] = (float)c[2] / 256.0f;
            o[3] = stereo ? magnitude : 1.0f;
        }
}

/* ------------------------------------------------------------------------------------------ */
/* GPUSpectrogram's fragment program  (widgets/gpu_spectrogram.rs:150-186)                    */
/* [third-party] OpenGL leaves GL_LINEAR's arithmetic to the implementation: restated are the  */
/* program text and the sampler state the reference sets (:282-289): the two nearest texels   */
/* per axis weighted by the fractional part of s * size - 0.5, REPEAT for the F16F16 ring,     */
/* CLAMP (to edge) for the 32 x 32 palette; float32.  NaN coordinates sample coordinate 0.     */
/* Not a parity target: the widget variant's checker.               */
/* ------------------------------------------------------------------------------------------ */

static float half_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu, bits;
    if (e == 0) {
        if (m == 0) bits = sign;
        else {                       /* subnormal half: renormalise */
            e = 127 - 15 + 1;
            while (!(m & 0x400u)) { m <<= 1; --e; }
            bits = sign | (e << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (e == 31) bits = sign | 0x7f800000u | (m << 13);
    else bits = sign | ((e + 127 - 15) << 23) | (m << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static int wrap_repeat(float f, int n)
{
    int i = (int)fmodf(f, (float)n);
    return i < 0 ? i + n : i;
}
static int clamp_edge(float f, int n) { return f < 0.0f ? 0 : (f > (float)(n - 1) ? n - 1 : (int)f); }
static float mixf(float u, float v, float w)
{
    volatile float d = v - u;
    volatile float t = d * w;
    return u + t;
}

void orc_glsl_fragments(const uint16_t *ring_f16, size_t M, size_t rows, size_t offset, const float *palette32,
                        float min_db, float max_db, size_t width, size_t height, float *out)
{
    const float min_frequency = 32.0f, max_frequency = 22030.0f; /* the locals that shadow the uniforms (:152-153) */
    for (size_t py = 0; py < height; ++py)
        for (size_t px = 0; px < width; ++px) {
            volatile float uvx = ((float)px + 0.5f) / (float)width, uvy = ((float)py + 0.5f) / (float)height;
            volatile float log_min = logf(min_frequency), log_max = logf(max_frequency);
            volatile float lrange = log_max - log_min;
            volatile float lf0 = uvy * lrange;
            volatile float log_frequency = lf0 + log_min;
            volatile float s = expf(log_frequency) / max_frequency;            /* log_frequency_mapped */
            volatile float t0 = uvx * (float)rows;
            volatile float t1 = t0 + (float)offset;
            volatile float t = t1 / (float)rows;                                /* time, with offset */
            /* texture(fft, coord.yx): x = frequency over M texels, y = time over `rows` texels, REPEAT, LINEAR */
            volatile float xs = s * (float)M, ys = t * (float)rows;
            volatile float x = xs - 0.5f, y = ys - 0.5f;
            const float fx0 = floorf(x), fy0 = floorf(y);
            volatile float ax = x - fx0, ay = y - fy0;
            const int x0 = wrap_repeat(fx0, (int)M), x1 = wrap_repeat(fx0 + 1.0f, (int)M);
            const int y0 = wrap_repeat(fy0, (int)rows), y1 = wrap_repeat(fy0 + 1.0f, (int)rows);
            float mag[2];
            for (int ch = 0; ch < 2; ++ch) {
                const float a = half_to_float(ring_f16[((size_t)y0 * M + x0) * 2 + ch]), b = half_to_float(ring_f16[((size_t)y0 * M + x1) * 2 + ch]);
                const float c = half_to_float(ring_f16[((size_t)y1 * M + x0) * 2 + ch]), d = half_to_float(ring_f16[((size_t)y1 * M + x1) * 2 + ch]);
                mag[ch] = mixf(mixf(a, b, ax), mixf(c, d, ax), ay);
            }
            volatile float p0 = mag[0] * mag[0], p1 = mag[1] * mag[1];
            volatile float power = p0 + p1;
            volatile float arg = power + 1e-7f;
            volatile float l10 = 10.0f * logf(arg);
            volatile float magnitude_log = l10 / logf(10.0f);
            volatile float num = magnitude_log - min_db, den = max_db - min_db;
            volatile float magnitude_db = num / den;
            volatile float lr = mag[0] + mag[1];
            volatile float pan = mag[1] / lr;
            /* texture(palette, vec2(pan, magnitude_db)): 32 x 32, CLAMP, LINEAR */
            float ps = pan, pt = magnitude_db;
            if (!(ps == ps)) ps = 0.0f;
            if (!(pt == pt)) pt = 0.0f;
            volatile float qx = ps * 32.0f, qy = pt * 32.0f;
            volatile float cx = qx - 0.5f, cy = qy - 0.5f;
            const float gx0 = floorf(cx), gy0 = floorf(cy);
            volatile float bx = cx - gx0, by = cy - gy0;
            const int u0 = clamp_edge(gx0, 32), u1 = clamp_edge(gx0 + 1.0f, 32), v0 = clamp_edge(gy0, 32), v1 = clamp_edge(gy0 + 1.0f, 32);
            float *o = out + 4 * (py * width + px);
            for (int ch = 0; ch < 4; ++ch) {
                const float a = palette32[(v0 * 32 + u0) * 4 + ch], b = palette32[(v0 * 32 + u1) * 4 + ch];
                const float c = palette32[(v1 * 32 + u0) * 4 + ch], d = palette32[(v1 * 32 + u1) * 4 + ch];
                o[ch] = mixf(mixf(a, b, bx), mixf(c, d, bx), by);
            }
        }
}

/* ------------------------------------------------------------------------------------------ */
/* SpectrumAnalyzer  (widgets/spectrum_analyzer.rs)                                           */
/* ------------------------------------------------------------------------------------------ */

float orc_log_space(float start, float end, size_t n, float base, size_t i)
{
    /* :20-36 -- f32::log(base) is ln(x) / ln(base); the iterator never stops (the `if i > n` is a no-op) */
    volatile float ls = logf(start) / logf(base);
    volatile float le = logf(end) / logf(base);
    volatile float span = le - ls;
    volatile float step = span / (float)n;
    volatile float off = step * (float)i;
    volatile float lin = ls + off;
    return powf(base, lin);
}

void orc_spectrum_levels(const float *mags, size_t M, uint32_t sample_rate, int interp, size_t n_bars, double *levels)
{
    /* push_frequencies, :46-68 */
    const float min = -70.0f, max = -10.0f;
    volatile float half = (float)sample_rate / 2.0f;       /* frequencies().end, interpolated_frequency_sample.rs:56-58 */
    const float end = half > 22050.0f ? half : 22050.0f;   /* .max(22050.0) */
    for (size_t i = 0; i < n_bars; ++i) {
        const float f0 = orc_log_space(32.0f, end, n_bars + 1, 10.0f, i);
        const float f1 = orc_log_space(32.0f, end, n_bars + 1, 10.0f, i + 1);
        float lr[2];
        orc_magnitude_in(mags, M, sample_rate, f0, f1, interp, lr);
        volatile float magnitude = hypotf(lr[0], lr[1]);    /* c32::new(l, r).norm() */
        volatile float biased = magnitude + 1e-7f;
        volatile float db = 10.0f * log10f(biased);
        volatile float num = db - min;
        volatile float t = num / (max - min);
        levels[i] = fmax((double)t, levels[i] * 0.99);      /* f64::max ignores a NaN operand, as fmax */
    }
}

/* ------------------------------------------------------------------------------------------ */
/* synthetic inputs -- not reference code                                         */
/* ------------------------------------------------------------------------------------------ */

uint32_t orc_lowbias32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

void orc_white_noise(uint32_t seed, uint64_t first, size_t n, float *out)
{
    for (size_t i = 0; i < n; ++i) {
        uint64_t idx = first + i;
        /* the high word only matters past 2^32 samples (config 5) */
        uint32_t s = seed + (uint32_t)(idx >> 32) * 0x9E3779B9U;
        uint32_t h = orc_lowbias32(s ^ (uint32_t)idx);
        out[i] = (float)(h >> 8) * 1.1920928955078125e-07f - 1.0f; /* 2^-23 */
    }
}

void orc_sine_sweep(size_t first, size_t n, float *out)
{
    /* x[n] = 0.5 sin(2 pi (f0 t + (f1-f0) t^2 / (2T))), t = n/48000, f0 = 20, f1 = 20000, T = 1 */
    const double f0 = 20.0, f1 = 20000.0, T = 1.0, fs = 48000.0;
    for (size_t i = 0; i < n; ++i) {
        double t = (double)(first + i) / fs;
        double ph = 2.0 * M_PI * (f0 * t + (f1 - f0) * t * t / (2.0 * T));
        out[i] = (float)(0.5 * sin(ph));
    }
}
