// sgx_live.hip -- live capture: the ring between the audio callback and the GUI tick, and the
// SpectrumAnalyzer consumer.  Host code only; the transforms are the context's ordinary launches.
//
//   producer   audio_input_list_model.rs:63-75   cpal callback -> HeapRb::push_iter
//   consumer   audio_transform.rs:34-42           hop loop over the ring (peek W, skip H)
//   analyzer   spectrum_analyzer.rs:20-68         128 log-spaced level bars
//
// Sample positions are counted in (l, r) pairs since creation:
//   skipped_  <= uploaded_ <= pushed_ <= skipped_ + capacity
//   [skipped_, uploaded_)  resident on the device, at d_cur[0 ..)
//   [uploaded_, pushed_)   in the pinned host ring, slot = position % capacity
// The producer writes pushed_, the consumer writes skipped_ (both std::atomic); uploaded_ and the
// device buffers belong to the consumer.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <new>

#include "live_ring.hpp"
#include "sgx_internal.hpp"

static_assert(sizeof(sgx::RingPair) == sizeof(float2), "the ring's pairs are the device's float2");

struct sgx_live {
    sgx_ctx *ctx = nullptr;
    size_t capacity = 0;
    uint32_t flags = 0;
    float2 *h_ring = nullptr;        // pinned, [capacity]
    float2 *d_cur = nullptr;         // [capacity] device image of the ring's unconsumed head
    float2 *d_alt = nullptr;         // [capacity] compaction target (ping-pong)
    void *d_out = nullptr;           // results of one tick
    size_t out_bytes = 0;
    sgx::LiveRingState ring;         // positions and the lock-free producer / consumer protocol (live_ring.hpp: host only, sanitizer-tested)
};

namespace {

int live_fail(sgx_live *l, int code, const char *msg)
{
    if (l && l->ctx) l->ctx->err = msg;
    return code;
}

int live_fail_hip(sgx_live *l, hipError_t e, const char *what)
{
    char buf[512];
    std::snprintf(buf, sizeof(buf), "%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
    if (l && l->ctx) l->ctx->err = buf;
    return SGX_ERR_HIP;
}

#define LIVE_HIP(l, call)                                              \
    do {                                                               \
        hipError_t e__ = (call);                                       \
        if (e__ != hipSuccess) return live_fail_hip((l), e__, #call);  \
    } while (0)

}  // namespace

extern "C" {

int sgx_live_create(sgx_ctx *c, size_t capacity_pairs, uint32_t flags, sgx_live **out)
{
    if (out) *out = nullptr;
    if (!c || !out) return SGX_ERR_INVALID_ARG;
    if (c->C != 2) {
        c->err = "sgx_live_create: the ring holds (l, r) pairs, the context must have channels = 2";
        return SGX_ERR_INVALID_ARG;
    }
    if (capacity_pairs < c->W) {
        c->err = "sgx_live_create: the ring must hold at least one window";
        return SGX_ERR_INVALID_ARG;
    }
    sgx_live *l = new (std::nothrow) sgx_live();
    if (!l) return SGX_ERR_NOMEM;
    l->ctx = c;
    l->capacity = capacity_pairs;
    l->flags = flags;
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&l->h_ring), capacity_pairs * sizeof(float2), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&l->d_cur), capacity_pairs * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&l->d_alt), capacity_pairs * sizeof(float2));
    if (e != hipSuccess) {
        const int rc = live_fail_hip(l, e, "sgx_live_create: allocation");
        sgx_live_destroy(l);
        return rc;
    }
    l->ring.slots = reinterpret_cast<sgx::RingPair *>(l->h_ring);
    l->ring.capacity = capacity_pairs;
    *out = l;
    return SGX_OK;
}

void sgx_live_destroy(sgx_live *l)
{
    if (!l) return;
    if (l->ctx) {
        (void)hipSetDevice(l->ctx->device);
        (void)hipStreamSynchronize(l->ctx->stream);
    }
    if (l->h_ring) (void)hipHostFree(l->h_ring);
    if (l->d_cur) (void)hipFree(l->d_cur);
    if (l->d_alt) (void)hipFree(l->d_alt);
    if (l->d_out) (void)hipFree(l->d_out);
    delete l;
}

long long sgx_live_push(sgx_live *l, const float *h_samples, size_t n_values, uint32_t channels)
{
    if (!l) return SGX_ERR_INVALID_ARG;
    if (channels != 1 && channels != 2) return SGX_ERR_UNSUPPORTED;  // "{}-channel input not supported!" (:73)
    if (n_values && !h_samples) return SGX_ERR_INVALID_ARG;
    return (long long)l->ring.push(h_samples, n_values, channels);
}

size_t sgx_live_occupied(const sgx_live *l)
{
    return l ? l->ring.occupied() : 0;
}

static int live_tick(sgx_live *l, int what, void *h_out, sgx_view *view, sgx_image *image, size_t max_frames, size_t *n_frames);

int sgx_live_tick(sgx_live *l, int what, void *h_out, size_t max_frames, size_t *n_frames)
{
    return live_tick(l, what, h_out, nullptr, nullptr, max_frames, n_frames);
}

// GPUSpectrogram::render (gpu_spectrogram.rs:255-275): this tick's frames go from the transform straight into the widget's ring
// texture -- half-pair rows, device to device on the context's stream; nothing crosses the bus but the new samples.
int sgx_live_tick_view(sgx_live *l, sgx_view *view, size_t max_frames, size_t *n_frames)
{
    if (!view) { if (n_frames) *n_frames = 0; return SGX_ERR_INVALID_ARG; }
    return live_tick(l, SGX_LIVE_MAGS_F16, nullptr, view, nullptr, max_frames, n_frames);
}

int sgx_live_tick_image(sgx_live *l, sgx_image *image, size_t max_frames, size_t *n_frames)
{
    if (!image) { if (n_frames) *n_frames = 0; return SGX_ERR_INVALID_ARG; }
    return live_tick(l, SGX_LIVE_RGBA, nullptr, nullptr, image, max_frames, n_frames);
}

static int live_tick(sgx_live *l, int what, void *h_out, sgx_view *view, sgx_image *image, size_t max_frames, size_t *n_frames)
{
    if (n_frames) *n_frames = 0;
    if (!l) return SGX_ERR_INVALID_ARG;
    sgx_ctx *c = l->ctx;
    // a view of another context has another row length (its M), another stream and possibly another device: refused before any work
    if (view && sgx::view_context(view) != c)
        return live_fail(l, SGX_ERR_INVALID_ARG, "sgx_live_tick_view: the view belongs to another (or a destroyed) context");
    if (image && sgx::image_context(image) != c)
        return live_fail(l, SGX_ERR_INVALID_ARG, "sgx_live_tick_image: the image belongs to another (or a destroyed) context");
    size_t frame_bytes;
    switch (what) {
    case SGX_LIVE_MAGS: frame_bytes = (size_t)c->M * 2 * sizeof(float); break;
    case SGX_LIVE_MAGS_F16: frame_bytes = (size_t)c->M * 2 * 2; break;
    case SGX_LIVE_RGBA: frame_bytes = (size_t)c->R * 4; break;
    default: return live_fail(l, SGX_ERR_INVALID_ARG, "sgx_live_tick: unknown output format");
    }
    LIVE_HIP(l, hipSetDevice(c->device));

    // the samples that arrived since the last tick
    const sgx::LiveRingState::Upload up = l->ring.begin_tick();
    if (up.first) {
        float2 *dst = l->d_cur + up.dst;
        LIVE_HIP(l, hipMemcpyAsync(dst, l->h_ring + up.slot, up.first * sizeof(float2), hipMemcpyHostToDevice, c->stream));
        if (up.second)
            LIVE_HIP(l, hipMemcpyAsync(dst + up.first, l->h_ring, up.second * sizeof(float2), hipMemcpyHostToDevice, c->stream));
    }
    const size_t occupied = up.occupied;

    // the hop loop: every complete frame, one launch
    size_t frames = sgx_num_frames(c, occupied);
    const bool truncated = frames > max_frames;
    if (truncated) frames = max_frames;
    if (frames) {
        if (!h_out && !view && !image) return live_fail(l, SGX_ERR_INVALID_ARG, "sgx_live_tick: null output buffer");
        const size_t need = frames * frame_bytes;
        if (need > l->out_bytes) {
            LIVE_HIP(l, hipStreamSynchronize(c->stream));
            if (l->d_out) { (void)hipFree(l->d_out); l->d_out = nullptr; l->out_bytes = 0; }
            LIVE_HIP(l, hipMalloc(&l->d_out, need));
            l->out_bytes = need;
        }
        const size_t n_samples = (frames - 1) * (size_t)c->H + c->W;
        const float *pcm = reinterpret_cast<const float *>(l->d_cur);
        size_t got = 0;
        int rc;
        if (what == SGX_LIVE_MAGS) rc = sgx_stft_batch(c, pcm, n_samples, 0, frames, static_cast<float *>(l->d_out), &got);
        else if (what == SGX_LIVE_MAGS_F16) rc = sgx_stft_batch_f16(c, pcm, n_samples, 0, frames, l->d_out, &got);
        else rc = sgx_render_batch(c, pcm, n_samples, 0, frames, static_cast<uint8_t *>(l->d_out), &got);
        if (rc != SGX_OK) return rc;
        if (got != frames) return live_fail(l, SGX_ERR_INVALID_ARG, "sgx_live_tick: frame count mismatch");
        if (view) {
            rc = sgx_view_write_rows(view, l->d_out, frames, nullptr);
            if (rc != SGX_OK) return rc;
        } else if (image) {
            rc = sgx_image_write_columns(image, static_cast<const uint8_t *>(l->d_out), frames, nullptr);
            if (rc != SGX_OK) return rc;
        } else {
            LIVE_HIP(l, hipMemcpyAsync(h_out, l->d_out, need, hipMemcpyDeviceToHost, c->stream));
        }
    }

    // ring.skip(H) per yielded frame, the reference's extra skip on the terminating read (live_ring.hpp)
    const size_t skip = sgx::LiveRingState::skip_of(frames, c->H, (l->flags & SGX_LIVE_REFERENCE_SKIP) != 0, truncated, occupied);
    const size_t keep = occupied - skip;
    if (skip && keep) {
        LIVE_HIP(l, hipMemcpyAsync(l->d_alt, l->d_cur + skip, keep * sizeof(float2), hipMemcpyDeviceToDevice, c->stream));
        float2 *t = l->d_cur;
        l->d_cur = l->d_alt;
        l->d_alt = t;
    }
    // the pinned slots just uploaded are reusable only once the copies above have run
    LIVE_HIP(l, hipStreamSynchronize(c->stream));
    l->ring.end_tick(up, skip);
    if (n_frames) *n_frames = frames;
    return SGX_OK;
}

int sgx_spectrum_levels(sgx_ctx *c, const float *d_column, uint32_t n_bars, double *h_levels)
{
    if (!c) return SGX_ERR_INVALID_ARG;
    if (n_bars == 0) return SGX_OK;
    if (!d_column || !h_levels) {
        c->err = "sgx_spectrum_levels: null buffer";
        return SGX_ERR_INVALID_ARG;
    }
    // log_space(32, frequencies().end.max(22050), n_bars + 1, 10) (spectrum_analyzer.rs:20-36,52-57), f32
    const float base = 10.0f;
    float end = (float)c->sr_u32 / 2.0f;  // frequencies().end (interpolated_frequency_sample.rs:56-58)
    if (!(end > 22050.0f)) end = 22050.0f;
    const float lo = logf(32.0f) / logf(base);
    const float hi = logf(end) / logf(base);
    const float step = (hi - lo) / (float)(n_bars + 1);
    std::vector<float> ranges((size_t)n_bars * 2);
    float prev = powf(base, lo + step * 0.0f);
    for (uint32_t i = 0; i < n_bars; ++i) {
        const float next = powf(base, lo + step * (float)(i + 1));
        ranges[2 * i] = prev;
        ranges[2 * i + 1] = next;
        prev = next;
    }
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess && n_bars > c->levels_cap) {
        e = hipStreamSynchronize(c->stream);
        if (c->d_levels) { (void)hipFree(c->d_levels); c->d_levels = nullptr; c->levels_cap = 0; }
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->d_levels), (size_t)n_bars * 2 * sizeof(float));
        if (e == hipSuccess) c->levels_cap = n_bars;
    }
    if (e != hipSuccess) {
        c->err = std::string("sgx_spectrum_levels: ") + hipGetErrorString(e);
        return SGX_ERR_HIP;
    }
    int rc = sgx_magnitude_in(c, d_column, 1, ranges.data(), n_bars, c->d_levels);
    if (rc != SGX_OK) return rc;
    std::vector<float> lr((size_t)n_bars * 2);
    e = hipMemcpyAsync(lr.data(), c->d_levels, lr.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        c->err = std::string("sgx_spectrum_levels: ") + hipGetErrorString(e);
        return SGX_ERR_HIP;
    }
    const float min_db = -70.0f, max_db = -10.0f;  // locals of push_frequencies (:47-48), not the colour scheme's
    for (uint32_t i = 0; i < n_bars; ++i) {
        float m = hypotf(lr[2 * i], lr[2 * i + 1]);          // c32::new(l, r).norm()   :60
        m = 10.0f * log10f(m + 1e-7f);                        //                          :61
        const double level = (double)((m - min_db) / (max_db - min_db));  //             :62
        const double decayed = h_levels[i] * 0.99;            //                          :64
        h_levels[i] = fmax(level, decayed);                   // f64::max: a NaN operand loses, as in fmax
    }
    return SGX_OK;
}

}  // extern "C"
