// sgx.hpp -- header-only C++ mirror of the reference's interface for the STFT path, over the C ABI
// of sgx.h.  The reference is Rust; this is the host side a C++ (or, transliterated, a Rust) caller
// programs against: same names, same argument meaning, same "None on short input" rule.
//
//   fourier::AudioTransform            src/fourier/audio_transform.rs:4-11     (abstract interface)
//   fourier::FastFourierTransform      src/fourier/fft.rs:11-99
//   fourier::AudioStreamTransform<T>   src/fourier/audio_transform.rs:14-43    (hop loop over a ring)
//   RingBuffer                         ringbuf::HeapRb<(f32, f32)> as used at src/devices/audio_input_list_model.rs:30,63-72
//   LiveRing                           the same ring, consumed side on the device (sgx_live_*)
//
// Host buffers in, host buffers out (the per-frame `process` of the trait works on host pairs); the
// batched `process` of AudioStreamTransform moves every complete frame of the ring through ONE
// device launch.  Needs libsgx.so and the HIP runtime (hipMalloc / hipMemcpy for staging).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <deque>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "sgx.h"

namespace sgx_host {

using StereoMagnitude = std::pair<float, float>;  // src/fourier/mod.rs:13
using Frequency = float;                          // :15
using Period = float;                             // :14

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// audio_transform.rs:4-11
class AudioTransform {
public:
    using Output = std::vector<StereoMagnitude>;
    virtual ~AudioTransform() = default;
    virtual Frequency sample_rate() const = 0;
    virtual std::size_t num_input_samples() const = 0;
    // `samples`: at least num_input_samples() (l, r) pairs for Some, fewer for None (fft.rs:72)
    virtual std::optional<Output> process(const StereoMagnitude *samples, std::size_t n_available) = 0;
};

// Rust `f32 as usize`
inline std::size_t f32_as_usize(float v) { return v > 0.0f ? (std::size_t)v : 0; }

// fft.rs:11-99
class FastFourierTransform : public AudioTransform {
public:
    // FastFourierTransform::new(sample_rate, period) (fft.rs:18); `stride` is the hop the stream wrapper
    // will use (audio_transform.rs:35) -- the context is created per (window, hop)
    FastFourierTransform(Frequency sample_rate, Period period, Period stride = 0.0f, int device = -1)
        : sample_rate_(sample_rate), period_(period)
    {
        sgx_config cfg;
        sgx_config_init(&cfg);
        cfg.sample_rate = sample_rate;
        cfg.period = period;
        cfg.window_samples = 0;
        cfg.stride = stride;
        cfg.hop_samples = stride > 0.0f ? 0u : 1u;
        cfg.channels = 2;  // process() receives (l, r) pairs
        cfg.device = device;
        int rc = sgx_create(&cfg, &ctx_);
        if (rc != SGX_OK) throw Error(rc, sgx_last_error(nullptr));
    }
    FastFourierTransform(const FastFourierTransform &) = delete;
    FastFourierTransform &operator=(const FastFourierTransform &) = delete;
    ~FastFourierTransform() override
    {
        if (d_in_) (void)hipFree(d_in_);
        if (d_out_) (void)hipFree(d_out_);
        sgx_destroy(ctx_);
    }

    Frequency sample_rate() const override { return sample_rate_; }
    std::size_t num_input_samples() const override { return f32_as_usize(period_ * sample_rate_); }  // fft.rs:41
    std::size_t num_output_frequencies() const { return num_input_samples() - 1; }                   // fft.rs:33

    std::optional<Output> process(const StereoMagnitude *samples, std::size_t n_available) override
    {
        const std::size_t w = num_input_samples();
        Output out(w - 1);
        const int rc = sgx_process_one(ctx_, reinterpret_cast<const float *>(samples), n_available < w ? n_available : w,
                                       reinterpret_cast<float *>(out.data()));
        if (rc < 0) throw Error(rc, sgx_last_error(ctx_));
        if (rc == 0) return std::nullopt;
        return out;
    }

    // every complete frame of `lr` (n pairs) at the context's hop, one launch: frames x (W-1) pairs
    std::vector<Output> process_stream(const StereoMagnitude *lr, std::size_t n)
    {
        const std::size_t frames = sgx_num_frames(ctx_, n), m = num_output_frequencies();
        std::vector<Output> out(frames, Output(m));
        if (!frames) return out;
        reserve(n * 2 * sizeof(float), frames * m * 2 * sizeof(float));
        check_hip(hipMemcpy(d_in_, lr, n * 2 * sizeof(float), hipMemcpyHostToDevice));
        std::size_t got = 0;
        int rc = sgx_stft_batch(ctx_, d_in_, n, 0, frames, d_out_, &got);
        if (rc != SGX_OK) throw Error(rc, sgx_last_error(ctx_));
        if ((rc = sgx_sync(ctx_)) != SGX_OK) throw Error(rc, sgx_last_error(ctx_));
        std::vector<float> flat(frames * m * 2);
        check_hip(hipMemcpy(flat.data(), d_out_, flat.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (std::size_t f = 0; f < frames; ++f)
            for (std::size_t j = 0; j < m; ++j) out[f][j] = {flat[(f * m + j) * 2], flat[(f * m + j) * 2 + 1]};
        return out;
    }

    sgx_ctx *ctx() { return ctx_; }

private:
    static void check_hip(hipError_t e)
    {
        if (e != hipSuccess) throw Error(SGX_ERR_HIP, hipGetErrorString(e));
    }
    void reserve(std::size_t in_bytes, std::size_t out_bytes)
    {
        if (in_bytes > in_cap_) {
            if (d_in_) (void)hipFree(d_in_);
            check_hip(hipMalloc(reinterpret_cast<void **>(&d_in_), in_bytes));
            in_cap_ = in_bytes;
        }
        if (out_bytes > out_cap_) {
            if (d_out_) (void)hipFree(d_out_);
            check_hip(hipMalloc(reinterpret_cast<void **>(&d_out_), out_bytes));
            out_cap_ = out_bytes;
        }
    }
    sgx_ctx *ctx_ = nullptr;
    Frequency sample_rate_;
    Period period_;
    float *d_in_ = nullptr, *d_out_ = nullptr;
    std::size_t in_cap_ = 0, out_cap_ = 0;
};

// the SPSC ring between the capture thread and the GUI thread (audio_input_list_model.rs:30,63-72)
class RingBuffer {
public:
    explicit RingBuffer(std::size_t capacity) : capacity_(capacity) {}
    std::size_t push_iter(const StereoMagnitude *s, std::size_t n)  // overflow is dropped (:70)
    {
        std::size_t pushed = 0;
        for (; pushed < n && data_.size() < capacity_; ++pushed) data_.push_back(s[pushed]);
        return pushed;
    }
    std::size_t skip(std::size_t n)
    {
        n = n < data_.size() ? n : data_.size();
        data_.erase(data_.begin(), data_.begin() + (std::ptrdiff_t)n);
        return n;
    }
    std::size_t occupied_len() const { return data_.size(); }
    std::vector<StereoMagnitude> peek(std::size_t n) const
    {
        n = n < data_.size() ? n : data_.size();
        return std::vector<StereoMagnitude>(data_.begin(), data_.begin() + (std::ptrdiff_t)n);
    }

private:
    std::size_t capacity_;
    std::deque<StereoMagnitude> data_;
};

// The same ring with its consumed side resident on the device (sgx_live_*): the capture callback pushes
// host samples (audio_input_list_model.rs:63-75), one tick() per GUI frame returns every complete frame
// (audio_transform.rs:34-42) and moves only the new samples to the GPU.
class LiveRing {
public:
    // reference_skip: also drop H samples on the terminating short read, exactly as audio_transform.rs:37-41
    LiveRing(FastFourierTransform &transform, std::size_t capacity = 4096, bool reference_skip = false)
        : transform_(transform)
    {
        const int rc = sgx_live_create(transform.ctx(), capacity, reference_skip ? SGX_LIVE_REFERENCE_SKIP : 0u, &live_);
        if (rc != SGX_OK) throw Error(rc, sgx_last_error(transform.ctx()));
    }
    LiveRing(const LiveRing &) = delete;
    LiveRing &operator=(const LiveRing &) = delete;
    ~LiveRing() { sgx_live_destroy(live_); }

    // interleaved samples of a 1- or 2-channel callback; returns the pairs accepted (overflow is dropped)
    std::size_t push(const float *data, std::size_t n_values, unsigned channels)
    {
        const long long rc = sgx_live_push(live_, data, n_values, channels);
        if (rc < 0) throw Error((int)rc, std::to_string(channels) + "-channel input not supported!");
        return (std::size_t)rc;
    }
    std::size_t occupied_len() const { return sgx_live_occupied(live_); }

    std::vector<AudioTransform::Output> tick(std::size_t max_frames = 64)
    {
        const std::size_t m = transform_.num_output_frequencies();
        std::vector<float> flat(max_frames * m * 2);
        std::size_t got = 0;
        const int rc = sgx_live_tick(live_, SGX_LIVE_MAGS, flat.data(), max_frames, &got);
        if (rc != SGX_OK) throw Error(rc, sgx_last_error(transform_.ctx()));
        std::vector<AudioTransform::Output> out(got, AudioTransform::Output(m));
        for (std::size_t f = 0; f < got; ++f)
            for (std::size_t j = 0; j < m; ++j) out[f][j] = {flat[(f * m + j) * 2], flat[(f * m + j) * 2 + 1]};
        return out;
    }

    sgx_live *raw() { return live_; }

private:
    FastFourierTransform &transform_;
    sgx_live *live_ = nullptr;
};

// SimpleSpectrogram's Pixbuf ring on the device (sgx_image_*): `buffer` + `offset` of simple_spectrogram.rs:60-66,89-94, the
// put_pixel loop with its offset update (:140-164) as tick(), the picture of the two sub-images (:181-209) as scrolled().
class ImageRing {
public:
    // (the height is never the caller's to choose: it is the context's row count, read back from the image)
    ImageRing(FastFourierTransform &transform, std::uint32_t width) : transform_(transform)
    {
        const int rc = sgx_image_create(transform.ctx(), width, &image_);
        if (rc != SGX_OK) throw Error(rc, sgx_last_error(transform.ctx()));
        width_ = sgx_image_width(image_);
        rows_ = sgx_image_height(image_);
    }
    ImageRing(const ImageRing &) = delete;
    ImageRing &operator=(const ImageRing &) = delete;
    ~ImageRing() { sgx_image_destroy(image_); }

    // one GUI tick: every complete frame of the capture ring becomes a pixel column at `offset`, device to device
    std::size_t tick(LiveRing &live)
    {
        std::size_t got = 0;
        const int rc = sgx_live_tick_image(live.raw(), image_, width_, &got);
        if (rc != SGX_OK) throw Error(rc, sgx_last_error(transform_.ctx()));
        return got;
    }
    std::size_t offset() const { return sgx_image_offset(image_); }
    std::uint32_t width() const { return width_; }
    std::uint32_t height() const { return rows_; }

    // [rows][width][4] bytes on the host: the buffer as it lies, or the scrolled picture
    std::vector<std::uint8_t> pixels(bool scrolled = false)
    {
        const std::size_t bytes = (std::size_t)rows_ * width_ * 4;
        void *d = nullptr;
        if (hipMalloc(&d, bytes) != hipSuccess) throw Error(SGX_ERR_HIP, "hipMalloc");
        std::vector<std::uint8_t> out(bytes);
        int rc = sgx_image_read(image_, scrolled ? 1 : 0, static_cast<std::uint8_t *>(d));
        if (rc == SGX_OK) rc = sgx_sync(transform_.ctx());
        const hipError_t e = rc == SGX_OK ? hipMemcpy(out.data(), d, bytes, hipMemcpyDeviceToHost) : hipSuccess;
        (void)hipFree(d);
        if (rc != SGX_OK) throw Error(rc, sgx_last_error(transform_.ctx()));
        if (e != hipSuccess) throw Error(SGX_ERR_HIP, hipGetErrorString(e));
        return out;
    }

private:
    FastFourierTransform &transform_;
    sgx_image *image_ = nullptr;
    std::uint32_t width_ = 0, rows_ = 0;
};

// audio_transform.rs:14-43; the three members are public and assignable, as in the reference
template <typename T>
struct AudioStreamTransform {
    RingBuffer &input_stream;
    T &transform;
    Period stride;

    AudioStreamTransform(RingBuffer &s, T &t, Period st) : input_stream(s), transform(t), stride(st) {}

    std::size_t stride_samples() const { return f32_as_usize(stride * transform.sample_rate()); }  // :35

    // every frame the reference's repeat_with / take_while loop would yield, from one batched launch;
    // the ring advances as the reference's does, including the skip of its terminating short read (:37-41)
    std::vector<typename T::Output> process()
    {
        const std::size_t h = stride_samples(), w = transform.num_input_samples(), n = input_stream.occupied_len();
        if (h == 0) throw Error(SGX_ERR_INVALID_ARG, "stride * sample_rate truncates to 0 samples");
        const std::size_t frames = n < w ? 0 : (n - w) / h + 1;
        std::vector<typename T::Output> out;
        if (frames) {
            const auto lr = input_stream.peek((frames - 1) * h + w);
            out = transform.process_stream(lr.data(), lr.size());
        }
        input_stream.skip((frames + 1) * h);
        return out;
    }
};

}  // namespace sgx_host
