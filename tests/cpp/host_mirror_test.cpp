// Exercises include/sgx.hpp (the C++ mirror of the reference's fourier:: interface) end to end on the
// GPU: a ring is filled with the shared counter-based noise, drained through AudioStreamTransform, and
// the frames are written to a file that tests/test_gpu_parity.py compares with the CPU oracle.
// usage: host_mirror_test <out.bin> <sample_rate> <period> <stride> <n_pairs>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "sgx.hpp"

static uint32_t lowbias32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    const float sr = (float)atof(argv[2]), period = (float)atof(argv[3]), stride = (float)atof(argv[4]);
    const size_t n = (size_t)atoll(argv[5]);
    using namespace sgx_host;
    try {
        FastFourierTransform fft(sr, period, stride);
        RingBuffer ring(1 << 20);
        std::vector<StereoMagnitude> lr(n);
        for (size_t i = 0; i < n; ++i) {  // interleaved (l, r) = samples 2i, 2i+1 of the noise stream, seed 0x5EED0001
            const float l = (float)(lowbias32(0x5EED0001u ^ (uint32_t)(2 * i)) >> 8) * 1.1920928955078125e-07f - 1.0f;
            const float r = (float)(lowbias32(0x5EED0001u ^ (uint32_t)(2 * i + 1)) >> 8) * 1.1920928955078125e-07f - 1.0f;
            lr[i] = {l, r};
        }
        if (ring.push_iter(lr.data(), n) != n) return 3;
        AudioStreamTransform<FastFourierTransform> stream(ring, fft, stride);
        auto frames = stream.process();
        // the trait's per-frame process() on the first frame must agree with the batched one bit for bit
        auto one = fft.process(lr.data(), n);
        if (!one || *one != frames.at(0)) { fprintf(stderr, "process() != batched frame 0\n"); return 4; }
        if (fft.process(lr.data(), fft.num_input_samples() - 1)) { fprintf(stderr, "short input must be None\n"); return 5; }
        // the device-resident ring: the same samples pushed in callback-sized pieces with a tick after each must
        // yield the same frames, in order (t * H framing across ticks)
        {
            LiveRing live(fft, 1 << 16);
            std::vector<AudioTransform::Output> got;
            for (size_t i = 0; i < n; i += 480) {
                const size_t m = n - i < 480 ? n - i : 480;
                if (live.push(reinterpret_cast<const float *>(lr.data() + i), 2 * m, 2) != m) return 6;
                for (auto &fr : live.tick()) got.push_back(std::move(fr));
            }
            if (got != frames) { fprintf(stderr, "LiveRing frames != batched frames (%zu vs %zu)\n", got.size(), frames.size()); return 7; }
            bool refused = false;
            try { live.push(reinterpret_cast<const float *>(lr.data()), 6, 3); } catch (const Error &) { refused = true; }
            if (!refused) return 8;
        }
        // SimpleSpectrogram's Pixbuf ring: the same pieces, a tick into a 16-column image after each; against the columns the host-side
        // tick returns, put in place one by one as the reference does (:150-164)
        {
            const uint32_t width = 16, rows = 1024;
            LiveRing live_a(fft, 1 << 16), live_b(fft, 1 << 16);
            ImageRing image(fft, width);
            if (image.width() != width || image.height() != rows) return 8;      // the height is the context's R, not an argument
            std::vector<uint8_t> want((size_t)rows * width * 4, 0), cols(64 * (size_t)rows * 4);
            size_t off = 0, total = 0;
            for (size_t i = 0; i < n; i += 480) {
                const size_t m = n - i < 480 ? n - i : 480;
                live_a.push(reinterpret_cast<const float *>(lr.data() + i), 2 * m, 2);
                live_b.push(reinterpret_cast<const float *>(lr.data() + i), 2 * m, 2);
                total += image.tick(live_a);
                size_t got = 0;
                if (sgx_live_tick(live_b.raw(), SGX_LIVE_RGBA, cols.data(), 64, &got) != SGX_OK) return 9;
                for (size_t c = 0; c < got; ++c) {
                    for (uint32_t y = 0; y < rows; ++y)
                        for (int b = 0; b < 4; ++b) want[((size_t)y * width + off) * 4 + b] = cols[(c * rows + y) * 4 + b];
                    off = (off + 1) % width;
                }
            }
            if (total != frames.size() || image.offset() != off) { fprintf(stderr, "image ring: %zu columns, offset %zu (want %zu, %zu)\n", total, image.offset(), frames.size(), off); return 10; }
            if (image.pixels() != want) { fprintf(stderr, "image ring: pixels differ from the column-by-column scatter\n"); return 11; }
            const auto scrolled = image.pixels(true);
            for (uint32_t y = 0; y < rows; ++y)
                for (uint32_t x = 0; x < width; ++x)
                    for (int b = 0; b < 4; ++b)
                        if (scrolled[((size_t)y * width + x) * 4 + b] != want[((size_t)y * width + (x + off) % width) * 4 + b]) { fprintf(stderr, "image ring: scrolled picture\n"); return 12; }
        }
        FILE *f = fopen(argv[1], "wb");
        const uint64_t hdr[4] = {frames.size(), fft.num_output_frequencies(), stream.stride_samples(), ring.occupied_len()};
        fwrite(hdr, sizeof(hdr), 1, f);
        for (auto &fr : frames) fwrite(fr.data(), sizeof(StereoMagnitude), fr.size(), f);
        fclose(f);
        printf("frames=%zu M=%zu H=%zu left_in_ring=%zu\n", frames.size(), fft.num_output_frequencies(), stream.stride_samples(),
               ring.occupied_len());
    } catch (const Error &e) {
        fprintf(stderr, "sgx error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
