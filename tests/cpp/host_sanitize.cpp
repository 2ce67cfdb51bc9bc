// host_sanitize.cpp -- the product's HOST code under the CPU sanitizers (GPU AddressSanitizer is not available on this pool).
//   ring    csrc/live_ring.hpp: the lock-free single-producer / single-consumer capture ring (audio_input_list_model.rs:30,63-72 on
//           the producer side, the hop loop of audio_transform.rs:34-42 on the consumer side) -- a producer thread pushing 1e6 values
//           in bursts, a consumer thread ticking as fast as it can, a 4096-pair ring that wraps hundreds of times and overflows
//           (drops) whenever the consumer is late; every pair the producer was told was accepted must arrive, once, in order.
//           The device copies of sgx_live.hip are a memcpy into a host image here.
//   tables  csrc/sgx_tables.cpp: every table builder at W in {86 ... 9600}, both interpolators, thresholds and segments.
// Built by tests/test_host_sanitizers.py with -fsanitize=thread and with -fsanitize=address,undefined.
// usage: host_sanitize ring|tables
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "live_ring.hpp"
#include "sgx_internal.hpp"

static int fail(const char *what, long long a = 0, long long b = 0)
{
    std::fprintf(stderr, "host_sanitize: %s (%lld, %lld)\n", what, a, b);
    return 1;
}

static int run_ring(bool reference_skip)
{
    const size_t capacity = 4096, W = 2048, H = 256;
    const size_t total_values = 1000000;                       // interleaved (l, r) values offered by the producer
    std::vector<sgx::RingPair> slots(capacity);
    sgx::LiveRingState ring;
    ring.slots = slots.data();
    ring.capacity = capacity;
    std::atomic<bool> done{false};
    std::atomic<unsigned long long> accepted{0}, offered{0};
    // the producer tags pair number p (counted over ACCEPTED pairs) as (p, -p): whatever is dropped never gets a number
    std::thread producer([&] {
        unsigned long long next = 0;
        unsigned x = 12345;
        std::vector<float> burst;
        size_t sent = 0;
        while (sent < total_values) {
            x = x * 1664525u + 1013904223u;
            size_t pairs = 1 + (x >> 20) % 700;                // bursts of 1 .. 700 pairs: sometimes more than the ring has room for
            if (2 * pairs > total_values - sent) pairs = (total_values - sent + 1) / 2;
            burst.resize(2 * pairs);
            // a push accepts a PREFIX of the burst: number the pairs as if all were accepted, renumber from what was
            for (size_t i = 0; i < pairs; ++i) { burst[2 * i] = (float)((next + i) & 0xffffff); burst[2 * i + 1] = -(float)((next + i) & 0xffffff); }
            const size_t n = ring.push(burst.data(), 2 * pairs, 2);
            next += n;
            sent += 2 * pairs;
            offered.fetch_add(pairs, std::memory_order_relaxed);
            accepted.store(next, std::memory_order_release);
            if ((x & 7) == 0) std::this_thread::yield();
        }
        done.store(true, std::memory_order_release);
    });
    // the consumer: the tick of sgx_live.hip with a host image in place of the device's
    std::vector<sgx::RingPair> image(capacity), alt(capacity);
    unsigned long long expect = 0;                             // number of the pair at image[0]
    unsigned long long frames_total = 0, ticks = 0;
    int rc = 0;
    for (;;) {
        const bool last = done.load(std::memory_order_acquire);
        const sgx::LiveRingState::Upload up = ring.begin_tick();
        if (up.first) std::memcpy(image.data() + up.dst, slots.data() + up.slot, up.first * sizeof(sgx::RingPair));
        if (up.second) std::memcpy(image.data() + up.dst + up.first, slots.data(), up.second * sizeof(sgx::RingPair));
        if (up.occupied > capacity) { rc = fail("occupied beyond capacity", (long long)up.occupied); break; }
        for (size_t i = 0; i < up.occupied; ++i) {             // everything resident is the accepted stream, in order
            const float want = (float)((expect + i) & 0xffffff);
            if (image[i].l != want || image[i].r != -want) { rc = fail("pair out of order", (long long)(expect + i), (long long)image[i].l); break; }
        }
        if (rc) break;
        size_t frames = up.occupied < W ? 0 : (up.occupied - W) / H + 1;
        const bool truncated = frames > 3;                     // max_frames = 3: some ticks end early
        if (truncated) frames = 3;
        const size_t skip = sgx::LiveRingState::skip_of(frames, H, reference_skip, truncated, up.occupied);
        const size_t keep = up.occupied - skip;
        if (skip && keep) { std::memcpy(alt.data(), image.data() + skip, keep * sizeof(sgx::RingPair)); image.swap(alt); }
        ring.end_tick(up, skip);
        expect += skip;
        frames_total += frames;
        ++ticks;
        if (frames == 0 && !last) std::this_thread::yield();     // (nothing to do: let the producer run)
        if (last && frames == 0 && (!reference_skip || up.occupied == 0 || skip == 0)) break;
        if (last && frames == 0 && ring.occupied() < W && !reference_skip) break;
    }
    producer.join();
    if (rc) return rc;
    const unsigned long long acc = accepted.load(), off = offered.load();
    if (acc > off || acc == 0) return fail("accepted count", (long long)acc, (long long)off);
    if (expect + ring.occupied() != acc) return fail("pairs lost or duplicated", (long long)(expect + ring.occupied()), (long long)acc);
    std::printf("ring ok (reference_skip %d): %llu pairs offered, %llu accepted (%llu dropped on overflow), %llu frames in %llu ticks, ring wrapped %llu times\n",
                (int)reference_skip, off, acc, off - acc, frames_total, ticks, acc / capacity);
    return 0;
}

static void grad(double t, uint8_t rgb[3], void *)
{
    const double c = t < 0 ? 0 : (t > 1 ? 1 : t);
    rgb[0] = (uint8_t)(255.0 * c); rgb[1] = (uint8_t)(255.0 * (1.0 - c)); rgb[2] = (uint8_t)(128.0 + 127.0 * c * c);
}

static int run_tables()
{
    const uint32_t windows[] = {86, 128, 256, 512, 751, 1024, 1102, 2048, 2205, 2400, 4096, 4410, 4800, 8192, 9600};
    const uint32_t rates[] = {8000, 22050, 44100, 48000, 96000, 192000};
    size_t built = 0;
    for (uint32_t W : windows)
        for (uint32_t sr : rates)
            for (uint32_t interp = 0; interp < 2; ++interp)
                for (uint32_t R : {7u, 1024u, 2048u}) {
                    sgx::Tables t;
                    sgx::build_tables(W, R, sr, 32.0, 22030.0, interp, t);
                    if (t.window.size() != W || t.edges.size() != R + 1 || t.rows.size() != R) return fail("table sizes", W, R);
                    size_t n = 0;
                    for (const auto &r : t.rows) {
                        if (r.first != n || r.count == 0) return fail("row table", W, (long long)n);
                        n += r.count;
                    }
                    if (n != t.samples.size()) return fail("sample table", W, (long long)n);
                    for (const auto &s : t.samples)
                        if (s.i0 < 0 || s.i0 > (int32_t)W - 2) return fail("tap index", W, s.i0);
                    ++built;
                }
    // axes far outside the spectrum, one row, tiny windows
    for (auto fr : {std::pair<double, double>{1.0, 90000.0}, {20.0, 24000.0}, {100.0, 101.0}}) {
        sgx::Tables t;
        sgx::build_tables(2048, 300, 48000, fr.first, fr.second, 0, t);
        sgx::build_tables(8, 1, 48000, fr.first, fr.second, 1, t);
        ++built;
    }
    // SpectrumAnalyzer ranges
    {
        std::vector<float> f0, f1;
        for (int i = 0; i < 128; ++i) { f0.push_back(32.0f * (float)(i + 1)); f1.push_back(32.0f * (float)(i + 2)); }
        std::vector<sgx::RowEntry> rows;
        std::vector<sgx::SampleEntry> samples;
        for (uint32_t W : windows) sgx::build_range_tables(W, 48000, 0, f0.data(), f1.data(), 128, rows, samples);
        if (rows.size() != 128) return fail("range rows", (long long)rows.size());
    }
    // palettes: a 256-entry table in every LUT mode, mono and diverging; a callback gradient bisected into segments
    for (int stereo = 0; stereo < 2; ++stereo)
        for (uint32_t mode = 0; mode < 3; ++mode) {
            sgx::Palette pal;
            pal.n = 256;
            pal.stereo = stereo;
            pal.rgb.resize(256 * 3);
            for (int i = 0; i < 256; ++i) grad(i / 255.0, &pal.rgb[3 * i], nullptr);
            sgx::build_palette_thresholds(-70.0f, -10.0f, mode, pal);
            if (!stereo && pal.lut_thr.size() != 255) return fail("lut thresholds", (long long)pal.lut_thr.size());
            if (stereo && pal.alpha_thr.size() != 255) return fail("alpha thresholds", (long long)pal.alpha_thr.size());
            for (size_t i = 1; i < pal.lut_thr.size(); ++i)
                if (!(pal.lut_thr[i] >= pal.lut_thr[i - 1])) return fail("thresholds not monotone", (long long)i);
        }
    for (int stereo = 0; stereo < 2; ++stereo) {
        sgx::Palette pal;
        pal.segments = true;
        pal.fn = grad;
        pal.stereo = stereo;
        sgx::build_palette_segments(-70.0f, -10.0f, pal);
        if (pal.n == 0 || pal.rgb.size() != (size_t)pal.n * 3) return fail("segments", pal.n);
    }
    std::printf("tables ok: %zu configurations\n", built);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc == 2 && !std::strcmp(argv[1], "ring")) return run_ring(false) || run_ring(true);
    if (argc == 2 && !std::strcmp(argv[1], "tables")) return run_tables();
    std::fprintf(stderr, "usage: host_sanitize ring|tables\n");
    return 2;
}
