// sgx_tables.cpp -- host-side, once-per-context evaluation of everything on the pixel path that
// depends only on the configuration (not on the audio): Hann table, twiddles, the log-frequency
// row edges, magnitude_in's sample positions and interpolation weights, and the power
// thresholds that replace log10 on the device.
//
// The arithmetic follows the reference's f32/f64 operation order (citations inline; file:line
// in the reference repository).  This file must be compiled without FMA contraction
// (-ffp-contract=off): every float operation below rounds exactly once, as rustc emits it.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "sgx_internal.hpp"

namespace sgx {

namespace {

// Rust `x as usize` / `as i32` for floats: truncate toward zero, saturate, NaN -> 0.
inline int64_t f32_as_index(float v)
{
    if (!(v > 0.0f)) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    return (int64_t)v;
}

// interpolated_frequency_sample.rs:52-54: 2.0 * len as f32 / sample_rate as f32
inline float period_of(uint32_t M, uint32_t sr) { return (2.0f * (float)M) / (float)sr; }

// :24-31: (frequency * period).clamp(0.0, (len - 1) as f32)
inline float index_of(float f, float period, uint32_t M)
{
    float idx = f * period;
    const float hi = (float)(M - 1);
    if (idx < 0.0f) idx = 0.0f;
    if (idx > hi) idx = hi;
    return idx;
}

}  // namespace

void build_tables(uint32_t W, uint32_t R, uint32_t sr, double f_min, double f_max, uint32_t interp, Tables &out)
{
    const uint32_t P = 2 * W;

    // fft.rs:61 -- 0.5 * (1.0 - ((TAU * i as f32) / (W as f32)).cos()), all f32, product first.
    out.window.resize(W);
    {
        const float tau = 6.28318530717958647692528676655900577f;
        const float wf = (float)W;
        for (uint32_t i = 0; i < W; ++i) {
            float prod = tau * (float)i;
            float q = prod / wf;
            float c = cosf(q);
            float om = 1.0f - c;
            out.window[i] = 0.5f * om;
        }
    }

    // Forward twiddles e^{-2 pi i j / P}, j < P, rounded from double (FFTW computes its
    // twiddles in extended precision too; fft.rs:20-24 Sign::Forward).
    // The full circle: a radix-4 stage uses w^j, w^2j and w^3j.
    out.twiddle.resize(P);
    for (uint32_t j = 0; j < P; ++j) {
        double ang = -2.0 * M_PI * (double)j / (double)P;
        double c = cos(ang), s = sin(ang);
        if (j == 0) { c = 1.0; s = 0.0; }
        if (4 * j == P) { c = 0.0; s = -1.0; }
        if (2 * j == P) { c = -1.0; s = 0.0; }
        if (4 * j == 3 * P) { c = 0.0; s = 1.0; }
        out.twiddle[j] = make_float2((float)c, (float)s);
    }

    // log_scaling.rs:160-191 (linear = ln(start)..ln(end), zero_point 0) and :114-119 (unmap),
    // through plotters' RangedCoordf64::unmap: (hi - lo) * ((p - 0) / (R - 0)) + lo; exp; the
    // cast to f32 is simple_spectrogram.rs:145.
    out.edges.resize(R + 1);
    {
        const double lo = log(f_min), hi = log(f_max);
        for (uint32_t p = 0; p <= R; ++p) {
            double off = (double)p / (double)R;
            double lin = (hi - lo) * off + lo;
            out.edges[p] = (float)exp(lin);
        }
    }

    // interpolated_frequency_sample.rs:60-75 for every row, :79-105 for the weights.
    std::vector<float> f0(R), f1(R);
    for (uint32_t py = 0; py < R; ++py) { f0[py] = out.edges[py]; f1[py] = out.edges[py + 1]; }
    build_range_tables(W, sr, interp, f0.data(), f1.data(), R, out.rows, out.samples);
}

void build_range_tables(uint32_t W, uint32_t sr, uint32_t interp, const float *range_f0, const float *range_f1, uint32_t n_ranges,
                        std::vector<RowEntry> &rows, std::vector<SampleEntry> &samples)
{
    // FrequencySample::magnitude_in(f0..f1) (interpolated_frequency_sample.rs:60-75) for each range:
    // sample count, lin_space positions, fractional indices and interpolation weights (:79-105)
    const uint32_t M = W - 1;
    rows.resize(n_ranges);
    samples.clear();
    const float period = period_of(M, sr);
    const float pi = 3.14159265358979323846264338327950288f;
    for (uint32_t py = 0; py < n_ranges; ++py) {
        const float f0 = range_f0[py], f1 = range_f1[py];
        const float i0 = index_of(f0, period, M), i1 = index_of(f1, period, M);
        float d = i1 - i0;
        int64_t n = f32_as_index(floorf(d));
        if (n < 1) n = 1;
        RowEntry re;
        re.first = (uint32_t)samples.size();
        re.count = (uint32_t)n;
        re.count_f = (float)n;
        re.pad = 0;
        rows[py] = re;
        // iter_num_tools lin_space over the half-open range: step = (end - start) / n,
        // x_i = start + i * step
        const float span = f1 - f0;
        const float step = span / (float)n;
        for (int64_t i = 0; i < n; ++i) {
            float off = (float)i * step;
            float f = f0 + off;
            float idx = index_of(f, period, M);
            SampleEntry se;
            std::memset(&se, 0, sizeof(se));
            float fl = floorf(idx);
            if (interp == SGX_INTERP_COSINE) {
                int64_t low = f32_as_index(fl);
                int64_t high = f32_as_index(ceilf(idx));
                if (high < low + 1) high = low + 1;
                if (high > (int64_t)M - 1) high = (int64_t)M - 1;
                float o = idx - (float)low;
                float ang = o * pi;
                float c = cosf(ang);
                float om = 1.0f - c;
                float o2 = om / 2.0f;
                se.i0 = (int32_t)low;
                se.i1 = (int32_t)high;
                se.w1 = 1.0f - o2;
                se.w2 = o2;
            } else {
                float mu = idx - fl;
                float mu2 = mu * mu;   // num_traits::pow(mu, 2)
                float mu3 = mu * mu2;  // num_traits::pow(mu, 3) = mu * (mu * mu)
                se.i0 = (int32_t)f32_as_index(fl);
                se.w0 = mu;
                se.w1 = mu2;
                se.w2 = mu3;
            }
            samples.push_back(se);
        }
    }
}

int lut_index_host(double t, uint32_t n, uint32_t mode)
{
    double x = (mode == SGX_LUT_ROUND_NM1) ? floor(t * (double)(n - 1) + 0.5) : floor(t * (double)n);
    if (!(x > 0.0)) return 0;  // NaN, negatives: Rust's saturating `as usize`
    if (x >= (double)n) return (int)n - 1;
    return (int)x;
}

uint8_t alpha_u8_host(float alpha)
{
    // simple_spectrogram.rs:159: (alpha * 255.0) as u8
    float v = alpha * 255.0f;
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

float bounded_db_host(float min_db, float max_db, float power)
{
    // colorscheme.rs:60-61
    float arg = power + 1e-7f;
    float db = 10.0f * log10f(arg);
    float num = db - min_db;
    float den = max_db - min_db;
    return num / den;
}

namespace {

// Smallest non-negative float p (by bit pattern, +inf included) with level(p) >= want; +inf's
// successor (a NaN pattern) is returned as +inf when no finite or infinite p reaches it.
template <typename F>
float first_power_reaching(int want, F level)
{
    uint32_t lo = 0, hi = 0x7f800000u;  // [+0, +inf]
    auto at = [&](uint32_t bits) {
        float p;
        std::memcpy(&p, &bits, 4);
        return level(p);
    };
    if (at(hi) < want) return std::numeric_limits<float>::quiet_NaN();  // unreachable level: never matches
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (at(mid) >= want) hi = mid;
        else lo = mid + 1;
    }
    float p;
    std::memcpy(&p, &lo, 4);
    return p;
}

}  // namespace

void build_palette_thresholds(float min_db, float max_db, uint32_t lut_mode, Palette &pal)
{
    // color_for (colorscheme.rs:55-71) is a monotone step function of the power l^2 + r^2.  The
    // device counts how many of these thresholds the power has reached instead of evaluating
    // log10f, so the byte it writes is the one the host's libm would have chosen.
    const uint32_t n = pal.n;
    pal.lut_thr.assign(n > 0 ? n - 1 : 0, 0.0f);
    for (uint32_t i = 1; i < n; ++i)
        pal.lut_thr[i - 1] = first_power_reaching((int)i, [&](float p) {
            return lut_index_host((double)bounded_db_host(min_db, max_db, p), n, lut_mode);
        });
    pal.alpha_thr.assign(255, 0.0f);
    for (int a = 1; a <= 255; ++a)
        pal.alpha_thr[a - 1] =
            first_power_reaching(a, [&](float p) { return (int)alpha_u8_host(bounded_db_host(min_db, max_db, p)); });
}


namespace {

inline bool same_rgb(const uint8_t a[3], const uint8_t b[3]) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; }
inline float bits_f32(uint32_t b) { float v; std::memcpy(&v, &b, 4); return v; }
inline uint32_t f32_bits(float v) { uint32_t b; std::memcpy(&b, &v, 4); return b; }

}  // namespace

void build_palette_segments(float min_db, float max_db, Palette &pal)
{
    // colour as a function of the power l^2 + r^2 (mono branch of color_for, colorscheme.rs:59-61,69):
    // a step function of bytes.  Sample it on a grid that is finer than its steps, then bisect over
    // float bit patterns inside every grid cell whose ends differ: the thresholds are the exact
    // switch points of  eval(bounded_dB(power))  as the host evaluates it.
    const int G = 8192;
    auto color_at = [&](float power, uint8_t c[3]) { pal.fn((double)bounded_db_host(min_db, max_db, power), c, pal.fn_user); };
    std::vector<uint32_t> grid;
    grid.push_back(0u);
    for (int g = 0; g <= G; ++g) {
        const double t = -0.02 + 1.04 * (double)g / (double)G;
        const double db = t * ((double)max_db - (double)min_db) + (double)min_db;
        double pw = pow(10.0, db / 10.0) - 1e-7;
        if (pw < 0.0) pw = 0.0;
        grid.push_back(f32_bits((float)pw));
    }
    grid.push_back(0x7f800000u);  // +inf
    std::sort(grid.begin(), grid.end());
    grid.erase(std::unique(grid.begin(), grid.end()), grid.end());

    pal.rgb.clear();
    pal.lut_thr.clear();
    uint8_t cur[3], cb[3], cm[3];
    color_at(bits_f32(grid[0]), cur);
    pal.rgb.insert(pal.rgb.end(), cur, cur + 3);
    for (size_t i = 0; i + 1 < grid.size(); ++i) {
        uint32_t lo = grid[i];
        const uint32_t b = grid[i + 1];
        color_at(bits_f32(b), cb);
        while (!same_rgb(cur, cb)) {
            uint32_t l = lo, h = b;  // colour(l) == cur, colour(h) != cur
            while (h - l > 1) {
                const uint32_t m = l + (h - l) / 2;
                color_at(bits_f32(m), cm);
                if (same_rgb(cm, cur)) l = m;
                else h = m;
            }
            color_at(bits_f32(h), cur);
            pal.lut_thr.push_back(bits_f32(h));
            pal.rgb.insert(pal.rgb.end(), cur, cur + 3);
            lo = h;
            if (pal.rgb.size() / 3 > 60000) break;  // not a byte-valued gradient
        }
    }
    pal.n = (uint32_t)(pal.rgb.size() / 3);
    pal.fn(std::numeric_limits<double>::quiet_NaN(), pal.nan_rgb, pal.fn_user);

    // alpha byte thresholds (diverging branch, simple_spectrogram.rs:159) -- as in the LUT mode
    pal.alpha_thr.assign(255, 0.0f);
    for (int a = 1; a <= 255; ++a)
        pal.alpha_thr[a - 1] =
            first_power_reaching(a, [&](float p) { return (int)alpha_u8_host(bounded_db_host(min_db, max_db, p)); });

    // diverging branch: colour as a function of t = l / (|l| + |r|) in [-1, 1] (colorscheme.rs:65-66)
    pal.t_thr.clear();
    if (pal.stereo) {
        std::vector<uint8_t> seg;
        auto col_t = [&](double t, uint8_t c[3]) { pal.fn(t, c, pal.fn_user); };
        col_t(-1.0, cur);
        seg.insert(seg.end(), cur, cur + 3);
        for (int g = 0; g < G; ++g) {
            double lo = -1.0 + 2.0 * (double)g / (double)G;
            const double b = -1.0 + 2.0 * (double)(g + 1) / (double)G;
            col_t(b, cb);
            while (!same_rgb(cur, cb)) {
                double l = lo, h = b;
                for (;;) {
                    const double m = l + (h - l) * 0.5;
                    if (!(m > l) || !(m < h)) break;  // adjacent doubles
                    col_t(m, cm);
                    if (same_rgb(cm, cur)) l = m;
                    else h = m;
                }
                col_t(h, cur);
                pal.t_thr.push_back(h);
                seg.insert(seg.end(), cur, cur + 3);
                lo = h;
                if (seg.size() / 3 > 60000) break;
            }
        }
        pal.rgb = seg;  // the diverging branch indexes colours by balance, not by power
        pal.n = (uint32_t)(seg.size() / 3);
        pal.lut_thr.assign(pal.n > 0 ? pal.n - 1 : 0, std::numeric_limits<float>::quiet_NaN());
    }
}

}  // namespace sgx
