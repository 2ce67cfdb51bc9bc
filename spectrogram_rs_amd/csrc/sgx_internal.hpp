// sgx_internal.hpp -- context layout and kernel launchers shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <array>
#include <string>
#include <vector>

#include "../../include/sgx.h"

namespace sgx {

// interpolated_frequency_sample.rs:89-105 on an (l, r) / (frame A, frame B) pair: both components in packed f32 instructions --
// v_pk_add_f32 / v_pk_mul_f32 round each half as the scalar instructions do (the library is built with -ffp-contract=off), so the bits
// are those of the two scalar evaluations; on the fused mono path 4-5 % of config 3's cubic leg (profiles/r05_pixel_lds.txt)
#ifdef __HIPCC__
typedef float sgx_f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 cubic_pair(float2 y0, float2 y1, float2 y2, float2 y3, float mu, float mu2, float mu3)
{
#if defined(SGX_PK_PAIR) && !SGX_PK_PAIR   // (A/B: the scalar form)
    float2 v;
    { const float a0 = ((y3.x - y2.x) - y0.x) + y1.x, a1 = (y0.x - y1.x) - a0, a2 = y2.x - y0.x; v.x = ((a0 * mu3) + (a1 * mu2)) + ((a2 * mu) + y1.x); }
    { const float a0 = ((y3.y - y2.y) - y0.y) + y1.y, a1 = (y0.y - y1.y) - a0, a2 = y2.y - y0.y; v.y = ((a0 * mu3) + (a1 * mu2)) + ((a2 * mu) + y1.y); }
    return v;
#endif
    const sgx_f2v Y0 = {y0.x, y0.y}, Y1 = {y1.x, y1.y}, Y2 = {y2.x, y2.y}, Y3 = {y3.x, y3.y};
    const sgx_f2v a0 = ((Y3 - Y2) - Y0) + Y1;
    const sgx_f2v a1 = (Y0 - Y1) - a0;
    const sgx_f2v a2 = Y2 - Y0;
    const sgx_f2v r = ((a0 * mu3) + (a1 * mu2)) + ((a2 * mu) + Y1);
    return make_float2(r.x, r.y);
}
#endif

// The grid over the balance t = l / (|l| + |r|) that shortens the search for its colour segment: a monotone cell number,
// the same IEEE double operations on the host (table) and on the device (lookup).
constexpr int kTCells = 512;
__host__ __device__ inline int sgx_t_cell(double t)
{
    const double x = (t + 1.0) * (double)(kTCells / 2);
    return x >= (double)kTCells ? kTCells - 1 : (x > 0.0 ? (int)x : 0);
}

// One row of the pixel column (py = 0 is the LOWEST frequency; it is written at image row R-1-py).
struct RowEntry {
    uint32_t first;  // index of the row's first entry in the sample table
    uint32_t count;  // n = max(1, floor(i1 - i0))           interpolated_frequency_sample.rs:63-64
    float count_f;   // n as f32, the divisor of the mean     :72
    uint32_t pad;
};

// One sample of magnitude_in's lin_space.  Everything that depends only on the configuration is
// evaluated once on the host with the reference's f32 operation order; the kernel is left with
// gathers and exactly-rounded adds / multiplies.
//   cubic  (:89-105): i0 = x1 (floor(index)), w = {mu, mu^2, mu^3}
//   cosine (:79-86) : i0 = low, i1 = high,    w = {1 - o', o'}
struct SampleEntry {
    int32_t i0;
    union { int32_t i1; float w0; };
    float w1;
    float w2;
};
static_assert(sizeof(SampleEntry) == 16, "SampleEntry must be 16 bytes");

struct Tables {
    std::vector<float> window;      // [W]      fft.rs:61
    std::vector<float2> twiddle;    // [P]      e^{-2 pi i j / P}, the full circle
    std::vector<float> edges;       // [R+1]    log_scaling.rs:114-119
    std::vector<RowEntry> rows;     // [R]
    std::vector<SampleEntry> samples;
};

struct Palette {
    std::vector<uint8_t> rgb;       // [n][3]
    uint32_t n = 0;
    int stereo = 0;
    std::vector<float> lut_thr;     // [n-1]   smallest power whose LUT index is >= i+1 (mono)
    std::vector<float> alpha_thr;   // [255]   smallest power whose alpha byte is >= i+1 (stereo)
    // continuous gradient given as a callback (sgx_set_gradient_fn): rgb[] then holds one colour per
    // constant SEGMENT of the colour function instead of one per uniform LUT step
    bool segments = false;
    sgx_gradient_fn fn = nullptr;
    void *fn_user = nullptr;
    std::vector<double> t_thr;      // stereo + segments: smallest t (as double) at which segment i+1 starts
    std::vector<uint16_t> t_cell;   // [kTCells + 1]: t_cell[c] = how many switch points fall in cells below c of the grid over t in [-1, 1] (sgx_t_cell)
    uint8_t nan_rgb[3] = {0, 0, 0}; // colour of t = NaN (l = r = 0 in the diverging branch)
};

void build_tables(uint32_t W, uint32_t R, uint32_t sample_rate_u32, double f_min, double f_max, uint32_t interp,
                  Tables &out);
void build_range_tables(uint32_t W, uint32_t sample_rate_u32, uint32_t interp, const float *range_f0, const float *range_f1,
                        uint32_t n_ranges, std::vector<RowEntry> &rows, std::vector<SampleEntry> &samples);
void build_palette_thresholds(float min_db, float max_db, uint32_t lut_mode, Palette &pal);
void build_palette_segments(float min_db, float max_db, Palette &pal);
int lut_index_host(double t, uint32_t n, uint32_t mode);
uint8_t alpha_u8_host(float alpha);
float bounded_db_host(float min_db, float max_db, float power);

}  // namespace sgx

struct sgx_ctx {
    sgx_config cfg{};
    uint32_t W = 0, P = 0, M = 0, H = 0, C = 0, pairs = 0, R = 0, sr_u32 = 0, logP = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    int stft_kernel = 0;  // 0 generic, 1 tuned 4096 wave-per-transform, 2 tuned 4096 workgroup-per-transform (scalar codelets), 3 the same with packed (re, im) arithmetic, 4 Bluestein (2W not a power of two), 5 tuned 16384, second design (four 4096-point residues), 6 mixed radix (2W = 2^a 3^b 5^c 7^d), 7 tuned 16384, first design (whole transform in LDS), 8 tuned 16384, third design (four time-decimated 4096-point transforms per quad of lanes; removed in round 6), 9 tuned 4800 (W = 2400; more than two channels: 6), 10 tuned 16384, fourth design (32 x 32 x 16, 512 threads: stft16384_w.hip)

    sgx::Tables tab;
    sgx::Palette pal;

    // device tables
    float *d_window = nullptr;
    float2 *d_twiddle = nullptr;
    sgx::RowEntry *d_rows = nullptr;
    sgx::SampleEntry *d_samples = nullptr;
    float *d_lut_thr = nullptr;    // [n-1]
    float *d_alpha_thr = nullptr;  // [255]
    uchar4 *d_lut_rgba = nullptr;  // [n]
    uint2 *d_pal_seed = nullptr;   // [256] {threshold to leave level i (NaN for 255), RGBA of level i}: 256-level palettes, mono branch
    double *d_t_thr = nullptr;     // [n-1], segment palettes with a diverging scheme only
    uint16_t *d_t_cell = nullptr;  // [kTCells + 1] (or null)
    void *d_fast = nullptr;        // tables of the wave-per-transform kernel (opaque here)
    void *d_fast_wg = nullptr;     // tables of the workgroup-per-transform kernel
    void *d_blu = nullptr;         // tables of the Bluestein (non-power-of-two) kernel
    void *d_mix = nullptr;         // tables of the mixed-radix (2, 3, 5, 7-smooth lengths) kernel
    void *d_w4800 = nullptr;       // tables of the tuned 4800-point kernel (W = 2400: the application's window at 48 kHz)
    void *d_real = nullptr;        // tables of the real-input 4096-point kernel (independent mono frames at W 2048 / H 256)
    void *d_chz = nullptr;         // chirp-z through the mixed-radix kernel's stages (or null: the radix-4 ladder of stft_bluestein.hip)
    void *d_w16k = nullptr;        // tables of the 16384-point kernel, 32 x 32 x 16 (stft16384_w.hip)

    // workspaces (grown on demand, kept)
    float *d_ws_mags = nullptr;
    size_t ws_frames = 0;
    float *d_one_in = nullptr, *d_one_out = nullptr;
    unsigned long long *d_cksum = nullptr;
    // cached tables of the last sgx_magnitude_in range set
    std::vector<float> bands_key;
    sgx::RowEntry *d_band_rows = nullptr;
    sgx::SampleEntry *d_band_samples = nullptr;
    float *d_levels = nullptr;  // sgx_spectrum_levels: the bands' (l, r) means, grown on demand
    uint32_t levels_cap = 0;

    unsigned long long palette_gen = 0;  // bumped by every palette upload (sgx_view rebuilds its palette texture on a change)

    // device limits, read once at sgx_create (not on the per-tick latency path)
    int n_cu = 256;
    size_t lds_optin = 64 * 1024;   // hipDeviceAttributeSharedMemPerBlockOptin: the largest LDS image a workgroup may ask for
    // resident workgroups per CU of a kernel instantiation at a block size and LDS image (hipOccupancyMaxActiveBlocksPerMultiprocessor, cached)
    mutable std::vector<std::pair<std::array<size_t, 3>, int>> occupancy_cache;
    // objects that hold a pointer into this context (sgx_view): sgx_destroy detaches them, their calls then fail cleanly
    std::vector<struct sgx_view *> views;
    std::vector<struct sgx_image *> images;   // sgx_image.hip: the image rings created on this context

    std::string err;
};

namespace sgx {

inline bool fast4096_supported(const sgx_ctx *c) { return c->W == 2048; }   // the tuned 4096-point kernels (stft4096_wg.hip)
// kernel launchers (each returns hipSuccess or the launch error)
hipError_t launch_stft_generic(const sgx_ctx *c, const float *d_pcm, uint32_t channels, uint32_t pairs, size_t first_frame,
                               size_t n_frames, size_t total_frames, float *d_mags);
hipError_t wg4096_init(sgx_ctx *c, void **out);
void wg4096_destroy(void *tables);
hipError_t launch_stft_wg4096(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                              size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags);
bool wg4096_can_fuse_render(const sgx_ctx *c, const void *tables);
hipError_t launch_deinterleave_pairs(const sgx_ctx *c, const float *d_pcm, float *d_planes, size_t plane_floats, size_t first_sample, size_t n_samples,
                                     uint32_t channels, uint32_t pairs);   // deinterleave.hip
bool wg4096_seed_is_within_one(const sgx_ctx *c);
void lut_seed_coefficients(const sgx_ctx *c, float &a, float &b);
namespace wg { bool seed_within_one(const std::vector<float> &thr, double guess_a, double guess_b); }   // stft4096_wg.hip: is floor(log2(p + 1e-7) a + b) within one of the threshold count for every power?   // LUT level ~ floor(log2(power + 1e-7) a + b): the seed of the threshold count
hipError_t launch_render_wg4096(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                                size_t first_frame, size_t n_frames, size_t total_frames, uint8_t *d_rgba);
// stft16384_w.hip: W = 8192 as 32 x 32 x 16 in one 512-thread workgroup, 32 points per thread
bool w16384_supported(const sgx_ctx *c);
hipError_t w16384_init(sgx_ctx *c, void **out);
void w16384_destroy(void *tables);
hipError_t launch_stft_w16384(const sgx_ctx *c, void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                              size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags);
// W = 2400 (48 kHz x 0.05 s): persistent 320-thread workgroups, 16 x 20 x 15 (stft4800_wg.hip); rows and half rows of one or two channels -- more
// channels and the fused PCM-to-pixel path go to the composite-radix kernel, whose tables such a context carries too
// stft4096_real.hip: independent mono frames at W 2048 / H 256 as 2048-point complex transforms of the real frame
hipError_t real4096_init(sgx_ctx *c, void **out);
void real4096_destroy(void *tables);
bool real4096_serves(const sgx_ctx *c, const float *d_pcm, uint32_t channels);   // (launched from stft4096_wg.hip: launch_wg)
bool w4800_supported(const sgx_ctx *c);
hipError_t w4800_init(sgx_ctx *c, void **out);
void w4800_destroy(void *tables);
hipError_t launch_stft_w4800(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, size_t first_frame, size_t n_frames,
                             size_t total_frames, float *d_mags, bool out_f16);
bool mixed_supported(uint32_t W);
hipError_t mixed_init(sgx_ctx *c, void **out);
void mixed_destroy(void *tables);
uint32_t mixed_fixed_plan(const void *tables);
bool mixed_real_serves(const sgx_ctx *c, const void *tables, uint32_t channels);   // real-input mode: a mono stream, every frame its own W-point transform
// chirp-z through the composite stages of the mixed-radix kernel (stft_mixed.hip): L = 512 .. 16384, i.e. W = 86 .. 5461
bool chirpz_supported(uint32_t W);
hipError_t chirpz_init(sgx_ctx *c, void **out);
void chirpz_destroy(void *tables);
bool chirpz_real_serves(const sgx_ctx *c, const void *tables, uint32_t channels);   // real-input mode, as mixed_real_serves
hipError_t launch_stft_chirpz(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                              size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags);
bool mixed_can_fuse_render(const sgx_ctx *c, const void *tables);   // one kernel from PCM to pixels at this length, palette and row table
hipError_t launch_render_mixed(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs, size_t first_frame,
                               size_t n_frames, size_t total_frames, uint8_t *d_rgba);   // the length whose compile-time plan serves this context, or 0 (run-time geometry)
hipError_t launch_stft_mixed(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                             size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags, bool out_f16 = false);   // out_f16: (l, r) half pairs, 4 B per bin
bool bluestein_supported(uint32_t W);
hipError_t bluestein_init(sgx_ctx *c, void **out);
void bluestein_destroy(void *tables);
hipError_t launch_stft_bluestein(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                                 size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags);
hipError_t launch_stft_wg4096_f16(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                                  size_t first_frame, size_t n_frames, size_t total_frames, void *d_mags_f16);
hipError_t launch_to_half(const sgx_ctx *c, const float *d_in, void *d_out, size_t n_pairs);
hipError_t launch_render(const sgx_ctx *c, const float *d_mags, size_t n_columns, uint8_t *d_rgba);
hipError_t launch_magnitude_in(const sgx_ctx *c, const float *d_mags, size_t n_columns, const RowEntry *d_rows,
                               const SampleEntry *d_samples, uint32_t n_ranges, float *d_out);
void detach_views(sgx_ctx *c);   // sgx_view.hip: every live view of the context forgets it
void detach_images(sgx_ctx *c);   // sgx_image.hip
sgx_ctx *image_context(const struct sgx_image *im);
sgx_ctx *view_context(const struct sgx_view *v);   // sgx_view.hip: the context a view was created on (nullptr once that context is gone)
hipError_t launch_white_noise(const sgx_ctx *c, float *d_out, uint64_t first, size_t n, uint32_t channels, uint32_t seed);
hipError_t launch_checksum(const sgx_ctx *c, const uint32_t *d_words, size_t n_words, uint64_t base_word,
                           unsigned long long *d_acc);

}  // namespace sgx

#ifdef __HIPCC__
// a (left, right) magnitude pair of an output row: written once, never read by the kernel
__device__ __forceinline__ void st_stream(float2 *p, float a, float b)
{
#ifdef SGX_GEN_NT
    typedef float f2s __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(f2s{a, b}, reinterpret_cast<f2s *>(p));
#else
    *p = make_float2(a, b);
#endif
}
#endif
