// sgx_kernels.hip -- gfx950 kernels that are not the tuned 4096-point STFT:
//   * stft_generic_kernel : any power-of-two transform length up to 16384 (one workgroup per
//                           frame and channel pair, the whole padded frame resident in LDS)
//   * render_kernel       : magnitude_in + color_for + put_pixel for one pixel column
//   * white-noise generator and checksum (harness helpers)
//
// Compiled with -ffp-contract=off: the pixel path must round every f32 operation exactly once
// so that its bytes equal the CPU restatement's; FFT code asks for FMAs explicitly.
#include <hip/hip_fp16.h>

#include "lds_fft.hpp"
#include "sgx_internal.hpp"

namespace sgx {

// ------------------------------------------------------------------------------------------------
// generic power-of-two STFT
// ------------------------------------------------------------------------------------------------

struct StftGenericParams {
    const float *pcm;      // [n][C]
    const float *window;   // [W]
    const float2 *twiddle; // [P]   e^{-2 pi i j / P}
    float *mags;           // [F][pairs][M][2]
    unsigned long long first_frame;
    // mono pairs (frames 2q and 2q+1 ride in the real and imaginary part of one transform, as in stft4096_wg.hip):
    unsigned long long pair_base, n_frames, total_frames;
    uint32_t mono_pairs;
    uint32_t W, logP, H, C, pairs;
    float half_scale;      // applied as (hypot * 0.5f) * scale
    float scale;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    // (a.x + i a.y)(b.x + i b.y) with two FMAs
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}

// Replaces FastFourierTransform::process (fft.rs:43-99) for one (frame, pair).
// Data flow: PCM -> (l + i r) * hann -> LDS; the zero padding is never materialised: with
// z[n] = 0 for n >= W the first decimation-in-frequency stage degenerates to
//   s[n] = z[n], s[n + W] = z[n] * w_P^n,
// after which the two halves are length-W problems: in-place radix-4 DIF stages leave F digit-reversed.
__global__ void __launch_bounds__(1024) stft_generic_kernel(StftGenericParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const uint32_t W = p.W, P = 2 * W, M = W - 1;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint32_t pair = blockIdx.y;
    // (l, r) of one frame -- or, for a mono stream, frames 2q and 2q+1 by GLOBAL index: the split that separates left
    // from right then separates the two frames, each is written as (m, m), and any sub-range writes the same bytes
    long long row_a, row_b = -1;   // output rows (frames relative to first_frame); < 0 or >= n_frames: not stored
    const float *src_a, *src_b;
    uint32_t cl, cr;
    bool data_b = true;
    if (p.mono_pairs) {
        const unsigned long long fa = 2 * (p.pair_base + blockIdx.x), fb = fa + 1;
        row_a = (long long)fa - (long long)p.first_frame;
        row_b = row_a + 1;
        data_b = fb < p.total_frames;
        src_a = p.pcm + (size_t)(fa * p.H);
        src_b = data_b ? src_a + p.H : src_a;
        cl = cr = 0;
    } else {
        row_a = (long long)blockIdx.x;
        src_a = src_b = p.pcm + (size_t)((p.first_frame + blockIdx.x) * p.H) * p.C;
        cl = p.C == 1 ? 0 : 2 * pair;
        cr = p.C == 1 ? 0 : 2 * pair + 1;
    }

    for (uint32_t n = tid; n < W; n += nt) {
        const float w = p.window[n];
        const float l = src_a[(size_t)n * p.C + cl];
        const float r = data_b ? src_b[(size_t)n * p.C + cr] : 0.0f;
        const float2 z = make_float2(l * w, r * w);  // complex * real, fft.rs:59-63
        s[n] = z;
        s[n + W] = cmul(z, p.twiddle[n]);
    }
    __syncthreads();

    // the two halves are independent length-W transforms now (even bins / odd bins): radix-4 stages in place
    // (lds_fft.hpp), each half's result in digit-reversed order
    if (p.logP >= 2) ldsfft::forward_dif(s, p.logP, p.logP - 1, p.twiddle, p.logP, tid, nt);

    // fft.rs:81-98: a = F[k], b = F[P - k]; left = |a + conj b| / 2, right = |a - conj b| / 2; * 2/W
    const bool st_a = row_a >= 0 && (unsigned long long)row_a < p.n_frames;
    const bool st_b = p.mono_pairs && row_b >= 0 && (unsigned long long)row_b < p.n_frames;
    float2 *out_a = reinterpret_cast<float2 *>(p.mags) + ((size_t)(st_a ? row_a : 0) * p.pairs + pair) * M;
    float2 *out_b = reinterpret_cast<float2 *>(p.mags) + ((size_t)(st_b ? row_b : 0) * p.pairs + pair) * M;
    // bin k: half k & 1 (the pruned first stage), then the digit-reversed position of k >> 1 inside it
    const uint32_t logW = p.logP - 1;
    for (uint32_t j = tid; j < M; j += nt) {
        const uint32_t k = j + 1, kp = P - k;
        const float2 a = s[(k & 1u) * W + ldsfft::pos_of(k >> 1, logW)];
        const float2 b = s[(kp & 1u) * W + ldsfft::pos_of(kp >> 1, logW)];
        const float sre = a.x + b.x, sim = a.y - b.y;
        const float dre = a.x - b.x, dim = a.y + b.y;
        const float left = sqrtf(fmaf(sre, sre, sim * sim)) * 0.5f * p.scale;
        const float right = sqrtf(fmaf(dre, dre, dim * dim)) * 0.5f * p.scale;
        if (p.mono_pairs) {
            if (st_a) st_stream(out_a + j, left, left);
            if (st_b) st_stream(out_b + j, right, right);
        } else {
            st_stream(out_a + j, left, right);
        }
    }
}

hipError_t launch_stft_generic(const sgx_ctx *c, const float *d_pcm, uint32_t channels, uint32_t pairs, size_t first_frame,
                               size_t n_frames, size_t total_frames, float *d_mags)
{
    if (n_frames == 0) return hipSuccess;
    StftGenericParams p{};
    p.pcm = d_pcm;
    p.window = c->d_window;
    p.twiddle = c->d_twiddle;
    p.mags = d_mags;
    p.first_frame = first_frame;
    p.W = c->W;
    p.logP = c->logP;
    p.H = c->H;
    p.C = channels;
    p.pairs = pairs;
    p.half_scale = 0.5f;
    p.scale = 2.0f / (float)c->W;
    const size_t lds = (size_t)c->P * sizeof(float2);
    // workgroup size by measurement (tools/quick_bench.py --generic-sizes): about one radix-4 butterfly per thread and
    // stage for short transforms, 16 waves per CU where the LDS image limits residency (2 x 512 at 8192, 1 x 1024 at 16384)
    const unsigned threads = c->P >= 16384 ? 1024u : (c->P >= 8192 ? 512u : (c->P >= 2048 ? 256u : (c->P >= 1024 ? 128u : 64u)));
    if (lds > 64 * 1024) {  // per launch: the attribute is per device, and a process may hold contexts on several
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft_generic_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    p.n_frames = n_frames;
    p.total_frames = total_frames;
    if (channels == 1 && (c->cfg.flags & SGX_FLAG_PAIRED_FRAMES)) {
        // one workgroup per frame PAIR (2q, 2q+1); an odd first frame / last frame shares its transform with a
        // neighbour outside the range, which is computed and not stored
        p.mono_pairs = 1;
        const unsigned long long q0 = first_frame / 2, q1 = (first_frame + n_frames + 1) / 2;
        const unsigned long long max_chunk = 1u << 30;
        for (unsigned long long q = q0; q < q1; q += max_chunk) {
            const unsigned long long chunk = q1 - q < max_chunk ? q1 - q : max_chunk;
            p.pair_base = q;
            hipLaunchKernelGGL(stft_generic_kernel, dim3((unsigned)chunk, 1), dim3(threads), lds, c->stream, p);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    // gridDim.x is limited to 2^31-1; frames are chunked far below that by the caller
    const size_t max_chunk = 1u << 30;
    size_t done = 0;
    while (done < n_frames) {
        size_t chunk = n_frames - done < max_chunk ? n_frames - done : max_chunk;
        StftGenericParams q = p;
        q.first_frame = first_frame + done;
        q.n_frames = chunk;
        q.mags = d_mags + done * (size_t)pairs * c->M * 2;
        hipLaunchKernelGGL(stft_generic_kernel, dim3((unsigned)chunk, pairs), dim3(threads), lds, c->stream, q);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        done += chunk;
    }
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// pixel column: magnitude_in -> color_for -> put_pixel
// ------------------------------------------------------------------------------------------------

struct RenderParams {
    const float *mags;          // [n_columns][M][2]
    const RowEntry *rows;       // [R]
    const SampleEntry *samples;
    const float *lut_thr;       // [n_lut - 1]
    const float *alpha_thr;     // [255]
    const uchar4 *lut_rgba;     // [n_lut]
    uint8_t *rgba;              // [n_columns][R][4]
    uint32_t M, R, n_lut, interp, stereo, lut_mode;
    const double *t_thr;        // segment palettes, diverging branch: [n_lut - 1] switch points of the balance t
    uint32_t segments;
    uchar4 nan_rgba;            // colour of t = NaN
    float guess_a, guess_b;     // seeded threshold counts: level (mono) or alpha byte (diverging) ~ floor(log2(power + 1e-7) a + b - 1/2), then one compare
    uint32_t alpha_seed;        // the alpha table passes the seed proof with (guess_a, guess_b)
    const uint16_t *t_cell;     // [kTCells + 1] or null (segment palettes, diverging branch)
};

// number of thresholds <= v in a sorted table (NaN thresholds sort last and never match)
__device__ __forceinline__ uint32_t count_reached(const float *thr, uint32_t n, float v)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (v >= thr[mid]) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// ColorScheme::color_for((l, r)) (colorscheme.rs:55-71) as threshold counts; `thr` / `athr` are the LDS copies
// `tthr` / `lut`: the diverging branch's switch points and the colours, wherever the caller keeps them (LDS or global)
// `tcell` (or null): the grid over t that brackets the segment (sgx_t_cell; table from upload_palette)
__device__ __forceinline__ uchar4 color_for(const RenderParams &p, const float *thr, const float *athr, const double *tthr, const uchar4 *lut,
                                            const uint16_t *tcell, float l, float r)
{
    // colorscheme.rs:59: norm_sqr = l*l + r*r, then the dB ramp as a threshold count
    const float power = (l * l) + (r * r);
    // (the pixel as ONE 32-bit word, alpha in its top byte: as a uchar4 whose .w is overwritten the compiler copied the other three bytes
    // of p.nan_rgba through scratch -- 8 bytes of private segment in every kernel of this file that can reach this function)
    const uint32_t *lutw = reinterpret_cast<const uint32_t *>(lut);
    const uint32_t nanw = (uint32_t)p.nan_rgba.x | ((uint32_t)p.nan_rgba.y << 8) | ((uint32_t)p.nan_rgba.z << 16);
    uint32_t px;
    if (p.stereo) {
        // :63-66
        const float l1 = fabsf(l) + fabsf(r);
        const double t = (double)l / (double)l1;
        if (p.segments) {
            if (t != t) {
                px = nanw;
            } else {
                uint32_t lo = 0, hi = p.n_lut - 1;  // number of switch points <= t
                if (tcell) {   // the cells below t's hold switch points t has passed, the cells above it ones it has not
                    const int cell = sgx_t_cell(t);
                    lo = tcell[cell];
                    hi = tcell[cell + 1];
                }
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (t >= tthr[mid]) lo = mid + 1;
                    else hi = mid;
                }
                px = lutw[lo];
            }
        } else {
            double x = (p.lut_mode == SGX_LUT_ROUND_NM1) ? floor(t * (double)(p.n_lut - 1) + 0.5) : floor(t * (double)p.n_lut);
            uint32_t idx = 0;
            if (x > 0.0) idx = x >= (double)p.n_lut ? p.n_lut - 1 : (uint32_t)x;
            px = lutw[idx];
        }
        uint32_t alpha;
        if (p.alpha_seed) {   // the seed proof holds on the alpha table (launch_render): one read and one compare instead of eight
            const float ua = fmaf(__builtin_amdgcn_logf(power + 1e-7f), p.guess_a, p.guess_b);
            int ia = (int)floorf(ua - 0.5f);
            ia = ia < 0 ? 0 : (ia > 254 ? 254 : ia);
            alpha = ((uint32_t)ia + (power >= athr[ia] ? 1u : 0u)) & 0xffu;
        } else {
            alpha = count_reached(athr, 255, power) & 0xffu;  // (alpha * 255.0) as u8, simple_spectrogram.rs:159
        }
        px = (px & 0x00ffffffu) | (alpha << 24);
    } else {
        // :67-70; alpha = 1.0 -> 255
        px = (p.segments && power != power) ? nanw : lutw[count_reached(thr, p.n_lut - 1, power)];
        px |= 0xff000000u;
    }
    uchar4 out;
    out.x = px & 0xff; out.y = (px >> 8) & 0xff; out.z = (px >> 16) & 0xff; out.w = px >> 24;
    return out;
}

// Replaces the body of `for py in 0..buffer.height()` (simple_spectrogram.rs:141-161).
__global__ void __launch_bounds__(256) render_kernel(RenderParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *m = reinterpret_cast<float2 *>(smem_raw);                 // [M]
    float *thr = reinterpret_cast<float *>(m + p.M + 1);              // [n_lut - 1]
    float *athr = thr + p.n_lut;                                      // [255]
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const size_t col = blockIdx.x;
    const float2 *src = reinterpret_cast<const float2 *>(p.mags) + col * p.M;
    for (uint32_t i = tid; i < p.M; i += nt) m[i] = src[i];
    for (uint32_t i = tid; i + 1 < p.n_lut; i += nt) thr[i] = p.lut_thr[i];
    for (uint32_t i = tid; i < 255; i += nt) athr[i] = p.alpha_thr[i];
    __syncthreads();

    const int32_t last = (int32_t)p.M - 1;
    uchar4 *dst = reinterpret_cast<uchar4 *>(p.rgba) + col * p.R;
    for (uint32_t py = tid; py < p.R; py += nt) {
        const RowEntry row = p.rows[py];
        float sl = 0.0f, sr = 0.0f;  // Complex::sum starts at zero (interpolated_frequency_sample.rs:70-72)
        for (uint32_t i = 0; i < row.count; ++i) {
            const SampleEntry se = p.samples[row.first + i];
            float vl, vr;
            if (p.interp == SGX_INTERP_COSINE) {
                // :79-86  data[low] * (1 - o') + data[high] * o'
                const float2 a = m[se.i0], b = m[se.i1];
                vl = a.x * se.w1 + b.x * se.w2;
                vr = a.y * se.w1 + b.y * se.w2;
            } else {
                // :89-105
                const int32_t x1 = se.i0;
                const int32_t x0 = x1 > 0 ? x1 - 1 : 0;
                const int32_t x2 = x1 + 1 < last ? x1 + 1 : last;
                const int32_t x3 = x1 + 2 < last ? x1 + 2 : last;
                const float2 y0 = m[x0], y1 = m[x1], y2 = m[x2], y3 = m[x3];
                const float mu = se.w0, mu2 = se.w1, mu3 = se.w2;
                const float2 vv = cubic_pair(y0, y1, y2, y3, mu, mu2, mu3);
                vl = vv.x; vr = vv.y;
            }
            sl = sl + vl;
            sr = sr + vr;
        }
        const float l = sl / row.count_f, r = sr / row.count_f;  // :72

        const uchar4 px = color_for(p, thr, athr, p.t_thr, p.lut_rgba, p.t_cell, l, r);
        dst[p.R - 1 - py] = px;  // simple_spectrogram.rs:150
    }
}

// The same column in two balanced passes over LDS (see stft4096_wg.hpp: one thread per magnitude_in SAMPLE,
// then one thread per row), by persistent workgroups that keep the threshold tables in LDS and request the
// next column's magnitudes while the current one is rendered.  Used whenever the column, its interpolated
// samples and the tables fit in LDS; render_kernel above is the general fallback.
//   NT    threads per workgroup: 256 where four workgroups fit a CU's LDS, 512 / 1024 where only two / one do (16 waves per
//         CU either way: a 130 KB column image with four waves on it ran at a fifth of the rate)
//   KPRE  bins per thread held in registers for the next column (M <= NT KPRE)
//   MODE  kGeneric: color_for as written above (thresholds bisected: eight dependent LDS reads per table and pixel,
//           the diverging branch in double precision)
//         kMonoSeed: a 256-level palette without the diverging branch whose dB thresholds pass the host's seed proof
//           (wg::seed_within_one): the level is floor(log2(power + 1e-7) a + b - 1/2) or one above it, one 16-byte LDS
//           access brings that entry's threshold and both candidate colours, one compare picks (the fused kernel's
//           pixel_for)
//         kStereoSeed: the diverging branch (colorscheme.rs:63-66) of a 256-level palette indexed floor(t n): the alpha
//           byte by the same seed + compare on the alpha thresholds (proof on that table); the level
//           floor(256 fl64(l / l1)) is the largest k with 256 l >= k l1 -- k / 256 is a double, so rounding the quotient
//           cannot cross it, and both products are exact in double -- found from a float32 quotient (off by one at
//           most) and two exact compares: no double-precision division, no floor
//   SPT   > 0: a thread's samples (tid + 256 k, k < SPT) and rows (tid + 256 i, i < 4) are the same for every column: their
//         table entries are read once and stay in registers (i0 + one weight per sample -- mu^2, mu^3, 1 - o' and the
//         upper tap are re-derived by the single-rounded operations the host table was built with; first | count << 16
//         per row), so a column costs no table loads and no dependent load -> gather chain.  0: tables streamed per column.
constexpr int kGeneric = 0, kMonoSeed = 1, kStereoSeed = 2;
template <int KPRE, int MODE, int SPT, int NT>
__global__ void __launch_bounds__(NT, 4) render_two_pass_kernel(RenderParams p, unsigned long long n_columns, uint32_t n_samples)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *m = reinterpret_cast<float2 *>(smem_raw);                 // [M]
    float2 *vbuf = m + p.M + 1;                                       // [n_samples]
    float *thr = reinterpret_cast<float *>(vbuf + n_samples);         // [n_lut - 1]      (bisection)
    float *athr = thr + p.n_lut;                                      // [255]
    uint2 *pal = reinterpret_cast<uint2 *>(vbuf + n_samples);         // [256] {threshold to leave level i, RGBA of level i}   (kMonoSeed)
    float *athr_s = reinterpret_cast<float *>(vbuf + n_samples);      // [256] alpha thresholds, then [256] RGBA words           (kStereoSeed)
    uint32_t *rgba_s = reinterpret_cast<uint32_t *>(athr_s + 256);
    const uint32_t tid = threadIdx.x;
    if (MODE == kMonoSeed) {
        if (tid < 256) pal[tid] = make_uint2(__float_as_uint(tid < 255 ? p.lut_thr[tid] : __builtin_nanf("")), *reinterpret_cast<const uint32_t *>(&p.lut_rgba[tid]));
    } else if (MODE == kStereoSeed) {
        if (tid < 256) {
            athr_s[tid] = tid < 255 ? p.alpha_thr[tid] : __builtin_nanf("");
            rgba_s[tid] = *reinterpret_cast<const uint32_t *>(&p.lut_rgba[tid]) & 0x00ffffffu;
        }
    } else {
        for (uint32_t i = tid; i + 1 < p.n_lut; i += NT) thr[i] = p.lut_thr[i];
        for (uint32_t i = tid; i < 255; i += NT) athr[i] = p.alpha_thr[i];
    }
    // kGeneric: the colours and the diverging branch's switch points behind the two float tables (8-byte aligned)
    uchar4 *lut_s = reinterpret_cast<uchar4 *>(thr + ((p.n_lut + 255 + 1) & ~1u));   // [n_lut]
    double *tthr_s = reinterpret_cast<double *>(lut_s + ((p.n_lut + 1) & ~1u));      // [n_lut - 1]
    uint16_t *tcell_s = reinterpret_cast<uint16_t *>(tthr_s + p.n_lut);               // [kTCells + 1]
    if (MODE == kGeneric) {
        for (uint32_t i = tid; i < p.n_lut; i += NT) lut_s[i] = p.lut_rgba[i];
        if (p.stereo && p.segments)
            for (uint32_t i = tid; i + 1 < p.n_lut; i += NT) tthr_s[i] = p.t_thr[i];
        if (p.t_cell)
            for (uint32_t i = tid; i <= (uint32_t)kTCells; i += NT) tcell_s[i] = p.t_cell[i];
    }

    const int32_t last = (int32_t)p.M - 1;
    int32_t s_i0[SPT > 0 ? SPT : 1];
    float s_w[SPT > 0 ? SPT : 1];
    uint32_t row_w[4] = {0u, 0u, 0u, 0u};
    if (SPT > 0) {
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
            const uint32_t sidx = tid + (uint32_t)NT * k;
            const SampleEntry se = p.samples[sidx < n_samples ? sidx : 0];
            s_i0[k] = se.i0;
            s_w[k] = p.interp == SGX_INTERP_COSINE ? se.w2 : se.w0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t py = tid + (uint32_t)NT * i;
            if (py < p.R) row_w[i] = p.rows[py].first | (p.rows[py].count << 16);
        }
    }
    float2 nxt[KPRE];
    auto request = [&](unsigned long long col) {
        const float2 *src = reinterpret_cast<const float2 *>(p.mags) + col * p.M;
#pragma unroll
        for (int j = 0; j < KPRE; ++j) {
            const uint32_t i = tid + (uint32_t)NT * j;
            nxt[j] = i < p.M ? src[i] : make_float2(0.0f, 0.0f);
        }
    };
    auto fill = [&]() {
#pragma unroll
        for (int j = 0; j < KPRE; ++j) {
            const uint32_t i = tid + (uint32_t)NT * j;
            if (i < p.M) m[i] = nxt[j];
        }
    };
    // Order of the vector-memory stream (vmcnt retires in order: a wait for a load is a wait for every store issued
    // before it): the magnitudes of column c + 2 are requested BEFORE column c's pixel stores, and they are waited for
    // only after column c + 1's sample pass -- by then the only stores ahead of them (column c - 1's ... none) have
    // long landed.  The first form of this loop requested behind the stores and waited at the top: every column paid a
    // full store round trip.
    unsigned long long col = blockIdx.x;
    if (col < n_columns) {
        request(col);
        fill();
        if (col + gridDim.x < n_columns) request(col + gridDim.x);
    }
    for (; col < n_columns; col += gridDim.x) {
        __syncthreads();  // m is filled; the previous column's row pass is done with vbuf
        // ---- sample pass (interpolated_frequency_sample.rs:79-105)
        if (SPT > 0) {
#pragma unroll
            for (int k = 0; k < SPT; ++k) {
                const uint32_t sidx = tid + (uint32_t)NT * k;
                if (sidx >= n_samples) break;
                // opaque copies: everything derived from a table entry (tap addresses, mu^2, mu^3) is re-derived per column;
                // left visible, the compiler hoists all of it out of the column loop -- 100 registers and a resident workgroup
                int32_t x1 = s_i0[k];
                float wk = s_w[k];
                asm volatile("" : "+v"(x1), "+v"(wk));
                float2 v;
                if (p.interp == SGX_INTERP_COSINE) {
                    const int32_t hi = x1 + 1 < last ? x1 + 1 : last;   // clamp(ceil(idx), low + 1, M - 1)  (:81)
                    const float o2 = wk, w1 = 1.0f - o2;
                    const float2 a = m[x1], b = m[hi];
                    v.x = a.x * w1 + b.x * o2;
                    v.y = a.y * w1 + b.y * o2;
                } else {
                    const int32_t x0 = x1 > 0 ? x1 - 1 : 0;
                    const int32_t x2 = x1 + 1 < last ? x1 + 1 : last;
                    const int32_t x3 = x1 + 2 < last ? x1 + 2 : last;
                    const float2 y0 = m[x0], y1 = m[x1], y2 = m[x2], y3 = m[x3];
                    const float mu = wk, mu2 = mu * mu, mu3 = mu * mu2;   // num_traits::pow (:93-94), as the host table
                    v = cubic_pair(y0, y1, y2, y3, mu, mu2, mu3);
                }
                vbuf[sidx] = v;
            }
        } else {
            uint32_t sidx = tid;
            SampleEntry se_cur = p.samples[sidx < n_samples ? sidx : 0];
            while (sidx < n_samples) {
                // the next step's table entry is requested before this step's gathers (one L1 latency overlapped)
                const SampleEntry se = se_cur;
                se_cur = p.samples[sidx + NT < n_samples ? sidx + NT : 0];
                float2 v;
                if (p.interp == SGX_INTERP_COSINE) {
                    const float2 a = m[se.i0], b = m[se.i1];
                    v.x = a.x * se.w1 + b.x * se.w2;
                    v.y = a.y * se.w1 + b.y * se.w2;
                } else {
                    const int32_t x1 = se.i0;
                    const int32_t x0 = x1 > 0 ? x1 - 1 : 0;
                    const int32_t x2 = x1 + 1 < last ? x1 + 1 : last;
                    const int32_t x3 = x1 + 2 < last ? x1 + 2 : last;
                    const float2 y0 = m[x0], y1 = m[x1], y2 = m[x2], y3 = m[x3];
                    const float mu = se.w0, mu2 = se.w1, mu3 = se.w2;
                    v = cubic_pair(y0, y1, y2, y3, mu, mu2, mu3);
                }
                vbuf[sidx] = v;
                sidx += NT;
            }
        }
        __syncthreads();  // vbuf is complete, m is free
        if (col + gridDim.x < n_columns) {
            fill();
            if (col + 2ull * gridDim.x < n_columns) request(col + 2ull * gridDim.x);
        }
        // ---- row pass (:60-75 the mean; colorscheme.rs:55-71; simple_spectrogram.rs:150-160)
        uchar4 *dst = reinterpret_cast<uchar4 *>(p.rgba) + col * p.R;
        int i_row = 0;
        for (uint32_t py = tid; py < p.R; py += NT, ++i_row) {
            uint32_t first, count;
            if (SPT > 0) {
                const uint32_t w = i_row == 0 ? row_w[0] : i_row == 1 ? row_w[1] : i_row == 2 ? row_w[2] : row_w[3];
                first = w & 0xffffu;
                count = w >> 16;
            } else {
                const RowEntry row = p.rows[py];
                first = row.first;
                count = row.count;
            }
            float sl = 0.0f, sr = 0.0f;  // Complex::sum starts at zero
            for (uint32_t i = 0; i < count; ++i) {
                const float2 v = vbuf[first + i];
                sl = sl + v.x;
                sr = sr + v.y;
            }
            float l = sl, r = sr;
            if (count > 1) {  // x / 1.0 == x: only rows that average several samples divide (:72)
                const float nf = (float)count;   // `n as f32`
                l = sl / nf;
                r = sr / nf;
            }
            if (MODE == kStereoSeed) {
                const float power = (l * l) + (r * r);   // colorscheme.rs:59
                const float l1 = fabsf(l) + fabsf(r);    // :65 l1_norm, float32
                uint32_t idx = 0;                        // t <= 0 or NaN: floor(t n) > 0 fails, level 0
                if (l > 0.0f && l1 < __builtin_inff()) {
                    int k = (int)((l / l1) * 256.0f);
                    k = k < 0 ? 0 : (k > 256 ? 256 : k);
                    const double L = (double)l * 256.0, D = (double)l1;
                    if (L < (double)k * D) --k;
                    else if (L >= (double)(k + 1) * D) ++k;
                    idx = k > 255 ? 255u : (uint32_t)k;
                } else if (l > 0.0f) {   // an infinite l1: the reference's own arithmetic
                    const double x = floor(((double)l / (double)l1) * 256.0);
                    if (x > 0.0) idx = x >= 256.0 ? 255u : (uint32_t)x;
                }
                const float ua = fmaf(__builtin_amdgcn_logf(power + 1e-7f), p.guess_a, p.guess_b);   // host: the alpha table's coefficients
                int ia = (int)floorf(ua - 0.5f);
                ia = ia < 0 ? 0 : (ia > 254 ? 254 : ia);
                const uint32_t alpha = (uint32_t)ia + (power >= athr_s[ia] ? 1u : 0u);   // (alpha * 255.0) as u8, simple_spectrogram.rs:159
                *reinterpret_cast<uint32_t *>(dst + (p.R - 1 - py)) = rgba_s[idx] | (alpha << 24);
            } else if (MODE == kMonoSeed) {
                const float power = (l * l) + (r * r);   // colorscheme.rs:59
                const float u = fmaf(__builtin_amdgcn_logf(power + 1e-7f), p.guess_a, p.guess_b);
                int idx = (int)floorf(u - 0.5f);
                idx = idx < 0 ? 0 : (idx > 254 ? 254 : idx);
                const uint2 e0 = pal[idx], e1 = pal[idx + 1];
                const uint32_t rgba = (power >= __uint_as_float(e0.x) ? e1.y : e0.y) | 0xff000000u;   // alpha = 1.0 -> 255
                *reinterpret_cast<uint32_t *>(dst + (p.R - 1 - py)) = (p.segments && power != power) ? (*reinterpret_cast<const uint32_t *>(&p.nan_rgba) | 0xff000000u) : rgba;
            } else {
                dst[p.R - 1 - py] = color_for(p, thr, athr, tthr_s, lut_s, p.t_cell ? tcell_s : nullptr, l, r);
            }
        }
    }
}

hipError_t launch_render(const sgx_ctx *c, const float *d_mags, size_t n_columns, uint8_t *d_rgba)
{
    if (n_columns == 0) return hipSuccess;
    RenderParams p;
    p.mags = d_mags;
    p.rows = c->d_rows;
    p.samples = c->d_samples;
    p.lut_thr = c->d_lut_thr;
    p.alpha_thr = c->d_alpha_thr;
    p.lut_rgba = c->d_lut_rgba;
    p.rgba = d_rgba;
    p.M = c->M;
    p.R = c->R;
    p.n_lut = c->pal.n;
    p.interp = c->cfg.interp;
    p.stereo = (uint32_t)c->pal.stereo;
    p.lut_mode = c->cfg.lut_index_mode;
    p.t_thr = c->d_t_thr;
    p.t_cell = (c->pal.segments && c->pal.stereo && !(c->cfg.flags & SGX_FLAG_LUT_WALK)) ? c->d_t_cell : nullptr;
    p.segments = c->pal.segments ? 1u : 0u;
    p.nan_rgba = make_uchar4(c->pal.nan_rgb[0], c->pal.nan_rgb[1], c->pal.nan_rgb[2], 255);
    const size_t n_samples = c->tab.samples.size();
    int mode = kGeneric;
    p.guess_a = p.guess_b = 0.0f;
    p.alpha_seed = 0;
    if (!c->pal.stereo && c->pal.n == 256 && wg4096_seed_is_within_one(c)) {
        mode = kMonoSeed;
        lut_seed_coefficients(c, p.guess_a, p.guess_b);
    } else if (c->pal.stereo && !(c->cfg.flags & SGX_FLAG_LUT_WALK)) {
        // alpha byte = (bounded * 255.0) as u8: bounded * 255 = log2(x) a + b with these coefficients; the same proof on its table
        const double span = (double)c->cfg.max_db - (double)c->cfg.min_db;
        const double a = 10.0 * log10(2.0) * 255.0 / span, b = -(double)c->cfg.min_db * 255.0 / span;
        if (wg::seed_within_one(c->pal.alpha_thr, a, b)) {
            p.alpha_seed = 1;
            p.guess_a = (float)a;
            p.guess_b = (float)b;
            if (!c->pal.segments && c->pal.n == 256 && c->cfg.lut_index_mode == SGX_LUT_FLOOR_N) mode = kStereoSeed;
        }
    }
    const size_t tail = mode != kGeneric ? 256 * sizeof(uint2)
                                         : (size_t)((c->pal.n + 255 + 1) & ~1u) * sizeof(float) + (size_t)((c->pal.n + 1) & ~1u) * sizeof(uchar4) +
                                               (size_t)c->pal.n * sizeof(double) + (size_t)(kTCells + 4) * sizeof(uint16_t);
    const size_t lds2 = (size_t)(c->M + 1 + n_samples) * sizeof(float2) + tail;
    // the largest LDS image a workgroup may ask for on THIS device (read once at sgx_create): a column that does not fit takes
    // the one-workgroup-per-column kernel below, as it did before the two-pass form accepted images above 64 KB
    const size_t lds_cap = c->lds_optin < (size_t)160 * 1024 ? c->lds_optin : (size_t)160 * 1024;
    if (c->M <= 1024 * 10 && lds2 <= lds_cap) {
        // persistent two-pass form: workgroups sized to the LDS image, each walks columns blockIdx.x, + grid, ...
        const int n_cu = c->n_cu;
        const size_t fit = lds_cap / lds2;
        unsigned nt = fit >= 4 ? 256u : (fit >= 2 ? 512u : 1024u);
        while (nt < 1024u && (c->M + nt - 1) / nt > 16) nt *= 2;
        auto go = [&](auto kernel) -> hipError_t {
            // as many persistent workgroups as the device keeps resident (LDS image and registers of THIS instantiation):
            // asked once per (instantiation, block size, image size) and kept -- this sits on the per-tick path of a live ring
            const std::array<size_t, 3> key = {(size_t)reinterpret_cast<uintptr_t>(reinterpret_cast<const void *>(kernel)), (size_t)nt, lds2};
            int per_cu = 0;
            for (const auto &kv : c->occupancy_cache)
                if (kv.first == key) per_cu = kv.second;
            if (per_cu == 0) {
                if (lds2 > 64 * 1024) {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
                    if (e != hipSuccess) { (void)hipGetLastError(); return hipErrorNotSupported; }   // -> the per-column kernel below
                }
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)nt, lds2) != hipSuccess || per_cu < 1) per_cu = 1;
                c->occupancy_cache.push_back({key, per_cu});
            }
            size_t blocks = (size_t)n_cu * (size_t)(per_cu > 8 ? 8 : per_cu);
            if (blocks > n_columns) blocks = n_columns;
            hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(nt), lds2, c->stream, p, (unsigned long long)n_columns, (uint32_t)n_samples);
            return hipGetLastError();
        };
        const uint32_t need = (c->M + nt - 1) / nt;
        // table entries in registers: 12 samples and 4 rows per thread at most, 16-bit row fields
        bool in_regs = nt == 256 && n_samples <= 256 * 12 && c->R <= 1024 && n_samples < 65536;
        for (const RowEntry &r : c->tab.rows) in_regs = in_regs && r.count < 65536 && r.first < 65536;
#define SGX_TWO_PASS(K, S, T) (mode == kMonoSeed ? go(render_two_pass_kernel<K, kMonoSeed, S, T>) : mode == kStereoSeed ? go(render_two_pass_kernel<K, kStereoSeed, S, T>) : go(render_two_pass_kernel<K, kGeneric, S, T>))
        hipError_t e2;
        if (nt == 256) {
            if (need <= 8) e2 = in_regs ? SGX_TWO_PASS(8, 12, 256) : SGX_TWO_PASS(8, 0, 256);
            else if (need <= 10) e2 = in_regs ? SGX_TWO_PASS(10, 12, 256) : SGX_TWO_PASS(10, 0, 256);
            else e2 = SGX_TWO_PASS(16, 0, 256);
        } else if (nt == 512) e2 = need <= 8 ? SGX_TWO_PASS(8, 0, 512) : SGX_TWO_PASS(16, 0, 512);
        else e2 = need <= 8 ? SGX_TWO_PASS(8, 0, 1024) : SGX_TWO_PASS(10, 0, 1024);
#undef SGX_TWO_PASS
        if (e2 != hipErrorNotSupported) return e2;   // (not supported: the image was refused; nothing was launched)
    }
    const size_t lds = (size_t)(c->M + 1) * sizeof(float2) + (size_t)(c->pal.n + 255) * sizeof(float);
    if (lds > 64 * 1024) {  // per launch: the attribute is per device, and a process may hold contexts on several
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(render_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const size_t max_chunk = 1u << 30;
    size_t done = 0;
    while (done < n_columns) {
        size_t chunk = n_columns - done < max_chunk ? n_columns - done : max_chunk;
        RenderParams q = p;
        q.mags = d_mags + done * (size_t)c->M * 2;
        q.rgba = d_rgba + done * (size_t)c->R * 4;
        hipLaunchKernelGGL(render_kernel, dim3((unsigned)chunk), dim3(256), lds, c->stream, q);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        done += chunk;
    }
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// FrequencySample::magnitude_in over arbitrary ranges (src/fourier/mod.rs:17-21): the pixel stage
// without the colour -- what SpectrumAnalyzer::push_frequencies consumes (spectrum_analyzer.rs:60-61)
// ------------------------------------------------------------------------------------------------

struct BandsParams {
    const float *mags;
    const RowEntry *rows;
    const SampleEntry *samples;
    float *out;  // [n_columns][n_ranges][2]
    uint32_t M, n_ranges, interp;
};

__global__ void __launch_bounds__(256) magnitude_in_kernel(BandsParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *m = reinterpret_cast<float2 *>(smem_raw);
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const size_t col = blockIdx.x;
    const float2 *src = reinterpret_cast<const float2 *>(p.mags) + col * p.M;
    for (uint32_t i = tid; i < p.M; i += nt) m[i] = src[i];
    __syncthreads();
    const int32_t last = (int32_t)p.M - 1;
    float2 *dst = reinterpret_cast<float2 *>(p.out) + col * p.n_ranges;
    for (uint32_t b = tid; b < p.n_ranges; b += nt) {
        const RowEntry row = p.rows[b];
        float sl = 0.0f, sr = 0.0f;
        for (uint32_t i = 0; i < row.count; ++i) {
            const SampleEntry se = p.samples[row.first + i];
            float vl, vr;
            if (p.interp == SGX_INTERP_COSINE) {
                const float2 a = m[se.i0], bb = m[se.i1];
                vl = a.x * se.w1 + bb.x * se.w2;
                vr = a.y * se.w1 + bb.y * se.w2;
            } else {
                const int32_t x1 = se.i0;
                const int32_t x0 = x1 > 0 ? x1 - 1 : 0;
                const int32_t x2 = x1 + 1 < last ? x1 + 1 : last;
                const int32_t x3 = x1 + 2 < last ? x1 + 2 : last;
                const float2 y0 = m[x0], y1 = m[x1], y2 = m[x2], y3 = m[x3];
                const float mu = se.w0, mu2 = se.w1, mu3 = se.w2;
                const float2 vv = cubic_pair(y0, y1, y2, y3, mu, mu2, mu3);
                vl = vv.x; vr = vv.y;
            }
            sl = sl + vl;
            sr = sr + vr;
        }
        dst[b] = make_float2(sl / row.count_f, sr / row.count_f);
    }
}

hipError_t launch_magnitude_in(const sgx_ctx *c, const float *d_mags, size_t n_columns, const RowEntry *d_rows,
                               const SampleEntry *d_samples, uint32_t n_ranges, float *d_out)
{
    if (n_columns == 0 || n_ranges == 0) return hipSuccess;
    BandsParams p;
    p.mags = d_mags;
    p.rows = d_rows;
    p.samples = d_samples;
    p.out = d_out;
    p.M = c->M;
    p.n_ranges = n_ranges;
    p.interp = c->cfg.interp;
    const size_t lds = (size_t)(c->M + 1) * sizeof(float2);
    if (lds > 64 * 1024) {  // per launch: the attribute is per device, and a process may hold contexts on several
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(magnitude_in_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const size_t max_chunk = 1u << 30;
    for (size_t done = 0; done < n_columns; done += max_chunk) {
        const size_t chunk = n_columns - done < max_chunk ? n_columns - done : max_chunk;
        BandsParams q = p;
        q.mags = d_mags + done * (size_t)c->M * 2;
        q.out = d_out + done * (size_t)n_ranges * 2;
        hipLaunchKernelGGL(magnitude_in_kernel, dim3((unsigned)chunk), dim3(256), lds, c->stream, q);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// f32 -> f16 magnitudes (only for STFT kernels that have no native half store)
// ------------------------------------------------------------------------------------------------

__global__ void to_half_kernel(const float2 *in, __half2 *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float2 v = in[i];
        out[i] = __floats2half2_rn(v.x, v.y);
    }
}

hipError_t launch_to_half(const sgx_ctx *c, const float *d_in, void *d_out, size_t n_pairs)
{
    if (n_pairs == 0) return hipSuccess;
    size_t blocks = (n_pairs + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, reinterpret_cast<const float2 *>(d_in),
                       static_cast<__half2 *>(d_out), n_pairs);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// harness helpers
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t lowbias32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__global__ void white_noise_kernel(float *out, unsigned long long first, size_t n, uint32_t channels, uint32_t seed)
{
    const size_t total = n * channels;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t i = e / channels;
        const uint32_t ch = (uint32_t)(e - i * channels);
        const unsigned long long idx = first + i;
        const uint32_t s = (seed + ch) + (uint32_t)(idx >> 32) * 0x9E3779B9U;
        const uint32_t h = lowbias32(s ^ (uint32_t)idx);
        out[e] = (float)(h >> 8) * 1.1920928955078125e-07f - 1.0f;
    }
}

hipError_t launch_white_noise(const sgx_ctx *c, float *d_out, uint64_t first, size_t n, uint32_t channels, uint32_t seed)
{
    if (n == 0) return hipSuccess;
    const size_t total = n * channels;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(white_noise_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, d_out, (unsigned long long)first, n,
                       channels, seed);
    return hipGetLastError();
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void checksum_kernel(const uint32_t *words, size_t n_words, unsigned long long base, unsigned long long *acc)
{
    unsigned long long local = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x)
        local += mix64(((base + i) << 32) ^ (unsigned long long)words[i] ^ ((base + i) >> 32));
    // wave reduction, then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(acc, local);
}

hipError_t launch_checksum(const sgx_ctx *c, const uint32_t *d_words, size_t n_words, uint64_t base_word,
                           unsigned long long *d_acc)
{
    if (n_words == 0) return hipSuccess;
    size_t blocks = (n_words + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(checksum_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, d_words, n_words,
                       (unsigned long long)base_word, d_acc);
    return hipGetLastError();
}

}  // namespace sgx
