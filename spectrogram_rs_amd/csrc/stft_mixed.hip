// stft_mixed.hip -- STFT for windows whose padded length P = 2W is not a power of two but has only the prime
// factors 2, 3, 5 and 7: a mixed-radix FFT of exactly P points, in place in LDS.
//
// The reference sizes its window from a duration: FastFourierTransform::new(sample_rate, 0.05)
// (gpu_spectrogram.rs:323, simple_spectrogram.rs:217) gives W = 2400 at 48 kHz (P = 4800 = 2^6 3 5^2) and
// W = 2205 at 44.1 kHz (P = 4410 = 2 3^2 5 7^2); FFTW takes any length (fft.rs:20-24).  The chirp-z kernel
// (stft_bluestein.hip) serves every length with two power-of-two transforms of >= 3W points; for the smooth
// lengths the application actually produces, this kernel does a quarter of that arithmetic.
//
// Decimation in frequency, one stage per radix R of the current block length Ns (m = Ns / R):
//   y_k = (sum_q x[q m + j] w_R^{q k}) * w_Ns^{j k},  stored at k m + j      (k < R, j < m)
// after which sub-block k is the length-m problem of the bins = k (mod R).  Bin K ends at pos[K] (mixed-radix
// digit reversal, a host table); the split reads F[k] and F[P - k] through it.
//
// Round 2: a stage's radix is a PRODUCT of two prime-ish factors (R = RA RB <= 28: 4800 = 20 x 15 x 16, 4410 =
// 21 x 15 x 14), its R-point DFT done in registers (RB transforms of RA points, constant twiddles, RA transforms of
// RB points), so a transform makes three trips through LDS instead of six; the first stage reads the windowed
// samples straight from the stream (no staging trip for the input), the last (m = 1) has no twiddles; a stage's
// twiddles w_Ns^{j k}, k = 1 .. R-1, lie side by side in a per-stage table (16-byte reads instead of R-1 gathers
// from the full circle); the LDS image is padded by one point per R_last points (the last stage reads at a lane stride of R
// points); the split reads both positions of a bin from one packed word.  Measured at W = 2400, hop 93, stereo
// (SQ counters, profiles/r02_mixed_radix.txt): the first version kept the LDS array busy 62 % of the launch, 58 %
// of that in bank conflicts, and the address unit 58 %.
//
// The lengths the usual device rates produce (and the powers of two without a tuned kernel) run instantiations whose
// whole plan is a compile-time constant (stft_mixed_fixed_kernel / stft_mixed_fixed4_kernel, MIX_FIXED_PLANS): the same
// stage() over FixGeo instead of DynGeo; half the registers, workgroups of 512 / 1024 threads, +24 % to +64 %.
//
// Round 4: REAL-INPUT MODE -- a mono stream in the default mode (every frame its own transform; include/sgx.h, "Mono streams") runs the
// W-point plan on z[m] = x[2m] + i x[2m+1] and an untangling epilogue (untangle_store / pixel_epilogue_real) instead of the 2W-point plan
// on (s, s): half the transform per frame.  The chirp-z kernels below (lengths with a prime factor above 7) take the same mode.
#include <algorithm>
#include <vector>

#include <hip/hip_fp16.h>

#include "mix_codelets.hpp"
#include "pixel_passes.hpp"
#include "sgx_internal.hpp"

namespace sgx {

namespace mix {

constexpr int kMaxStages = 8;
constexpr uint32_t kMaxRadix = 28;

struct MixTables {
    uint32_t *d_split = nullptr;  // [M] padded position of bin k | padded position of bin P - k << 16   (k = j + 1)
    float2 *d_tw = nullptr;       // per stage: [m][R - 1] w_Ns^{j k}
    uint32_t n_stages = 0, pad_every = 0, lds_points = 0, threads = 0;
    int fixed = 0;                // P when the host's plan is the compile-time one of a fixed kernel (MIX_FIXED_PLANS), else 0
    // real-input mode (a mono stream, every frame its own transform): the plan of the W-point transform of z[m] = x[2m] + i x[2m+1]
    // with its own d_split ([W / 2]: positions of Z[k] and Z[W - k], k = j + 1) and the untangling twiddles w_2W^k
    MixTables *half = nullptr;
    float2 *d_twr = nullptr;
    uint32_t ra[kMaxStages] = {}, rb[kMaxStages] = {}, m[kMaxStages] = {}, tw_off[kMaxStages] = {};
    uint32_t q_stride[kMaxStages] = {}, blk_stride[kMaxStages] = {};   // padded LDS positions: see stage()
    float inv_m[kMaxStages] = {};
};

struct Params {
    const float *pcm;
    const float *window;
    const float2 *tw;
    const uint32_t *split;
    float *mags;
    unsigned long long first_frame, pair_base, n_frames, total_frames;
    uint32_t mono_pairs, W, P, H, C, pairs, n_stages, vec2;
    uint32_t real;       // real-input mode: P = W points, the first stage reads sample PAIRS, the epilogue untangles (untangle_store)
    const float2 *twr;   // [W / 2] w_2W^k, k = j + 1
    uint32_t out_f16;   // magnitudes are stored as (l, r) half pairs, 4 B per bin (the F16F16 ring of gpu_spectrogram.rs:218-226)
    float scale, inv_pad;
    // fused pixel stage (fixed plans only): magnitudes never leave LDS
    uint32_t render, R, n_samples, interp;
    const RowEntry *rows;
    const SampleEntry *samples;
    const uint2 *pal;          // [256] {threshold to leave level i, RGBA of level i}
    float guess_a, guess_b;    // level ~ floor(log2(power + 1e-7) a + b - 1/2), then one compare (wg::seed_within_one holds)
    uint8_t *rgba;             // [F][pairs][R][4]
    // chirp-z through the same stages (lengths with a prime factor above 7): the transform length is a compile-time power of two,
    // P stays 2 W, the first stage multiplies by the chirp, the split reads natural-order positions and multiplies again
    const float2 *chirp;       // [P] exp(-i pi n^2 / P), or null
    const float2 *bhat;        // [LDS image] FFT_L(conj chirp) / L at the padded digit-reversed positions, zero in the padding
    uint32_t ra[kMaxStages], rb[kMaxStages], m[kMaxStages], tw_off[kMaxStages], q_stride[kMaxStages], blk_stride[kMaxStages];
    float inv_m[kMaxStages];
};

typedef float v2f_a4 __attribute__((ext_vector_type(2), aligned(4)));   // an 8-byte word at a 4-byte aligned address

struct Source {   // where the first stage finds (l + i r) * hann (fft.rs:53-63); zeros from W on (fft.rs:65-69)
    const float *a, *b;
    uint32_t cl, cr;
    bool data_b;
};

// LDS positions.  The last stage (m = 1) reads R consecutive points per lane -- a lane stride of R points, 32-way bank
// conflicts when R is even -- so the image carries one point of padding per D = R_last points: position(i) = i + i / D.
// Every earlier stage's m is a multiple of R_last, hence of D, which makes the R accesses of a butterfly an arithmetic
// sequence:
//   position(blk m R + j + q m) = blk * blk_stride + j + j / D + q * q_stride      (one add per access)
// with blk_stride = m R (1 + 1 / D) and q_stride = m (1 + 1 / D) from the host (last stage: R + 1 and 1); j / D by one float
// multiply (inv_pad = 0 when the image is not padded).
// A stage's geometry: read from the launch parameters (any supported length), or compile-time constants (the two lengths
// the application itself produces: every index multiply, LDS offset and "is this row inside the window" test folds
// away, and rows of the first stage that lie in the zero padding prune the butterfly at compile time).
struct DynGeo {
    uint32_t m_, count_, qs_, bs_, W_, nt_;
    float inv_m_, inv_pad_;
    bool first_;
    __device__ __forceinline__ uint32_t m() const { return m_; }
    __device__ __forceinline__ uint32_t count() const { return count_; }
    __device__ __forceinline__ uint32_t qs() const { return qs_; }
    __device__ __forceinline__ uint32_t W() const { return W_; }
    __device__ __forceinline__ uint32_t nt() const { return nt_; }
    __device__ __forceinline__ bool first() const { return first_; }
    __device__ __forceinline__ uint32_t blk_of(uint32_t b) const { return (uint32_t)(((float)b + 0.5f) * inv_m_); }  // b / m: exact for every supported length (tests/test_host_logic.py)
    __device__ __forceinline__ uint32_t base_of(uint32_t blk, uint32_t j) const { return blk * bs_ + j + (uint32_t)(((float)j + 0.5f) * inv_pad_); }
};
template <uint32_t M, uint32_t COUNT, uint32_t QS, uint32_t BS, uint32_t WN, uint32_t PAD, uint32_t NT, bool FIRST>
struct FixGeo {
    uint32_t w_rt = 0;   // WN = 0: the number of non-zero inputs is a run-time value (chirp-z)
    __device__ __forceinline__ constexpr uint32_t m() const { return M; }
    __device__ __forceinline__ constexpr uint32_t count() const { return COUNT; }
    __device__ __forceinline__ constexpr uint32_t qs() const { return QS; }
    __device__ __forceinline__ constexpr uint32_t W() const { return WN ? WN : w_rt; }
    __device__ __forceinline__ constexpr uint32_t nt() const { return NT; }
    __device__ __forceinline__ constexpr bool first() const { return FIRST; }
    __device__ __forceinline__ uint32_t blk_of(uint32_t b) const { return b / M; }
    __device__ __forceinline__ uint32_t base_of(uint32_t blk, uint32_t j) const { return blk * BS + j + (PAD ? j / (PAD ? PAD : 1u) : 0u); }
};

// REAL: 1 / 0 at compile time, -1 = p.real (the run-time plan).  Real-input mode: complex sample n of the first stage is
// (x[2n] hann[2n], x[2n + 1] hann[2n + 1]) of the one mono frame, g.W() = ceil(W / 2) of them; p.vec2 = both as 8-byte words.
template <int RA, int RB, typename Geo, int REAL = 0>
__device__ __forceinline__ void stage(float2 *s, const Params &p, const float2 *tw, const Geo g, const Source &src, uint32_t tid)
{
    constexpr int R = RA * RB;
    const bool real = REAL < 0 ? p.real != 0 : REAL != 0;
    const uint32_t m = g.m(), qs = g.qs();
    const uint32_t q_nz = g.first() ? (g.W() + m - 1) / m : (uint32_t)R;   // first stage: rows q >= q_nz lie wholly in the padding
    for (uint32_t b = tid; b < g.count(); b += g.nt()) {
        const uint32_t blk = g.first() ? 0u : g.blk_of(b);
        const uint32_t j = b - blk * m;
        float2 *at = s + g.base_of(blk, j);
        float2 x[R];
        if (g.first()) {   // blk = 0: sample n = q m + j; a lane past the window loads sample W - 1 and drops it
#pragma unroll
            for (int q = 0; q < R; ++q) {
                x[q] = make_float2(0.0f, 0.0f);
                if ((uint32_t)q < q_nz) {   // uniform
                    const uint32_t n = q * m + j, nc = n < g.W() ? n : g.W() - 1;
                    if (real) {   // uniform
                        float2 x2, w2;
                        if (p.vec2) {   // an even W: whole pairs.  The samples' 8-byte words are 4-byte aligned at odd hops / stream offsets:
                                        // loads of two dwords need dword alignment only (v2f_a4 tells the compiler as much)
                            const v2f_a4 xv = *reinterpret_cast<const v2f_a4 *>(src.a + 2 * nc);
                            x2 = make_float2(xv.x, xv.y);
                            w2 = *reinterpret_cast<const float2 *>(p.window + 2 * nc);
                        } else {   // an odd W ends on half a pair
                            const bool whole = 2 * nc + 1 < p.W;
                            x2 = make_float2(src.a[2 * nc], whole ? src.a[2 * nc + 1] : 0.0f);
                            w2 = make_float2(p.window[2 * nc], whole ? p.window[2 * nc + 1] : 0.0f);
                        }
                        if (n < g.W()) x[q] = p.chirp ? cmul(make_float2(x2.x * w2.x, x2.y * w2.y), p.chirp[nc]) : make_float2(x2.x * w2.x, x2.y * w2.y);
                        continue;
                    }
                    const float w = p.window[nc];
                    float l, r;
                    if (p.vec2) {   // uniform
                        const float2 lr = *reinterpret_cast<const float2 *>(src.a + (nc * p.C + src.cl));
                        l = lr.x;
                        r = lr.y;
                    } else {
                        l = src.a[nc * p.C + src.cl];
                        r = src.data_b ? src.b[nc * p.C + src.cr] : 0.0f;
                    }
                    if (n < g.W()) x[q] = p.chirp ? cmul(make_float2(l * w, r * w), p.chirp[nc]) : make_float2(l * w, r * w);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < R; ++q) x[q] = at[q * qs];
        }
        dft_composite<RA, RB>(x);
        at[0] = x[0];
        if (m == 1 || j == 0) {
#pragma unroll
            for (int k = 1; k < R; ++k) at[k * qs] = x[k];
        } else {
            const float2 *twj = tw + (size_t)j * (R - 1);
#pragma unroll
            for (int k = 1; k < R; ++k) at[k * qs] = cmul(x[k], twj[k - 1]);
        }
    }
    __syncthreads();
}

// The inverse of stage() on the same positions: x[q m + j] = sum_k (y_k conj(w_Ns^{j k})) conj(w_R^{q k}), unnormalised, through
// the forward butterfly: IDFT(v) = conj(DFT(conj(v))).
template <int RA, int RB, typename Geo>
__device__ __forceinline__ void stage_inv(float2 *s, const float2 *tw, const Geo g, uint32_t tid)
{
    constexpr int R = RA * RB;
    const uint32_t m = g.m(), qs = g.qs();
    for (uint32_t b = tid; b < g.count(); b += g.nt()) {
        const uint32_t blk = g.blk_of(b);
        const uint32_t j = b - blk * m;
        float2 *at = s + g.base_of(blk, j);
        float2 x[R];
        const float2 y0 = at[0];
        x[0] = make_float2(y0.x, -y0.y);
        if (m == 1 || j == 0) {
#pragma unroll
            for (int k = 1; k < R; ++k) { const float2 y = at[k * qs]; x[k] = make_float2(y.x, -y.y); }
        } else {
            const float2 *twj = tw + (size_t)j * (R - 1);
#pragma unroll
            for (int k = 1; k < R; ++k) { const float2 y = at[k * qs]; x[k] = cmul(make_float2(y.x, -y.y), twj[k - 1]); }
        }
        dft_composite<RA, RB>(x);
#pragma unroll
        for (int q = 0; q < R; ++q) at[q * qs] = make_float2(x[q].x, -x[q].y);
    }
    __syncthreads();
}

#define MIX_STAGE_CASES(X) X(7, 4) X(7, 3) X(7, 2) X(5, 5) X(5, 4) X(5, 3) X(5, 2) X(4, 4) X(4, 3) X(4, 2) X(3, 3) X(3, 2) \
                           X(7, 1) X(5, 1) X(4, 1) X(3, 1) X(2, 1)

// (l, r) of one frame -- or, for a mono stream, frames 2q and 2q+1 by GLOBAL index (see sgx_kernels.hip)
__device__ __forceinline__ void frame_source(const Params &p, uint32_t pair, Source &src, long long &row_a, long long &row_b)
{
    row_b = -1;
    src.data_b = true;
    if (p.mono_pairs) {
        const unsigned long long fa = 2 * (p.pair_base + blockIdx.x), fb = fa + 1;
        row_a = (long long)fa - (long long)p.first_frame;
        row_b = row_a + 1;
        src.data_b = fb < p.total_frames;
        src.a = p.pcm + (size_t)(fa * p.H);
        src.b = src.data_b ? src.a + p.H : src.a;
        src.cl = src.cr = 0;
    } else {
        row_a = (long long)blockIdx.x;
        src.a = src.b = p.pcm + (size_t)((p.first_frame + blockIdx.x) * p.H) * p.C;
        src.cl = p.C == 1 ? 0 : 2 * pair;
        src.cr = p.C == 1 ? 0 : 2 * pair + 1;
    }
}

// |re + i im| * hs through the hardware square root (1 ulp; the tuned kernels' choice: the correctly rounded sqrtf is a dozen more
// vector instructions per bin -- 12-14 % of this kernel's instruction count, round 4), hs = scale / 2 (an exact halving)
__device__ __forceinline__ float mag_of(float re, float im, float hs) { return __builtin_amdgcn_sqrtf(fmaf(re, re, im * im)) * hs; }

// split + magnitude + scale (fft.rs:81-98); k = 1 .. W-1 kept
__device__ __forceinline__ void split_store(const Params &p, const float2 *s, uint32_t pair, long long row_a, long long row_b, uint32_t tid, uint32_t nt)
{
    const uint32_t M = p.W - 1;
    const bool st_a = row_a >= 0 && (unsigned long long)row_a < p.n_frames;
    const bool st_b = p.mono_pairs && row_b >= 0 && (unsigned long long)row_b < p.n_frames;
    const size_t off_a = ((size_t)(st_a ? row_a : 0) * p.pairs + pair) * M, off_b = ((size_t)(st_b ? row_b : 0) * p.pairs + pair) * M;
    float2 *out_a = reinterpret_cast<float2 *>(p.mags) + off_a, *out_b = reinterpret_cast<float2 *>(p.mags) + off_b;
    __half2 *half_a = reinterpret_cast<__half2 *>(p.mags) + off_a, *half_b = reinterpret_cast<__half2 *>(p.mags) + off_b;
    for (uint32_t j = tid; j < M; j += nt) {
        float2 a, b;
        if (p.chirp) {   // natural order after the inverse stages, one point of padding in 16; F[k] = c[k] y[k]
            const uint32_t k = j + 1, kp = p.P - k;
            a = cmul(s[k + (k >> 4)], p.chirp[k]);
            b = cmul(s[kp + (kp >> 4)], p.chirp[kp]);
        } else {
            const uint32_t w = p.split[j];
            a = s[w & 0xffffu];
            b = s[w >> 16];
        }
        const float sre = a.x + b.x, sim = a.y - b.y;
        const float dre = a.x - b.x, dim = a.y + b.y;
        const float left = mag_of(sre, sim, 0.5f * p.scale), right = mag_of(dre, dim, 0.5f * p.scale);
        if (p.out_f16) {   // round to nearest even, as the conversion pass of the kernels without a native half store
            if (p.mono_pairs) {
                if (st_a) half_a[j] = __floats2half2_rn(left, left);
                if (st_b) half_b[j] = __floats2half2_rn(right, right);
            } else {
                half_a[j] = __floats2half2_rn(left, right);
            }
        } else if (p.mono_pairs) {
            if (st_a) st_stream(out_a + j, left, left);
            if (st_b) st_stream(out_b + j, right, right);
        } else {
            st_stream(out_a + j, left, right);
        }
    }
}

// Real-input mode: the image holds Z = FFT_W(z), z[m] = x[2m] + i x[2m+1] of ONE real frame x.  With E / O the transforms of the
// even / odd samples, 2 E[k] = Z[k] + conj Z[W-k], 2 O[k] = -i (Z[k] - conj Z[W-k]), the frame's 2W-point spectrum is
// S[k] = E[k] + w_2W^k O[k] and S[W-k] = conj(E[k] - w_2W^k O[k]): one (k, W-k) pair of the image gives bins k and W-k, k = 1 .. W/2
// (the same identity as stft4096_real.hip; W even: k = W/2 is its own partner and both formulas give the same magnitude).
__device__ __forceinline__ float2 untangle(const Params &p, const float2 *s, uint32_t k1)
{
    float2 a, b;
    if (p.chirp) {   // chirp-z: natural order, one point of padding in 16; Z[k] = c[k] y[k] (P = W here)
        const uint32_t k = k1 + 1, kp = p.P - k;
        a = cmul(s[k + (k >> 4)], p.chirp[k]);
        b = cmul(s[kp + (kp >> 4)], p.chirp[kp]);
    } else {
        const uint32_t w = p.split[k1];
        a = s[w & 0xffffu];
        b = s[w >> 16];
    }
    const float2 t = p.twr[k1];
    const float sre = a.x + b.x, sim = a.y - b.y;     // 2 E
    const float dre = a.x - b.x, dim = a.y + b.y;     // 2 i O
    const float tx = t.x * dim + t.y * dre, ty = t.y * dim - t.x * dre;   // w (dim, -dre) = 2 w O
    const float ux = sre + tx, uy = sim + ty, vx = sre - tx, vy = sim - ty;
    return make_float2(mag_of(ux, uy, 0.5f * p.scale), mag_of(vx, vy, 0.5f * p.scale));
}

__device__ __forceinline__ void untangle_store(const Params &p, const float2 *s, long long row, uint32_t tid, uint32_t nt)
{
    const uint32_t M = p.W - 1, K = p.W / 2;
    float2 *out = reinterpret_cast<float2 *>(p.mags) + (size_t)row * M;
    __half2 *half = reinterpret_cast<__half2 *>(p.mags) + (size_t)row * M;
    for (uint32_t k1 = tid; k1 < K; k1 += nt) {   // bin k = k1 + 1 is row element k1, bin W - k element M - 1 - k1
        const float2 m = untangle(p, s, k1);
        if (p.out_f16) {
            half[k1] = __floats2half2_rn(m.x, m.x);
            half[M - 1 - k1] = __floats2half2_rn(m.y, m.y);
        } else {
            st_stream(out + k1, m.x, m.x);
            st_stream(out + (M - 1 - k1), m.y, m.y);
        }
    }
}

// ... and the pixel stage on it: the column (s, s) of the one frame, then pixel_passes as below (the launch asks for LDS for the column
// and its samples, which the W-point image alone would not hold).
template <uint32_t NT, uint32_t WN>
__device__ __forceinline__ void pixel_epilogue_real(const Params &p, float2 *s, long long row, uint32_t tid)
{
    constexpr uint32_t M = WN - 1, K = WN / 2, kPer = (K + NT - 1) / NT;
    float2 mg[kPer];
#pragma unroll
    for (uint32_t i = 0; i < kPer; ++i) {
        const uint32_t k1 = tid + NT * i;
        mg[i] = k1 < K ? untangle(p, s, k1) : make_float2(0.0f, 0.0f);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < kPer; ++i) {
        const uint32_t k1 = tid + NT * i;
        if (k1 < K) {
            s[k1] = make_float2(mg[i].x, mg[i].x);
            s[M - 1 - k1] = make_float2(mg[i].y, mg[i].y);
        }
    }
    __syncthreads();
    pixel_passes<NT>(p, s, s + M + 1, M, false, 0u, row, -1, tid);
}

// The pixel stage on the transform's own LDS image (magnitude_in -> color_for -> put_pixel, simple_spectrogram.rs:141-161), as
// the two-pass kernel of sgx_kernels.hip does it on magnitudes from HBM: every thread first takes its bins' magnitudes
// into registers (all reads of the transform happen before anything is written over it), the column goes to s[0 .. M), the
// interpolated samples behind it, then one thread per row.  A mono stream carries two frames per transform: .x / .y of a
// bin are the two columns, each an (s, s) pixel.  256-level palettes without the diverging branch whose thresholds pass
// the seed proof only (mixed_can_fuse_render); everything else takes the two-kernel route.
template <uint32_t NT, uint32_t WN>
__device__ __forceinline__ void pixel_epilogue(const Params &p, float2 *s, uint32_t pair, long long row_a, long long row_b, uint32_t tid)
{
    constexpr uint32_t M = WN - 1, kPer = (M + NT - 1) / NT;
    float2 mg[kPer];
#pragma unroll
    for (uint32_t i = 0; i < kPer; ++i) {
        const uint32_t j = tid + NT * i;
        mg[i] = make_float2(0.0f, 0.0f);
        if (j < M) {
            const uint32_t w = p.split[j];
            const float2 a = s[w & 0xffffu], b = s[w >> 16];
            const float sre = a.x + b.x, sim = a.y - b.y;
            const float dre = a.x - b.x, dim = a.y + b.y;
            mg[i] = make_float2(mag_of(sre, sim, 0.5f * p.scale), mag_of(dre, dim, 0.5f * p.scale));
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < kPer; ++i) {
        const uint32_t j = tid + NT * i;
        if (j < M) s[j] = mg[i];
    }
    __syncthreads();
    pixel_passes<NT>(p, s, s + M + 1, M, p.mono_pairs != 0, pair, row_a, row_b, tid);
}

__global__ void __launch_bounds__(1024) stft_mixed_kernel(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint32_t pair = blockIdx.y;
    long long row_a, row_b;
    Source src;
    frame_source(p, pair, src, row_a, row_b);

    for (uint32_t st = 0; st < p.n_stages; ++st) {
        const uint32_t code = p.ra[st] * 8 + p.rb[st];  // uniform
        DynGeo g;
        g.m_ = p.m[st];
        g.count_ = p.P / (p.ra[st] * p.rb[st]);
        g.qs_ = p.q_stride[st];
        g.bs_ = p.blk_stride[st];
        g.W_ = p.real ? (p.W + 1) / 2 : p.W;
        g.nt_ = nt;
        g.inv_m_ = p.inv_m[st];
        g.inv_pad_ = p.inv_pad;
        g.first_ = st == 0;
        const float2 *tw = p.tw + p.tw_off[st];
        switch (code) {
#define X(A, B) case A * 8 + B: stage<A, B, DynGeo, -1>(s, p, tw, g, src, tid); break;
            MIX_STAGE_CASES(X)
#undef X
        default: break;
        }
    }
    if (p.real) untangle_store(p, s, row_a, tid, nt);
    else split_store(p, s, pair, row_a, row_b, tid, nt);
}

// The two lengths the application produces (0.05 s at 48 and 44.1 kHz), three stages each, everything about the plan a
// compile-time constant.  PAD = R2 (even) -> position(i) = i + i / R2; the host checks that its own plan for the length is
// exactly this one before it launches these (mixed_init).
template <int P_, int R0A, int R0B, int R1A, int R1B, int R2A, int R2B, int NT_>
struct Fixed3 {
    static constexpr uint32_t P = P_, W = (P_ + 1) / 2, NT = NT_;   // W: the non-zero inputs (real-input mode runs odd P)
    static constexpr uint32_t R0 = R0A * R0B, R1 = R1A * R1B, R2 = R2A * R2B;
    static constexpr uint32_t M0 = P / R0, M1 = M0 / R1, M2 = 1;
    static constexpr uint32_t PAD = R2 % 2 == 0 ? R2 : 0;   // an odd lane stride needs no padding
    static constexpr uint32_t pp(uint32_t i) { return i + (PAD ? i / (PAD ? PAD : 1) : 0); }
    static constexpr uint32_t TW1 = M0 * (R0 - 1);   // offset of the second stage's twiddle rows
    static_assert(M1 == R2 && M0 % R2 == 0 && R0 * R1 * R2 == P, "plan shape");
};

template <int P_, int R0A, int R0B, int R1A, int R1B, int R2A, int R2B, int R3A, int R3B, int NT_>
struct Fixed4 {
    static constexpr uint32_t P = P_, W = (P_ + 1) / 2, NT = NT_;   // W: the non-zero inputs (real-input mode runs odd P)
    static constexpr uint32_t R0 = R0A * R0B, R1 = R1A * R1B, R2 = R2A * R2B, R3 = R3A * R3B;
    static constexpr uint32_t M0 = P / R0, M1 = M0 / R1, M2 = M1 / R2;
    static constexpr uint32_t PAD = R3 % 2 == 0 ? R3 : 0;
    static constexpr uint32_t pp(uint32_t i) { return i + (PAD ? i / (PAD ? PAD : 1) : 0); }
    static constexpr uint32_t TW1 = M0 * (R0 - 1), TW2 = TW1 + M1 * (R1 - 1);
    static_assert(M2 == R3 && M1 % R3 == 0 && M0 % R3 == 0 && R0 * R1 * R2 * R3 == P, "plan shape");
};

template <typename F, int R0A, int R0B, int R1A, int R1B, int R2A, int R2B, bool REAL>
__global__ void __launch_bounds__(F::NT, F::NT <= 256 ? 4 : 8) stft_mixed_fixed_kernel(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const uint32_t tid = threadIdx.x;
    const uint32_t pair = blockIdx.y;
    long long row_a, row_b;
    Source src;
    frame_source(p, pair, src, row_a, row_b);
    using G0 = FixGeo<F::M0, F::P / F::R0, F::pp(F::M0), F::pp(F::P), F::W, F::PAD, F::NT, true>;
    stage<R0A, R0B, G0, REAL>(s, p, p.tw, G0{}, src, tid);
    stage<R1A, R1B>(s, p, p.tw + F::TW1, FixGeo<F::M1, F::P / F::R1, F::pp(F::M1), F::pp(F::M0), F::W, F::PAD, F::NT, false>{}, src, tid);
    stage<R2A, R2B>(s, p, p.tw, FixGeo<1, F::P / F::R2, 1, F::pp(F::M1), F::W, F::PAD, F::NT, false>{}, src, tid);
    if constexpr (REAL) {   // P is the WINDOW here
        if (p.render) pixel_epilogue_real<F::NT, F::P>(p, s, row_a, tid);
        else untangle_store(p, s, row_a, tid, F::NT);
    } else if (p.render) {
        pixel_epilogue<F::NT, F::W>(p, s, pair, row_a, row_b, tid);
    } else {
        split_store(p, s, pair, row_a, row_b, tid, F::NT);
    }
}

// Real-input mode from PCM to pixels, TWO frames per workgroup (the two application plans, MIX_REAL2_RENDER_PLANS): each half of the
// workgroup transforms one frame on its own W-point image; then the pixel stage runs ONCE, with every thread, over both columns -- .x / .y
// of a bin, as for a frame pair -- where a mono column alone would carry the same value in both components and do every sum twice.
// (waves per SIMD the LDS lets stay resident -- two images, or the column with ~2300 samples behind it -- so that the register budget is
// no tighter than the occupancy the kernel can have anyway)
template <typename F>
constexpr unsigned real2_waves_per_simd()
{
    constexpr size_t lds = (size_t)(2 * F::pp(F::P) > F::P + 2304 ? 2 * F::pp(F::P) : F::P + 2304) * sizeof(float2);
    constexpr size_t wgs = 160 * 1024 / lds < 1 ? 1 : 160 * 1024 / lds;
    constexpr size_t waves = wgs * (2 * F::NT / 64) / 4;
    return waves < 1 ? 1u : (waves > 8 ? 8u : (unsigned)waves);
}

template <typename F, int R0A, int R0B, int R1A, int R1B, int R2A, int R2B>
__global__ void __launch_bounds__(2 * F::NT, real2_waves_per_simd<F>()) stft_mixed_real2_render_kernel(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    constexpr uint32_t NT = F::NT, IMG = F::pp(F::P), M = F::P - 1, K = F::P / 2, kPer = (K + NT - 1) / NT;
    const uint32_t tid = threadIdx.x, half = tid >= NT ? 1u : 0u, ltid = tid - half * NT;
    float2 *img = s + half * IMG;
    const unsigned long long fa = 2ull * blockIdx.x, f = fa + half;              // rows fa, fa + 1 of this launch
    const unsigned long long fc = f < p.n_frames ? f : p.n_frames - 1;            // no second frame: the last one again, never stored
    Source src;
    src.a = src.b = p.pcm + (size_t)((p.first_frame + fc) * p.H);
    src.cl = src.cr = 0;
    src.data_b = true;
    using G0 = FixGeo<F::M0, F::P / F::R0, F::pp(F::M0), F::pp(F::P), F::W, F::PAD, NT, true>;
    stage<R0A, R0B, G0, 1>(img, p, p.tw, G0{}, src, ltid);
    stage<R1A, R1B>(img, p, p.tw + F::TW1, FixGeo<F::M1, F::P / F::R1, F::pp(F::M1), F::pp(F::M0), F::W, F::PAD, NT, false>{}, src, ltid);
    stage<R2A, R2B>(img, p, p.tw, FixGeo<1, F::P / F::R2, 1, F::pp(F::M1), F::W, F::PAD, NT, false>{}, src, ltid);
    float2 mg[kPer];
#pragma unroll
    for (uint32_t i = 0; i < kPer; ++i) {
        const uint32_t k1 = ltid + NT * i;
        mg[i] = k1 < K ? untangle(p, img, k1) : make_float2(0.0f, 0.0f);
    }
    __syncthreads();
    float *col = reinterpret_cast<float *>(s);   // column element j: (frame fa, frame fa + 1) = col[2 j], col[2 j + 1]
#pragma unroll
    for (uint32_t i = 0; i < kPer; ++i) {
        const uint32_t k1 = ltid + NT * i;
        if (k1 < K) {
            col[2 * k1 + half] = mg[i].x;
            col[2 * (M - 1 - k1) + half] = mg[i].y;
        }
    }
    __syncthreads();
    pixel_passes<2 * NT>(p, s, s + M + 1, M, true, 0u, (long long)fa, (long long)fa + 1, tid);
}

template <typename F, int R0A, int R0B, int R1A, int R1B, int R2A, int R2B, int R3A, int R3B, bool REAL>
__global__ void __launch_bounds__(F::NT, F::NT == 256 ? 4 : (F::NT == 512 ? 8 : 4)) stft_mixed_fixed4_kernel(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const uint32_t tid = threadIdx.x;
    const uint32_t pair = blockIdx.y;
    long long row_a, row_b;
    Source src;
    frame_source(p, pair, src, row_a, row_b);
    using G0 = FixGeo<F::M0, F::P / F::R0, F::pp(F::M0), F::pp(F::P), F::W, F::PAD, F::NT, true>;
    stage<R0A, R0B, G0, REAL>(s, p, p.tw, G0{}, src, tid);
    stage<R1A, R1B>(s, p, p.tw + F::TW1, FixGeo<F::M1, F::P / F::R1, F::pp(F::M1), F::pp(F::M0), F::W, F::PAD, F::NT, false>{}, src, tid);
    stage<R2A, R2B>(s, p, p.tw + F::TW2, FixGeo<F::M2, F::P / F::R2, F::pp(F::M2), F::pp(F::M1), F::W, F::PAD, F::NT, false>{}, src, tid);
    stage<R3A, R3B>(s, p, p.tw, FixGeo<1, F::P / F::R3, 1, F::pp(F::M2), F::W, F::PAD, F::NT, false>{}, src, tid);
    if constexpr (REAL) {
        if (p.render) pixel_epilogue_real<F::NT, F::P>(p, s, row_a, tid);
        else untangle_store(p, s, row_a, tid, F::NT);
    } else if (p.render) {
        pixel_epilogue<F::NT, F::W>(p, s, pair, row_a, row_b, tid);
    } else {
        split_store(p, s, pair, row_a, row_b, tid, F::NT);
    }
}

// ---- chirp-z (Bluestein) through the same stages: F[k] = c[k] sum_{n<W} (z[n] c[n]) conj(c)[k - n], c[n] = exp(-i pi n^2 / P), as a
// circular convolution of length L = pow2 >= P + W - 1 (stft_bluestein.hip has the derivation and the first implementation, a
// radix-4 ladder): forward stages (the first one reads z[n] hann[n] c[n] from the stream, rows past W are zero), pointwise
// product with B^ = FFT_L(conj chirp) / L stored at the image's own positions, the stages inverted in reverse order (natural
// order back at position n + n / 16), split with the second chirp factor.
template <typename F, int R0A, int R0B, int R1A, int R1B, int R2A, int R2B, bool REAL>
__global__ void __launch_bounds__(F::NT, F::NT >= 256 ? 4 : 2) chirpz3_kernel(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const uint32_t tid = threadIdx.x;
    const uint32_t pair = blockIdx.y;
    long long row_a, row_b;
    Source src;
    frame_source(p, pair, src, row_a, row_b);
    using G0f = FixGeo<F::M0, F::P / F::R0, F::pp(F::M0), F::pp(F::P), 0, F::PAD, F::NT, true>;
    using G0 = FixGeo<F::M0, F::P / F::R0, F::pp(F::M0), F::pp(F::P), 0, F::PAD, F::NT, false>;
    using G1 = FixGeo<F::M1, F::P / F::R1, F::pp(F::M1), F::pp(F::M0), 0, F::PAD, F::NT, false>;
    using G2 = FixGeo<1, F::P / F::R2, 1, F::pp(F::M1), 0, F::PAD, F::NT, false>;
    stage<R0A, R0B, G0f, REAL>(s, p, p.tw, G0f{REAL ? (p.W + 1) / 2 : p.W}, src, tid);   // (real-input mode: ceil(W / 2) sample pairs)
    stage<R1A, R1B>(s, p, p.tw + F::TW1, G1{}, src, tid);
    stage<R2A, R2B>(s, p, p.tw, G2{}, src, tid);
    for (uint32_t i = tid; i < F::pp(F::P); i += F::NT) s[i] = cmul(s[i], p.bhat[i]);
    __syncthreads();
    stage_inv<R2A, R2B>(s, p.tw, G2{}, tid);
    stage_inv<R1A, R1B>(s, p.tw + F::TW1, G1{}, tid);
    stage_inv<R0A, R0B>(s, p.tw, G0{}, tid);
    if constexpr (REAL) untangle_store(p, s, row_a, tid, F::NT);
    else split_store(p, s, pair, row_a, row_b, tid, F::NT);
}

template <typename F, int R0A, int R0B, int R1A, int R1B, int R2A, int R2B, int R3A, int R3B, bool REAL>
__global__ void __launch_bounds__(F::NT, 4) chirpz4_kernel(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const uint32_t tid = threadIdx.x;
    const uint32_t pair = blockIdx.y;
    long long row_a, row_b;
    Source src;
    frame_source(p, pair, src, row_a, row_b);
    using G0f = FixGeo<F::M0, F::P / F::R0, F::pp(F::M0), F::pp(F::P), 0, F::PAD, F::NT, true>;
    using G0 = FixGeo<F::M0, F::P / F::R0, F::pp(F::M0), F::pp(F::P), 0, F::PAD, F::NT, false>;
    using G1 = FixGeo<F::M1, F::P / F::R1, F::pp(F::M1), F::pp(F::M0), 0, F::PAD, F::NT, false>;
    using G2 = FixGeo<F::M2, F::P / F::R2, F::pp(F::M2), F::pp(F::M1), 0, F::PAD, F::NT, false>;
    using G3 = FixGeo<1, F::P / F::R3, 1, F::pp(F::M2), 0, F::PAD, F::NT, false>;
    stage<R0A, R0B, G0f, REAL>(s, p, p.tw, G0f{REAL ? (p.W + 1) / 2 : p.W}, src, tid);
    stage<R1A, R1B>(s, p, p.tw + F::TW1, G1{}, src, tid);
    stage<R2A, R2B>(s, p, p.tw + F::TW2, G2{}, src, tid);
    stage<R3A, R3B>(s, p, p.tw, G3{}, src, tid);
    for (uint32_t i = tid; i < F::pp(F::P); i += F::NT) s[i] = cmul(s[i], p.bhat[i]);
    __syncthreads();
    stage_inv<R3A, R3B>(s, p.tw, G3{}, tid);
    stage_inv<R2A, R2B>(s, p.tw + F::TW2, G2{}, tid);
    stage_inv<R1A, R1B>(s, p.tw + F::TW1, G1{}, tid);
    stage_inv<R0A, R0B>(s, p.tw, G0{}, tid);
    if constexpr (REAL) untangle_store(p, s, row_a, tid, F::NT);
    else split_store(p, s, pair, row_a, row_b, tid, F::NT);
}

// L, the stages, threads (W 86 .. 5461; shorter windows keep the radix-4 ladder of stft_bluestein.hip)
#define CHIRP_PLANS3(X) X(512, 4, 1, 4, 2, 4, 4, 128) X(1024, 4, 1, 4, 4, 4, 4, 256) X(2048, 4, 2, 4, 4, 4, 4, 256) X(4096, 4, 4, 4, 4, 4, 4, 256)
#define CHIRP_PLANS4(X) X(8192, 4, 1, 4, 2, 4, 4, 4, 4, 512) X(16384, 4, 1, 4, 4, 4, 4, 4, 4, 1024)

struct ChirpTables {
    float2 *d_chirp = nullptr, *d_bhat = nullptr, *d_tw = nullptr;
    uint32_t L = 0, lds_points = 0;
    // real-input mode (a mono stream, every frame its own transform): the chirp-z transform of W points over ceil(W / 2) sample pairs
    // -- a convolution of half the length -- and the untangling twiddles w_2W^k
    ChirpTables *half = nullptr;
    float2 *d_twr = nullptr;
};

// P, the three stages (RA, RB), threads.  0.05 s at 48 / 44.1 / 32 / 16 / 8 / 88.2 kHz.  512 threads where a stage has more than
// 256 butterflies (62-64 registers: eight waves per SIMD; same-device A/B at 4800 points: 256 threads 1.41 ms per 100 000
// stereo frames, 320 1.31, 512 1.33; at 4410: 1.21 / 1.17 / 1.14).
#define MIX_FIXED_PLANS(X) X(4800, 5, 4, 5, 3, 4, 4, 512) X(4410, 7, 3, 5, 3, 7, 2, 512) X(3200, 5, 4, 5, 2, 4, 4, 512) \
                           X(1600, 5, 4, 5, 1, 4, 4, 512) X(800, 5, 2, 5, 1, 4, 4, 256) X(8820, 7, 3, 7, 3, 5, 4, 512) \
                           X(2048, 4, 2, 4, 4, 4, 4, 256) X(1024, 4, 1, 4, 4, 4, 4, 256) \
                           X(2400, 5, 3, 5, 2, 4, 4, 256) X(2205, 7, 3, 5, 3, 7, 1, 192) \
                           X(4096, 4, 4, 4, 4, 4, 4, 256) X(512, 4, 1, 4, 2, 4, 4, 128)
// (2400, 2205, 4096, 512: the W-point transforms of real-input mode at 48 and 44.1 kHz and at W 4096 / 512; every plan is instantiated for both modes.  Same-device
// A/B, mono, 262 144 frames at hop 93, rows / PCM -> pixels: 2400 points at 192 threads 2.07 / 3.86 ms, 256 2.03 / 3.25, 320 2.17 / 3.07,
// 512 2.23 / 2.77; 2205 points at 160 2.02 / 4.31, 192 1.97 / 3.89, 256 1.99 / 3.46, 320 2.22 / 3.27 -- the transform wants one
// butterfly per thread, the pixel stage behind it every thread it can get: real-input mode to pixels runs these two plans wider)
#define MIX_REAL_RENDER_PLANS(X) X(2400, 5, 3, 5, 2, 4, 4, 512) X(2205, 7, 3, 5, 3, 7, 1, 320)
// ... and, unless SGX_KM_REAL1 (A/B), every three-stage plan with TWO frames per workgroup (threads per FRAME here):
// stft_mixed_real2_render_kernel
#define MIX_REAL2_RENDER_PLANS(X) X(2400, 5, 3, 5, 2, 4, 4, 256) X(2205, 7, 3, 5, 3, 7, 1, 192) X(4800, 5, 4, 5, 3, 4, 4, 512) \
                                  X(4410, 7, 3, 5, 3, 7, 2, 512) X(4096, 4, 4, 4, 4, 4, 4, 256) X(1024, 4, 1, 4, 4, 4, 4, 256) \
                                  X(512, 4, 1, 4, 2, 4, 4, 128) X(1600, 5, 4, 5, 1, 4, 4, 256) X(800, 5, 2, 5, 1, 4, 4, 128) \
                                  X(3200, 5, 4, 5, 2, 4, 4, 256) X(8820, 7, 3, 7, 3, 5, 4, 512)
// four stages: 0.05 s at 96 / 192 / 176.4 kHz, and the 8192-point power of two
#define MIX_FIXED4_PLANS(X) X(9600, 4, 3, 5, 2, 5, 1, 4, 4, 1024) X(19200, 5, 3, 5, 1, 4, 4, 4, 4, 1024) \
                            X(17640, 5, 3, 7, 2, 4, 3, 7, 1, 1024) X(8192, 4, 1, 4, 2, 4, 4, 4, 4, 512)

}  // namespace mix

bool mixed_supported(uint32_t W)
{
    uint32_t n = 2 * W;
    if (W < 4 || n > 20480) return false;  // the transform lives in LDS: 8 bytes per point, all 160 KB of a CU at most
                                           // (192 kHz x 0.05 s: 2W = 19 200)
    for (uint32_t f : {2u, 3u, 5u, 7u})
        while (n % f == 0) n /= f;
    return n == 1;
}

namespace mix {

// The stage plan: P's factors 7, 5, 3, 4 (pairs of twos) and a last 2, grouped into stages of one or two factors with
// a product <= kMaxRadix -- fewest stages, then the smallest largest radix (registers), then the smallest sum.
// Stages with an odd factor come first (largest first), powers of two last (smallest first).  tests/test_host_logic.py restates it.
struct Plan { std::vector<std::pair<uint32_t, uint32_t>> stages; };

static void plan_search(std::vector<uint32_t> &rest, std::vector<std::pair<uint32_t, uint32_t>> &cur,
                        std::vector<std::pair<uint32_t, uint32_t>> &best, uint64_t &best_key)
{
    if (rest.empty()) {
        uint64_t mx = 0, sum = 0;
        for (auto &g : cur) { mx = std::max<uint64_t>(mx, g.first * g.second); sum += g.first * g.second; }
        const uint64_t key = ((uint64_t)cur.size() << 40) | (mx << 20) | sum;
        if (key < best_key) { best_key = key; best = cur; }
        return;
    }
    const uint32_t f = rest.back();   // the smallest remaining factor goes alone or with any other
    rest.pop_back();
    cur.push_back({f, 1});
    plan_search(rest, cur, best, best_key);
    cur.pop_back();
    for (size_t i = 0; i < rest.size(); ++i) {
        if (i > 0 && rest[i] == rest[i - 1]) continue;
        const uint32_t g = rest[i];
        if (f * g > kMaxRadix || (f == 2 && g == 2)) continue;
        rest.erase(rest.begin() + (long)i);
        cur.push_back({std::max(f, g), std::min(f, g)});
        plan_search(rest, cur, best, best_key);
        cur.pop_back();
        rest.insert(rest.begin() + (long)i, g);
    }
    rest.push_back(f);
}

static bool make_plan(uint32_t P, Plan &plan)
{
    std::vector<uint32_t> factors;   // descending
    uint32_t n = P;
    for (uint32_t f : {7u, 5u})
        while (n % f == 0) { factors.push_back(f); n /= f; }
    std::vector<uint32_t> threes;
    while (n % 3 == 0) { threes.push_back(3); n /= 3; }
    while (n % 4 == 0) { factors.push_back(4); n /= 4; }
    factors.insert(factors.end(), threes.begin(), threes.end());
    if (n % 2 == 0) { factors.push_back(2); n /= 2; }
    if (n != 1) return false;
    std::sort(factors.begin(), factors.end(), std::greater<uint32_t>());
    std::vector<std::pair<uint32_t, uint32_t>> cur, best;
    uint64_t best_key = ~0ull;
    plan_search(factors, cur, best, best_key);
    auto odd = [](const std::pair<uint32_t, uint32_t> &g) { return ((g.first * g.second) & (g.first * g.second - 1)) != 0; };
    std::stable_sort(best.begin(), best.end(), [&](const auto &a, const auto &b) {
        if (odd(a) != odd(b)) return odd(a);
        if (odd(a)) return a.first * a.second > b.first * b.second;
        return a.first * a.second < b.first * b.second;   // powers of two: the largest last (its padding is the thinnest)
    });
    plan.stages = best;
    return !best.empty() && best.size() <= (size_t)kMaxStages;
}

}  // namespace mix

// Tables of a P-point plan.  real: P is the WINDOW of real-input mode -- the split table pairs Z[k] with Z[P - k], k = 1 .. P / 2, and
// the untangling twiddles w_2P^k come with it.  hipErrorInvalidValue: no plan for P (the caller of the real-input half goes without).
static hipError_t build_tables(uint32_t P, bool real, mix::MixTables **out)
{
    using namespace mix;
    auto *t = new MixTables();
    const uint32_t M = real ? P / 2 : P / 2 - 1;
    Plan plan;
    if (!make_plan(P, plan)) { delete t; return hipErrorInvalidValue; }
    t->n_stages = (uint32_t)plan.stages.size();
    std::vector<uint32_t> radix(t->n_stages);
    std::vector<float2> tw;
    uint32_t ns = P;
    for (uint32_t i = 0; i < t->n_stages; ++i) {
        t->ra[i] = plan.stages[i].first;
        t->rb[i] = plan.stages[i].second;
        radix[i] = t->ra[i] * t->rb[i];
        t->m[i] = ns / radix[i];
        t->inv_m[i] = 1.0f / (float)t->m[i];
        t->tw_off[i] = (uint32_t)tw.size();
        if (t->m[i] > 1)
            for (uint32_t j = 0; j < t->m[i]; ++j)
                for (uint32_t k = 1; k < radix[i]; ++k) {
                    const unsigned long long e = ((unsigned long long)j * k) % ns;
                    const double ang = -2.0 * M_PI * (double)e / (double)ns;
                    double cs = cos(ang), sn = sin(ang);
                    if (e == 0) { cs = 1.0; sn = 0.0; }
                    if (4 * e == ns) { cs = 0.0; sn = -1.0; }
                    if (2 * e == ns) { cs = -1.0; sn = 0.0; }
                    if (4 * e == 3ull * ns) { cs = 0.0; sn = 1.0; }
                    tw.push_back(make_float2((float)cs, (float)sn));
                }
        ns = t->m[i];
    }
    // one point of padding per R_last points (see stage()) when R_last is even and the padding costs no resident
    // workgroup (it cannot be afforded at the largest lengths: P = 20480 fills the 160 KB by itself)
    const size_t kLds = 160 * 1024;
    const uint32_t r_last = radix[t->n_stages - 1];
    auto resident_of = [&](uint32_t points) {
        return (unsigned)std::max<size_t>(1, std::min<size_t>(8, kLds / ((size_t)points * sizeof(float2))));
    };
    auto threads_of = [&](uint32_t points) {
        // the widest stage's butterflies in one round where the resident workgroups leave room (16 waves per CU at this
        // kernel's register budget), whole waves  (measured at 4800 points: 256 threads 1.72 ms, 192 2.13, 320 2.11)
        unsigned widest = 0;
        for (uint32_t i = 0; i < t->n_stages; ++i) widest = std::max(widest, P / radix[i]);
        unsigned cap = std::max((1024u / resident_of(points)) / 64 * 64, 64u);
        return std::max(std::min(cap, (widest + 63) / 64 * 64), 64u);
    };
    uint32_t best_d = 0;
    // (round 4: "no resident workgroup" counts up to four -- the registers hold no more than four 256-thread workgroups of the run-time
    // geometry or 512-thread ones of a compile-time plan anyway.  Same-device A/B: 4000 points +15 %, 3600 +-0, 3888 -4 %, and the
    // 4096-point plan of real-input mode at W 4096 becomes the compile-time one: 41 -> 68 M frames/s.  gpurun_out/r4_km_pad.log)
    if (r_last % 2 == 0 && (size_t)(P + P / r_last) * sizeof(float2) <= kLds && resident_of(P + P / r_last) >= std::min(resident_of(P), 4u)) best_d = r_last;
    t->pad_every = best_d;
    t->lds_points = best_d ? P + P / best_d : P;
    auto padpos = [&](uint32_t i) { return i + (t->pad_every ? i / t->pad_every : 0u); };
    for (uint32_t i = 0; i < t->n_stages; ++i) {
        t->q_stride[i] = t->m[i] == 1 ? 1u : padpos(t->m[i]);          // m > 1 is a multiple of R_last, hence of D
        t->blk_stride[i] = padpos(t->m[i] * radix[i]);
    }
    t->threads = threads_of(t->lds_points);
    auto is_plan = [&](uint32_t Pn, std::initializer_list<uint32_t> ab) {   // the compiled plan of a fixed kernel == the host's
        if (P != Pn || t->n_stages * 2 != ab.size()) return false;
        uint32_t i = 0, off = 0;
        for (auto it = ab.begin(); it != ab.end(); ++i) {
            const uint32_t a = *it++, b = *it++;
            if (t->ra[i] != a || t->rb[i] != b || (t->m[i] > 1 && t->tw_off[i] != off)) return false;
            off += t->m[i] * (a * b - 1);
        }
        const uint32_t rl = radix[t->n_stages - 1];
        return t->pad_every == (rl % 2 == 0 ? rl : 0u);
    };
#ifndef SGX_MIX_NO_FIXED   // (A/B builds)
#define X(Pn, A0, B0, A1, B1, A2, B2, N) if (is_plan(Pn, {A0, B0, A1, B1, A2, B2})) t->fixed = Pn;
    MIX_FIXED_PLANS(X)
#undef X
#define X(Pn, A0, B0, A1, B1, A2, B2, A3, B3, N) if (is_plan(Pn, {A0, B0, A1, B1, A2, B2, A3, B3})) t->fixed = Pn;
    MIX_FIXED4_PLANS(X)
#undef X
#endif
    // bin K = k1 + r1 (k2 + r2 (k3 + ...)) ends at k1 m1 + k2 m2 + ...
    std::vector<uint32_t> pos(P);
    for (uint32_t K = 0; K < P; ++K) {
        uint32_t k = K, at = 0;
        for (uint32_t i = 0; i < t->n_stages; ++i) {
            at += (k % radix[i]) * t->m[i];
            k /= radix[i];
        }
        pos[K] = padpos(at);
    }
    std::vector<uint32_t> split(M);
    for (uint32_t j = 0; j < M; ++j) split[j] = pos[j + 1] | (pos[P - (j + 1)] << 16);   // positions < 21 120 < 2^16
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&t->d_split), (size_t)std::max<uint32_t>(M, 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(t->d_split, split.data(), (size_t)M * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&t->d_tw), std::max<size_t>(tw.size(), 1) * sizeof(float2));
    if (e == hipSuccess && !tw.empty()) e = hipMemcpy(t->d_tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e == hipSuccess && real) {
        std::vector<float2> twr(std::max<uint32_t>(M, 1));
        for (uint32_t j = 0; j < M; ++j) {
            const uint32_t k = j + 1;   // w_2P^k, k <= P / 2: the first quadrant of the circle, its end exact
            const double ang = -M_PI * (double)k / (double)P;
            twr[j] = 2 * k == P ? make_float2(0.0f, -1.0f) : make_float2((float)cos(ang), (float)sin(ang));
        }
        e = hipMalloc(reinterpret_cast<void **>(&t->d_twr), twr.size() * sizeof(float2));
        if (e == hipSuccess) e = hipMemcpy(t->d_twr, twr.data(), twr.size() * sizeof(float2), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        mixed_destroy(t);
        return e;
    }
    *out = t;
    return hipSuccess;
}

hipError_t mixed_init(sgx_ctx *c, void **out)
{
    mix::MixTables *t = nullptr;
    hipError_t e = build_tables(c->P, false, &t);
    if (e != hipSuccess) return e;
    if (c->C == 1 && c->W >= 8) {   // a mono stream: the W-point plan of real-input mode, where W has one
        e = build_tables(c->W, true, &t->half);
        if (e != hipSuccess && e != hipErrorInvalidValue) {
            mixed_destroy(t);
            return e;
        }
    }
    *out = t;
    return hipSuccess;
}

uint32_t mixed_fixed_plan(const void *tables) { return tables ? (uint32_t)static_cast<const mix::MixTables *>(tables)->fixed : 0u; }

void mixed_destroy(void *tables)
{
    auto *t = static_cast<mix::MixTables *>(tables);
    if (!t) return;
    if (t->half) mixed_destroy(t->half);
    if (t->d_split) (void)hipFree(t->d_split);
    if (t->d_tw) (void)hipFree(t->d_tw);
    if (t->d_twr) (void)hipFree(t->d_twr);
    delete t;
}

// a mono stream whose frames each get their own transform (the default; SGX_FLAG_COMPLEX_MONO: as (s, s) through the 2W-point plan)
bool mixed_real_serves(const sgx_ctx *c, const void *tables, uint32_t channels)
{
    const auto *t = static_cast<const mix::MixTables *>(tables);
    return t && t->half && channels == 1 && !(c->cfg.flags & SGX_FLAG_PAIRED_FRAMES) && !(c->cfg.flags & SGX_FLAG_COMPLEX_MONO);
}

// real-input mode from PCM to pixels: a compile-time W-point plan, the palette conditions of the fused pixel stage, W / 2 bin pairs in
// registers (ten per thread at most); the launch sizes its LDS for the column and its samples
static bool real_fuses_render(const sgx_ctx *c, const mix::MixTables *t)
{
    const auto *h = t->half;
    if (!h || !h->fixed || c->pal.stereo || c->pal.segments || c->pal.n != 256 || !c->d_pal_seed || !wg4096_seed_is_within_one(c)) return false;
    unsigned nt = 0;
#define X(Pn, A0, B0, A1, B1, A2, B2, N) if (h->fixed == Pn) nt = N;
    MIX_FIXED_PLANS(X)
    MIX_REAL_RENDER_PLANS(X)
#undef X
#define X(Pn, A0, B0, A1, B1, A2, B2, A3, B3, N) if (h->fixed == Pn) nt = N;
    MIX_FIXED4_PLANS(X)
#undef X
    return nt && c->W / 2 <= nt * 10u && ((size_t)c->M + 1 + c->tab.samples.size()) * sizeof(float2) <= 160 * 1024;
}

bool mixed_can_fuse_render(const sgx_ctx *c, const void *tables)
{
    const auto *t = static_cast<const mix::MixTables *>(tables);
    // (a mono stream that real-input mode cannot take to pixels goes the two-kernel route on real-input ROWS, not through the 2W-point
    // plan as an (s, s) frame: the pixels of a context are those of its rows)
    if (mixed_real_serves(c, tables, c->C)) return real_fuses_render(c, t);
    if (!t || !t->fixed || c->pal.stereo || c->pal.segments || c->pal.n != 256 || !c->d_pal_seed || !wg4096_seed_is_within_one(c)) return false;
    // the column and its interpolated samples on the transform's LDS image; ten bins per thread in registers at most
    unsigned nt = 0;
#define X(Pn, A0, B0, A1, B1, A2, B2, N) if (t->fixed == Pn) nt = N;
    MIX_FIXED_PLANS(X)
#undef X
#define X(Pn, A0, B0, A1, B1, A2, B2, A3, B3, N) if (t->fixed == Pn) nt = N;
    MIX_FIXED4_PLANS(X)
#undef X
    return nt && (size_t)c->M + 1 + c->tab.samples.size() <= t->lds_points && c->M <= nt * 10u;
}

static hipError_t launch_mixed(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                               size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags, bool out_f16, uint8_t *d_rgba);

hipError_t launch_stft_mixed(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                             size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags, bool out_f16)
{
    return launch_mixed(c, tables, d_pcm, channels, pairs, first_frame, n_frames, total_frames, d_mags, out_f16, nullptr);
}

hipError_t launch_render_mixed(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs, size_t first_frame,
                               size_t n_frames, size_t total_frames, uint8_t *d_rgba)
{
    return launch_mixed(c, tables, d_pcm, channels, pairs, first_frame, n_frames, total_frames, nullptr, false, d_rgba);
}

static hipError_t launch_mixed(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                               size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags, bool out_f16, uint8_t *d_rgba)
{
    using namespace mix;
    if (n_frames == 0) return hipSuccess;
    const auto *t = static_cast<const MixTables *>(tables);
    // a mono stream, every frame its own transform (the default): real-input mode on the W-point plan, where there is one
    const bool real = mixed_real_serves(c, tables, channels) && (!d_rgba || real_fuses_render(c, t));
    if (real) t = t->half;
    Params p{};
    if (d_rgba) {
        p.render = 1;
        p.rgba = d_rgba;
        p.R = c->R;
        p.n_samples = (uint32_t)c->tab.samples.size();
        p.interp = c->cfg.interp;
        p.rows = c->d_rows;
        p.samples = c->d_samples;
        p.pal = c->d_pal_seed;
        lut_seed_coefficients(c, p.guess_a, p.guess_b);
    }
    p.pcm = d_pcm;
    p.window = c->d_window;
    p.tw = t->d_tw;
    p.split = t->d_split;
    p.W = c->W;
    p.P = real ? c->W : c->P;
    p.real = real ? 1u : 0u;
    p.twr = t->d_twr;
    p.H = c->H;
    p.C = channels;
    p.pairs = pairs;
    p.scale = 2.0f / (float)c->W;
    p.n_stages = t->n_stages;
    p.out_f16 = out_f16 ? 1u : 0u;
    p.inv_pad = t->pad_every ? 1.0f / (float)t->pad_every : 0.0f;
    for (uint32_t i = 0; i < t->n_stages; ++i) {
        p.ra[i] = t->ra[i];
        p.rb[i] = t->rb[i];
        p.m[i] = t->m[i];
        p.tw_off[i] = t->tw_off[i];
        p.inv_m[i] = t->inv_m[i];
        p.q_stride[i] = t->q_stride[i];
        p.blk_stride[i] = t->blk_stride[i];
    }
    p.first_frame = first_frame;
    p.n_frames = n_frames;
    p.total_frames = total_frames;
    // (l, r) of a channel pair as one 8-byte word: even channel count and an 8-byte aligned stream
    p.vec2 = (channels >= 2 && channels % 2 == 0 && reinterpret_cast<uintptr_t>(d_pcm) % 8 == 0) ? 1u : 0u;
    if (real) p.vec2 = c->W % 2 == 0 ? 1u : 0u;   // sample PAIRS of one channel as 8-byte words (any hop, any alignment: see stage())
    size_t lds = (size_t)t->lds_points * sizeof(float2);
    if (real && d_rgba) lds = std::max(lds, ((size_t)c->M + 1 + c->tab.samples.size()) * sizeof(float2));   // the column and its samples
    bool two_frames = false;   // real-input mode to pixels at the two application plans: two frames (two images) per workgroup
#ifndef SGX_KM_REAL1
#define X(Pn, A0, B0, A1, B1, A2, B2, N) if (real && d_rgba && t->fixed == Pn) two_frames = true;
    MIX_REAL2_RENDER_PLANS(X)
#undef X
#endif
    if (two_frames) lds = std::max(lds, 2 * (size_t)t->lds_points * sizeof(float2));
    const unsigned threads = t->threads;
    hipError_t attr_err = hipSuccess;
    auto go = [&](auto kernel, unsigned nt, dim3 grid) {
        if (lds > 64 * 1024) {  // per launch: the attribute is per device, and a process may hold contexts on several
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { attr_err = e; return; }
        }
        hipLaunchKernelGGL(kernel, grid, dim3(nt), lds, c->stream, p);
    };
    auto launch = [&](dim3 grid) {
        if (two_frames)
            switch (t->fixed) {
#define X(Pn, A0, B0, A1, B1, A2, B2, N)                                                                                                  \
    case Pn:                                                                                                                              \
        go(stft_mixed_real2_render_kernel<Fixed3<Pn, A0, B0, A1, B1, A2, B2, N>, A0, B0, A1, B1, A2, B2>, 2 * N, dim3((grid.x + 1) / 2, 1)); \
        return;
                MIX_REAL2_RENDER_PLANS(X)
#undef X
            default: break;
            }
        if (real && d_rgba)
            switch (t->fixed) {
#define X(Pn, A0, B0, A1, B1, A2, B2, N) \
    case Pn: go(stft_mixed_fixed_kernel<Fixed3<Pn, A0, B0, A1, B1, A2, B2, N>, A0, B0, A1, B1, A2, B2, true>, N, grid); return;
                MIX_REAL_RENDER_PLANS(X)
#undef X
            default: break;
            }
        switch (t->fixed) {
#define X(Pn, A0, B0, A1, B1, A2, B2, N)                                                                                      \
    case Pn:                                                                                                                  \
        if (real) go(stft_mixed_fixed_kernel<Fixed3<Pn, A0, B0, A1, B1, A2, B2, N>, A0, B0, A1, B1, A2, B2, true>, N, grid);   \
        else go(stft_mixed_fixed_kernel<Fixed3<Pn, A0, B0, A1, B1, A2, B2, N>, A0, B0, A1, B1, A2, B2, false>, N, grid);      \
        break;
            MIX_FIXED_PLANS(X)
#undef X
#define X(Pn, A0, B0, A1, B1, A2, B2, A3, B3, N)                                                                                               \
    case Pn:                                                                                                                                   \
        if (real) go(stft_mixed_fixed4_kernel<Fixed4<Pn, A0, B0, A1, B1, A2, B2, A3, B3, N>, A0, B0, A1, B1, A2, B2, A3, B3, true>, N, grid);   \
        else go(stft_mixed_fixed4_kernel<Fixed4<Pn, A0, B0, A1, B1, A2, B2, A3, B3, N>, A0, B0, A1, B1, A2, B2, A3, B3, false>, N, grid);      \
        break;
            MIX_FIXED4_PLANS(X)
#undef X
        default: go(stft_mixed_kernel, threads, grid); break;
        }
    };
    const size_t max_chunk = 1u << 30;
    if (channels == 1 && (c->cfg.flags & SGX_FLAG_PAIRED_FRAMES)) {
        p.mono_pairs = 1;
        p.mags = d_mags;
        const unsigned long long q0 = first_frame / 2, q1 = (first_frame + n_frames + 1) / 2;
        for (unsigned long long q = q0; q < q1; q += max_chunk) {
            const unsigned long long chunk = q1 - q < max_chunk ? q1 - q : max_chunk;
            p.pair_base = q;
            launch(dim3((unsigned)chunk, 1));
            hipError_t e = attr_err != hipSuccess ? attr_err : hipGetLastError();
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    for (size_t done = 0; done < n_frames; done += max_chunk) {
        const size_t chunk = n_frames - done < max_chunk ? n_frames - done : max_chunk;
        p.first_frame = first_frame + done;
        p.n_frames = chunk;
        p.mags = d_mags ? d_mags + done * (size_t)pairs * c->M * (out_f16 ? 1 : 2) : nullptr;
        if (d_rgba) p.rgba = d_rgba + done * (size_t)pairs * c->R * 4;
        launch(dim3((unsigned)chunk, pairs));
        hipError_t e = attr_err != hipSuccess ? attr_err : hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ---- chirp-z through the composite stages (see chirpz3_kernel) --------------------------------------------------------

namespace mix {

// host float64 radix-2 FFT (table set-up only)
static void fft_host(std::vector<double> &re, std::vector<double> &im)
{
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / (double)len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const double wr = cos(ang * (double)k), wi = sin(ang * (double)k);
                const size_t a = i + k, b = i + k + len / 2;
                const double tr = re[b] * wr - im[b] * wi, ti = re[b] * wi + im[b] * wr;
                re[b] = re[a] - tr; im[b] = im[a] - ti;
                re[a] += tr; im[a] += ti;
            }
    }
}

static uint32_t chirp_length(uint32_t W)
{
    uint32_t L = 1;
    while (L < 3 * W - 1) L <<= 1;   // P + W - 1
    return L;
}

// P points out of nz non-zero inputs: any power of two >= P + nz - 1 that has a plan
static uint32_t conv_length(uint32_t P, uint32_t nz)
{
    uint32_t L = 512;
    while (L < P + nz - 1) L <<= 1;
    return L;
}

}  // namespace mix

bool chirpz_supported(uint32_t W)
{
    if (W < 4 || 3ull * W - 1 > 16384) return false;
    const uint32_t L = mix::chirp_length(W);
    return L == 512 || L == 1024 || L == 2048 || L == 4096 || L == 8192 || L == 16384;
}

// Tables of the chirp-z transform of P points out of nz non-zero inputs (2W and W for an (l, r) frame; real: W and ceil(W / 2) sample
// pairs, with the untangling twiddles w_2P^k).  hipErrorInvalidValue: no plan for the convolution length.
static hipError_t build_chirp(uint32_t P, uint32_t nz, bool real, mix::ChirpTables **out)
{
    using namespace mix;
    const uint32_t L = real ? conv_length(P, nz) : chirp_length(nz);
    std::vector<uint32_t> radix;
#define X(Ln, A0, B0, A1, B1, A2, B2, N) if (L == Ln) radix = {A0 * B0, A1 * B1, A2 * B2};
    CHIRP_PLANS3(X)
#undef X
#define X(Ln, A0, B0, A1, B1, A2, B2, A3, B3, N) if (L == Ln) radix = {A0 * B0, A1 * B1, A2 * B2, A3 * B3};
    CHIRP_PLANS4(X)
#undef X
    if (radix.empty()) return hipErrorInvalidValue;
    auto *t = new ChirpTables();
    t->L = L;
    t->lds_points = L + L / 16;
    auto padpos = [](uint32_t i) { return i + i / 16; };
    // per-stage twiddle rows [m][R - 1] and the digit reversal, exactly as mixed_init lays them out for the dynamic plan
    std::vector<float2> tw;
    std::vector<uint32_t> ms(radix.size());
    uint32_t ns = L;
    for (size_t i = 0; i < radix.size(); ++i) {
        ms[i] = ns / radix[i];
        if (ms[i] > 1)
            for (uint32_t j = 0; j < ms[i]; ++j)
                for (uint32_t k = 1; k < radix[i]; ++k) {
                    const unsigned long long e = ((unsigned long long)j * k) % ns;
                    const double ang = -2.0 * M_PI * (double)e / (double)ns;
                    double cs = cos(ang), sn = sin(ang);
                    if (e == 0) { cs = 1.0; sn = 0.0; }
                    if (4 * e == ns) { cs = 0.0; sn = -1.0; }
                    if (2 * e == ns) { cs = -1.0; sn = 0.0; }
                    if (4 * e == 3ull * ns) { cs = 0.0; sn = 1.0; }
                    tw.push_back(make_float2((float)cs, (float)sn));
                }
        ns = ms[i];
    }
    std::vector<float2> chirp(P), bhat(t->lds_points, make_float2(0.0f, 0.0f));
    std::vector<double> cr(P), ci(P);
    for (uint32_t n = 0; n < P; ++n) {
        const unsigned long long q = ((unsigned long long)n * n) % (2ull * P);  // n^2 mod 2P: exact
        const double ang = -M_PI * (double)q / (double)P;
        cr[n] = cos(ang); ci[n] = sin(ang);
        chirp[n] = make_float2((float)cr[n], (float)ci[n]);
    }
    // b[m] = conj(c[|m|]) for m in [-(nz-1), P-1], wrapped modulo L
    std::vector<double> br(L, 0.0), bi(L, 0.0);
    for (uint32_t m = 0; m < P; ++m) { br[m] = cr[m]; bi[m] = -ci[m]; }
    for (uint32_t m = 1; m < nz; ++m) { br[L - m] = cr[m]; bi[L - m] = -ci[m]; }
    fft_host(br, bi);
    for (uint32_t K = 0; K < L; ++K) {   // bin K = k1 + r1 (k2 + r2 (...)) ends at k1 m1 + k2 m2 + ...
        uint32_t k = K, at = 0;
        for (size_t i = 0; i < radix.size(); ++i) {
            at += (k % radix[i]) * ms[i];
            k /= radix[i];
        }
        bhat[padpos(at)] = make_float2((float)(br[K] / (double)L), (float)(bi[K] / (double)L));
    }
    auto up = [](float2 **dst, const std::vector<float2> &v) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), v.size() * sizeof(float2));
        if (e == hipSuccess) e = hipMemcpy(*dst, v.data(), v.size() * sizeof(float2), hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(&t->d_chirp, chirp);
    if (e == hipSuccess) e = up(&t->d_bhat, bhat);
    if (e == hipSuccess) e = up(&t->d_tw, tw);
    if (e == hipSuccess && real) {
        std::vector<float2> twr(std::max<uint32_t>(P / 2, 1));
        for (uint32_t j = 0; j < P / 2; ++j) {
            const uint32_t k = j + 1;   // w_2P^k, k <= P / 2
            const double ang = -M_PI * (double)k / (double)P;
            twr[j] = 2 * k == P ? make_float2(0.0f, -1.0f) : make_float2((float)cos(ang), (float)sin(ang));
        }
        e = up(&t->d_twr, twr);
    }
    if (e != hipSuccess) {
        chirpz_destroy(t);
        return e;
    }
    *out = t;
    return hipSuccess;
}

hipError_t chirpz_init(sgx_ctx *c, void **out)
{
    mix::ChirpTables *t = nullptr;
    hipError_t e = build_chirp(c->P, c->W, false, &t);
    if (e != hipSuccess) return e;
    if (c->C == 1) {   // a mono stream: real-input mode
        e = build_chirp(c->W, (c->W + 1) / 2, true, &t->half);
        if (e != hipSuccess && e != hipErrorInvalidValue) {
            chirpz_destroy(t);
            return e;
        }
    }
    *out = t;
    return hipSuccess;
}

bool chirpz_real_serves(const sgx_ctx *c, const void *tables, uint32_t channels)
{
    const auto *t = static_cast<const mix::ChirpTables *>(tables);
    return t && t->half && channels == 1 && !(c->cfg.flags & SGX_FLAG_PAIRED_FRAMES) && !(c->cfg.flags & SGX_FLAG_COMPLEX_MONO);
}

void chirpz_destroy(void *tables)
{
    auto *t = static_cast<mix::ChirpTables *>(tables);
    if (!t) return;
    if (t->half) chirpz_destroy(t->half);
    if (t->d_twr) (void)hipFree(t->d_twr);
    if (t->d_chirp) (void)hipFree(t->d_chirp);
    if (t->d_bhat) (void)hipFree(t->d_bhat);
    if (t->d_tw) (void)hipFree(t->d_tw);
    delete t;
}

hipError_t launch_stft_chirpz(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                              size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags)
{
    using namespace mix;
    if (n_frames == 0) return hipSuccess;
    const auto *t = static_cast<const ChirpTables *>(tables);
    const bool real = chirpz_real_serves(c, tables, channels);   // a mono stream, every frame its own transform (the default)
    if (real) t = t->half;
    Params p{};
    p.pcm = d_pcm;
    p.window = c->d_window;
    p.tw = t->d_tw;
    p.chirp = t->d_chirp;
    p.bhat = t->d_bhat;
    p.twr = t->d_twr;
    p.real = real ? 1u : 0u;
    p.W = c->W;
    p.P = real ? c->W : c->P;
    p.H = c->H;
    p.C = channels;
    p.pairs = pairs;
    p.scale = 2.0f / (float)c->W;
    p.first_frame = first_frame;
    p.n_frames = n_frames;
    p.total_frames = total_frames;
    p.vec2 = (channels >= 2 && channels % 2 == 0 && reinterpret_cast<uintptr_t>(d_pcm) % 8 == 0) ? 1u : 0u;
    if (real) p.vec2 = c->W % 2 == 0 ? 1u : 0u;   // sample PAIRS of one channel as 8-byte words
    const size_t lds = (size_t)t->lds_points * sizeof(float2);
    hipError_t attr_err = hipSuccess;
    auto go = [&](auto kernel, unsigned nt, dim3 grid) {
        if (lds > 64 * 1024) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { attr_err = e; return; }
        }
        hipLaunchKernelGGL(kernel, grid, dim3(nt), lds, c->stream, p);
    };
    auto launch = [&](dim3 grid) {
        switch (t->L) {
#define X(Ln, A0, B0, A1, B1, A2, B2, N)                                                                             \
    case Ln:                                                                                                         \
        if (real) go(chirpz3_kernel<Fixed3<Ln, A0, B0, A1, B1, A2, B2, N>, A0, B0, A1, B1, A2, B2, true>, N, grid);   \
        else go(chirpz3_kernel<Fixed3<Ln, A0, B0, A1, B1, A2, B2, N>, A0, B0, A1, B1, A2, B2, false>, N, grid);      \
        break;
            CHIRP_PLANS3(X)
#undef X
#define X(Ln, A0, B0, A1, B1, A2, B2, A3, B3, N)                                                                                      \
    case Ln:                                                                                                                          \
        if (real) go(chirpz4_kernel<Fixed4<Ln, A0, B0, A1, B1, A2, B2, A3, B3, N>, A0, B0, A1, B1, A2, B2, A3, B3, true>, N, grid);    \
        else go(chirpz4_kernel<Fixed4<Ln, A0, B0, A1, B1, A2, B2, A3, B3, N>, A0, B0, A1, B1, A2, B2, A3, B3, false>, N, grid);       \
        break;
            CHIRP_PLANS4(X)
#undef X
        default: attr_err = hipErrorInvalidValue; break;
        }
    };
    const size_t max_chunk = 1u << 30;
    if (channels == 1 && (c->cfg.flags & SGX_FLAG_PAIRED_FRAMES)) {
        p.mono_pairs = 1;
        p.mags = d_mags;
        const unsigned long long q0 = first_frame / 2, q1 = (first_frame + n_frames + 1) / 2;
        for (unsigned long long q = q0; q < q1; q += max_chunk) {
            const unsigned long long chunk = q1 - q < max_chunk ? q1 - q : max_chunk;
            p.pair_base = q;
            launch(dim3((unsigned)chunk, 1));
            const hipError_t e = attr_err != hipSuccess ? attr_err : hipGetLastError();
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    for (size_t done = 0; done < n_frames; done += max_chunk) {
        const size_t chunk = n_frames - done < max_chunk ? n_frames - done : max_chunk;
        p.first_frame = first_frame + done;
        p.n_frames = chunk;
        p.mags = d_mags + done * (size_t)pairs * c->M * 2;
        launch(dim3((unsigned)chunk, pairs));
        const hipError_t e = attr_err != hipSuccess ? attr_err : hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace sgx
