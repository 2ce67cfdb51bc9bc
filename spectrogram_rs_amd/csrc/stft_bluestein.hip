// stft_bluestein.hip -- STFT for windows whose padded length P = 2W is NOT a power of two.
//
// The reference sizes its window from a duration: FastFourierTransform::new(sample_rate, 0.05)
// (gpu_spectrogram.rs:323, simple_spectrogram.rs:217) gives W = 2400 at 48 kHz (P = 4800 = 2^6 3 5^2)
// and W = 2205 at 44.1 kHz (P = 4410 = 2 3^2 5 7^2); FFTW handles any length (fft.rs:20-24).  Here
// the length-P DFT of the zero-padded frame is evaluated as a chirp-z (Bluestein) convolution with
// power-of-two FFTs that live entirely in LDS:
//
//   F[k] = c[k] * sum_{n<W} (z[n] c[n]) conj(c)[k - n],      c[n] = exp(-i pi n^2 / P)
//
// Only n < W inputs are non-zero (the padding), and only k < P outputs are needed, so the circular
// convolution length is L = pow2 >= P + W - 1 (8192 for both windows above: 64 KB of LDS).
//   1. a[n] = z[n] c[n] -> LDS, zero to L           2. forward DIF FFT_L, radix 4 (digit-reversed result)
//   3. multiply by B^ = FFT_L(conj chirp) / L, precomputed in float64, stored in the same digit-reversed order
//   4. inverse DIT FFT_L from digit-reversed input (natural result)      5. F[k] = c[k] y[k], split.
// Chirp angles use n^2 mod 2P in integers, so they are exact before the single float rounding.
#include "lds_fft.hpp"
#include "sgx_internal.hpp"

namespace sgx {

namespace blu {

struct BluTables {
    float2 *d_chirp = nullptr;  // [P]    c[n]
    float2 *d_bhat = nullptr;   // [L]    FFT_L(b) / L at ldsfft::pos_of(k)
    float2 *d_tw = nullptr;     // [L]    e^{-2 pi i j / L}
    uint32_t L = 0, logL = 0;
};

struct Params {
    const float *pcm;
    const float *window;
    const float2 *chirp, *bhat, *tw;
    float *mags;
    unsigned long long first_frame;
    // mono pairs (frames 2q and 2q+1 ride in the real and imaginary part of one transform, as in stft4096_wg.hip):
    unsigned long long pair_base, n_frames, total_frames;
    uint32_t mono_pairs;
    uint32_t W, P, L, logL, H, C, pairs;
    float scale;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b)  // a * conj(b)
{
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -(a.x * b.y)));
}

__global__ void __launch_bounds__(1024) stft_bluestein_kernel(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const uint32_t W = p.W, P = p.P, L = p.L, M = W - 1;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint32_t pair = blockIdx.y;
    // (l, r) of one frame -- or, for a mono stream, frames 2q and 2q+1 by GLOBAL index (see sgx_kernels.hip)
    long long row_a, row_b = -1;
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

    // 1. (l + i r) * hann (fft.rs:53-63), times the chirp; zeros up to L
    for (uint32_t n = tid; n < L; n += nt) {
        float2 v = make_float2(0.0f, 0.0f);
        if (n < W) {
            const float w = p.window[n];
            const float l = src_a[(size_t)n * p.C + cl];
            const float r = data_b ? src_b[(size_t)n * p.C + cr] : 0.0f;
            v = cmul(make_float2(l * w, r * w), p.chirp[n]);
        }
        s[n] = v;
    }
    __syncthreads();

    // 2. forward FFT_L in place (radix 4): natural in, digit-reversed out
    ldsfft::forward_dif(s, p.logL, p.tw, tid, nt);

    // 3. spectrum of the convolution (B^ carries the 1/L of the inverse transform and is stored in the same order)
    for (uint32_t i = tid; i < L; i += nt) s[i] = cmul(s[i], p.bhat[i]);
    __syncthreads();

    // 4. inverse FFT_L in place with conjugate twiddles: digit-reversed in, natural out
    ldsfft::inverse_dit(s, p.logL, p.tw, tid, nt);

    // 5. F[k] = c[k] y[k]; split + magnitude + scale (fft.rs:81-98); k = 1 .. W-1 kept
    const bool st_a = row_a >= 0 && (unsigned long long)row_a < p.n_frames;
    const bool st_b = p.mono_pairs && row_b >= 0 && (unsigned long long)row_b < p.n_frames;
    float2 *out_a = reinterpret_cast<float2 *>(p.mags) + ((size_t)(st_a ? row_a : 0) * p.pairs + pair) * M;
    float2 *out_b = reinterpret_cast<float2 *>(p.mags) + ((size_t)(st_b ? row_b : 0) * p.pairs + pair) * M;
    for (uint32_t j = tid; j < M; j += nt) {
        const uint32_t k = j + 1;
        const float2 a = cmul(s[k], p.chirp[k]);
        const float2 b = cmul(s[P - k], p.chirp[P - k]);
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

}  // namespace blu

bool bluestein_supported(uint32_t W)
{
    // L = pow2 >= P + W - 1 = 3W - 1 complex points must fit the 160 KB LDS: L <= 16384
    return W >= 4 && 3ull * W - 1 <= 16384;
}

hipError_t bluestein_init(sgx_ctx *c, void **out)
{
    using namespace blu;
    auto *t = new BluTables();
    const uint32_t W = c->W, P = c->P;
    uint32_t L = 1, logL = 0;
    while (L < P + W - 1) { L <<= 1; ++logL; }
    t->L = L;
    t->logL = logL;
    std::vector<float2> chirp(P), bhat(L), tw(L);
    std::vector<double> cr(P), ci(P);
    for (uint32_t n = 0; n < P; ++n) {
        const unsigned long long q = ((unsigned long long)n * n) % (2ull * P);  // n^2 mod 2P: exact
        const double ang = -M_PI * (double)q / (double)P;
        cr[n] = cos(ang); ci[n] = sin(ang);
        chirp[n] = make_float2((float)cr[n], (float)ci[n]);
    }
    // b[m] = conj(c[|m|]) for m in [-(W-1), P-1], wrapped modulo L
    std::vector<double> br(L, 0.0), bi(L, 0.0);
    for (uint32_t m = 0; m < P; ++m) { br[m] = cr[m]; bi[m] = -ci[m]; }
    for (uint32_t m = 1; m < W; ++m) { br[L - m] = cr[m]; bi[L - m] = -ci[m]; }
    fft_host(br, bi);
    for (uint32_t i = 0; i < L; ++i)
        bhat[ldsfft::pos_of(i, logL)] = make_float2((float)(br[i] / (double)L), (float)(bi[i] / (double)L));
    for (uint32_t j = 0; j < L; ++j) {  // the full circle: a radix-4 stage uses w^j, w^2j and w^3j
        const double ang = -2.0 * M_PI * (double)j / (double)L;
        double cs = cos(ang), sn = sin(ang);
        if (4 * j == L) { cs = 0.0; sn = -1.0; }
        if (2 * j == L) { cs = -1.0; sn = 0.0; }
        if (4 * j == 3 * L) { cs = 0.0; sn = 1.0; }
        tw[j] = make_float2((float)cs, (float)sn);
    }
    auto up = [](float2 **dst, const std::vector<float2> &v) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), v.size() * sizeof(float2));
        if (e == hipSuccess) e = hipMemcpy(*dst, v.data(), v.size() * sizeof(float2), hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(&t->d_chirp, chirp);
    if (e == hipSuccess) e = up(&t->d_bhat, bhat);
    if (e == hipSuccess) e = up(&t->d_tw, tw);
    if (e == hipSuccess && (size_t)L * sizeof(float2) > 64 * 1024)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft_bluestein_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)L * sizeof(float2)));
    if (e != hipSuccess) {
        bluestein_destroy(t);
        return e;
    }
    *out = t;
    return hipSuccess;
}

void bluestein_destroy(void *tables)
{
    auto *t = static_cast<blu::BluTables *>(tables);
    if (!t) return;
    if (t->d_chirp) (void)hipFree(t->d_chirp);
    if (t->d_bhat) (void)hipFree(t->d_bhat);
    if (t->d_tw) (void)hipFree(t->d_tw);
    delete t;
}

hipError_t launch_stft_bluestein(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                                 size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags)
{
    using namespace blu;
    if (n_frames == 0) return hipSuccess;
    const auto *t = static_cast<const BluTables *>(tables);
    Params p{};
    p.pcm = d_pcm;
    p.window = c->d_window;
    p.chirp = t->d_chirp;
    p.bhat = t->d_bhat;
    p.tw = t->d_tw;
    p.W = c->W;
    p.P = c->P;
    p.L = t->L;
    p.logL = t->logL;
    p.H = c->H;
    p.C = channels;
    p.pairs = pairs;
    p.scale = 2.0f / (float)c->W;
    const size_t lds = (size_t)t->L * sizeof(float2);
    const size_t max_chunk = 1u << 30;
    // one radix-4 butterfly per thread and stage where the transform is long enough (short windows: fewer idle lanes)
    unsigned threads = t->L / 4;
    threads = threads > 1024u ? 1024u : (threads < 64u ? 64u : threads);
    p.first_frame = first_frame;
    p.n_frames = n_frames;
    p.total_frames = total_frames;
    if (channels == 1 && (c->cfg.flags & SGX_FLAG_PAIRED_FRAMES)) {
        p.mono_pairs = 1;
        p.mags = d_mags;
        const unsigned long long q0 = first_frame / 2, q1 = (first_frame + n_frames + 1) / 2;
        for (unsigned long long q = q0; q < q1; q += max_chunk) {
            const unsigned long long chunk = q1 - q < max_chunk ? q1 - q : max_chunk;
            p.pair_base = q;
            hipLaunchKernelGGL(stft_bluestein_kernel, dim3((unsigned)chunk, 1), dim3(threads), lds, c->stream, p);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    for (size_t done = 0; done < n_frames; done += max_chunk) {
        const size_t chunk = n_frames - done < max_chunk ? n_frames - done : max_chunk;
        p.first_frame = first_frame + done;
        p.n_frames = chunk;
        p.mags = d_mags + done * (size_t)pairs * c->M * 2;
        hipLaunchKernelGGL(stft_bluestein_kernel, dim3((unsigned)chunk, pairs), dim3(threads), lds, c->stream, p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace sgx
