// stft_mixed.hip -- STFT for windows whose padded length P = 2W is not a power of two but has only the prime
// factors 2, 3, 5 and 7: a mixed-radix FFT of exactly P points, in place in LDS.
//
// The reference sizes its window from a duration: FastFourierTransform::new(sample_rate, 0.05)
// (gpu_spectrogram.rs:323, simple_spectrogram.rs:217) gives W = 2400 at 48 kHz (P = 4800 = 2^6 3 5^2) and
// W = 2205 at 44.1 kHz (P = 4410 = 2 3^2 5 7^2); FFTW takes any length (fft.rs:20-24).  The chirp-z kernel
// (stft_bluestein.hip) serves every length with two power-of-two transforms of >= 3W points; for the smooth
// lengths the application actually produces, this kernel does a quarter of that arithmetic.
//
// Decimation in frequency, one stage per factor r of the current block length Ns (m = Ns / r):
//   y_k = (sum_q x[q m + j] w_r^{q k}) * w_Ns^{j k},  stored at k m + j      (k < r, j < m)
// after which sub-block k is the length-m problem of the bins = k (mod r).  Bin K ends at pos[K] (mixed-radix
// digit reversal, a host table); the split reads F[k] and F[P - k] through it.  Twiddles come from the context's
// full-circle table e^{-2 pi i j / P}: the index j k P / Ns never leaves [0, P).
#include "sgx_internal.hpp"

namespace sgx {

namespace mix {

constexpr int kMaxStages = 16;

struct MixTables {
    uint32_t *d_pos = nullptr;  // [P] position of bin k after the stages
    uint32_t n_stages = 0;
    uint32_t radix[kMaxStages] = {}, m[kMaxStages] = {}, tws[kMaxStages] = {};
    float inv_m[kMaxStages] = {};
};

struct Params {
    const float *pcm;
    const float *window;
    const float2 *tw;       // [P] e^{-2 pi i j / P}
    const uint32_t *pos;    // [P]
    float *mags;
    unsigned long long first_frame, pair_base, n_frames, total_frames;
    uint32_t mono_pairs, W, P, H, C, pairs, n_stages;
    float scale;
    uint32_t radix[kMaxStages], m[kMaxStages], tws[kMaxStages];
    float inv_m[kMaxStages];
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// a - i b and a + i b
__device__ __forceinline__ float2 sub_i(float2 a, float2 b) { return make_float2(a.x + b.y, a.y - b.x); }
__device__ __forceinline__ float2 add_i(float2 a, float2 b) { return make_float2(a.x - b.y, a.y + b.x); }
__device__ __forceinline__ float2 scale2(float2 a, float c) { return make_float2(a.x * c, a.y * c); }

// forward r-point DFTs (kernel e^{-2 pi i q k / r}), in place on x[0 .. r)
__device__ __forceinline__ void dft2(float2 *x)
{
    const float2 a = x[0], b = x[1];
    x[0] = cadd(a, b);
    x[1] = csub(a, b);
}
__device__ __forceinline__ void dft3(float2 *x)
{
    const float s = 0.86602540378443864676f;  // sin(2 pi / 3)
    const float2 t = cadd(x[1], x[2]), d = scale2(csub(x[1], x[2]), s);
    const float2 a = make_float2(fmaf(t.x, -0.5f, x[0].x), fmaf(t.y, -0.5f, x[0].y));
    x[0] = cadd(x[0], t);
    x[1] = sub_i(a, d);
    x[2] = add_i(a, d);
}
__device__ __forceinline__ void dft4(float2 *x)
{
    const float2 b0 = cadd(x[0], x[2]), b1 = csub(x[0], x[2]), b2 = cadd(x[1], x[3]), d = csub(x[1], x[3]);
    x[0] = cadd(b0, b2);
    x[2] = csub(b0, b2);
    x[1] = sub_i(b1, d);
    x[3] = add_i(b1, d);
}
__device__ __forceinline__ void dft5(float2 *x)
{
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;  // cos(2 pi / 5), cos(4 pi / 5)
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;   // sin(2 pi / 5), sin(4 pi / 5)
    const float2 t1 = cadd(x[1], x[4]), t2 = cadd(x[2], x[3]), t3 = csub(x[1], x[4]), t4 = csub(x[2], x[3]);
    const float2 a1 = make_float2(fmaf(t2.x, c2, fmaf(t1.x, c1, x[0].x)), fmaf(t2.y, c2, fmaf(t1.y, c1, x[0].y)));
    const float2 a2 = make_float2(fmaf(t2.x, c1, fmaf(t1.x, c2, x[0].x)), fmaf(t2.y, c1, fmaf(t1.y, c2, x[0].y)));
    const float2 b1 = make_float2(fmaf(t4.x, s2, t3.x * s1), fmaf(t4.y, s2, t3.y * s1));
    const float2 b2 = make_float2(fmaf(t4.x, -s1, t3.x * s2), fmaf(t4.y, -s1, t3.y * s2));
    x[0] = cadd(x[0], cadd(t1, t2));
    x[1] = sub_i(a1, b1);
    x[4] = add_i(a1, b1);
    x[2] = sub_i(a2, b2);
    x[3] = add_i(a2, b2);
}
__device__ __forceinline__ void dft7(float2 *x)
{
    const float c1 = 0.62348980185873353053f, c2 = -0.22252093395631440429f, c3 = -0.90096886790241912624f;  // cos(2 pi k / 7)
    const float s1 = 0.78183148246802980871f, s2 = 0.97492791218182360702f, s3 = 0.43388373911755812048f;    // sin(2 pi k / 7)
    const float2 t1 = cadd(x[1], x[6]), t2 = cadd(x[2], x[5]), t3 = cadd(x[3], x[4]);
    const float2 u1 = csub(x[1], x[6]), u2 = csub(x[2], x[5]), u3 = csub(x[3], x[4]);
    // a_k = x0 + sum_q cos(2 pi q k / 7) t_q ; b_k = sum_q sin(2 pi q k / 7) u_q   (q k mod 7 folded to 1..3 with sign)
    const float2 a1 = make_float2(fmaf(t3.x, c3, fmaf(t2.x, c2, fmaf(t1.x, c1, x[0].x))), fmaf(t3.y, c3, fmaf(t2.y, c2, fmaf(t1.y, c1, x[0].y))));
    const float2 a2 = make_float2(fmaf(t3.x, c1, fmaf(t2.x, c3, fmaf(t1.x, c2, x[0].x))), fmaf(t3.y, c1, fmaf(t2.y, c3, fmaf(t1.y, c2, x[0].y))));
    const float2 a3 = make_float2(fmaf(t3.x, c2, fmaf(t2.x, c1, fmaf(t1.x, c3, x[0].x))), fmaf(t3.y, c2, fmaf(t2.y, c1, fmaf(t1.y, c3, x[0].y))));
    const float2 b1 = make_float2(fmaf(u3.x, s3, fmaf(u2.x, s2, u1.x * s1)), fmaf(u3.y, s3, fmaf(u2.y, s2, u1.y * s1)));
    const float2 b2 = make_float2(fmaf(u3.x, -s1, fmaf(u2.x, -s3, u1.x * s2)), fmaf(u3.y, -s1, fmaf(u2.y, -s3, u1.y * s2)));
    const float2 b3 = make_float2(fmaf(u3.x, s2, fmaf(u2.x, -s1, u1.x * s3)), fmaf(u3.y, s2, fmaf(u2.y, -s1, u1.y * s3)));
    x[0] = cadd(x[0], cadd(t1, cadd(t2, t3)));
    x[1] = sub_i(a1, b1);
    x[6] = add_i(a1, b1);
    x[2] = sub_i(a2, b2);
    x[5] = add_i(a2, b2);
    x[3] = sub_i(a3, b3);
    x[4] = add_i(a3, b3);
}

template <int R>
__device__ __forceinline__ void stage(float2 *s, const Params &p, int st, uint32_t tid, uint32_t nt)
{
    const uint32_t m = p.m[st], tws = p.tws[st], count = p.P / R;
    const float inv_m = p.inv_m[st];
    for (uint32_t b = tid; b < count; b += nt) {
        const uint32_t blk = (uint32_t)(((float)b + 0.5f) * inv_m);  // b / m: exact for every supported length (tests/test_host_logic.py)
        const uint32_t j = b - blk * m;
        float2 *base = s + blk * m * R + j;
        float2 x[R];
#pragma unroll
        for (int q = 0; q < R; ++q) x[q] = base[q * m];
        if (R == 2) dft2(x);
        else if (R == 3) dft3(x);
        else if (R == 4) dft4(x);
        else if (R == 5) dft5(x);
        else dft7(x);
        base[0] = x[0];
#pragma unroll
        for (int k = 1; k < R; ++k) base[k * m] = j == 0 ? x[k] : cmul(x[k], p.tw[j * k * tws]);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(1024) stft_mixed_kernel(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const uint32_t W = p.W, P = p.P, M = W - 1;
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

    // (l + i r) * hann (fft.rs:53-63); zeros up to P (fft.rs:65-69)
    for (uint32_t n = tid; n < P; n += nt) {
        float2 v = make_float2(0.0f, 0.0f);
        if (n < W) {
            const float w = p.window[n];
            const float l = src_a[(size_t)n * p.C + cl];
            const float r = data_b ? src_b[(size_t)n * p.C + cr] : 0.0f;
            v = make_float2(l * w, r * w);
        }
        s[n] = v;
    }
    __syncthreads();

    for (uint32_t st = 0; st < p.n_stages; ++st) {
        switch (p.radix[st]) {  // uniform
        case 2: stage<2>(s, p, st, tid, nt); break;
        case 3: stage<3>(s, p, st, tid, nt); break;
        case 4: stage<4>(s, p, st, tid, nt); break;
        case 5: stage<5>(s, p, st, tid, nt); break;
        default: stage<7>(s, p, st, tid, nt); break;
        }
    }

    // split + magnitude + scale (fft.rs:81-98); k = 1 .. W-1 kept
    const bool st_a = row_a >= 0 && (unsigned long long)row_a < p.n_frames;
    const bool st_b = p.mono_pairs && row_b >= 0 && (unsigned long long)row_b < p.n_frames;
    float2 *out_a = reinterpret_cast<float2 *>(p.mags) + ((size_t)(st_a ? row_a : 0) * p.pairs + pair) * M;
    float2 *out_b = reinterpret_cast<float2 *>(p.mags) + ((size_t)(st_b ? row_b : 0) * p.pairs + pair) * M;
    for (uint32_t j = tid; j < M; j += nt) {
        const uint32_t k = j + 1;
        const float2 a = s[p.pos[k]], b = s[p.pos[P - k]];
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

hipError_t mixed_init(sgx_ctx *c, void **out)
{
    using namespace mix;
    auto *t = new MixTables();
    const uint32_t P = c->P;
    // stage order: the odd factors first (long strides), then radix 4, then a last radix 2
    std::vector<uint32_t> radices;
    uint32_t n = P;
    for (uint32_t f : {7u, 5u, 3u})
        while (n % f == 0) { radices.push_back(f); n /= f; }
    while (n % 4 == 0) { radices.push_back(4); n /= 4; }
    if (n % 2 == 0) { radices.push_back(2); n /= 2; }
    if (n != 1 || radices.size() > (size_t)kMaxStages) { delete t; return hipErrorInvalidValue; }
    t->n_stages = (uint32_t)radices.size();
    uint32_t ns = P;
    for (uint32_t i = 0; i < t->n_stages; ++i) {
        t->radix[i] = radices[i];
        t->m[i] = ns / radices[i];
        t->tws[i] = P / ns;
        t->inv_m[i] = 1.0f / (float)t->m[i];
        ns = t->m[i];
    }
    // bin K = k1 + r1 (k2 + r2 (k3 + ...)) ends at k1 m1 + k2 m2 + ...
    std::vector<uint32_t> pos(P);
    for (uint32_t K = 0; K < P; ++K) {
        uint32_t k = K, at = 0;
        for (uint32_t i = 0; i < t->n_stages; ++i) {
            at += (k % t->radix[i]) * t->m[i];
            k /= t->radix[i];
        }
        pos[K] = at;
    }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&t->d_pos), (size_t)P * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(t->d_pos, pos.data(), (size_t)P * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        mixed_destroy(t);
        return e;
    }
    *out = t;
    return hipSuccess;
}

void mixed_destroy(void *tables)
{
    auto *t = static_cast<mix::MixTables *>(tables);
    if (!t) return;
    if (t->d_pos) (void)hipFree(t->d_pos);
    delete t;
}

hipError_t launch_stft_mixed(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                             size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags)
{
    using namespace mix;
    if (n_frames == 0) return hipSuccess;
    const auto *t = static_cast<const MixTables *>(tables);
    Params p{};
    p.pcm = d_pcm;
    p.window = c->d_window;
    p.tw = c->d_twiddle;
    p.pos = t->d_pos;
    p.W = c->W;
    p.P = c->P;
    p.H = c->H;
    p.C = channels;
    p.pairs = pairs;
    p.scale = 2.0f / (float)c->W;
    p.n_stages = t->n_stages;
    for (uint32_t i = 0; i < t->n_stages; ++i) {
        p.radix[i] = t->radix[i];
        p.m[i] = t->m[i];
        p.tws[i] = t->tws[i];
        p.inv_m[i] = t->inv_m[i];
    }
    p.first_frame = first_frame;
    p.n_frames = n_frames;
    p.total_frames = total_frames;
    const size_t lds = (size_t)c->P * sizeof(float2);
    if (lds > 64 * 1024) {  // per launch: the attribute is per device, and a process may hold contexts on several
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft_mixed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
    }
    // about one radix-4 butterfly per thread and stage, and as many resident workgroups as the LDS image allows
    // (2048 threads per CU): independent workgroups fill each other's barrier waits
    unsigned threads = ((c->P / 4 + 63) / 64) * 64;
    const unsigned resident = (unsigned)((160 * 1024) / (lds ? lds : 1));
    if (resident >= 2) {
        const unsigned cap = (2048u / (resident > 8 ? 8 : resident)) / 64 * 64;
        if (threads > cap) threads = cap;
    }
    threads = threads > 1024u ? 1024u : (threads < 64u ? 64u : threads);
    const size_t max_chunk = 1u << 30;
    if (channels == 1 && !(c->cfg.flags & SGX_FLAG_INDEPENDENT_FRAMES)) {
        p.mono_pairs = 1;
        p.mags = d_mags;
        const unsigned long long q0 = first_frame / 2, q1 = (first_frame + n_frames + 1) / 2;
        for (unsigned long long q = q0; q < q1; q += max_chunk) {
            const unsigned long long chunk = q1 - q < max_chunk ? q1 - q : max_chunk;
            p.pair_base = q;
            hipLaunchKernelGGL(stft_mixed_kernel, dim3((unsigned)chunk, 1), dim3(threads), lds, c->stream, p);
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
        hipLaunchKernelGGL(stft_mixed_kernel, dim3((unsigned)chunk, pairs), dim3(threads), lds, c->stream, p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace sgx
