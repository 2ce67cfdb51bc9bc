// stft16384_wg.hip -- tuned STFT for W = 8192 (P = 16384): one 1024-thread workgroup per transform,
// 16 points per thread, radix 16 x 16 x 16 x 4 (BASELINE config 4: 16384-point, hop 512, 8 interleaved
// channels = 4 (l, r) pairs per hop position; the "LDS-pressure" case: the transform alone is 128 KB).
//
// Replaces FastFourierTransform::process (fft.rs:43-99) + the hop loop (audio_transform.rs:34-42).
// Same design rules as stft4096_wg.hip: 4 waves per SIMD (<= 128 VGPRs), persistent workgroups (one
// per CU: the LDS image is 141 KB), padding never materialised, next transform's samples requested
// before this transform's stores, LDS-only barriers.
//
//   n = t + 1024 a           (t = thread, a < 8 non-zero rows)
//   pass 1  thread t          : 16-point DFT over a (two 8-point FFTs) -> q1; twiddle w_16384^{t q1} (VGPRs)
//   pass 2  wave q1, lane t0  : t = t0 + 64 t1; FFT16 over t1 -> q2; twiddle w_1024^{t0 q2}   (LDS table)
//   pass 3  wave q1, lane (q2, u0): t0 = u0 + 4 u1; FFT16 over u1 -> q3; twiddle w_64^{u0 q3} (LDS table)
//           passes 2 -> 3 exchange inside ONE wave's 1088-element region: no workgroup barrier
//   pass 4  thread (q1, q2, g): for q3 = 4g + j: 4-point DFT over u0 -> q4
//           bin k = q1 + 16 q2 + 256 q3 + 4096 q4
//   split   partner of (klow, q4) is (4096 - klow, 3 - q4): the upper half (q4 = 2, 3) is published
//           through LDS as D[q4 - 2][klow]; D[0][4096] aliases D[1][0], which is exactly the partner
//           of klow = 0 -- no special case (k = 0, DC, is not an output)
#include "sgx_internal.hpp"

namespace sgx {

namespace wg16k {

typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float cl_fma(float a, float c, float u) { return fmaf(a, c, u); }
__device__ __forceinline__ f2v cl_fma(f2v a, float c, f2v u) { return __builtin_elementwise_fma(a, f2v{c, c}, u); }

#include "fft_codelets.inc"

constexpr int kW = 8192, kP = 16384, kM = 8191;
constexpr int kSA = 1088;              // per-q1 region: 16 rows of 64 (+4 pad) complex, also holds the flat [1024] image
constexpr int kSB = 68;                // row stride of the wave-local image [q2][t0]
constexpr int kSU = 276, kSQ = 17;     // image C: [(q3 * 4 + u0) * 276 + q2 * 17 + q1]
constexpr int kBufComplex = 64 * kSU;  // 17 664 complex = 141 312 B (>= 16 * 1088 and >= 2 * 4096 + 1)
constexpr size_t kLdsBytes = (size_t)(kBufComplex + 16 * 64 + 16 * 4) * sizeof(float2);

struct Params {
    const float *pcm;
    const float2 *tw1;   // [16][1024]  w_16384^{t q1}
    const float2 *tw2;   // [16][64]    w_1024^{t0 q2}
    const float2 *tw3;   // [16][4]     w_64^{u0 q3}
    const float *window; // [8192]
    float *mags;
    unsigned long long first_frame, n_frames, total_frames, pair_base, n_jobs;
    uint32_t H, C, pairs;
};

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void lds_wave_fence()
{
    // LDS operations of one wave execute in order; this only stops the compiler from moving them
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ float2 cmulf(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}

template <bool MONO>
__global__ void __launch_bounds__(1024) stft16384_wg_kernel(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *buf = reinterpret_cast<float2 *>(smem_raw);
    float2 *tw2 = buf + kBufComplex;   // [16][64]
    float2 *tw3 = tw2 + 16 * 64;       // [16][4]

    const int tid = threadIdx.x;
    tw2[tid] = p.tw2[tid];
    if (tid < 64) tw3[tid] = p.tw3[tid];

    float win[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) win[a] = p.window[tid + 1024 * a];
    float2 tw1[16];
#pragma unroll
    for (int q = 1; q < 16; ++q) tw1[q] = p.tw1[q * 1024 + tid];

    const int wave = tid >> 6, lane = tid & 63;       // passes 2 and 3: q1 = wave
    const int q2_3 = lane >> 2, u0_3 = lane & 3;       // pass-3 role inside the wave
    const int q1_4 = tid & 15, q2_4 = (tid >> 4) & 15, g_4 = tid >> 8;  // pass-4 role
    const int klow_base = q1_4 + 16 * q2_4 + 1024 * g_4;                // klow for j = 0 (q3 = 4g + j -> + 256 j)
    const float inv_w = 1.0f / (float)kW;
    __syncthreads();

    // job -> (frame or frame pair, channel pair); consecutive jobs of a workgroup are the channel
    // pairs of one hop position, so the interleaved PCM lines are reused from L1/L2
    float sa[8], sb[8];
    auto fetch = [&](unsigned long long job) {
        if (MONO) {
            const unsigned long long f = 2 * (p.pair_base + job);
            const bool second = f + 1 < p.total_frames;
            const float *s0 = p.pcm + f * p.H;
            const float *s1 = second ? s0 + p.H : s0;
#pragma unroll
            for (int a = 0; a < 8; ++a) { sa[a] = s0[tid + 1024 * a]; sb[a] = s1[tid + 1024 * a]; }
        } else {
            const unsigned long long hop = job / p.pairs;
            const uint32_t pair = (uint32_t)(job - hop * p.pairs);
            const float *s0 = p.pcm + (p.first_frame + hop) * p.H * p.C + 2 * pair;
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const float2 v = *reinterpret_cast<const float2 *>(s0 + (size_t)(tid + 1024 * a) * p.C);
                sa[a] = v.x; sb[a] = v.y;
            }
        }
    };

    unsigned long long job = blockIdx.x;
    if (job < p.n_jobs) fetch(job);
    for (; job < p.n_jobs; job += gridDim.x) {
        // ---- Hann (fft.rs:53-63)
        long long f0, f1;
        bool have_first = true, have_second = true, data_second = true;
        uint32_t pair = 0;
        if (MONO) {
            f0 = (long long)(2 * (p.pair_base + job)) - (long long)p.first_frame;
            f1 = f0 + 1;
            have_first = f0 >= 0;
            have_second = f1 < (long long)p.n_frames;
            data_second = (unsigned long long)(f1 + (long long)p.first_frame) < p.total_frames;
        } else {
            f0 = (long long)(job / p.pairs);
            f1 = f0;
            pair = (uint32_t)(job - (unsigned long long)f0 * p.pairs);
        }
        float er[8], ei[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            er[a] = sa[a] * win[a];
            ei[a] = data_second ? sb[a] * win[a] : 0.0f;
        }

        // ---- pass 1: 16-point DFT over a (a >= 8 is the zero padding): even q1 = FFT8(z), odd q1 = FFT8(z w_16^a)
        float orr[8], oi[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) { orr[a] = er[a]; oi[a] = ei[a]; }
        pretwiddle8_w16(orr, oi);
        fft8(er, ei);
        fft8(orr, oi);
        lds_barrier();  // the previous transform's partner reads are complete
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pos = FFT8_OUT[j];
            const float2 ve = make_float2(er[pos], ei[pos]);
            const float2 vo = make_float2(orr[pos], oi[pos]);
            buf[(2 * j) * kSA + tid] = j == 0 ? ve : cmulf(ve, tw1[2 * j]);
            buf[(2 * j + 1) * kSA + tid] = cmulf(vo, tw1[2 * j + 1]);
        }
        lds_barrier();

        // ---- pass 2 (wave q1, lane t0): FFT16 over t1, twiddle w_1024^{t0 q2}
        float2 *reg = buf + wave * kSA;
        float xr[16], xi[16];
#pragma unroll
        for (int t1 = 0; t1 < 16; ++t1) {
            const float2 v = reg[lane + 64 * t1];
            xr[t1] = v.x; xi[t1] = v.y;
        }
        fft16(xr, xi);
        lds_wave_fence();  // this wave's region only: reads above are complete, in-order LDS does the rest
#pragma unroll
        for (int q2 = 0; q2 < 16; ++q2) {
            const int pos = FFT16_OUT[q2];
            const float2 v = make_float2(xr[pos], xi[pos]);
            reg[q2 * kSB + lane] = q2 == 0 ? v : cmulf(v, tw2[q2 * 64 + lane]);
        }
        lds_wave_fence();

        // ---- pass 3 (wave q1, lane (q2, u0)): FFT16 over u1, twiddle w_64^{u0 q3}
#pragma unroll
        for (int u1 = 0; u1 < 16; ++u1) {
            const float2 v = reg[q2_3 * kSB + u0_3 + 4 * u1];
            xr[u1] = v.x; xi[u1] = v.y;
        }
        fft16(xr, xi);
        lds_barrier();  // every wave has finished with its region: image C overwrites all of them
#pragma unroll
        for (int q3 = 0; q3 < 16; ++q3) {
            const int pos = FFT16_OUT[q3];
            const float2 v = make_float2(xr[pos], xi[pos]);
            buf[(q3 * 4 + u0_3) * kSU + q2_3 * kSQ + wave] = (q3 == 0 || u0_3 == 0) ? v : cmulf(v, tw3[q3 * 4 + u0_3]);
        }
        lds_barrier();

        // ---- pass 4 (thread (q1, q2, g)): 4-point DFT over u0 for q3 = 4g + j -> q4
        float Xr[4][4], Xi[4][4];  // [j][q4]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q3 = 4 * g_4 + j;
            const float2 a0 = buf[(q3 * 4 + 0) * kSU + q2_4 * kSQ + q1_4];
            const float2 a1 = buf[(q3 * 4 + 1) * kSU + q2_4 * kSQ + q1_4];
            const float2 a2 = buf[(q3 * 4 + 2) * kSU + q2_4 * kSQ + q1_4];
            const float2 a3 = buf[(q3 * 4 + 3) * kSU + q2_4 * kSQ + q1_4];
            const float s02r = a0.x + a2.x, s02i = a0.y + a2.y, d02r = a0.x - a2.x, d02i = a0.y - a2.y;
            const float s13r = a1.x + a3.x, s13i = a1.y + a3.y, d13r = a1.x - a3.x, d13i = a1.y - a3.y;
            Xr[j][0] = s02r + s13r; Xi[j][0] = s02i + s13i;
            Xr[j][2] = s02r - s13r; Xi[j][2] = s02i - s13i;
            Xr[j][1] = d02r + d13i; Xi[j][1] = d02i - d13r;   // d02 - i d13
            Xr[j][3] = d02r - d13i; Xi[j][3] = d02i + d13r;   // d02 + i d13
        }
        if (job + gridDim.x < p.n_jobs) fetch(job + gridDim.x);  // ahead of this transform's stores
        lds_barrier();  // image C has been read
        // partner exchange: publish the upper half (q4 = 2, 3) as D[q4 - 2][klow]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            buf[klow_base + 256 * j] = make_float2(Xr[j][2], Xi[j][2]);
            buf[4096 + klow_base + 256 * j] = make_float2(Xr[j][3], Xi[j][3]);
        }
        lds_barrier();

        // ---- split + magnitude + store (fft.rs:81-98): k = klow + 4096 q4, q4 in {0, 1}
        char *row0 = reinterpret_cast<char *>(p.mags + (((size_t)(have_first ? f0 : 0) * p.pairs + pair) * (size_t)kM) * 2) - 8;
        char *row1 = reinterpret_cast<char *>(p.mags + (((size_t)f1 * p.pairs + pair) * (size_t)kM) * 2) - 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                const int klow = klow_base + 256 * j;
                // partner (4096 - klow, 3 - q4) = D[1 - q4][4096 - klow]; for klow = 0 this is D[.][4096] = D[. + 1][0]
                const float2 b = buf[(1 - q4) * 4096 + 4096 - klow];
                const float ar = Xr[j][q4], ai = Xi[j][q4];
                const float pr = ar + b.x, pi = ai - b.y;
                const float qr = ar - b.x, qi = ai + b.y;
                const float left = __builtin_amdgcn_sqrtf(fmaf(pr, pr, pi * pi)) * inv_w;
                const float right = __builtin_amdgcn_sqrtf(fmaf(qr, qr, qi * qi)) * inv_w;
                const int k = klow + 4096 * q4;
                if (k >= 1) {
                    if (MONO) {
                        if (have_first) *reinterpret_cast<float2 *>(row0 + (size_t)k * 8) = make_float2(left, left);
                        if (have_second) *reinterpret_cast<float2 *>(row1 + (size_t)k * 8) = make_float2(right, right);
                    } else {
                        *reinterpret_cast<float2 *>(row0 + (size_t)k * 8) = make_float2(left, right);
                    }
                }
            }
        }
    }
}

struct Tables16k {
    float2 *d_tw1 = nullptr, *d_tw2 = nullptr, *d_tw3 = nullptr;
};

}  // namespace wg16k

bool wg16384_supported(const sgx_ctx *c)
{
    // (l, r) pairs are loaded as one 8-byte word: the stream must be mono or have an even channel count
    return c->W == wg16k::kW && (c->C == 1 || (c->C & 1) == 0);
}

hipError_t wg16384_init(sgx_ctx *c, void **out)
{
    using namespace wg16k;
    auto *t = new Tables16k();
    auto unit = [](unsigned long long idx, unsigned long long N) {
        idx %= N;
        const double ang = -2.0 * M_PI * (double)idx / (double)N;
        double cs = cos(ang), sn = sin(ang);
        if (idx == 0) { cs = 1.0; sn = 0.0; }
        if (4 * idx == N) { cs = 0.0; sn = -1.0; }
        if (2 * idx == N) { cs = -1.0; sn = 0.0; }
        if (4 * idx == 3 * N) { cs = 0.0; sn = 1.0; }
        return make_float2((float)cs, (float)sn);
    };
    std::vector<float2> tw1(16 * 1024), tw2(16 * 64), tw3(16 * 4);
    for (int q = 0; q < 16; ++q) {
        for (int tt = 0; tt < 1024; ++tt) tw1[q * 1024 + tt] = unit((unsigned long long)tt * q, kP);
        for (int t0 = 0; t0 < 64; ++t0) tw2[q * 64 + t0] = unit((unsigned long long)t0 * q, 1024);
        for (int u0 = 0; u0 < 4; ++u0) tw3[q * 4 + u0] = unit((unsigned long long)u0 * q, 64);
    }
    auto up = [](float2 **dst, const std::vector<float2> &v) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), v.size() * sizeof(float2));
        if (e == hipSuccess) e = hipMemcpy(*dst, v.data(), v.size() * sizeof(float2), hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(&t->d_tw1, tw1);
    if (e == hipSuccess) e = up(&t->d_tw2, tw2);
    if (e == hipSuccess) e = up(&t->d_tw3, tw3);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft16384_wg_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft16384_wg_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e != hipSuccess) {
        wg16384_destroy(t);
        return e;
    }
    (void)c;
    *out = t;
    return hipSuccess;
}

void wg16384_destroy(void *tables)
{
    auto *t = static_cast<wg16k::Tables16k *>(tables);
    if (!t) return;
    if (t->d_tw1) (void)hipFree(t->d_tw1);
    if (t->d_tw2) (void)hipFree(t->d_tw2);
    if (t->d_tw3) (void)hipFree(t->d_tw3);
    delete t;
}

hipError_t launch_stft_wg16384(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                               size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags)
{
    using namespace wg16k;
    if (n_frames == 0) return hipSuccess;
    const auto *t = static_cast<const Tables16k *>(tables);
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
    Params p{};
    p.pcm = d_pcm;
    p.tw1 = t->d_tw1;
    p.tw2 = t->d_tw2;
    p.tw3 = t->d_tw3;
    p.window = c->d_window;
    p.mags = d_mags;
    p.first_frame = first_frame;
    p.n_frames = n_frames;
    p.total_frames = total_frames;
    p.H = c->H;
    p.C = channels;
    p.pairs = pairs;
    const bool mono = channels == 1 && !(c->cfg.flags & SGX_FLAG_INDEPENDENT_FRAMES);
    if (channels == 1 && !mono) return hipErrorNotSupported;  // caller falls back to the generic kernel
    p.pair_base = mono ? first_frame / 2 : 0;
    p.n_jobs = mono ? (first_frame + n_frames + 1) / 2 - first_frame / 2 : (unsigned long long)n_frames * pairs;
    // one persistent workgroup per CU (141 KB of LDS); jobs are dealt round-robin so that the channel
    // pairs of one hop position run on neighbouring CUs at the same time (shared lines hit L2)
    unsigned long long blocks = (unsigned long long)n_cu;
    if (blocks > p.n_jobs) blocks = p.n_jobs;
    const dim3 grid((unsigned)blocks), block(1024);
    if (mono) hipLaunchKernelGGL((stft16384_wg_kernel<true>), grid, block, kLdsBytes, c->stream, p);
    else hipLaunchKernelGGL((stft16384_wg_kernel<false>), grid, block, kLdsBytes, c->stream, p);
    return hipGetLastError();
}

}  // namespace sgx
