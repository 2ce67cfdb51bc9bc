// stft4096.hip -- tuned STFT for W = 2048 (P = 4096): one wavefront per transform.
//
// Replaces FastFourierTransform::process (fft.rs:43-99) + the hop loop (audio_transform.rs:34-42)
// for BASELINE config A.  One 64-lane wave owns one 4096-point complex FFT from PCM load to
// magnitude store; there is no workgroup barrier on the hot path.
//
//   n = 64 a + b  (a < 32 non-zero rows: the zero padding is never touched),   k = k1 + 64 k2
//   step 1  lane b : Y_b[k1] = sum_a z[64a+b] w_64^{a k1}        two 32-point FFTs (even / odd k1)
//           twiddle: T_b[k1] = Y_b[k1] * w_4096^{b k1}            table in LDS, [k1][b]
//   exchange through a per-wave LDS plane, real parts then imaginary parts (17 KB per wave)
//   step 2  lane k1: X[k1 + 64 k2] = sum_b T_b[k1] w_64^{b k2}   radix-2 stage + two 32-point FFTs
//   split   F[k], F[P-k] -> |L^[k]|, |R^[k]| (fft.rs:81-89): the partner bin lives in lane
//           (64 - k1) % 64, fetched with ds_bpermute; scale 2/W (fft.rs:92)
//
// Mono streams (the reference duplicates a mono sample into (s, s), audio_input_list_model.rs:67-69)
// use the same transform for TWO frames: frame 2j rides in the real part, frame 2j+1 in the
// imaginary part, and the split that separates left from right separates the two frames.
#include "sgx_internal.hpp"

namespace sgx {

typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float cl_fma(float a, float c, float u) { return fmaf(a, c, u); }
__device__ __forceinline__ f2v cl_fma(f2v a, float c, f2v u) { return __builtin_elementwise_fma(a, f2v{c, c}, u); }

#include "fft_codelets.inc"

namespace {

constexpr int kW = 2048, kP = 4096, kM = 2047;
constexpr int kPlaneStride = 68;                      // floats per exchange row (64 + 4: conflict-free b128 reads)
constexpr int kPlaneFloats = 64 * kPlaneStride;       // 4352 floats = 17408 B per wave
constexpr int kTwFloats = 64 * 64 * 2;                // [k1][b] complex
constexpr int kWinFloats = kW;

struct Fast4096Tables {
    float2 *d_tw;  // [64][64]: w_4096^{b k1} at [k1][b]
};

struct Fast4096Params {
    const float *pcm;
    const float2 *tw;
    const float *window;
    float *mags;
    unsigned long long first_frame;
    unsigned long long n_frames;
    unsigned long long n_jobs;
    unsigned long long jobs_per_block;
    unsigned long long pair_base;  // mono: global index of the first frame pair
    unsigned long long total_frames;
    uint32_t H, C, pair_l, pair_r, pairs, pair;
};

template <int NWAVES, bool MONO>
__global__ void __launch_bounds__(NWAVES * 64) stft4096_kernel(Fast4096Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *tw = reinterpret_cast<float2 *>(smem_raw);                           // 32 KB
    float *win = reinterpret_cast<float *>(smem_raw) + kTwFloats;                 // 8 KB
    float *planes = win + kWinFloats;                                            // NWAVES * 17 KB

    const int tid = threadIdx.x;
    for (int i = tid; i < 64 * 64; i += NWAVES * 64) tw[i] = p.tw[i];
    for (int i = tid; i < kW; i += NWAVES * 64) win[i] = p.window[i];
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    float *plane = planes + wave * kPlaneFloats;
    const int partner_addr = ((64 - lane) & 63) << 2;   // ds_bpermute byte address of the partner lane
    const bool lane0 = lane == 0;
    const float inv_w = 1.0f / (float)kW;               // (hypot / 2) * (2 / W), exact powers of two

    const unsigned long long job_begin = (unsigned long long)blockIdx.x * p.jobs_per_block;
    unsigned long long job_end = job_begin + p.jobs_per_block;
    if (job_end > p.n_jobs) job_end = p.n_jobs;

    for (unsigned long long job = job_begin + wave; job < job_end; job += NWAVES) {
        // ---- load + Hann (fft.rs:53-63); real part = left / frame f0, imaginary part = right / frame f0+1
        float er[32], ei[32], orr[32], oi[32];
        long long f0;  // local (output) index of the first frame of the job; -1 when it precedes the range
        bool have_first = true, have_second = true;
        if (MONO) {
            // pairs follow the GLOBAL frame index so that results do not depend on where a range starts
            const unsigned long long g0 = 2 * (p.pair_base + job);
            f0 = (long long)g0 - (long long)p.first_frame;
            have_first = f0 >= 0;
            have_second = f0 + 1 < (long long)p.n_frames;
            const bool data_second = g0 + 1 < p.total_frames;  // the partner is transformed whenever the stream holds it
            const float *s0 = p.pcm + g0 * p.H + lane;
            const float *s1 = data_second ? s0 + p.H : s0;
#pragma unroll
            for (int a = 0; a < 32; ++a) {
                const float w = win[64 * a + lane];
                er[a] = s0[64 * a] * w;
                ei[a] = data_second ? s1[64 * a] * w : 0.0f;
            }
        } else {
            f0 = job;
            const float *s0 = p.pcm + ((p.first_frame + f0) * p.H + lane) * p.C;
#pragma unroll
            for (int a = 0; a < 32; ++a) {
                const float w = win[64 * a + lane];
                er[a] = s0[(size_t)(64 * a) * p.C + p.pair_l] * w;
                ei[a] = s0[(size_t)(64 * a) * p.C + p.pair_r] * w;
            }
        }

        // ---- step 1: 64-point DFT over a with 32 non-zero inputs = FFT32(z) and FFT32(z * w_64^a)
#pragma unroll
        for (int a = 0; a < 32; ++a) { orr[a] = er[a]; oi[a] = ei[a]; }
        pretwiddle32_w64(orr, oi);
        fft32(er, ei);
        fft32(orr, oi);

        // ---- twiddle by w_4096^{b k1}, then transpose through LDS: real plane, then imaginary plane
        float tr[64], ti[64];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int pos = FFT32_OUT[j];
            const float2 t0 = tw[(2 * j) * 64 + lane];
            const float2 t1 = tw[(2 * j + 1) * 64 + lane];
            tr[2 * j] = fmaf(er[pos], t0.x, -(ei[pos] * t0.y));
            ti[2 * j] = fmaf(er[pos], t0.y, ei[pos] * t0.x);
            tr[2 * j + 1] = fmaf(orr[pos], t1.x, -(oi[pos] * t1.y));
            ti[2 * j + 1] = fmaf(orr[pos], t1.y, oi[pos] * t1.x);
        }
        float xr[64], xi[64];
#pragma unroll
        for (int k1 = 0; k1 < 64; ++k1) plane[k1 * kPlaneStride + lane] = tr[k1];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float4 v = *reinterpret_cast<const float4 *>(plane + lane * kPlaneStride + 4 * c);
            xr[4 * c] = v.x; xr[4 * c + 1] = v.y; xr[4 * c + 2] = v.z; xr[4 * c + 3] = v.w;
        }
#pragma unroll
        for (int k1 = 0; k1 < 64; ++k1) plane[k1 * kPlaneStride + lane] = ti[k1];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float4 v = *reinterpret_cast<const float4 *>(plane + lane * kPlaneStride + 4 * c);
            xi[4 * c] = v.x; xi[4 * c + 1] = v.y; xi[4 * c + 2] = v.z; xi[4 * c + 3] = v.w;
        }

        // ---- step 2: 64-point FFT over b = radix-2 DIF stage, then FFT32 on sums (even k2) and on
        //      twiddled differences (odd k2)
        float sr[32], si[32], dr[32], di[32];
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            sr[b] = xr[b] + xr[b + 32]; si[b] = xi[b] + xi[b + 32];
            dr[b] = xr[b] - xr[b + 32]; di[b] = xi[b] - xi[b + 32];
        }
        pretwiddle32_w64(dr, di);
        fft32(sr, si);
        fft32(dr, di);
        // now X[lane + 64 (2m)] = (sr, si)[FFT32_OUT[m]],  X[lane + 64 (2m+1)] = (dr, di)[FFT32_OUT[m]]

        // ---- split + magnitude + store (fft.rs:81-98); only k = 1 .. W-1 is kept (k2 < 32)
        float *row0, *row1;
        if (MONO) {
            row0 = p.mags + (((size_t)(have_first ? f0 : 0) * p.pairs + p.pair) * (size_t)kM) * 2;
            row1 = p.mags + (((size_t)(f0 + 1) * p.pairs + p.pair) * (size_t)kM) * 2;
        } else {
            row0 = p.mags + ((f0 * p.pairs + p.pair) * (size_t)kM) * 2;
            row1 = row0;
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int own = FFT32_OUT[m];
            const int mir = FFT32_OUT[31 - m];
            const int mir0 = FFT32_OUT[(32 - m) & 31];  // lane 0 only: bin 64 * (64 - 2m)
            // even k2 = 2m: own = S[m]; partner bin P-k sits in lane (64-k1)%64 as D[31-m]
            // (lane 0 is its own partner and needs S[32-m] instead)
            {
                const float send_r = lane0 ? sr[mir0] : dr[mir];
                const float send_i = lane0 ? si[mir0] : di[mir];
                const float br = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner_addr, __builtin_bit_cast(int, send_r)));
                const float bi = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner_addr, __builtin_bit_cast(int, send_i)));
                const float ar = sr[own], ai = si[own];
                const float pr = ar + br, pi = ai - bi;   // a + conj(b) = 2 L^
                const float qr = ar - br, qi = ai + bi;   // a - conj(b) = 2i R^
                const float left = __builtin_amdgcn_sqrtf(fmaf(pr, pr, pi * pi)) * inv_w;
                const float right = __builtin_amdgcn_sqrtf(fmaf(qr, qr, qi * qi)) * inv_w;
                const int k = lane + 64 * (2 * m);
                if (k >= 1) {
                    if (MONO) {
                        if (have_first) reinterpret_cast<float2 *>(row0)[k - 1] = make_float2(left, left);
                        if (have_second) reinterpret_cast<float2 *>(row1)[k - 1] = make_float2(right, right);
                    } else {
                        reinterpret_cast<float2 *>(row0)[k - 1] = make_float2(left, right);
                    }
                }
            }
            // odd k2 = 2m+1: own = D[m]; partner is S[31-m] of lane (64-k1)%64 (lane 0: its own D[31-m])
            {
                const float send_r = lane0 ? dr[mir] : sr[mir];
                const float send_i = lane0 ? di[mir] : si[mir];
                const float br = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner_addr, __builtin_bit_cast(int, send_r)));
                const float bi = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner_addr, __builtin_bit_cast(int, send_i)));
                const float ar = dr[own], ai = di[own];
                const float pr = ar + br, pi = ai - bi;
                const float qr = ar - br, qi = ai + bi;
                const float left = __builtin_amdgcn_sqrtf(fmaf(pr, pr, pi * pi)) * inv_w;
                const float right = __builtin_amdgcn_sqrtf(fmaf(qr, qr, qi * qi)) * inv_w;
                const int k = lane + 64 * (2 * m + 1);
                if (MONO) {
                    if (have_first) reinterpret_cast<float2 *>(row0)[k - 1] = make_float2(left, left);
                    if (have_second) reinterpret_cast<float2 *>(row1)[k - 1] = make_float2(right, right);
                } else {
                    reinterpret_cast<float2 *>(row0)[k - 1] = make_float2(left, right);
                }
            }
        }
    }
}

constexpr int kWaves = 6;
constexpr size_t kLdsBytes = (size_t)(kTwFloats + kWinFloats + kWaves * kPlaneFloats) * sizeof(float);

}  // namespace

static_assert(kW == 2048, "fast4096_supported (sgx_internal.hpp) names this window");

hipError_t fast4096_init(sgx_ctx *c)
{
    auto *t = new Fast4096Tables();
    std::vector<float2> tw(64 * 64);
    for (int k1 = 0; k1 < 64; ++k1)
        for (int b = 0; b < 64; ++b) {
            const int idx = (b * k1) % kP;
            const double ang = -2.0 * M_PI * (double)idx / (double)kP;
            double cs = cos(ang), sn = sin(ang);
            if (idx == 0) { cs = 1.0; sn = 0.0; }
            if (idx == kP / 4) { cs = 0.0; sn = -1.0; }
            if (idx == kP / 2) { cs = -1.0; sn = 0.0; }
            if (idx == 3 * kP / 4) { cs = 0.0; sn = 1.0; }
            tw[k1 * 64 + b] = make_float2((float)cs, (float)sn);
        }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&t->d_tw), tw.size() * sizeof(float2));
    if (e == hipSuccess) e = hipMemcpy(t->d_tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft4096_kernel<kWaves, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft4096_kernel<kWaves, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e != hipSuccess) {
        if (t->d_tw) (void)hipFree(t->d_tw);
        delete t;
        return e;
    }
    c->d_fast = t;
    return hipSuccess;
}

void fast4096_destroy(sgx_ctx *c)
{
    auto *t = static_cast<Fast4096Tables *>(c->d_fast);
    if (!t) return;
    if (t->d_tw) (void)hipFree(t->d_tw);
    delete t;
    c->d_fast = nullptr;
}

hipError_t launch_stft_fast4096(const sgx_ctx *c, const float *d_pcm, uint32_t channels, uint32_t pairs, size_t first_frame,
                                size_t n_frames, size_t total_frames, float *d_mags)
{
    if (n_frames == 0) return hipSuccess;
    const auto *t = static_cast<const Fast4096Tables *>(c->d_fast);
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
    for (uint32_t pair = 0; pair < pairs; ++pair) {
        Fast4096Params p;
        p.pcm = d_pcm;
        p.tw = t->d_tw;
        p.window = c->d_window;
        p.mags = d_mags;
        p.first_frame = first_frame;
        p.n_frames = n_frames;
        p.total_frames = total_frames;
        p.H = c->H;
        p.C = channels;
        p.pairs = pairs;
        p.pair = pair;
        p.pair_l = channels == 1 ? 0 : 2 * pair;
        p.pair_r = channels == 1 ? 0 : 2 * pair + 1;
        const bool mono = channels == 1 && !(c->cfg.flags & SGX_FLAG_INDEPENDENT_FRAMES);
        p.pair_base = mono ? first_frame / 2 : 0;
        p.n_jobs = mono ? (first_frame + n_frames + 1) / 2 - first_frame / 2 : n_frames;
        // one persistent workgroup per CU; each owns a contiguous run of jobs, a multiple of the
        // wave count so that its waves stay on neighbouring frames (shared audio stays in L1)
        unsigned long long blocks = (unsigned long long)n_cu;
        unsigned long long per = (p.n_jobs + blocks - 1) / blocks;
        per = (per + kWaves - 1) / kWaves * kWaves;
        blocks = (p.n_jobs + per - 1) / per;
        p.jobs_per_block = per;
        if (mono)
            hipLaunchKernelGGL((stft4096_kernel<kWaves, true>), dim3((unsigned)blocks), dim3(kWaves * 64), kLdsBytes, c->stream, p);
        else
            hipLaunchKernelGGL((stft4096_kernel<kWaves, false>), dim3((unsigned)blocks), dim3(kWaves * 64), kLdsBytes, c->stream, p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace sgx
