// stft4800_wg.hip -- tuned STFT for W = 2400 (P = 4800 = 16 x 20 x 15): the window the application itself runs,
// FastFourierTransform::new(48 kHz, 0.05 s) (gpu_spectrogram.rs:323, simple_spectrogram.rs:217).  One persistent 320-thread
// workgroup per transform, three trips through LDS, in the manner of stft4096_wg.hip; replaces fft.rs:43-99 + audio_transform.rs:34-42.
//
//   sample index  n = t + 300 a             (t < 300, a < 8 non-zero rows: 2400 = 8 x 300, the padding is never touched)
//   pass 1  thread t < 300      : 16-point DFT over a (8 non-zero inputs = two 8-point FFTs) -> q1; twiddle w_4800^{t q1}
//                                 (15 per-thread constants in VGPRs)
//   pass 2  thread (q1, t0) < 240: t = t0 + 15 t1; 20-point DFT over t1 -> q2; twiddle w_300^{t0 q2} (LDS table)
//   pass 3  thread u = q1 + 16 q2: 15-point DFT over t0 -> q3;  bin k = u + 320 q3
//   split   F[k] and F[P - k] -> |L^[k]|, |R^[k]| (fft.rs:81-89): the partner of thread u is thread (320 - u) % 320, exchanged
//           through LDS (registers q3 = 7 .. 14 only; k = 1 .. 2399 is kept: q3 < 7 whole, q3 = 7 for u < 160)
//
// Against the composite-radix kernel of stft_mixed.hip (any smooth length; one workgroup per frame, twiddles and window read from
// L2 per butterfly, bins found through a digit-reversal table): resident twiddles, the next frame's samples requested ahead of this
// frame's stores, natural-order bins in registers (coalesced row stores straight from them): 4 480 instead of 7 230 vector
// instructions per transform.  Same-device A/B, 262 144 frames at hop 93 (profiles/r03_app_point.txt): (l, r) rows 81.6 -> 89 M
// frames/s, mono rows 132 -> 161 M.  Rows and half rows only: PCM -> RGBA stays with the composite-radix kernel's fused path
// (this kernel's, on the generic pixel passes, ran at 45 M frames/s against 74 M there), and so do streams of more than two channels.
// What the shape costs: 5 waves on 4 SIMDs (one SIMD carries two waves of every phase) and three workgroups per CU (LDS) -- the
// vector pipe issues one instruction per ~4 clocks here, as in stft4096_wg.hip, against ~3 at the composite kernel's 6 waves per SIMD.
//
// Mono streams (a mono sample is duplicated into (s, s): audio_input_list_model.rs:67-69) pack TWO consecutive frames into one
// transform: frame 2j in the real part, frame 2j+1 in the imaginary part; the split that separates left from right separates them.
#include <cmath>
#include <vector>

#include <hip/hip_fp16.h>

#include "mix_codelets.hpp"
#include <type_traits>

#include "sgx_internal.hpp"

namespace sgx {

namespace w48 {

constexpr int kP = 4800, kW = 2400, kM = kW - 1;
constexpr int kT = 320;                 // threads
constexpr int kN1 = 300, kN2 = 240;     // threads with a butterfly in pass 1 / pass 2
constexpr int kS1 = 302;                // image 1 [16][300]: row stride.  Pass 2 reads (q1, t0) at q1 kS1 + t0 + 15 t1 with q1 the fast
                                        // lane index: 302 = 14 (mod 32), so the 16 rows start in 16 different even 8-byte banks
constexpr int kS2 = 336;                // image 2 [15][320]: row stride = 16 (mod 32): pass 2 writes rows t0, t0 + 1 side by side
constexpr int kBuf = 15 * kS2;          // 5040 complex: image 2 (image 1: 4832; partner slots: 2560; column + samples of the pixel stage)
constexpr size_t kLdsBytes = (size_t)(kBuf + 15 * 20) * sizeof(float2) + (size_t)kW * sizeof(float);   // + w_300^{t0 q2} at [t0][q2] + the Hann
                                                                       // window: 52 320 B, three workgroups per CU

typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float cl_fma(float a, float c, float u) { return fmaf(a, c, u); }
__device__ __forceinline__ f2v cl_fma(f2v a, float c, f2v u) { return __builtin_elementwise_fma(a, f2v{c, c}, u); }

#include "fft_codelets.inc"

using mix::cmul;

struct W48Tables {
    float2 *d_tw1 = nullptr;   // [8][320][2]  w_4800^{t q1} at [q1 / 2][t][q1 % 2]: a lane takes (q1, q1 + 1) as one 16-byte word
    float2 *d_tw2 = nullptr;   // [15][20]   w_300^{t0 q2} at [t0][q2]
};

struct Params {
    const float *pcm;
    const float2 *tw1, *tw2;
    const float *window;       // [2400]
    float *mags;
    unsigned long long first_frame, n_frames, total_frames, pair_base, n_jobs, jobs_per_block;
    uint32_t H, pairs;
    float half_scale;          // (hypot / 2) (2 / W)
};

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS only: outstanding global stores are not waited for
}

// a raw buffer descriptor over a wave-uniform address: loads and stores then take an SGPR base + one 32-bit lane offset + a scalar
// offset, no per-lane 64-bit address arithmetic
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void *base)
{
    const unsigned long long a = (unsigned long long)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}

// MODE: 0 an (l, r) stream, one frame per transform; 1 a mono stream, frames (2j, 2j+1) per transform; 2 a mono stream, every
// frame its own (s, s) transform (the default: the reference's dataflow)
#ifndef W48_ABL
#define W48_ABL 0   // ablation builds (profiles/r03_app_point.txt): 1 no row stores, 2 no sample loads, 4 / 8 no butterfly in pass 2 / 3
#endif
template <int MODE, bool F16>
__global__ void __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(4, 4))) stft4800_wg_kernel(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *buf = reinterpret_cast<float2 *>(smem_raw);
    float2 *tw2 = buf + kBuf;

    float *winl = reinterpret_cast<float *>(tw2 + 15 * 20);   // [2400] the Hann window (fft.rs:61)

    const int tid0 = threadIdx.x;
    if (tid0 < 300) tw2[tid0] = p.tw2[tid0];
    for (int i = tid0; i < kW; i += kT) winl[i] = p.window[i];
    // Everything a lane derives from its index (roles, LDS offsets, store offsets) is derived again where it is used, from a copy of
    // the index the compiler cannot see through: hoisted out of the loop these values are a dozen registers that live through the
    // 20-point butterfly of pass 2, and what does not fit there is spilled -- a spill reload is a vector-memory load, and its wait
    // (vmcnt(0)) waits for the row stores in flight as well.
    auto lane = [&]() { int t = tid0; asm volatile("" : "+v"(t)); return t; };
    // (the 20 lanes without a pass-1 butterfly load what lane 299 loads and never store it)
    auto lane1 = [&]() { const int t = lane(); return t < kN1 ? t : kN1 - 1; };

    // per-thread constants, in registers for the life of the (persistent) workgroup
    float2 tw1[16];
    auto load_tw1 = [&]() {
        const __amdgpu_buffer_rsrc_t rt = uniform_rsrc(p.tw1);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rt, 16 * tid0, 16 * kT * g, 0);
            tw1[2 * g] = make_float2(__uint_as_float(w.x), __uint_as_float(w.y));
            tw1[2 * g + 1] = make_float2(__uint_as_float(w.z), __uint_as_float(w.w));
        }
    };
    load_tw1();
    __syncthreads();

    const unsigned long long job_begin = (unsigned long long)blockIdx.x * p.jobs_per_block;
    unsigned long long job_end = job_begin + p.jobs_per_block;
    if (job_end > p.n_jobs) job_end = p.n_jobs;
    if (job_begin >= job_end) return;

    // Software pipeline: the samples of transform j+1 are requested while transform j is in its last pass, BEFORE j's stores (vmcnt
    // retires in issue order: a load behind the stores waits for every one of them), and consumed (Hann, fft.rs:53-63) at the END of
    // iteration j, in straight-line code behind the stores: the compiler then waits with vmcnt(stores issued since) -- at the loop
    // header it would merge the entry path and fall back to vmcnt(0) (measured on the 16384-point kernel of round 3: 0.93 ms per 20 000 transforms against 0.61 without the stores, profiles/r03_k16_ablation.txt).
    float pl[8], pr[8];
    struct JobIn { const float *base; bool data_second; };
    auto job_in = [&](unsigned long long job) {
        JobIn j{nullptr, true};
        if (MODE == 1) {
            const unsigned long long f = 2 * (p.pair_base + job);
            j.data_second = f + 1 < p.total_frames;
            j.base = p.pcm + f * p.H;
        } else {
            j.base = p.pcm + (p.first_frame + job) * p.H * (MODE == 0 ? 2 : 1);
        }
        return j;
    };
    const int second_off = (int)(p.H * 4);   // mono pairs: the second frame starts H samples on
    auto prefetch = [&](const JobIn &j) {
        const int tl = lane1();
        const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(j.base);
        const int sec = j.data_second ? second_off : 0;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            if (W48_ABL & 2) { pl[a] = (float)(a + tl); pr[a] = (float)(tl - a); continue; }
            if (MODE == 0) {
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, 8 * tl, 8 * kN1 * a, 0);
                pl[a] = __uint_as_float(v.x); pr[a] = __uint_as_float(v.y);
            } else {
                pl[a] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, 4 * tl, 4 * kN1 * a, 0));
                if (MODE == 1) pr[a] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, 4 * tl, 4 * kN1 * a + sec, 0));
            }
        }
    };
    float er[8], ei[8];
    auto take = [&](bool data_second) {
        // behind the stores: the samples pass through a statement the compiler may not move above them (a memory clobber), so
        // neither may the products -- nor their wait
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            if (MODE == 2) asm volatile("" : "+v"(pl[a]) :: "memory");
            else asm volatile("" : "+v"(pl[a]), "+v"(pr[a]) :: "memory");
        }
        const float *wl = winl + lane1();
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const float w = wl[kN1 * a];
            er[a] = pl[a] * w;
            ei[a] = MODE == 2 ? er[a] : ((MODE == 1 && !data_second) ? 0.0f : pr[a] * w);
        }
        // (the products are formed HERE: left to itself the compiler sinks them to their use at the head of the next iteration and keeps
        // the prefetch registers and their wait alive across the loop edge)
#pragma unroll
        for (int a = 0; a < 8; ++a) asm volatile("" : "+v"(er[a]), "+v"(ei[a]));
    };
    {
        const JobIn first = job_in(job_begin);
        prefetch(first);
        take(first.data_second);
    }

    for (unsigned long long job = job_begin; job < job_end; ++job) {
        // local (output) frame indices; mono pairs: f0 may be -1 (the pair's first frame precedes the range)
        const long long f0 = MODE == 1 ? (long long)(2 * (p.pair_base + job)) - (long long)p.first_frame : (long long)job;
        const long long f1 = f0 + 1;
        const bool have_first = MODE != 1 || f0 >= 0;
        const bool have_second = MODE == 1 && f1 < (long long)p.n_frames;
        const bool more = job + 1 < job_end;
        const JobIn nxt = job_in(more ? job + 1 : job);

        // ---- pass 1: 16-point DFT over a, inputs a >= 8 are the zero padding: even q1 = FFT8(z), odd q1 = FFT8(z w_16^a)
        float orr[8], oi[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) { orr[a] = er[a]; oi[a] = ei[a]; }
        pretwiddle8_w16(orr, oi);
        fft8(er, ei);
        fft8(orr, oi);
        lds_barrier();  // the previous transform's partner reads are complete
        if (const int tid = lane(); tid < kN1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int pos = FFT8_OUT[j];
                const float2 ve = make_float2(er[pos], ei[pos]);
                const float2 vo = make_float2(orr[pos], oi[pos]);
                buf[(2 * j) * kS1 + tid] = j == 0 ? ve : cmul(ve, tw1[2 * j]);
                buf[(2 * j + 1) * kS1 + tid] = cmul(vo, tw1[2 * j + 1]);
            }
        }
        lds_barrier();

        // ---- pass 2: thread (q1, t0): 20-point DFT over t1, then twiddle w_300^{t0 q2}
        float2 x[20];
        if (const int tid = lane(); tid < kN2) {
            const float2 *r2 = buf + (tid & 15) * kS1 + (tid >> 4);
#pragma unroll
            for (int t1 = 0; t1 < 20; ++t1) x[t1] = r2[15 * t1];
            if (!(W48_ABL & 4)) mix::dft_composite<5, 4>(x);
        }
        lds_barrier();  // everyone has read image 1
        if (const int tid = lane(); tid < kN2) {
            float2 *w2 = buf + (tid >> 4) * kS2 + (tid & 15);
            const float2 *tw = tw2 + 20 * (tid >> 4);
#pragma unroll
            for (int q2 = 0; q2 < 20; ++q2) w2[16 * q2] = q2 == 0 ? x[0] : cmul(x[q2], tw[q2]);
        }
        lds_barrier();

        // ---- pass 3: thread u = q1 + 16 q2: 15-point DFT over t0 -> bins k = u + 320 q3
        float2 y[15];
#pragma unroll
        for (int t0 = 0; t0 < 15; ++t0) y[t0] = buf[t0 * kS2 + lane()];
        if (!(W48_ABL & 8)) mix::dft_composite<5, 3>(y);
        // (the bins are FINISHED before the next transform's samples are requested: the compiler otherwise moves the request above the
        // butterfly, where the sixteen extra live registers do not fit)
#pragma unroll
        for (int i = 0; i < 15; ++i) asm volatile("" : "+v"(y[i].x), "+v"(y[i].y) :: "memory");
        lds_barrier();  // everyone has read image 2
        // partner exchange: publish q3 = 7 .. 14 (the bins P - k of the kept half)
#pragma unroll
        for (int j = 0; j < 8; ++j) buf[j * kT + lane()] = y[7 + j];
        // the next transform's samples: ahead of this transform's stores, behind the publish (eight of the fifteen bins are dead there);
        // unconditional -- the last transform of a run requests its own samples again --, so that the prefetch registers are dead from
        // `take` to here (with `if (more)` they are a phi of old and new values, live around the whole loop)
        prefetch(nxt);
        lds_barrier();

        // ---- split + magnitude (fft.rs:81-98).  F[P - k] of k = u + 320 q3: thread 320 - u holds it as q3' = 14 - q3 (slot 7 - q3);
        //      thread 0 is its own partner one slot up (q3' = 15 - q3; its q3 = 0 is bin 0, never stored)
        const int tid = lane();
        const int pcol = tid == 0 ? kT : kT - tid;   // slot s of the partner at buf[s kT + (pcol mod kT)]: thread 0 reads one slot up
        // Every segment is stored as soon as its bins are split (order 1, 0, 2 .. 7: segment 0's dropped lane re-stores its segment-1 bin):
        // eight stores spread over the split instead of a burst of eight behind it -- a wave stalls at the issue of a store while the
        // CU's vector-memory path drains the previous ones (profiles/r05_k1_stereo.txt; W48_BURST restores the burst for A/B).
        float ml[8], mr[8];
        float2 pb[8];
#pragma unroll
        for (int q3 = 0; q3 < 8; ++q3) {
            const int at = (7 - q3) * kT + pcol;     // thread 0, q3 = 0: slot 8 does not exist -- reads slot 7 of column 0 instead
            pb[q3] = buf[(q3 == 0 && tid == 0) ? 7 * kT : at];
        }
        // ---- store [F][pairs][M][2]: uniform row base (SGPR) + one 32-bit lane offset + a scalar offset per q3.
        // STRAIGHT-LINE code: the wait for the prefetched samples in `take` below is vmcnt(stores issued since), and the compiler
        // can only count stores it does not have to branch around -- with one branch in here it waits for every store to be
        // acknowledged by memory, once per transform (measured: 3.4 us per transform instead of 1.9).  So:
        //   * lanes whose bin is not an output (bin 0 = lane 0 at q3 = 0; bins >= 2400 = lanes >= 160 at q3 = 7) store a bin they
        //     do own a second time (lane 0: its q3 = 1 bin; lanes >= 160: their q3 = 6 bin) -- same address, same value.  The
        //     q3 = 6 bin lies one kT-slab BELOW the q3 = 7 scalar offset: the descriptors start one slab in front of the row and
        //     every lane offset carries + kT bins, so that no lane offset is ever negative (as the unsigned 32-bit voffset of a
        //     raw buffer store a negative one is ~4 GiB: dropped by the range check today, a stray write if that ever changed);
        //   * a mono pair whose first or second frame lies outside the requested range stores the other row twice.
        char *base = reinterpret_cast<char *>(p.mags);
        const int bin_bytes = F16 ? 4 : 8;
        const bool sa = have_first, sb = MODE == 1 ? have_second : false;
        const long long fa = sa ? f0 : f1, fb = sb ? f1 : f0;            // (row, values) of the two stores of a mono pair
        const __amdgpu_buffer_rsrc_t ra = uniform_rsrc(base + ((long long)((size_t)fa * p.pairs * (size_t)kM) - 1 - kT) * bin_bytes);
        const __amdgpu_buffer_rsrc_t rb = uniform_rsrc(base + ((long long)((size_t)fb * p.pairs * (size_t)kM) - 1 - kT) * bin_bytes);
        const bool drop0 = tid == 0, drop7 = tid >= kW - 7 * kT;
        auto split = [&](const int q3) {
            const float2 b = pb[q3];
            const float ar = y[q3].x, ai = y[q3].y;
            const float sr_ = ar + b.x, si_ = ai - b.y;   // a + conj(b) = 2 L^
            const float dr_ = ar - b.x, di_ = ai + b.y;   // a - conj(b) = 2i R^
            ml[q3] = __builtin_amdgcn_sqrtf(fmaf(sr_, sr_, si_ * si_)) * p.half_scale;
            mr[q3] = __builtin_amdgcn_sqrtf(fmaf(dr_, dr_, di_ * di_)) * p.half_scale;
        };
        auto store = [&](const int q3) {
            float l = ml[q3], r = mr[q3];
            if ((W48_ABL & 1) && l != -12345.0f) return;
            int lane_off = bin_bytes * (tid + kT);
            if (q3 == 0) { l = drop0 ? ml[1] : l; r = drop0 ? mr[1] : r; lane_off = drop0 ? bin_bytes * 2 * kT : lane_off; }
            if (q3 == 7) { l = drop7 ? ml[6] : l; r = drop7 ? mr[6] : r; lane_off = drop7 ? bin_bytes * tid : lane_off; }
            const float va = MODE == 1 ? (sa ? l : r) : l, vb = MODE == 1 ? (sb ? r : l) : r;
            if (F16) {
                const __half2 ha = MODE == 1 ? __floats2half2_rn(va, va) : __floats2half2_rn(l, r);
                __builtin_amdgcn_raw_buffer_store_b32(*reinterpret_cast<const uint32_t *>(&ha), ra, lane_off, bin_bytes * kT * q3, 0);
                if (MODE == 1) {
                    const __half2 hb = __floats2half2_rn(vb, vb);
                    __builtin_amdgcn_raw_buffer_store_b32(*reinterpret_cast<const uint32_t *>(&hb), rb, lane_off, bin_bytes * kT * q3, 0);
                }
            } else {
                const u32x2 da = MODE == 1 ? u32x2{__float_as_uint(va), __float_as_uint(va)} : u32x2{__float_as_uint(l), __float_as_uint(r)};
                __builtin_amdgcn_raw_buffer_store_b64(da, ra, lane_off, bin_bytes * kT * q3, 2);
                if (MODE == 1) __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(vb), __float_as_uint(vb)}, rb, lane_off, bin_bytes * kT * q3, 2);
            }
        };
#ifdef W48_BURST
#pragma unroll
        for (int q3 = 0; q3 < 8; ++q3) split(q3);
#pragma unroll
        for (int q3 = 0; q3 < 8; ++q3) store(q3);
#else
        split(1);
        split(0);
        store(1);
        __builtin_amdgcn_sched_barrier(0);
        store(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q3 = 2; q3 < 8; ++q3) {
            split(q3);
            store(q3);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        take(nxt.data_second);
    }
}

}  // namespace w48

bool w4800_supported(const sgx_ctx *c) { return c->P == (uint32_t)w48::kP && c->W == (uint32_t)w48::kW; }

hipError_t w4800_init(sgx_ctx *c, void **out)
{
    using namespace w48;
    (void)c;
    auto *t = new W48Tables();
    std::vector<float2> tw1((size_t)16 * kT), tw2((size_t)15 * 20);
    auto unit = [](unsigned long long e, unsigned long long n) {   // e^{-2 pi i e / n}, exact on the axes
        e %= n;
        if (e == 0) return make_float2(1.0f, 0.0f);
        if (4 * e == n) return make_float2(0.0f, -1.0f);
        if (2 * e == n) return make_float2(-1.0f, 0.0f);
        if (4 * e == 3 * n) return make_float2(0.0f, 1.0f);
        const double ang = -2.0 * M_PI * (double)e / (double)n;
        return make_float2((float)cos(ang), (float)sin(ang));
    };
    for (int q1 = 0; q1 < 16; ++q1)
        for (int th = 0; th < kT; ++th) tw1[((size_t)(q1 / 2) * kT + th) * 2 + q1 % 2] = unit((unsigned long long)q1 * (th < kN1 ? th : 0), kP);
    for (int t0 = 0; t0 < 15; ++t0)
        for (int q2 = 0; q2 < 20; ++q2) tw2[(size_t)t0 * 20 + q2] = unit((unsigned long long)t0 * q2, 300);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&t->d_tw1), tw1.size() * sizeof(float2));
    if (e == hipSuccess) e = hipMemcpy(t->d_tw1, tw1.data(), tw1.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&t->d_tw2), tw2.size() * sizeof(float2));
    if (e == hipSuccess) e = hipMemcpy(t->d_tw2, tw2.data(), tw2.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        w4800_destroy(t);
        return e;
    }
    *out = t;
    return hipSuccess;
}

void w4800_destroy(void *tables)
{
    auto *t = static_cast<w48::W48Tables *>(tables);
    if (!t) return;
    if (t->d_tw1) (void)hipFree(t->d_tw1);
    if (t->d_tw2) (void)hipFree(t->d_tw2);
    delete t;
}

hipError_t launch_stft_w4800(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, size_t first_frame, size_t n_frames,
                               size_t total_frames, float *d_mags, bool out_f16)
{
    using namespace w48;
    if (n_frames == 0) return hipSuccess;
    if (channels > 2) return hipErrorInvalidValue;   // (the caller sends more channels to the composite-radix kernel)
    const auto *t = static_cast<const W48Tables *>(tables);
    Params p{};
    p.pcm = d_pcm;
    p.tw1 = t->d_tw1;
    p.tw2 = t->d_tw2;
    p.window = c->d_window;
    p.mags = d_mags;
    p.first_frame = first_frame;
    p.n_frames = n_frames;
    p.total_frames = total_frames;
    p.H = c->H;
    p.pairs = 1;
    p.half_scale = 0.5f * (2.0f / (float)c->W);
    const int mode = channels == 2 ? 0 : ((c->cfg.flags & SGX_FLAG_PAIRED_FRAMES) ? 1 : 2);
    p.pair_base = mode == 1 ? first_frame / 2 : 0;
    p.n_jobs = mode == 1 ? (first_frame + n_frames + 1) / 2 - first_frame / 2 : n_frames;
    // persistent workgroups, three per CU, each with a contiguous run of transforms: neighbouring frames share 96 % of their samples
    unsigned long long blocks = (unsigned long long)(c->n_cu > 0 ? c->n_cu : 256) * 3;
    unsigned long long per = (p.n_jobs + blocks - 1) / blocks;
    if (per < 1) per = 1;
    blocks = (p.n_jobs + per - 1) / per;
    p.jobs_per_block = per;
    const dim3 grid((unsigned)blocks), block(kT);
#define W48_GO(MODE_) \
    do { \
        if (out_f16) hipLaunchKernelGGL((stft4800_wg_kernel<MODE_, true>), grid, block, kLdsBytes, c->stream, p); \
        else hipLaunchKernelGGL((stft4800_wg_kernel<MODE_, false>), grid, block, kLdsBytes, c->stream, p); \
    } while (0)
    if (mode == 0) W48_GO(0);
    else if (mode == 1) W48_GO(1);
    else W48_GO(2);
#undef W48_GO
    return hipGetLastError();
}

}  // namespace sgx
