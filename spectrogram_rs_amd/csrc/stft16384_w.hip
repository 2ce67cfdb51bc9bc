// stft16384_w.hip -- tuned STFT for W = 8192 (P = 16384), fourth design: ONE 512-thread workgroup per transform, 32 points per thread,
// plan 32 x 32 x 16, two exchanges through LDS, three workgroup barriers, no recombination stage, no partner exchange, no staged row.
// (BASELINE config 4: 16384-point, hop 512, 8 interleaved channels = 4 (l, r) pairs per hop position.)
//
// Replaces FastFourierTransform::process (fft.rs:43-99) + the hop loop (audio_transform.rs:34-42).
//
// Why.  The third design (rounds 3-5, removed in round 6: 1024 threads x 16 points, four time-decimated 4096-point transforms in the lanes of a quad)
// pays per transform five workgroup barriers, a two-stage DPP recombination, a partner exchange through LDS and a row staged in LDS and
// read back: 48 LDS writes and 78 LDS reads per thread, 1024 threads.  Its vector pipe (8 550 cycles per CU and transform) and its LDS
// pipe (9 340) take turns (15 000 - 16 000 measured: profiles/r05_k16.txt).  With 512 threads a thread may use 256 registers, enough
// for 32 points, and 16384 = 32 x 32 x 16 needs two exchanges only:
//
//   n = c + 512 a           (c < 512, a < 32; a >= 16 is the zero padding, fft.rs:65-69)
//   c = c0 + 16 c1          (c0 < 16, c1 < 32)
//   k = q1 + 32 q2 + 1024 q3
//   F[k] = sum_c0 w_16^{c0 q3} w_512^{c0 q2} sum_c1 w_32^{c1 q2} w_16384^{c q1} sum_a w_32^{a q1} z[c + 512 a]
//
//   pass 1  thread c        : 32-point DFT over a with 16 non-zero inputs = two 16-point FFTs (even q1: FFT16(z); odd q1: FFT16(z w_32^a));
//                             twiddle w_16384^{q1 c} (31 per thread, resident in registers); image row q1, slot c
//   pass 2  thread (q1, c0) : q1 = tid & 31, c0 = tid >> 5; FFT32 over c1 -> q2; twiddle w_512^{q2 c0} (LDS, a broadcast read per half
//                             wave); written back IN PLACE: the 32 slots c0 + 16 j of row q1 are this thread's own, read as j = c1,
//                             written as j = q2 -- no barrier between the reads and the writes
//   pass 3  thread (q1, q2A): q2A = tid >> 5 < 16; TWO 16-point FFTs over c0: u_A = q1 + 32 q2A (< 512) and u_B = 1024 - u_A, whose
//                             registers hold each other's partners F[P - k] (k = u + 1024 q3 pairs with (1024 - u) + 1024 (15 - q3)):
//                             the L/R split (fft.rs:81-98) needs nothing from another thread.  Thread 0 holds the two self-paired
//                             columns u = 0 and u = 512.
//   rows    lanes of a half wave hold 32 consecutive bins: every store instruction writes 2 x 256 contiguous bytes, straight from the
//           registers (8 kept bins of u_A, 8 of u_B: 16 stores of 8 bytes per thread)
//
// Row stride 513 (odd): pass-1 writes are lane-consecutive; in pass 2 and 3 the 32 lanes of a half wave sit in 32 different rows at the
// same slot: banks 2 q1 mod 64 -- conflict-free 8-byte reads and writes.  131 KB of image + 4 KB of pass-2 twiddles: one workgroup per CU,
// 8 waves = 2 per SIMD.  Per thread 64 LDS writes and 95 reads (31 of them the broadcast twiddles) -- per transform 2/3 of the third
// design's writes and 0.6 of its reads -- and about 0.8 of its vector instructions.
#include <type_traits>

#include "sgx_internal.hpp"

namespace sgx {

namespace w16k {

typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float cl_fma(float a, float c, float u) { return fmaf(a, c, u); }
__device__ __forceinline__ f2v cl_fma(f2v a, float c, f2v u) { return __builtin_elementwise_fma(a, f2v{c, c}, u); }

#include "fft_codelets.inc"

constexpr int kW = 8192, kP = 16384, kM = 8191;
constexpr uint32_t kXcdHint = 8;         // XCDs of an MI355X (SPX): workgroup i runs on XCD i % 8.  Used for job locality only
constexpr int kS = 513;                  // row stride (complex) of the image [q1][slot]: odd
constexpr int kImg = 32 * kS;            // 16 416 complex = 131 328 B
constexpr size_t kLdsBytes = (size_t)(kImg + 512) * sizeof(float2);   // + tw2 [c0][q2]: 135 424 B

struct Params {
    const float *pcm;        // MONO: [n] floats; DIRECT: the interleaved stream [n][C]; else per-pair planes of (l, r): plane p starts at pcm + p * plane_floats
    size_t plane_floats;
    uint32_t stride_floats;  // DIRECT: floats from one sample of a pair to the next (the stream's channel count)
    long long sample_base;   // absolute sample index of pcm[0] (a duplicated mono plane holds a sub-range)
    const float2 *T1;        // [32][512]  w_16384^{q1 c} at [q1][c]
    const float2 *tw2;       // [16][32]   w_512^{q2 c0} at [c0][q2]
    const float *win16;      // [4][512][4] hann[c + 512 a] / W at [a / 4][c][a % 4]   (fft.rs:61; the scale (hypot / 2) (2 / W) = 2^-13 rides along)
    float *mags;
    unsigned long long first_frame, n_frames, total_frames, pair_base, n_jobs, jobs_per_xcd;
    uint32_t xcds;           // kXcdHint when the grid is a multiple of it, else 1 (plain round-robin)
    uint32_t H, pairs;
    unsigned long long run_len;   // SLIDE: hop positions per workgroup (a contiguous run of ONE pair's transforms)
};

__device__ __forceinline__ void lds_barrier()
{
#ifdef W_ABL_NOBAR
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}
// (ablation builds: W_ABL_NOLDSW keeps an image write's value live without the write)
__device__ __forceinline__ void lds_put(float2 *dst, float2 v)
{
#ifdef W_ABL_NOLDSW
    asm volatile("" ::"v"(v.x), "v"(v.y));
#else
    *dst = v;
#endif
}
__device__ __forceinline__ float2 cmulf(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
// An 8-byte LDS read that stays one (a merged ds_read2_b64 holds the LDS pipe for 8 cycles, two ds_read_b64 for 2 each): the empty
// statement makes the base a new value for every read.
typedef const f2v __attribute__((address_space(3))) lds_cfloat2;
__device__ __forceinline__ lds_cfloat2 *lds_ptr(const float2 *p) { return (lds_cfloat2 *)p; }
#ifndef W_NO_READ2
#define W_NO_READ2 1
#endif
__device__ __forceinline__ float2 lds_read_alone(lds_cfloat2 *&base, int idx)
{
    if (W_NO_READ2) asm("" : "+v"(base));
#ifdef W_ABL_NOLDSR
    float a_ = (float)idx, b_ = 1.0f;
    asm volatile("" : "+v"(a_), "+v"(b_) : "v"(base));
    return make_float2(a_, b_);
#endif
    const f2v v = base[idx];
    return make_float2(v.x, v.y);
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// every global access goes through a raw buffer descriptor: a wave-uniform base (4 SGPRs) + ONE 32-bit lane offset + a scalar offset
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void *base, uint32_t records = 0x7fffffffu)
{
    const unsigned long long a = (unsigned long long)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0, (int)records, 0x00020000);
}
#ifndef W_OUT_AUX
#define W_OUT_AUX 2   // nt: a write-once stream
#endif
#ifdef W_ABL_NOSTORE
#define W_STORE_OK(v) ((v) == 12345.678f)   // ablation builds: (practically) never true, but the value stays live
#else
#define W_STORE_OK(v) true
#endif

#if SGX_STAMPS
// diagnostic build only (tools/k16_phases.py): per-phase wave cycles (s_memtime), summed over all waves and iterations
__device__ unsigned long long g_phase_cycles16w[24];
#define SGX_STAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[i] += now_ - st_last; st_last = now_; }
#else
#define SGX_STAMP(i)
#endif

// a mono sample range as one (s, s) plane: what the reference's capture callback does to a mono device (audio_input_list_model.rs:67-69)
__global__ void __launch_bounds__(256) duplicate_mono_kernel(const float *pcm, float *plane, size_t first, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float s = pcm[first + i];
        reinterpret_cast<float2 *>(plane)[i] = make_float2(s, s);
    }
}

// MONO: SGX_FLAG_PAIRED_FRAMES -- frames 2j and 2j + 1 of a mono stream in the real and the imaginary part of one transform.
// DIRECT: more than two interleaved channels, pair p = channels (2 p, 2 p + 1), read where they lie: 8-byte loads at a stride of C floats
// (the four pairs of a hop position run on CUs of one XCD at the same time and share the lines in its L2; HBM traffic 1.02 x algorithmic).
// SLIDE (H = 512, the thread stride of pass 1): a workgroup takes a contiguous run of hop positions of ONE pair.  Thread c holds samples
// c + 512 a, a < 16, and the next hop position's sample a is this one's a + 1: the raw window stays in 32 registers, moves down one place
// per transform, and ONE sample per thread is requested instead of sixteen (8-byte loads at a stride of 4 C bytes use a quarter of every
// line they touch: without any request the kernel ran 9.6 against 10.4 ms per 400 000 transforms, profiles/r06_k16.txt).
template <bool MONO, bool DIRECT = false, bool SLIDE = false>
__global__ void __launch_bounds__(512, 2) stft16384_w_kernel(Params p)
{
    static_assert(!(MONO && SLIDE), "frame pairs of a mono stream do not slide");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *buf = reinterpret_cast<float2 *>(smem_raw);
    float2 *tw2 = buf + kImg;

    const int tid = threadIdx.x;
    tw2[tid] = p.tw2[tid];

    // Per-lane constants.  The pass-1 twiddle w_16384^{q1 c} of q1 = l + 4 h is the product of w^{l c} (l = 1 .. 3) and w^{4 h c}
    // (h = 1 .. 7): TEN complex values resident instead of 31 (62 registers: with them the next transform's samples could only be
    // requested behind pass 3, and a wave waited 2.7 ms per 400 000 transforms for them -- two waves per SIMD hide little).  Cost: a
    // second multiply on the 21 elements with l > 0 and h > 0 (+84 of ~1 700 vector instructions), one more float32 rounding on them.
    float2 twl[4], twh[8];
#pragma unroll
    for (int l = 1; l < 4; ++l) twl[l] = p.T1[l * 512 + tid];
#pragma unroll
    for (int h = 1; h < 8; ++h) twh[h] = p.T1[4 * h * 512 + tid];
    float win[16];
    {
        const __amdgpu_buffer_rsrc_t rw = uniform_rsrc(p.win16);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rw, 16 * tid, g * (512 * 16), 0);
            win[4 * g] = __uint_as_float(w.x); win[4 * g + 1] = __uint_as_float(w.y);
            win[4 * g + 2] = __uint_as_float(w.z); win[4 * g + 3] = __uint_as_float(w.w);
        }
    }
    __syncthreads();

    // Software pipeline: the samples of the NEXT transform are requested before this transform's stores (vmcnt retires in issue order).
    float pl[16], pr[16];
    struct JobIn { const float *base; bool data_second; };   // base: wave-uniform
    auto job_in = [&](unsigned long long job, unsigned long long hop, uint32_t pair) {
        JobIn j{nullptr, true};
        if (MONO) {
            const unsigned long long f = 2 * (p.pair_base + job);
            j.data_second = f + 1 < p.total_frames;
            j.base = p.pcm + ((long long)(f * p.H) - p.sample_base);
        } else if (DIRECT) {
            j.base = p.pcm + (size_t)((p.first_frame + hop) * p.H) * p.stride_floats + 2 * pair;
        } else {
            j.base = p.pcm + (size_t)pair * p.plane_floats + 2 * ((long long)((p.first_frame + hop) * p.H) - p.sample_base);
        }
        return j;
    };
    const int second_off = MONO ? (int)(p.H * 4) : 0;   // mono: the pair's second frame starts H samples on
    auto prefetch = [&](const JobIn &j, int a_lo = 0, int a_hi = 16) {
        const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(j.base);
        const int sec = j.data_second ? second_off : 0;
        const int lane_bytes = DIRECT ? (int)(4u * p.stride_floats) * tid : 8 * tid;                  // one sample of the pair per lane
        const int row_bytes = DIRECT ? (int)(2048u * p.stride_floats) : 4096;                        // 512 samples on
#pragma unroll
        for (int a = a_lo; a < a_hi; ++a) {
#ifdef W_ABL_NOLOAD
            pl[a] = (float)(a + 1) * 1e-3f; pr[a] = (float)tid * 1e-3f;
            (void)rs; (void)sec; (void)lane_bytes; (void)row_bytes;
            continue;
#endif
            if (MONO) {
                pl[a] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, 4 * tid, 2048 * a, 0));
                pr[a] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, 4 * tid, 2048 * a + sec, 0));
            } else {
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_bytes, row_bytes * a, 0);
                pl[a] = __uint_as_float(v.x); pr[a] = __uint_as_float(v.y);
            }
        }
    };
    // Where the vector-memory instructions sit decides this kernel (profiles/r06_k16.txt section 3: with two waves per SIMD a BURST of
    // them stalls the issuing wave at the memory pipeline's queue and nothing else is there to run -- same device, per 400 000 transforms:
    // all 16 requests and all 16 stores at the top of the iteration 13.6 ms; spread as below 10.4).  kSpread, every stream but mono frame
    // pairs: the FFT codelets call back behind every butterfly (`hook`) and for every finished bin (`done`), and
    //   * ONE sample request sits behind each of the 16 butterflies of pass 1's two FFT16,
    //   * ONE store of the previous transform's row behind every second butterfly of pass 2's FFT32,
    //   * every LDS write of pass 1 and pass 2 is issued the moment its value is final, inside the arithmetic,
    //   * the pass-2 twiddles are requested eight at a time, one sub-block of the FFT32 ahead of their use.
    // Mono frame pairs (MONO: twice the requests and stores, an opt-in mode) keep them in groups: spread, that instantiation spills.
    constexpr bool kSpread = !MONO;
    // The prefetched values are consumed (Hann, fft.rs:53-63) at the END of the iteration that requested them, behind its stores, in
    // straight-line code, and pinned there (consumed at the loop head, where the entry path is merged in, the compiler's wait becomes vmcnt(0):
    // every store of the previous transform acknowledged by memory, once per transform -- 0.93 against 0.61 ms per 20 000 transforms on round 3's kernel)
    float er[16], ei[16];
    auto take = [&](bool data_second) {
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            er[a] = pl[a] * win[a];
            ei[a] = (MONO && !data_second) ? 0.0f : pr[a] * win[a];
        }
#pragma unroll
        for (int a = 0; a < 16; ++a) asm volatile("" : "+v"(er[a]), "+v"(ei[a]));
    };
    // Job order.  Jobs are hop-major (the pairs of one hop position, then the next hop) and consecutive hop positions share 15/16 of their
    // samples.  Workgroup i runs on XCD i % 8 (MI355X in SPX mode: kXcdHint), each XCD has its own L2: every XCD takes one contiguous eighth of
    // the jobs and deals it round-robin to its workgroups, so that at any time the 32 CUs of an XCD work on ~8 neighbouring hop positions and a
    // sample is fetched from the fabric once.  A LOCALITY hint only: on a part with another XCD count every job is still done exactly once.
    // SLIDE: workgroup number w (consecutive inside an XCD) takes pair w % pairs of run w / pairs -- the pairs of a run read the same
    // lines at about the same time from the same L2; a job is a hop position of that run.
    const unsigned long long nx = p.xcds, xcd = blockIdx.x % nx, local = blockIdx.x / nx;
    const unsigned long long wg = xcd * (gridDim.x / nx) + local;
    const unsigned long long job_step = SLIDE ? 1 : gridDim.x / nx;
    const unsigned long long job_begin = SLIDE ? (wg / p.pairs) * p.run_len : xcd * p.jobs_per_xcd + local;
    const unsigned long long job_end = SLIDE ? (job_begin + p.run_len < p.n_frames ? job_begin + p.run_len : p.n_frames)
                                             : ((xcd + 1) * p.jobs_per_xcd < p.n_jobs ? (xcd + 1) * p.jobs_per_xcd : p.n_jobs);
    unsigned long long hop_c = MONO ? 0 : SLIDE ? job_begin : job_begin / p.pairs;       // (hop, pair) of the current job
    uint32_t pair_c = MONO ? 0 : SLIDE ? (uint32_t)(wg % p.pairs) : (uint32_t)(job_begin - hop_c * p.pairs);
    const unsigned long long step_hops = MONO ? 0 : SLIDE ? 1 : job_step / p.pairs;      // ... and of one step of the loop
    const uint32_t step_pairs = (MONO || SLIDE) ? 0 : (uint32_t)(job_step - step_hops * p.pairs);
    if (job_begin < job_end) {
        const JobIn first = job_in(job_begin, hop_c, pair_c);
        prefetch(first);
        take(first.data_second);
    }
    // The resident constants are waited for HERE (empty statements that read them), so that the loop header carries no pending load of
    // the entry path: merged with the back edge, a twiddle still pending at the entry became vmcnt(33) .. vmcnt(4) at its first use
    // inside the loop -- in every iteration, where the only vector-memory operations in flight are the previous transform's row
    // stores: every transform waited for them to be acknowledged (read off the ISA; stft4096_wg.hip has the same pins)
#pragma unroll
    for (int l = 1; l < 4; ++l) asm volatile("" ::"v"(twl[l].x), "v"(twl[l].y));
#pragma unroll
    for (int h = 1; h < 8; ++h) asm volatile("" ::"v"(twh[h].x), "v"(twh[h].y));
#pragma unroll
    for (int a = 0; a < 16; ++a) asm volatile("" ::"v"(win[a]));

    // roles that do not change over the loop
    const int q1 = tid & 31, hi5 = tid >> 5;                 // pass 2: (q1, c0 = hi5); pass 3: (q1, q2A = hi5)
    const int q1B = (32 - q1) & 31;
    const int q2B = tid == 0 ? 16 : (q1 ? 31 - hi5 : 32 - hi5);
    const int uA = q1 + 32 * hi5, uB = q1B + 32 * q2B;      // thread 0: 0 and 512; else uB = 1024 - uA
    // lane offsets of the kept bins in their row (bin k at byte 8 (k - 1))
    const int voffA = tid == 0 ? 8184 : 8 * (uA - 1);                // thread 0: its bin 1024 (j + 1) in the store of j (see the split)
    const int voffA7 = tid == 0 ? (int)0x7ffffffc : 8 * (uA - 1);    // ... and nothing in the store of j = 7 (out of range: dropped)
    const int voffB = 8 * (uB - 1);

    // The finished row of a transform waits in 32 registers and is stored by the NEXT iteration (kSpread: through its pass 2; mono pairs:
    // four stores at a time between the stages of its pass 1), BEHIND its sample requests: issued at the end of the transform, all eight waves' 128 store instructions
    // (64 KB through a store path of ~34 B / clock / CU) went out at once and the next requests queued behind them -- same device,
    // loads and stores each alone +0.1 / +1.2 ms per 400 000 transforms, together +4.3.  An absent row
    // (nothing pending yet; the missing frame of a mono pair) is a descriptor of zero records: its stores are dropped.
    float pm[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) pm[i] = 0.0f;
    __amdgpu_buffer_rsrc_t pend0 = uniform_rsrc(p.mags, 0u), pend1 = uniform_rsrc(p.mags, 0u);
    auto store_bin = [&](float ml, float mr, const __amdgpu_buffer_rsrc_t &r0, const __amdgpu_buffer_rsrc_t &r1, int voff, int soff) {
        if (!W_STORE_OK(ml)) return;
        if (MONO) {   // (ml, mr) = the bin of frames 2j and 2j + 1: each row holds (s, s) pairs (audio_input_list_model.rs:67-69)
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(ml), __float_as_uint(ml)}, r0, voff, soff, W_OUT_AUX);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(mr), __float_as_uint(mr)}, r1, voff, soff, W_OUT_AUX);
        } else {
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(ml), __float_as_uint(mr)}, r0, voff, soff, W_OUT_AUX);
        }
    };
    auto flush_group = [&](int g) {   // bins 4 g .. 4 g + 3 of the pending row: i = 2 q3 + side
#pragma unroll
        for (int i = 4 * g; i < 4 * g + 4; ++i) store_bin(pm[2 * i], pm[2 * i + 1], pend0, pend1, (i & 1) ? voffB : (i == 14 ? voffA7 : voffA), 8192 * (i >> 1));
        __builtin_amdgcn_sched_barrier(0);   // (the groups stay where they are put)
    };
    // ---- pass 3 in two halves: its 32 reads (p3_read) and its arithmetic (p3_finish: two FFT16 over c0 for the columns u_A and u_B = 1024 - u_A,
    // then the split).  kDefer3 (every stream but mono frame pairs): the arithmetic of transform t - 1 runs in iteration t, BEHIND the requests
    // for pass 2's 32 words -- right behind barrier B1 all eight waves ask for their words at once, the LDS delivers them over ~1 000 clocks and
    // nothing else was there to issue; pass 3's words (64 registers) wait through pass 1 instead of the finished row (32).
    constexpr bool kDefer3 = kSpread;
    float ar[16], ai[16], br[16], bi[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ar[i] = ai[i] = br[i] = bi[i] = 0.0f;
    __amdgpu_buffer_rsrc_t row0 = uniform_rsrc(p.mags, 0u), row1 = uniform_rsrc(p.mags, 0u);   // the rows of the transform whose pass-3 words are in ar .. bi
    auto p3_read = [&]() {
        lds_cfloat2 *rA = lds_ptr(buf + q1 * kS + 16 * hi5);
        lds_cfloat2 *rB = lds_ptr(buf + q1B * kS + 16 * q2B);
#pragma unroll
        for (int c0 = 0; c0 < 16; ++c0) {
            const float2 v = lds_read_alone(rA, c0);
            ar[c0] = v.x; ai[c0] = v.y;
        }
#pragma unroll
        for (int c0 = 0; c0 < 16; ++c0) {
            const float2 v = lds_read_alone(rB, c0);
            br[c0] = v.x; bi[c0] = v.y;
        }
    };
    auto p3_finish = [&]() {
        fft16(ar, ai);
        fft16(br, bi);
        // ---- split + magnitude (fft.rs:81-98): bin k = u + 1024 q3 (q3 < 8) with its partner F[P - k] = register 15 - q3 of the other column
        auto split = [&](float xr_, float xi_, float yr_, float yi_, float &ml, float &mr) {
            const float sr_ = xr_ + yr_, si_ = xi_ - yi_;   // a + conj(b) = 2 L^
            const float dr_ = xr_ - yr_, di_ = xi_ + yi_;   // a - conj(b) = 2i R^
            ml = __builtin_amdgcn_sqrtf(fmaf(sr_, sr_, si_ * si_));  // the scale 1 / W rides on the window
            mr = __builtin_amdgcn_sqrtf(fmaf(dr_, dr_, di_ * di_));
        };
        // Thread 0 holds the two self-paired columns: u = 0 (k = 1024 q3 pairs with 1024 (16 - q3), register 16 - q3 of its OWN column; k = 0,
        // DC, is not an output: fft.rs:81) and u = 512 (k = 512 + 1024 q3 pairs with 512 + 1024 (15 - q3), own column again).  Selects on
        // the split's inputs instead of a branch of its own: a divergent branch (250 instructions, 15 stores, one lane) made wave 0 the
        // last at every barrier.  Its u = 0 bins ride one store instruction early (bin q3 + 1 in the slot of q3, lane offset 8184 =
        // 8192 - 8: a lane offset of -8 would be dropped, profiles/r05_bufrange.txt); instruction 7 drops its lane.
        const bool z = tid == 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pa = FFT16_OUT[j], pa1 = FFT16_OUT[j < 7 ? j + 1 : j], pb = FFT16_OUT[15 - j];
            split(z ? ar[pa1] : ar[pa], z ? ai[pa1] : ai[pa], z ? ar[pb] : br[pb], z ? ai[pb] : bi[pb], pm[4 * j], pm[4 * j + 1]);   // k = u_A + 1024 j, partner (1024 - u_A) + 1024 (15 - j)
            split(br[pa], bi[pa], z ? br[pb] : ar[pb], z ? bi[pb] : ai[pb], pm[4 * j + 2], pm[4 * j + 3]);                           // k = u_B + 1024 j, partner u_A + 1024 (15 - j)
        }
        pend0 = row0;
        pend1 = row1;
    };
#if SGX_STAMPS
    unsigned long long st_acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_readcyclecounter(), st_iters = 0;
#endif
    for (unsigned long long job = job_begin; job < job_end; job += job_step) {
#if SGX_STAMPS
        ++st_iters;
#endif
        SGX_STAMP(19)   // loop control
        long long f0, f1;
        bool have_first = true, have_second = true;
        uint32_t pair = 0;
        if (MONO) {
            f0 = (long long)(2 * (p.pair_base + job)) - (long long)p.first_frame;
            f1 = f0 + 1;
            have_first = f0 >= 0;
            have_second = f1 < (long long)p.n_frames;
        } else {
            f0 = (long long)hop_c;
            f1 = f0;
            pair = pair_c;
        }
        const bool more = job + job_step < job_end;
        if (!MONO && more) {   // the next job's (hop, pair)
            pair_c += step_pairs;
            hop_c += step_hops;
            if (pair_c >= p.pairs) { pair_c -= p.pairs; hop_c += 1; }
        }
        const JobIn nxt = job_in(more ? job + job_step : job, hop_c, pair_c);
        // the NEXT transform's samples, a whole iteration ahead of their use (`take`, behind this transform's row stores): in front of
        // this transform's stores (vmcnt retires in issue order) and with ~14 000 cycles to arrive
        if (!kSpread && more) prefetch(nxt);

        lds_barrier();  // B0: every wave's pass-3 reads of the previous transform are complete -- the image may be written again.  (At the
                        // top of the iteration; kDefer3: the reads were the last thing of the previous one.  Moved into pass 1, in front of
                        // its first image write, so that the pre-twiddle and the first FFT16 stage run while the words arrive: +-0.)
        SGX_STAMP(1)
        // ---- pass 1: 32-point DFT over a, inputs a >= 16 are the zero padding: even q1 = FFT16(z), odd q1 = FFT16(z * w_32^a)
        float orr[16], oi[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) { orr[a] = er[a]; oi[a] = ei[a]; }
        float2 *w1 = buf + tid;
        auto tw_of = [&](float2 v, int q) {       // v * w^{q c}, q = l + 4 h
            const int l = q & 3, h = q >> 2;
            if (l) v = cmulf(v, twl[l]);
            if (h) v = cmulf(v, twh[h]);
            return v;
        };
        pretwiddle16_w32(orr, oi);
        if constexpr (kSpread) {
            // (the last iteration requests its own samples again: in bounds, never used -- no branch inside the arithmetic)
            if constexpr (SLIDE) {   // the window moves down one place: sample a of the next hop position is sample a + 1 of this one
#pragma unroll
                for (int a = 0; a < 15; ++a) { pl[a] = pl[a + 1]; pr[a] = pr[a + 1]; }
            }
            fft16h(er, ei, [&](auto k) {
                if constexpr (SLIDE && decltype(k)::value != 0) return;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (SLIDE) prefetch(nxt, 15, 16);   // ... and its last one is the only request
                else prefetch(nxt, decltype(k)::value, decltype(k)::value + 1);
                __builtin_amdgcn_sched_barrier(0);
            }, [&](auto m, float re, float im) {   // even rows q1 = 2 m, written while the second FFT16 is still to come
                lds_put(&w1[(2 * decltype(m)::value) * kS], tw_of(make_float2(re, im), 2 * decltype(m)::value));
            });
            fft16h(orr, oi, [&](auto k) {
                if constexpr (SLIDE) return;
                __builtin_amdgcn_sched_barrier(0);
                prefetch(nxt, 8 + decltype(k)::value, 9 + decltype(k)::value);
                __builtin_amdgcn_sched_barrier(0);
            }, [&](auto m, float re, float im) {
                lds_put(&w1[(2 * decltype(m)::value + 1) * kS], tw_of(make_float2(re, im), 2 * decltype(m)::value + 1));
            });
        } else {
            flush_group(0);
            fft16(er, ei);
            flush_group(1);
            flush_group(2);
            fft16(orr, oi);
            flush_group(3);
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int pos = FFT16_OUT[m];
                w1[(2 * m) * kS] = tw_of(make_float2(er[pos], ei[pos]), 2 * m);
                w1[(2 * m + 1) * kS] = tw_of(make_float2(orr[pos], oi[pos]), 2 * m + 1);
            }
        }
        SGX_STAMP(2)    // pass 1: arithmetic, sample requests, twiddles, image writes
        // the pass-2 twiddles of this thread's c0 (a broadcast read per half wave): read one by one where they are used, each exposed its LDS
        // latency (31 x lgkmcnt(0) per wave and transform).  kSpread: streamed from inside the FFT32, below; else all in front of the barrier
        float2 t2[32];
        lds_cfloat2 *tw2p = lds_ptr(tw2 + 32 * hi5);
        if (!kSpread) {
#pragma unroll
            for (int q2 = 1; q2 < 32; ++q2) t2[q2] = lds_read_alone(tw2p, q2);
        }
        lds_barrier();  // B1: the image is complete
        if constexpr (kDefer3) {
            // (the previous transform's pass-3 words ARE there -- B0 and B1 waited with lgkmcnt(0) -- but the compiler does not read the
            // statement: told here, it puts its own wait in front of pass 2's requests, where it costs nothing; left alone it waited at the
            // first use, behind them, with lgkmcnt(14): for 18 of the 32 words just requested)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(ar[i]), "v"(ai[i]), "v"(br[i]), "v"(bi[i]));
        }
        SGX_STAMP(3)

        // ---- pass 2: thread (q1, c0): FFT32 over c1 -> q2, twiddle w_512^{q2 c0}, back into its own 32 slots
        {
            float xr[32], xi[32];
            lds_cfloat2 *r2 = lds_ptr(buf + q1 * kS + hi5);
#pragma unroll
            for (int c1 = 0; c1 < 32; ++c1) {
                const float2 v = lds_read_alone(r2, 16 * c1);
                xr[c1] = v.x; xi[c1] = v.y;
            }
            float2 *w2 = buf + q1 * kS + hi5;
            if constexpr (kDefer3) {
                __builtin_amdgcn_sched_barrier(0);
                p3_finish();   // the previous transform's (the first iteration: zeros, and a row descriptor of zero records)
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (kSpread) {
                // call-back points of the FFT32 (4 x 4 x 2, depth first): 0 .. 7 its first stage, then per sub-block s (bins q2 = s mod 4)
                // 8 + 6 s, 9 + 6 s (second stage) and 10 + 6 s .. 13 + 6 s (third stage: two bins final behind each)
                fft32h(xr, xi, [&](auto k) {
                    constexpr int kk = decltype(k)::value;
                    if constexpr (kk == 2 || kk == 8 || kk == 14 || kk == 20) {   // the twiddles of sub-block (kk - 2) / 6, one sub-block ahead
                        constexpr int sb = (kk - 2) / 6;
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (sb + 4 * j > 0) t2[sb + 4 * j] = lds_read_alone(tw2p, sb + 4 * j);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr ((kk & 1) == 0) {
                        constexpr int i = kk >> 1;       // bin 2 q3 + side of the pending row
                        __builtin_amdgcn_sched_barrier(0);
                        store_bin(pm[2 * i], pm[2 * i + 1], pend0, pend1, (i & 1) ? voffB : (i == 14 ? voffA7 : voffA), 8192 * (i >> 1));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }, [&](auto q2, float re, float im) {   // back into the thread's own slot the moment bin q2 is final
                    constexpr int qq = decltype(q2)::value;
                    lds_put(&w2[16 * qq], qq == 0 ? make_float2(re, im) : cmulf(make_float2(re, im), t2[qq]));
                });
            } else {
                fft32(xr, xi);
#pragma unroll
                for (int q2 = 0; q2 < 32; ++q2) {
                    const int pos = FFT32_OUT[q2];
                    const float2 v = make_float2(xr[pos], xi[pos]);
                    w2[16 * q2] = q2 == 0 ? v : cmulf(v, t2[q2]);
                }
            }
        }
        SGX_STAMP(5)    // pass 2: image reads, FFT32, row stores, twiddles, writes in place
        lds_barrier();  // B2: every thread's slots hold pass-2 results
        SGX_STAMP(6)

        // ---- pass 3 (the lambdas in front of the loop): this transform's words; kDefer3: their arithmetic in the next iteration
        row0 = uniform_rsrc(p.mags + (((size_t)(have_first ? f0 : 0) * p.pairs + pair) * (size_t)kM) * 2, (MONO && !have_first) ? 0u : 0x7fffffffu);
        row1 = uniform_rsrc(p.mags + (((size_t)f1 * p.pairs + pair) * (size_t)kM) * 2, (MONO && !have_second) ? 0u : 0x7fffffffu);
        p3_read();
        if constexpr (!kDefer3) p3_finish();
        SGX_STAMP(8)    // pass 3
        SGX_STAMP(9)    // split
        if (more) take(nxt.data_second);
        SGX_STAMP(10)   // wait for the next samples + Hann
    }
    if constexpr (kDefer3) {
        if (job_begin < job_end) p3_finish();       // the last transform's pass 3
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) flush_group(g);   // the last transform's row
#if SGX_STAMPS
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 20; ++i) atomicAdd(&g_phase_cycles16w[i], st_acc[i]);
        atomicAdd(&g_phase_cycles16w[20], st_iters);
    }
#endif
}

struct TablesW {
    float2 *d_T1 = nullptr, *d_tw2 = nullptr;
    float *d_win16 = nullptr;
    float *d_planes = nullptr;   // the (s, s) plane of a mono stream whose frames are not paired, grown on demand
    size_t planes_floats = 0;
};

}  // namespace w16k

#if SGX_STAMPS
extern "C" __attribute__((visibility("default"))) int sgx_debug_phase_cycles16w(unsigned long long *h_out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(h_out, HIP_SYMBOL(w16k::g_phase_cycles16w), sizeof(unsigned long long) * 24);
    if (e == hipSuccess && reset) {
        unsigned long long zero[24] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(w16k::g_phase_cycles16w), zero, sizeof(zero));
    }
    return e == hipSuccess ? 0 : -1;
}
#endif

bool w16384_supported(const sgx_ctx *c)
{
    // (l, r) pairs are moved as 8-byte words: the stream must be mono or have an even channel count
    return c->W == w16k::kW && (c->C == 1 || (c->C & 1) == 0) && c->lds_optin >= w16k::kLdsBytes;
}

hipError_t w16384_init(sgx_ctx *c, void **out)
{
    using namespace w16k;
    auto *t = new TablesW();
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
    std::vector<float2> T1(32 * 512), tw2(512);
    std::vector<float> win16((size_t)kW);
    for (int q = 0; q < 32; ++q)
        for (int cc = 0; cc < 512; ++cc) T1[q * 512 + cc] = unit((unsigned long long)q * cc, kP);
    for (int c0 = 0; c0 < 16; ++c0)
        for (int q2 = 0; q2 < 32; ++q2) tw2[c0 * 32 + q2] = unit((unsigned long long)q2 * c0, 512);
    // the scale (hypot / 2) * (2 / W) = 1 / W is a power of two and commutes with every rounding
    for (int a = 0; a < 16; ++a)
        for (int cc = 0; cc < 512; ++cc) win16[((size_t)(a >> 2) * 512 + cc) * 4 + (a & 3)] = c->tab.window[cc + 512 * a] * (1.0f / (float)kW);
    auto up = [](float2 **dst, const std::vector<float2> &v) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), v.size() * sizeof(float2));
        if (e == hipSuccess) e = hipMemcpy(*dst, v.data(), v.size() * sizeof(float2), hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(&t->d_T1, T1);
    if (e == hipSuccess) e = up(&t->d_tw2, tw2);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&t->d_win16), win16.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(t->d_win16, win16.data(), win16.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft16384_w_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft16384_w_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft16384_w_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft16384_w_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft16384_w_kernel<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e != hipSuccess) {
        w16384_destroy(t);
        return e;
    }
    *out = t;
    return hipSuccess;
}

void w16384_destroy(void *tables)
{
    auto *t = static_cast<w16k::TablesW *>(tables);
    if (!t) return;
    if (t->d_T1) (void)hipFree(t->d_T1);
    if (t->d_tw2) (void)hipFree(t->d_tw2);
    if (t->d_win16) (void)hipFree(t->d_win16);
    if (t->d_planes) (void)hipFree(t->d_planes);
    delete t;
}

hipError_t launch_stft_w16384(const sgx_ctx *c, void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                              size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags)
{
    using namespace w16k;
    if (n_frames == 0) return hipSuccess;
    auto *t = static_cast<TablesW *>(tables);
    Params p{};
    p.T1 = t->d_T1;
    p.tw2 = t->d_tw2;
    p.win16 = t->d_win16;
    p.mags = d_mags;
    p.first_frame = first_frame;
    p.n_frames = n_frames;
    p.total_frames = total_frames;
    p.H = c->H;
    p.pairs = pairs;
    const bool mono = channels == 1 && (c->cfg.flags & SGX_FLAG_PAIRED_FRAMES);
    // a mono stream whose frames are not paired (the default): every frame the (s, s) transform of the reference
    // (audio_input_list_model.rs:67-69) -- the sample range duplicated into one (s, s) plane, then the two-channel kernel
    const bool dup = channels == 1 && !mono;
    p.pair_base = mono ? first_frame / 2 : 0;
    p.n_jobs = mono ? (first_frame + n_frames + 1) / 2 - first_frame / 2 : (unsigned long long)n_frames * pairs;
    const bool direct = channels > 2;     // every pair read where it lies: no workspace, no second kernel
    // lane offsets are 32-bit: 4 C (511 + 512 * 15) + 8 bytes must stay below the descriptor's 2^31 - 1 records (sgx_create caps C)
    if (direct && (unsigned long long)channels * 32764ull + 8ull >= 0x7fffffffull) return hipErrorInvalidValue;
    if (dup) {
        const size_t first_sample = first_frame * (size_t)c->H;
        const size_t n_samp = (n_frames - 1) * (size_t)c->H + kW;
        const size_t plane = (2 * n_samp + 63) & ~(size_t)63;  // floats
        if (plane > t->planes_floats) {
            hipError_t e = hipStreamSynchronize(c->stream);  // a previous launch may still read the old plane
            if (e != hipSuccess) return e;
            if (t->d_planes) { (void)hipFree(t->d_planes); t->d_planes = nullptr; t->planes_floats = 0; }
            e = hipMalloc(reinterpret_cast<void **>(&t->d_planes), plane * sizeof(float));
            if (e != hipSuccess) return e;
            t->planes_floats = plane;
        }
        const unsigned blocks = (unsigned)std::min<size_t>((n_samp + 255) / 256, (size_t)c->n_cu * 16);
        hipLaunchKernelGGL(w16k::duplicate_mono_kernel, dim3(blocks), dim3(256), 0, c->stream, d_pcm, t->d_planes, first_sample, n_samp);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        p.pcm = t->d_planes;
        p.plane_floats = plane;
        p.sample_base = (long long)first_sample;
    } else {
        p.pcm = d_pcm;
        p.plane_floats = 0;
        p.sample_base = 0;
        p.stride_floats = channels;
    }
    // persistent workgroups, one per CU (132 KB of LDS); jobs are dealt round-robin in output-row order
    unsigned long long blocks = (unsigned long long)c->n_cu;
    if (blocks > p.n_jobs) blocks = p.n_jobs;
    p.xcds = (blocks % kXcdHint == 0 && p.n_jobs >= kXcdHint * blocks) ? kXcdHint : 1u;   // (short launches: plain round-robin keeps every workgroup busy)
    const unsigned long long group = mono ? 1 : pairs;                   // a hop position's pairs stay together
    p.jobs_per_xcd = ((p.n_jobs + p.xcds - 1) / p.xcds + group - 1) / group * group;
    // H = 512 (config 4's hop): the sliding window -- runs of hop positions, one pair per workgroup
    const bool slide = !mono && c->H == 512 && pairs <= (unsigned long long)c->n_cu;
    if (slide) {
        unsigned long long runs = (unsigned long long)c->n_cu / pairs;
        if (runs > n_frames) runs = n_frames;
        p.run_len = (n_frames + runs - 1) / runs;
        runs = (n_frames + p.run_len - 1) / p.run_len;
        blocks = runs * pairs;
        p.xcds = blocks % kXcdHint == 0 ? kXcdHint : 1u;
    }
    const dim3 grid((unsigned)blocks), block(512);
    if (mono) hipLaunchKernelGGL((stft16384_w_kernel<true>), grid, block, kLdsBytes, c->stream, p);
    else if (slide && direct) hipLaunchKernelGGL((stft16384_w_kernel<false, true, true>), grid, block, kLdsBytes, c->stream, p);
    else if (slide) hipLaunchKernelGGL((stft16384_w_kernel<false, false, true>), grid, block, kLdsBytes, c->stream, p);
    else if (direct) hipLaunchKernelGGL((stft16384_w_kernel<false, true>), grid, block, kLdsBytes, c->stream, p);
    else hipLaunchKernelGGL((stft16384_w_kernel<false>), grid, block, kLdsBytes, c->stream, p);
    return hipGetLastError();
}

}  // namespace sgx
