// stft4096_wg.hip -- tuned STFT for W = 2048 (P = 4096): one 256-thread workgroup per transform,
// 16 points per thread, three radix-16 passes.
//
// Why this shape (measured on MI355X, tools/microbench.hip): a single wave per SIMD issues one
// VALU instruction every ~7 cycles, four waves per SIMD one every ~2.  64 points per lane (the
// wave-per-transform kernel in stft4096.hip) needs > 128 VGPRs and so caps at 2 waves per SIMD;
// 16 points per thread fits 4 waves per SIMD and keeps 16 waves per CU in flight to cover LDS,
// barrier and memory latency.
//
// Replaces FastFourierTransform::process (fft.rs:43-99) + the hop loop (audio_transform.rs:34-42).
//
//   sample index  n = t + 256 a           (t = thread, a < 8 non-zero rows: padding never touched)
//   pass 1  thread t        : 16-point DFT over a (8 non-zero inputs = two 8-point FFTs), -> q1
//                             twiddle w_4096^{t q1}            (15 per-thread constants in VGPRs)
//   pass 2  thread (q1, t0) : t = t0 + 16 t1; 16-point FFT over t1 -> q2; twiddle w_256^{t0 q2} (LDS)
//   pass 3  thread q1+16 q2 : 16-point FFT over t0 -> q3;  bin k = q1 + 16 q2 + 256 q3
//   split   F[k] and F[P-k] -> |L^[k]|, |R^[k]| (fft.rs:81-89): the partner of thread u is thread
//           (256 - u) % 256, exchanged through LDS (upper 8 registers only; k = 1..2047 is kept)
//
// Mono streams (a mono sample is duplicated into (s, s): audio_input_list_model.rs:67-69) pack
// TWO consecutive frames into one transform: frame 2j in the real part, frame 2j+1 in the
// imaginary part; the same split that separates left from right separates the two frames.
#include <cmath>
#include <type_traits>

#include "stft4096_wg.hpp"

namespace sgx {

namespace wg {

typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float cl_fma(float a, float c, float u) { return fmaf(a, c, u); }
__device__ __forceinline__ f2v cl_fma(f2v a, float c, f2v u) { return __builtin_elementwise_fma(a, f2v{c, c}, u); }

#include "fft_codelets.inc"

__device__ __forceinline__ float2 cmulf(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}

#ifndef SGX_PRIO_A
#define SGX_PRIO_A 1   // (l, r) / (s, s) transforms: wave priority from the end of pass 2 (A) and from the image-2 writes (B) on; 3 from the split on
#define SGX_PRIO_B 2
#endif
#if SGX_STAMPS
// diagnostic build only (tools/k1_phases.py): per-phase wave cycles (s_memtime), summed over all waves and iterations
__device__ unsigned long long g_phase_cycles[20];
#define SGX_STAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[i] += now_ - st_last; st_last = now_; }
#else
#define SGX_STAMP(i)
#endif

// PIX: kPixNone = rows (float or half pairs); else the fused pixel path with that pixel code (stft4096_wg.hpp)
#ifndef SGX_ABL_LDS
#define SGX_ABL_LDS 0
#endif
#ifndef SGX_ADDTID
#define SGX_ADDTID 1   // (0: the (l, r) sliding kernel with the b64 transposes of every other instantiation, for A/B)
#endif
// rows q and q + 1 of the real and the imaginary plane, this wave's 64 words of each: LDS address = M0 + offset + 4 * lane
// (M0 is set inside the statement.  It cannot be DECLARED clobbered: clang answers "inline asm clobber list contains reserved registers: m0 ...
// may not be preserved" (ROCm 7.2) -- so that nothing of the compiler's may depend on M0 around these statements is checked on the ISA of
// every build instead: tools/isa_check_addtid.py fails on any instruction that reads M0 by name OR implicitly (indexed moves, LDS-DMA,
// GWS, s_sendmsg, interpolation) in a kernel that issues add-TID stores)
__device__ __forceinline__ void addtid_rows(float2 a, float2 b, uint32_t m0_wave, int q)
{
#if SGX_ABL_LDS & 1   // timing only (diagnostic builds): the values live, no LDS write
    asm volatile("" ::"v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y));
    return;
#endif
    asm volatile("s_mov_b32 m0, %4\n\t"
                 "s_nop 0\n\t"            // one wait state between a scalar write of M0 and an add-TID LDS instruction (the compiler does not see into the statement)
                 "ds_write_addtid_b32 %0 offset:%5\n\t"
                 "ds_write_addtid_b32 %1 offset:%6\n\t"
                 "ds_write_addtid_b32 %2 offset:%7\n\t"
                 "ds_write_addtid_b32 %3 offset:%8"
                 :
                 : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y), "s"(m0_wave), "i"(1088 * q), "i"(1088 * q + 17408), "i"(1088 * (q + 1)), "i"(1088 * (q + 1) + 17408)
                 : "memory");
}
// the 16 values of this thread's next transform: 16 consecutive words of its row in either plane
__device__ __forceinline__ void read_planes(const float4 *rd4, float (&xr)[16], float (&xi)[16])
{
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#if SGX_ABL_LDS & 2   // timing only: no LDS read
        float4 r = {1.0f, 2.0f, 3.0f, (float)c}, i = {4.0f, 3.0f, 2.0f, 1.0f};
        asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w), "+v"(i.x), "+v"(i.y), "+v"(i.z), "+v"(i.w) : "v"(rd4));
#else
        const float4 r = rd4[c], i = rd4[c + 1088];
#endif
        xr[4 * c] = r.x; xr[4 * c + 1] = r.y; xr[4 * c + 2] = r.z; xr[4 * c + 3] = r.w;
        xi[4 * c] = i.x; xi[4 * c + 1] = i.y; xi[4 * c + 2] = i.z; xi[4 * c + 3] = i.w;
    }
}

template <bool MONO, int PAIRING, bool C2, int PIX>
__global__ void __launch_bounds__(256, 4) stft4096_wg_kernel(Params p)
{
    constexpr bool RENDER = PIX != kPixNone && PIX != kPixRowsF16;
    constexpr bool F16 = PIX == kPixRowsF16;    // rows as (l, r) half pairs (compile-time: the row stores are straight-line code)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *buf = reinterpret_cast<float2 *>(smem_raw);
    float2 *tw2 = buf + kBufComplex;

    uint2 *pal = reinterpret_cast<uint2 *>(tw2 + 256);          // RENDER only: [256] {threshold, RGBA} (pixel_for)

    const int tid = threadIdx.x;
    constexpr bool kSlide2 = !MONO && C2 && PAIRING == kPairAdjacentRow;   // an (l, r) stream at H = 256: sliding register window
    // TR: the two LDS transposes as real / imaginary PLANES written with ds_write_addtid_b32 (no address register, 2 cycles per wave and
    // dword against 6 per ds_write_b64) and read back as 16-byte pieces.  A plane row is the 256 threads of the writing pass in thread
    // order, 4 words of padding behind every wave (272 words: 16-byte reads of 16 consecutive threads' words are conflict-free within
    // and across the lane groups of ds_read_b128), so the 16 values a thread of the NEXT pass transforms must sit in 16 consecutive
    // threads of THIS one: pass 1 runs column t = (tid >> 4) + 16 (tid & 15) instead of t = tid.  Only the sliding window can afford
    // that: its one 8-byte load per transform is the only one whose lanes then stride 128 bytes.
    constexpr bool TR = kSlide2 && SGX_ADDTID;
    const int t_p1 = TR ? (tid >> 4) + 16 * (tid & 15) : tid;     // pass-1 column of this thread
    float *plane = reinterpret_cast<float *>(smem_raw);           // TR: re [16][272], im [16][272] behind it (34 816 B, the images' bytes)
    const uint32_t m0_wave = __builtin_amdgcn_readfirstlane((uint32_t)(tid >> 6) * 272u);   // TR: byte offset of this wave inside a plane row
    const float4 *rd4 = reinterpret_cast<const float4 *>(plane + 272 * (tid >> 4) + 68 * ((tid & 15) >> 2) + 16 * (tid & 3));   // TR: both read sides
    tw2[tid] = p.tw2[tid];
    uint32_t row_words[4] = {0u, 0u, 0u, 0u};  // RENDER: the table words of this thread's rows tid + 256 i
    if (RENDER) {
        pal[tid] = make_uint2(__float_as_uint(tid < 255 ? p.lut_thr[tid] : __builtin_nanf("")), *reinterpret_cast<const uint32_t *>(&p.lut_rgba[tid]));
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((uint32_t)tid + 256u * i < p.R) row_words[i] = p.rows[tid + 256 * i];
    }

    // per-thread constants, kept in registers for the life of the (persistent) workgroup
    // the output scale (hypot / 2) * (2 / W) = 2^-11 rides on the window: a power of two commutes with every
    // rounding below (products, sums, the square root of a sum of squares), so the bits are the same and the
    // 16 multiplies per thread after the square roots are gone
    const float inv_w = 1.0f / (float)kW;
    static_assert((kW & (kW - 1)) == 0, "the scale must be a power of two to move it");
    float win[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) win[a] = p.window[t_p1 + 256 * a] * inv_w;
    float2 tw1[16];
#pragma unroll
    for (int q = 1; q < 16; ++q) tw1[q] = p.tw1[q * 256 + t_p1];

    const int q1_2 = tid >> 4, t0_2 = tid & 15;                 // pass-2 role
    __syncthreads();

    const unsigned long long job_begin = (unsigned long long)blockIdx.x * p.jobs_per_block;
    unsigned long long job_end = job_begin + p.jobs_per_block;
    if (job_end > p.n_jobs) job_end = p.n_jobs;

    // Software pipeline: the samples of transform j+1 are requested while transform j is still in
    // its FFT passes, i.e. BEFORE j's magnitude stores.  vmcnt retires in issue order, so a load
    // issued after 16-32 stores would have to wait for all of them to reach memory first.
    float sa[(MONO && PAIRING == kPairAdjacentRow) ? 9 : 8], sb[8];
    float ld0 = 0.0f, ld1 = 0.0f;   // sliding window: the two rows requested for the next transform
    bool pending = false;
    uint32_t issued_since = 0;      // vector-memory instructions this wave issued after requesting ld0 / ld1
    auto fetch = [&](unsigned long long job, bool sequential) {
        if (MONO) {
            // frames are paired by their GLOBAL index (2q, 2q+1), so the bytes do not depend on where a
            // range starts: an odd first_frame computes frame first_frame-1 too and simply does not store it
            const unsigned long long f = 2 * (p.pair_base + job);
            const unsigned long long fb = f + 1;
            const bool second = fb < p.total_frames;  // the partner is transformed whenever the stream holds it
            const float *s0 = p.pcm + f * p.H;
            if (PAIRING == kPairAdjacentRow) {
                // H = 256 = one row: frame f+1 row a is frame f row a+1, and the next transform
                // (two frames on) re-uses rows 2..8 of this one: slide the register window and
                // load only the two new rows -- every sample is fetched once per workgroup
                if (kSlideWindow && sequential) {
                    // Only REQUEST the two new rows here.  They land in two pending registers and are rotated
                    // into the window at the top of the next iteration, AFTER this transform's stores have been
                    // issued: the wait for them is then `vmcnt(stores issued since)`, which the older loads
                    // satisfy while the stores are still in flight.  The compiler cannot express that wait (at
                    // the loop header it merges the entry path and falls back to vmcnt(0), i.e. it drains the
                    // store stream once per transform), so the two loads and their wait are written by hand.
                    const float *r7 = s0 + 256 * 7;
                    const float *r8 = second ? r7 + 256 : r7;  // no partner frame: any valid address, the value is unused
                    asm volatile("global_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %4"
                                 : "=&v"(ld0), "=&v"(ld1)
                                 : "v"(tid * 4), "s"(r7), "s"(r8)
                                 : "memory");
                    pending = true;
                } else {
#pragma unroll
                    for (int a = 0; a < 8; ++a) sa[a] = s0[tid + 256 * a];
                    sa[8] = second ? s0[tid + 256 * 8] : 0.0f;
                }
            } else {
                // (a uniform base in a buffer descriptor + one 32-bit lane offset: a per-lane 64-bit pointer kept across the loop is
                // what the fused variants of this path spilled -- and its reload is a vector-memory load waited for with vmcnt(0))
                const __amdgpu_buffer_rsrc_t r0 = pcm_rsrc(s0), r1 = pcm_rsrc(second ? p.pcm + fb * p.H : s0);
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    sa[a] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r0, tid * 4, 1024 * a, 0));
                    sb[a] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, tid * 4, 1024 * a, 0));
                }
            }
        } else {
            const float *s0 = p.pcm + (p.first_frame + job) * p.H * p.C;
            if (C2) {
                const __amdgpu_buffer_rsrc_t r0 = pcm_rsrc(s0);
                if (kSlide2 && sequential) {
                    // H = 256 = one row of (l, r) columns: the next frame's rows 0 .. 6 are this frame's rows 1 .. 7 -- the window slides in
                    // registers and ONE 8-byte load per thread fetches the new row 7 (round 1 tried this at the register cap and lost 8 %;
                    // the kernel has 15 registers to spare now, and what the seven saved loads relieve is the CU's vector-memory path,
                    // which the row stores share: profiles/r05_k1_stereo.txt)
                    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r0, t_p1 * 8, 2048 * 7, 0);
                    ld0 = __uint_as_float(v.x); ld1 = __uint_as_float(v.y);
                    pending = true;
                } else {
#pragma unroll
                    for (int a = 0; a < 8; ++a) {
                        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r0, t_p1 * 8, 2048 * a, 0);
                        sa[a] = __uint_as_float(v.x); sb[a] = __uint_as_float(v.y);
                    }
                }
            } else {
                // one channel, every frame its own (s, s) transform (the default; audio_input_list_model.rs:67-69).
                // (More than two interleaved channels never come here: their pairs are split into planes first and each
                // plane runs the C2 kernel -- launch_wg.)
#pragma unroll
                for (int a = 0; a < 8; ++a) sa[a] = s0[tid + 256 * a];
            }
        }
    };
    if (job_begin < job_end) fetch(job_begin, false);
    {
        // The first window is waited for HERE (an empty asm that reads it), so that the loop header carries no pending load of the entry
        // path.  Merged with the back edge -- where the same registers are followed by the transform's row stores -- a pending entry load
        // made the compiler wait at the top of EVERY iteration with vmcnt(6) .. vmcnt(0): for every row store just issued to be
        // acknowledged by memory, once per transform (round 5: the (l, r) and (s, s) streams, 5.5 -> 4.6 ms per 1e6 frames; the sliding
        // mono window has had this wait since round 1).
        asm volatile("" ::"v"(sa[0]), "v"(sa[1]), "v"(sa[2]), "v"(sa[3]), "v"(sa[4]), "v"(sa[5]), "v"(sa[6]), "v"(sa[7]),
                     "v"(sa[(MONO && PAIRING == kPairAdjacentRow) ? 8 : 7]));
        if ((MONO && PAIRING != kPairAdjacentRow) || (!MONO && C2))
            asm volatile("" ::"v"(sb[0]), "v"(sb[1]), "v"(sb[2]), "v"(sb[3]), "v"(sb[4]), "v"(sb[5]), "v"(sb[6]), "v"(sb[7]));
        // (the resident constants too: a pass-1 twiddle still pending at the loop entry became a vmcnt(14) .. vmcnt(0) at its first use
        // inside the loop -- in every iteration, where the only vector-memory operations in flight are the previous transform's stores)
#pragma unroll
        for (int q = 1; q < 16; ++q) asm volatile("" ::"v"(tw1[q].x), "v"(tw1[q].y));
#pragma unroll
        for (int a = 0; a < 8; ++a) asm volatile("" ::"v"(win[a]));
    }

    // Wave priorities.  The four waves of a SIMD belong to four workgroups in four different phases; left to the
    // default arbitration they share the VALU evenly, so every transform reaches its stores as late as possible.  The
    // waves of a transform that has finished its third pass are raised to priority 3 and keep it through the split,
    // the stores and the next transform's first pass (its row loads are already in flight); they drop to 0 once that
    // pass is in LDS.  A transform that is nearly done is finished first, its stores are issued earlier and the store
    // stream overlaps the other workgroups' arithmetic better: +6 % (mono), and with the intermediate steps 1 and 2 for
    // the second pass +12 % for stereo input (same-device A/B; the steps cost mono 1 %).  The fused pixel path holds
    // priority 1 through its sample pass and 3 from its row pass on: +12 % over no priorities.
    // The next transform's samples, requested in front of the stores, are waited for behind them, in straight-line code: that is
    // vmcnt(stores issued since), which the older loads satisfy while the stores are still in flight.  Left pending across the back
    // edge the compiler merged them with the entry path and waited at the top of every iteration with vmcnt(5) .. vmcnt(0): for every
    // row store of the transform to be acknowledged by memory.  (The sliding mono window does the same by hand, below.)
    auto next_samples_are_here = [&]() {
        if (kSlide2) {
            asm volatile("" : "+v"(ld0), "+v"(ld1));
        } else {
            asm volatile("" : "+v"(sa[0]), "+v"(sa[1]), "+v"(sa[2]), "+v"(sa[3]), "+v"(sa[4]), "+v"(sa[5]), "+v"(sa[6]), "+v"(sa[7]));
            if ((MONO && PAIRING != kPairAdjacentRow) || (!MONO && C2))
                asm volatile("" : "+v"(sb[0]), "+v"(sb[1]), "+v"(sb[2]), "+v"(sb[3]), "+v"(sb[4]), "+v"(sb[5]), "+v"(sb[6]), "+v"(sb[7]));
        }
    };
#if SGX_STAMPS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_readcyclecounter();
#endif
    for (unsigned long long job = job_begin; job < job_end; ++job) {
        SGX_STAMP(15)   // back edge: loop control
        if (MONO && PAIRING == kPairAdjacentRow && kSlideWindow && pending) {
            // this transform = the previous one moved on by two rows.  vmcnt counts in issue order: the two row
            // loads are complete once no more than `issued_since` younger instructions are outstanding.
            // ONE asm statement chooses among the three waits with a scalar branch of its own.  As three statements in
            // three C++ branches (round 1) the compiler merged their results in a phi and placed the copies
            // `v_mov v0, v97` in FRONT of the waits of the two short paths: a read of a register whose load was still
            // pending (gfx9 has no interlock for it).  tools/isa_check_prefetch.py (run by the Makefile) fails the build
            // if anything touches the two registers between request and wait again.
            asm volatile("s_cmp_ge_u32 %2, 16\n\t"
                         "s_cbranch_scc1 4f\n\t"
                         "s_cmp_ge_u32 %2, 14\n\t"
                         "s_cbranch_scc1 1f\n\t"
                         "s_cmp_ge_u32 %2, 7\n\t"
                         "s_cbranch_scc1 2f\n\t"
                         "s_waitcnt vmcnt(0)\n\t"
                         "s_branch 3f\n"
                         "2:\n\t"
                         "s_waitcnt vmcnt(7)\n\t"
                         "s_branch 3f\n"
                         "1:\n\t"
                         "s_waitcnt vmcnt(14)\n\t"
                         "s_branch 3f\n"
                         "4:\n\t"
                         "s_waitcnt vmcnt(16)\n"
                         "3:"
                         : "+v"(ld0), "+v"(ld1)
                         : "s"(__builtin_amdgcn_readfirstlane(issued_since))
                         : "scc");
#pragma unroll
            for (int a = 0; a < 7; ++a) sa[a] = sa[a + 2];
            sa[7] = ld0;
            sa[(MONO && PAIRING == kPairAdjacentRow) ? 8 : 7] = ld1;
        }
        // ---- Hann (fft.rs:53-63) on the prefetched samples
        float er[8], ei[8];
        // local (output) frame indices; for mono f0 may be -1 (the pair's first frame precedes the range)
        const long long f0 = MONO ? (long long)(2 * (p.pair_base + job)) - (long long)p.first_frame : (long long)job;
        const long long f1 = f0 + 1;
        const bool have_first = !MONO || f0 >= 0;
        const bool data_second = !MONO || (unsigned long long)(f1 + (long long)p.first_frame) < p.total_frames;
        const bool have_second = !MONO || f1 < (long long)p.n_frames;  // ... but stored only inside the requested range
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            er[a] = sa[a] * win[a];
            if (MONO && PAIRING == kPairAdjacentRow) ei[a] = data_second ? sa[a + 1] * win[a] : 0.0f;
            else if (!MONO && !C2) ei[a] = er[a];   // (s, s)
            else ei[a] = data_second ? sb[a] * win[a] : 0.0f;
        }
        const int col = tid;                           // pass-3 / output column of this thread
        const int pcol = col == 0 ? 256 : 256 - col;   // partner column (column 0 is its own partner, one row up)

        // ---- pass 1: 16-point DFT over a, inputs a >= 8 are the zero padding:
        //      even q1 = FFT8(z), odd q1 = FFT8(z * w_16^a)
        float orr[8], oi[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) { orr[a] = er[a]; oi[a] = ei[a]; }
        pretwiddle8_w16(orr, oi);
        fft8(er, ei);
        fft8(orr, oi);
        SGX_STAMP(0)    // prefetch wait + Hann + pass-1 arithmetic

        lds_barrier();  // the previous transform's partner reads are complete
        SGX_STAMP(1)    // barrier 0
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pos = FFT8_OUT[j];
            const float2 ve = make_float2(er[pos], ei[pos]);
            const float2 vo = make_float2(orr[pos], oi[pos]);
            if (TR) {
                const float2 a = j == 0 ? ve : cmulf(ve, tw1[2 * j]), b = cmulf(vo, tw1[2 * j + 1]);
                addtid_rows(a, b, m0_wave, 2 * j);
            } else {
                buf[(2 * j) * kS1 + tid] = j == 0 ? ve : cmulf(ve, tw1[2 * j]);
                buf[(2 * j + 1) * kS1 + tid] = cmulf(vo, tw1[2 * j + 1]);
            }
        }
        __builtin_amdgcn_s_setprio(0);  // (wave priorities: see the note at the head of the loop)
        SGX_STAMP(2)    // twiddles + image-1 writes (to completion: the stamp drains lgkmcnt)
        lds_barrier();
        SGX_STAMP(3)    // barrier 1

        // ---- pass 2: thread (q1, t0): 16-point FFT over t1, then twiddle w_256^{t0 q2}
        float xr[16], xi[16];
        if (TR) {
            read_planes(rd4, xr, xi);
        } else {
#pragma unroll
            for (int t1 = 0; t1 < 16; ++t1) {
                const float2 v = buf[q1_2 * kS1 + t0_2 + 16 * t1];
                xr[t1] = v.x; xi[t1] = v.y;
            }
        }
        fft16(xr, xi);
        if (!MONO) __builtin_amdgcn_s_setprio(RENDER ? 2 : SGX_PRIO_A);   // (fused (l, r) pixels: 2 / 2 measured 3 % ahead of 1 / 2, rows the other way round)
        SGX_STAMP(4)    // image-1 reads + FFT16
        lds_barrier();  // everyone has read image 1
        SGX_STAMP(5)    // barrier 2
        if (TR) {
            lds_cfloat2 *tw2p = lds_ptr(tw2 + t0_2);
#pragma unroll
            for (int q2 = 0; q2 < 16; q2 += 2) {
                const int pa = FFT16_OUT[q2], pb2 = FFT16_OUT[q2 + 1];
                const float2 va = make_float2(xr[pa], xi[pa]), vb = make_float2(xr[pb2], xi[pb2]);
                addtid_rows(q2 == 0 ? va : cmulf(va, lds_read_alone(tw2p, q2 * 16)), cmulf(vb, lds_read_alone(tw2p, (q2 + 1) * 16)), m0_wave, q2);
            }
        } else {
#pragma unroll
            for (int q2 = 0; q2 < 16; ++q2) {
                const int pos = FFT16_OUT[q2];
                const float2 v = make_float2(xr[pos], xi[pos]);
                buf[t0_2 * kS2 + q1_2 + 16 * q2] = q2 == 0 ? v : cmulf(v, tw2[q2 * 16 + t0_2]);
            }
        }
        if (!MONO) __builtin_amdgcn_s_setprio(SGX_PRIO_B);
        SGX_STAMP(6)    // twiddles (LDS reads) + image-2 writes
        lds_barrier();
        SGX_STAMP(7)    // barrier 3

        // ---- pass 3: thread u = q1 + 16 q2: 16-point FFT over t0 -> bins k = u + 256 q3
        if (TR) {
            read_planes(rd4, xr, xi);
        } else {
#pragma unroll
            for (int t0 = 0; t0 < 16; ++t0) {
                const float2 v = buf[t0 * kS2 + col];
                xr[t0] = v.x; xi[t0] = v.y;
            }
        }
        fft16(xr, xi);
        if (job + 1 < job_end) fetch(job + 1, true);  // ahead of this transform's stores (see above)
        if (RENDER) __builtin_amdgcn_s_setprio(1);  // the pixel passes are long: 3 only from the row pass (the pixel stores) on
        else __builtin_amdgcn_s_setprio(3);
        SGX_STAMP(8)    // image-2 reads + FFT16 + the next transform's sample requests
        lds_barrier();  // everyone has read image 2
        SGX_STAMP(9)    // barrier 4
        // partner exchange: publish q3 = 8..15 (the bins P-k of the kept half)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pos = FFT16_OUT[8 + j];
#if SGX_ABL_LDS & 1
            asm volatile("" ::"v"(xr[pos]), "v"(xi[pos]));
#else
            buf[j * 256 + col] = make_float2(xr[pos], xi[pos]);
#endif
        }
        SGX_STAMP(10)   // partner writes
        lds_barrier();
        SGX_STAMP(11)   // barrier 5

        // ---- split + magnitude (fft.rs:81-98)
        float ml[8], mr[8];
        // (l, r) rows: every segment is stored as soon as it is computed -- eight stores spread over the split instead of one burst of eight
        // behind it (a wave stalls at the issue of a store while the CU's vector-memory path drains the previous ones: 1 083 cycles per
        // transform for the burst, profiles/r05_k1_stereo.txt; same device 5.10 -> 4.87 ms per 1e6 frames together with the sliding window)
        constexpr bool kInterleave = !MONO && !RENDER;
        float2 pb[8] = {};
        if (kInterleave) {
#pragma unroll
            for (int q3 = 0; q3 < 8; ++q3) {
#if SGX_ABL_LDS & 2
                pb[q3] = make_float2((float)q3, 1.0f);
                asm volatile("" : "+v"(pb[q3].x), "+v"(pb[q3].y) : "v"(pcol));
#else
                pb[q3] = buf[(7 - q3) * 256 + pcol];
#endif
            }
        }
#if SGX_ABL_STORES == 1   // timing only: every row lands in the first 64 rows
        const long long f0_il = f0 & 63;
#else
        const long long f0_il = f0;
#endif
        const __amdgpu_buffer_rsrc_t r_il = row_rsrc(reinterpret_cast<char *>(p.mags), (long long)(((size_t)f0_il * p.pairs + p.pair) * ((size_t)kM * (F16 ? 4 : 8))) - (F16 ? 4 : 8));
#pragma unroll
        for (int q3 = 0; q3 < 8; ++q3) {
            const int pos = FFT16_OUT[q3];
            // F[P-k]: thread 256-u holds it as q3' = 15 - q3 (row 7 - q3); thread 0 as q3' = 16 - q3
            const float2 b = kInterleave ? pb[q3] : buf[(7 - q3) * 256 + pcol];
            const float ar = xr[pos], ai = xi[pos];
            const float pr = ar + b.x, pi = ai - b.y;   // a + conj(b) = 2 L^
            const float qr = ar - b.x, qi = ai + b.y;   // a - conj(b) = 2i R^
            ml[q3] = __builtin_amdgcn_sqrtf(fmaf(pr, pr, pi * pi));  // already scaled by 1 / W (see `win`)
            mr[q3] = __builtin_amdgcn_sqrtf(fmaf(qr, qr, qi * qi));
            if (kInterleave) {
                // k = 0 (DC) is not part of the output (fft.rs:81): thread 0's first store carries a lane offset beyond the descriptor's records
                if (F16) {
                    const __half2 h = __floats2half2_rn(ml[q3], mr[q3]);
                    __builtin_amdgcn_raw_buffer_store_b32(*reinterpret_cast<const uint32_t *>(&h), r_il, ((q3 == 0 && col == 0) ? (int)0x80000000 : col * 4) + 1024 * (q3 & 3), 4096 * (q3 >> 2), 0);
                } else {
#if SGX_ABL_STORES == 3     // timing only: every value computed and live, (practically) no store executed
                    if (ml[q3] == 12345.678f)
#endif
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(ml[q3]), __float_as_uint(mr[q3])}, r_il, ((q3 == 0 && col == 0) ? (int)0x80000000 : col * 8) + 2048 * (q3 & 1), 4096 * (q3 >> 1), kAuxNt);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        SGX_STAMP(12)   // partner reads + split arithmetic ((l, r) rows: and the row stores, one per segment)
        // every row is 8 store instructions in every wave, whether the row is stored or not (an absent row's descriptor holds zero
        // records, the lane of bin 0 carries an offset beyond the records); the fused pixel path issues table loads and pixel stores
        // of its own
        // (the fused pixel path: every wave issues at least n_samples / 256 table loads in the sample pass and R / 256 pixel stores per
        // stored column in the row pass AFTER the two row loads were requested -- a lower bound is all the wait needs; with "unknown = 0"
        // the head of the next transform waited for the pixel stores just issued to be acknowledged by memory)
        issued_since = RENDER ? (p.n_samples >> 8) + (p.R >> 8) * ((have_first ? 1u : 0u) + ((MONO && have_second) ? 1u : 0u))
                              : (MONO ? 16u : 8u);    // rows: eight store instructions per row in every wave, absent rows included (dropped by their descriptor)
        if (!RENDER) {
            // ---- store [F][pairs][M][2]: uniform row base (SGPR) + one 32-bit lane offset
            if (F16) {
                char *base = reinterpret_cast<char *>(p.mags);
                const long long row0 = (long long)(((size_t)(have_first ? f0 : 0) * p.pairs + p.pair) * (size_t)kM * 4) - 4;
                if (MONO) {
                    store_row_f16<true>(base, row0, col, ml, ml, have_first);
                    store_row_f16<true>(base, (long long)((f1 * p.pairs + p.pair) * (size_t)kM * 4) - 4, col, mr, mr, have_second);
                }   // ((l, r) rows: stored inside the split)
            } else {
                // byte offset of bin k = 0 of the row (bin k lives 8 k bytes on; k = 0 is never stored)
                char *base = reinterpret_cast<char *>(p.mags);
                constexpr size_t kPitch = (size_t)kM * 8;
                const long long row0 = (long long)(((size_t)(have_first ? f0 : 0) * p.pairs + p.pair) * kPitch) - 8;
                if (MONO) {
                    store_row<true>(base, row0, col, ml, ml, have_first);
                    store_row<true>(base, (long long)((f1 * p.pairs + p.pair) * kPitch) - 8, col, mr, mr, have_second);
                }   // ((l, r) rows: stored inside the split)
            }
        } else {
            // ---- fused pixel column(s): magnitude_in -> color_for -> put_pixel
            //      (simple_spectrogram.rs:141-161), magnitudes staged in LDS only
            float2 *m2 = reinterpret_cast<float2 *>(buf);  // the padded column [bin]: (l, r), or for mono (frame f0, frame f0 + 1)
            float2 *vbuf = m2 + kColSlots;                  // [sample]: the interpolated pair
            lds_barrier();  // partner reads done: the buffer can be overwritten
#pragma unroll
            for (int q3 = 0; q3 < 8; ++q3) {
                const int k = col + 256 * q3;
                if (k >= 1) m2[k] = make_float2(ml[q3], mr[q3]);
                if (q3 == 0 && col == 1) m2[0] = make_float2(ml[0], mr[0]);                            // bin 1 again in front
                if (q3 == 7 && col == 255) m2[kM + 1] = m2[kM + 2] = make_float2(ml[7], mr[7]);      // bin 2047 twice behind
            }

            lds_barrier();
            sample_pass<PIX>(p, m2, vbuf, tid);
            // the fused pixel path waits for the next transform's samples HERE: the sample pass has just waited for its table words with
            // vmcnt(0), so they are there -- behind the row pass, whose pixel stores sit in a loop the compiler cannot count through,
            // the same wait is a vmcnt(0) again: every pixel store just issued acknowledged by memory, once per column
            if (!(MONO && PAIRING == kPairAdjacentRow && kSlideWindow)) next_samples_are_here();
            lds_barrier();
            uchar4 *rgba = reinterpret_cast<uchar4 *>(p.rgba);
            uchar4 *dst_a = rgba + ((size_t)(have_first ? f0 : 0) * p.pairs + p.pair) * (size_t)p.R;
            uchar4 *dst_b = rgba + ((size_t)f1 * p.pairs + p.pair) * (size_t)p.R;
            __builtin_amdgcn_s_setprio(3);
            row_pass<MONO, PIX>(p, row_words, vbuf, dst_a, dst_b, have_first, have_second, pal, tid);
        }
        SGX_STAMP(13)   // row stores issued (fused pixel path: the pixel passes)
        if (!RENDER && !(MONO && PAIRING == kPairAdjacentRow && kSlideWindow)) next_samples_are_here();   // rows: behind the stores, vmcnt(stores issued since)
        if (kSlide2) {
            // the (l, r) window moves down one row HERE, in every iteration (behind the last transform nothing was requested and nothing reads
            // the window again): at the loop head, under `pending`, the compiler copied the window out at the back edge and in again on
            // either side of the branch -- 30 moves per transform for 14
#pragma unroll
            for (int a = 0; a < 7; ++a) { sa[a] = sa[a + 1]; sb[a] = sb[a + 1]; }
            sa[7] = ld0;
            sb[7] = ld1;
        }
    }
#if SGX_STAMPS
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) atomicAdd(&g_phase_cycles[i], st_acc[i]);
        atomicAdd(&g_phase_cycles[16], (unsigned long long)(job_end > job_begin ? job_end - job_begin : 0));   // wave-iterations
    }
#endif
}

// Is the kernel's LUT-index seed floor(log2(power + 1e-7) a + b) (pixel_for) within one of the exact threshold count
// for EVERY power?  u(p) = log2(p + 1e-7) a + b is monotone in p and the count steps from e to e + 1 at lut_thr[e]
// (the exact switch point of the host's float32 evaluation, found by bisection over bit patterns), so it is enough
// to look at the switch points: with u(lut_thr[e]) inside (e + 0.5, e + 1.5) for every e, a power between two
// neighbouring switch points has u inside (count - 0.5, count + 1.5) and its floor is count - 1, count or count + 1.
// The device's v_log_f32 (1 ulp), its float32 add and fma move u by less than 1e-3 -- far inside the half index kept
// in hand.  Thresholds at +0 (levels every power reaches: dB ranges that start below the 1e-7 floor) are fine as long
// as u(0) is not below their count; unreachable levels (NaN) or anything else unusual: walk instead.
bool seed_within_one(const std::vector<float> &lut_thr, double guess_a, double guess_b)
{
    if (lut_thr.size() != 255) return false;
    auto u = [&](double pw) { return log2(pw + 1e-7) * guess_a + guess_b; };
    size_t zeros = 0;
    for (size_t e = 0; e < lut_thr.size(); ++e) {
        const float t = lut_thr[e];
        if (!(t == t) || t < 0.0f || !std::isfinite(t)) return false;
        if (e > 0 && t < lut_thr[e - 1]) return false;
        if (t == 0.0f) { zeros = e + 1; continue; }
        const double ue = u((double)t);
        if (!(ue > (double)e + 0.51 && ue < (double)e + 1.49)) return false;   // (the device's log, add and fma move u by < 1e-3)
    }
    return u(0.0) > (double)zeros - 0.49;
}

}  // namespace wg

#if SGX_STAMPS
extern "C" __attribute__((visibility("default"))) int sgx_debug_phase_cycles(unsigned long long *h_out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(h_out, HIP_SYMBOL(wg::g_phase_cycles), sizeof(unsigned long long) * 20);
    if (e == hipSuccess && reset) {
        unsigned long long zero[20] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(wg::g_phase_cycles), zero, sizeof(zero));
    }
    return e == hipSuccess ? 0 : -1;
}
#endif

hipError_t wg4096_init(sgx_ctx *c, void **out)
{
    using namespace wg;
    auto *t = new WgTables();
    std::vector<float2> tw1(16 * 256), tw2(256);
    auto unit = [](int idx, int N) {
        idx %= N;
        const double ang = -2.0 * M_PI * (double)idx / (double)N;
        double cs = cos(ang), sn = sin(ang);
        if (idx == 0) { cs = 1.0; sn = 0.0; }
        if (4 * idx == N) { cs = 0.0; sn = -1.0; }
        if (2 * idx == N) { cs = -1.0; sn = 0.0; }
        if (4 * idx == 3 * N) { cs = 0.0; sn = 1.0; }
        return make_float2((float)cs, (float)sn);
    };
    for (int q = 0; q < 16; ++q)
        for (int tt = 0; tt < 256; ++tt) tw1[q * 256 + tt] = unit(tt * q, kP);
    for (int q = 0; q < 16; ++q)
        for (int t0 = 0; t0 < 16; ++t0) tw2[q * 16 + t0] = unit(t0 * q, 256);

    // packed tables of the fused pixel path: 4 B per row, 8 B per LDS slot (the kernel re-derives mu^2, mu^3 and 1 - o' with the
    // same single-rounded operations the host table holds).  Slots = the rows' samples in lin_space order; after a row whose
    // count is even and at least 4 comes one PAD slot -- a repeat of the row's last sample that no row reads -- so that the next
    // row starts an odd number of slots on: the row pass reads slot first + i of 32 consecutive rows at once, and an even stride
    // of 8-byte slots folds those 32 addresses onto a few bank pairs (stride 8: four of them).  Host table only: the kernel is
    // the one of round 2.  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.21 -> 0.13 (profiles/r03_pixel_ab.txt).
    std::vector<uint32_t> rows(c->tab.rows.size());
    std::vector<PackedSample> samples;
    bool fusable = c->tab.rows.size() <= 1024;
    auto packed = [&](const SampleEntry &se) {
        PackedSample ps;
        ps.i0 = se.i0;   // the taps are slots i0 .. i0 + 3 of the padded column: the clamps at both ends of the spectrum are its repeated end bins
        ps.w = c->cfg.interp == SGX_INTERP_COSINE ? se.w2 : se.w0;
        if (se.i0 < 0 || se.i0 > (int32_t)c->M - 1) fusable = false;   // (index_of clamps to [0, M - 1]: cannot happen)
        return ps;
    };
    for (size_t i = 0; i < rows.size(); ++i) {
        const auto &r = c->tab.rows[i];
        // (the row word holds 16 bits of count; only the SGX_ROW_BATCH row pass keeps a BYTE per block of 256 rows, block_max_cnt)
        if (r.count >= (SGX_ROW_BATCH ? 256u : 65536u) || samples.size() >= 65536) fusable = false;
        rows[i] = ((uint32_t)samples.size() & 0xffffu) | ((r.count & 0xffffu) << 16);
        for (uint32_t j = 0; j < r.count; ++j) samples.push_back(packed(c->tab.samples[r.first + j]));
        if (r.count >= 4 && (r.count & 1u) == 0) samples.push_back(samples.back());   // the pad slot
    }
    // the interpolated samples of a column sit in LDS behind the column itself
    fusable = fusable && samples.size() <= (size_t)kMaxFusedSamples;
    if (samples.empty()) samples.push_back(PackedSample{0, 0.0f});
    t->fusable = fusable;
    t->n_samples = (uint32_t)samples.size();
    for (uint32_t blk = 0; blk < 4; ++blk) {          // blocks of 256 rows whose every row is ONE sample: no loop, no divide in the row pass
        bool single = true;
        for (size_t i = 256 * blk; i < 256 * (blk + 1) && i < rows.size(); ++i) single = single && c->tab.rows[i].count == 1;
        if (single && 256 * blk < rows.size()) t->single_rows |= 1u << blk;
        uint32_t mx = 0;
        for (size_t i = 256 * blk; i < 256 * (blk + 1) && i < rows.size(); ++i) mx = mx > c->tab.rows[i].count ? mx : c->tab.rows[i].count;
        t->block_max_cnt |= (mx > 255 ? 255u : mx) << (8 * blk);     // (a row of 256 samples or more: not fusable, below)
    }

    auto up = [](auto **dst, const auto &v) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), v.size() * sizeof(v[0]));
        if (e == hipSuccess) e = hipMemcpy(*dst, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(&t->d_tw1, tw1);
    if (e == hipSuccess) e = up(&t->d_tw2, tw2);
    if (e == hipSuccess) e = up(&t->d_rows, rows);
    if (e == hipSuccess) e = up(&t->d_samples, samples);
    if (e != hipSuccess) {
        wg4096_destroy(t);
        return e;
    }
    *out = t;
    return hipSuccess;
}

void wg4096_destroy(void *tables)
{
    auto *t = static_cast<wg::WgTables *>(tables);
    if (!t) return;
    if (t->d_tw1) (void)hipFree(t->d_tw1);
    if (t->d_tw2) (void)hipFree(t->d_tw2);
    if (t->d_rows) (void)hipFree(t->d_rows);
    if (t->d_samples) (void)hipFree(t->d_samples);
    if (t->d_planes) (void)hipFree(t->d_planes);
    delete t;
}

bool wg4096_can_fuse_render(const sgx_ctx *c, const void *tables)
{
    const auto *t = static_cast<const wg::WgTables *>(tables);
    // mono (sequential) colour schemes with a 256-entry ramp; diverging schemes take the two-kernel path
    return t && t->fusable && c->pal.n == 256 && !c->pal.stereo && !c->pal.segments;
}

void lut_seed_coefficients(const sgx_ctx *c, float &a, float &b)
{
    // t * n = (10 log10(x) - min_db) * n / (max_db - min_db) = log2(x) * a + b   (seed only)
    const double span = (double)c->cfg.max_db - (double)c->cfg.min_db;
    const double n = c->cfg.lut_index_mode == SGX_LUT_ROUND_NM1 ? 255.0 : 256.0;
    a = (float)(10.0 * log10(2.0) * n / span);
    b = (float)(-(double)c->cfg.min_db * n / span + (c->cfg.lut_index_mode == SGX_LUT_ROUND_NM1 ? 0.5 : 0.0));
}

bool wg4096_seed_is_within_one(const sgx_ctx *c)
{
    float a, b;
    lut_seed_coefficients(c, a, b);
    return !(c->cfg.flags & SGX_FLAG_LUT_WALK) && wg::seed_within_one(c->pal.lut_thr, (double)a, (double)b);
}

namespace {

template <bool RENDER>
hipError_t launch_wg(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                     size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags, uint8_t *d_rgba, bool out_f16 = false)
{
    using namespace wg;
    if (n_frames == 0) return hipSuccess;
    const auto *t = static_cast<const WgTables *>(tables);
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
    // More than two channels: the (l, r) pairs are split into planes first and every pair runs the two-channel kernel on its
    // own plane.  (Reading a pair at a stride of C floats kept the strided variant at the register cap with 60-68 bytes of
    // scratch: 8 channels ran at 80 M transforms/s against 145-180 M for a stereo stream.)
    size_t plane_floats = 0;
    if (channels > 2) {
        const size_t first_sample = first_frame * (size_t)c->H, n_samp = (n_frames - 1) * (size_t)c->H + kW;
        plane_floats = (2 * n_samp + 63) & ~(size_t)63;
        if (plane_floats * pairs > t->planes_floats) {
            hipError_t e = hipStreamSynchronize(c->stream);  // a previous launch may still read the old planes
            if (e != hipSuccess) return e;
            if (t->d_planes) { (void)hipFree(t->d_planes); t->d_planes = nullptr; t->planes_floats = 0; }
            e = hipMalloc(reinterpret_cast<void **>(&t->d_planes), plane_floats * pairs * sizeof(float));
            if (e != hipSuccess) return e;
            t->planes_floats = plane_floats * pairs;
        }
        const hipError_t e = launch_deinterleave_pairs(c, d_pcm, t->d_planes, plane_floats, first_sample, n_samp, channels, pairs);
        if (e != hipSuccess) return e;
    }
    for (uint32_t pair = 0; pair < pairs; ++pair) {
        Params p{};
        p.pcm = d_pcm;
        p.tw1 = t->d_tw1;
        p.tw2 = t->d_tw2;
        p.window = c->d_window;
        p.mags = d_mags;
        p.out_f16 = out_f16 ? 1u : 0u;
        p.first_frame = first_frame;
        p.n_frames = n_frames;
        p.total_frames = total_frames;
        p.H = c->H;
        p.C = channels;
        p.pairs = pairs;
        p.pair = pair;
        p.pair_l = channels == 1 ? 0 : 2 * pair;
        p.pair_r = channels == 1 ? 0 : 2 * pair + 1;
        if (channels > 2) {   // this pair's plane as a two-channel stream whose sample 0 is the call's first sample
            p.pcm = t->d_planes + (size_t)pair * plane_floats - first_frame * (size_t)c->H * 2;
            p.C = 2;
            p.pair_l = 0;
            p.pair_r = 1;
        }
        if (RENDER) {
            p.rows = t->d_rows;
            p.samples = t->d_samples;
            p.n_samples = t->n_samples;
            p.lut_thr = c->d_lut_thr;
            p.lut_rgba = c->d_lut_rgba;
            p.rgba = d_rgba;
            p.R = c->R;
            p.interp = c->cfg.interp;
            lut_seed_coefficients(c, p.guess_a, p.guess_b);
            p.seed_pm1 = wg4096_seed_is_within_one(c) ? 1u : 0u;
            p.single_rows = t->single_rows;
            p.block_max_cnt = t->block_max_cnt;
        }
        // A one-channel stream (include/sgx.h, "Mono streams"): by default every frame its own real-input transform
        // (stft4096_real.hip); SGX_FLAG_PAIRED_FRAMES: two frames per transform; SGX_FLAG_COMPLEX_MONO: every frame as its own (s, s)
        // transform
        const uint32_t fl = c->cfg.flags;      // (sgx_create: INDEPENDENT is set unless PAIRED was asked for)
        const bool own_transform = !(fl & SGX_FLAG_PAIRED_FRAMES) || (fl & SGX_FLAG_COMPLEX_MONO) != 0;
        if (channels == 1 && own_transform && !(fl & SGX_FLAG_COMPLEX_MONO) && real4096_serves(c, d_pcm, channels))
            return launch_real4096(c, c->d_real, p, out_f16, RENDER);
        const bool mono = channels == 1 && (fl & SGX_FLAG_PAIRED_FRAMES) && !(fl & SGX_FLAG_COMPLEX_MONO);
        p.pair_base = mono ? first_frame / 2 : 0;
        p.n_jobs = mono ? (first_frame + n_frames + 1) / 2 - first_frame / 2 : n_frames;
        // persistent workgroups, 4 per CU; each owns a contiguous run of transforms so that the
        // overlapping audio of consecutive frames is re-read from L1/L2, not HBM
        unsigned long long blocks = (unsigned long long)n_cu * 4;
        unsigned long long per = (p.n_jobs + blocks - 1) / blocks;
        if (per < 1) per = 1;
        blocks = (p.n_jobs + per - 1) / per;
        p.jobs_per_block = per;
        const dim3 grid((unsigned)blocks), block(256);
        const size_t lds = RENDER ? kLdsBytesRender : kLdsBytes;
        // the pixel code of the instantiation: the interpolator and the seed-only LUT search are compile-time (kPixCubic / kPixCosine);
        // SGX_FLAG_LUT_WALK and palettes whose seed proof fails run kPixGeneric
        const int pix = !RENDER ? kPixNone : (!p.seed_pm1 ? kPixGeneric : (p.interp == SGX_INTERP_COSINE ? kPixCosine : kPixCubic));
        auto launch = [&](auto mono_c, auto pairing_c, auto c2_c) {
            constexpr bool M_ = decltype(mono_c)::value, C2_ = decltype(c2_c)::value;
            constexpr int P_ = decltype(pairing_c)::value;
            if (!RENDER && out_f16) hipLaunchKernelGGL((stft4096_wg_kernel<M_, P_, C2_, kPixRowsF16>), grid, block, lds, c->stream, p);
            else if (!RENDER) hipLaunchKernelGGL((stft4096_wg_kernel<M_, P_, C2_, kPixNone>), grid, block, lds, c->stream, p);
            else if (pix == kPixCubic) hipLaunchKernelGGL((stft4096_wg_kernel<M_, P_, C2_, RENDER ? kPixCubic : kPixNone>), grid, block, lds, c->stream, p);
            else if (pix == kPixCosine) hipLaunchKernelGGL((stft4096_wg_kernel<M_, P_, C2_, RENDER ? kPixCosine : kPixNone>), grid, block, lds, c->stream, p);
            else hipLaunchKernelGGL((stft4096_wg_kernel<M_, P_, C2_, RENDER ? kPixGeneric : kPixNone>), grid, block, lds, c->stream, p);
        };
        using T = std::true_type;
        using F = std::false_type;
        if (mono) {
            if (c->H == 256) launch(T{}, std::integral_constant<int, kPairAdjacentRow>{}, F{});
            else launch(T{}, std::integral_constant<int, kPairAdjacent>{}, F{});
        } else if (channels == 1) {   // every mono frame as its own (s, s) transform
            launch(F{}, std::integral_constant<int, kPairAdjacent>{}, F{});
#ifndef SGX_NO_STEREO_SLIDE
        } else if (c->H == 256) {
            launch(F{}, std::integral_constant<int, kPairAdjacentRow>{}, T{});    // (l, r) at H = 256: sliding register window
#endif
        } else {
            launch(F{}, std::integral_constant<int, kPairAdjacent>{}, T{});
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

hipError_t launch_stft_wg4096(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                              size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags)
{
    return launch_wg<false>(c, tables, d_pcm, channels, pairs, first_frame, n_frames, total_frames, d_mags, nullptr);
}

hipError_t launch_stft_wg4096_f16(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                                  size_t first_frame, size_t n_frames, size_t total_frames, void *d_mags_f16)
{
    return launch_wg<false>(c, tables, d_pcm, channels, pairs, first_frame, n_frames, total_frames, static_cast<float *>(d_mags_f16), nullptr, true);
}

hipError_t launch_render_wg4096(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                                size_t first_frame, size_t n_frames, size_t total_frames, uint8_t *d_rgba)
{
    return launch_wg<true>(c, tables, d_pcm, channels, pairs, first_frame, n_frames, total_frames, nullptr, d_rgba);
}

}  // namespace sgx
