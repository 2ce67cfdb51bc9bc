// stft4096_wg.hpp -- declarations shared by the two workgroup-per-transform kernels
// (stft4096_wg.hip: scalar codelets; stft4096_wgp.hip: packed (re, im) codelets).
#pragma once
#include <hip/hip_fp16.h>

#include "sgx_internal.hpp"

namespace sgx {
namespace wg {

constexpr int kW = 2048, kP = 4096, kM = 2047;
#ifndef SGX_ROW_BATCH
#define SGX_ROW_BATCH 0   // row pass, 1: four samples per trip with a launch-uniform trip count (measured round 5: 3.90 vs 3.82 ms per 1e6 frames -- the selects cost more than the latency saved)
#endif
#ifndef SGX_SAMPLE_PRELOAD
#define SGX_SAMPLE_PRELOAD 9   // sample pass: a thread's table words are requested this many steps ahead (9 = all at once; 0: the loop of rounds 1-4, one step ahead.  Same device, config 3 cosine / cubic: 0: 3.85 / 4.13 ms, 3: 3.94 / 4.16, 9: 3.84 / 4.07)
#endif
// Sliding the sample window in registers (2 new rows per mono transform instead of 9 loads): every
// sample is fetched once per workgroup.  It pins 7 VGPRs across the FFT passes (the scalar kernel
// then spills 3 registers) but the launch is bound by total HBM traffic, and dropping the overlap
// re-reads (3.3 -> 1.0 KB per frame) was worth +5 % (same-device A/B, 1e6 frames).
constexpr bool kSlideWindow = true;
constexpr int kS1 = 272;            // row stride (complex) of the pass-1 -> pass-2 image [q1][t]
constexpr int kS2 = 257;            // row stride (complex) of the pass-2 -> pass-3 image [t0][q1 + 16 q2]
constexpr int kBufComplex = 16 * kS1;  // 4352 complex = 34 816 B (also holds 16*257 and 9*256)
constexpr size_t kLdsBytes = (size_t)(kBufComplex + 256) * sizeof(float2);
constexpr size_t kLdsBytesRender = kLdsBytes + 256 * sizeof(uint2);   // + the palette table of pixel_for: {threshold to leave index i, RGBA of index i}

struct PackedSample {
    int32_t i0;   // cubic: floor(index) = x1; cosine: low.  The taps are the slots i0 .. i0 + 3 (cubic) / i0 + 1, i0 + 2 (cosine) of the padded column
    float w;      // cubic: mu;           cosine: o' (the cosine-eased offset)
};

struct Params {
    const float *pcm;
    const float2 *tw1;   // [16][256]  w_4096^{t q1}
    const float2 *tw2;   // [16][16]   w_256^{t0 q2} at [q2][t0]
    const float *window; // [2048]
    float *mags;
    unsigned long long first_frame, n_frames, n_jobs, jobs_per_block;
    unsigned long long pair_base;  // mono: global index of the first frame PAIR (first_frame / 2)
    unsigned long long total_frames;  // frames the stream holds (a pair's second frame is transformed whenever it exists)
    uint32_t H, C, pair_l, pair_r, pairs, pair;
    uint32_t out_f16;          // magnitudes are stored as (l, r) half pairs, 4 B per bin (the F16F16 ring of gpu_spectrogram.rs:218-226)
    // fused pixel path (RENDER): magnitudes never leave LDS
    const uint32_t *rows;      // [R]  first slot | count << 16
    const PackedSample *samples;   // one entry per LDS slot: the rows' samples in lin_space order, plus pad slots (see wg4096_init)
    uint32_t n_samples;        // slots per column (sum of the rows' counts + pads)
    const float *lut_thr;      // [255]
    const uchar4 *lut_rgba;    // [256]
    uint8_t *rgba;             // [F][pairs][R][4]
    uint32_t R, interp;
    float guess_a, guess_b;    // LUT index ~ floor(log2(power + 1e-7) * a + b), then exact fix-up
    uint32_t seed_pm1;         // the host has shown that this seed is never off by more than one (seed_within_one): one compare pair fixes it
    uint32_t single_rows;      // bit i: every row of block i (rows 256 i .. 256 i + 255) averages exactly one sample
    uint32_t block_max_cnt;    // byte i: the largest sample count of a row of block i (row pass: a launch-uniform trip count per block)
    // the real-input kernel (stft4096_real.hip: every mono frame its own transform; tw1 is then [8][256] w_2048^{t q1})
    const float2 *twu;         // [8][128] w_4096^{u + 128 q3} at [q3][u]; [0][0] holds w_4096^{1024} = -i
    unsigned long long stream_samples;   // samples the frames of the stream cover: columns past them read as zero
};

// stft4096_real.hip
hipError_t launch_real4096(const sgx_ctx *c, const void *real_tables, Params p, bool out_f16, bool render);

// Which two mono frames share a transform: always (2j, 2j+1).
//   kPairAdjacentRow : H = 256: frame 2j+1's rows are frame 2j's rows shifted by one (9 rows feed both)
//   kPairAdjacent    : any other hop (16 row loads)
// (Tried and rejected, same-device A/B on 1e6 frames: pairing (f, f+16) plus a per-row lane rotation
// so that both rows' stores are 128-byte aligned: -10 %, the extra row loads cost more than the
// alignment buys; 16-byte stores via a DPP lane-pair exchange: -8 %.  The launch is bound by total
// HBM traffic, not by store alignment or store instruction count.)
constexpr int kPairAdjacentRow = 0, kPairAdjacent = 1;

struct WgTables {
    float2 *d_tw1 = nullptr;
    float2 *d_tw2 = nullptr;
    uint32_t *d_rows = nullptr;        // packed row table for the fused pixel path
    PackedSample *d_samples = nullptr;
    uint32_t n_samples = 0;
    uint32_t single_rows = 0;          // Params::single_rows
    uint32_t block_max_cnt = 0;        // Params::block_max_cnt
    bool fusable = false;
    mutable float *d_planes = nullptr;   // more than two channels: (l, r) pair planes of the sample range of a call, grown on demand
    mutable size_t planes_floats = 0;
};

__device__ __forceinline__ void lds_barrier()
{
    // LDS-only workgroup barrier: outstanding global stores are NOT waited for
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// An 8-byte LDS read that stays one: the compiler merges two reads off one base register into a ds_read2_b64, which occupies the LDS
// pipe for 8 cycles where two ds_read_b64 take 2 each (MI355X_MICROARCH.md, LDS table).  The empty statement makes the base a new value
// for every read (no instruction; the register allocator keeps it in place).
#ifndef SGX_PK_CUBIC
#define SGX_PK_CUBIC 1   // cubic sample pass: both components of a sample in packed f32 instructions (0: scalar, for A/B).  (The cosine taps, the row sums and the two mono powers packed the same way: config 3 cosine 3.62-3.70 -> 3.77-3.83 ms, and the cubic gain halved: not taken)
#endif
#ifndef SGX_NO_READ2
#define SGX_NO_READ2 1
#endif
typedef float lds_f2v __attribute__((ext_vector_type(2)));
typedef const lds_f2v __attribute__((address_space(3))) lds_cfloat2;   // (an LDS pointer by type: behind the statement nothing else says so)
__device__ __forceinline__ lds_cfloat2 *lds_ptr(const float2 *p) { return (lds_cfloat2 *)p; }
__device__ __forceinline__ float2 lds_read_alone(lds_cfloat2 *&base, int idx)
{
#if SGX_NO_READ2
    asm("" : "+v"(base));
#endif
#if defined(SGX_ABL_LDS) && (SGX_ABL_LDS & 2)   // timing only (diagnostic builds): no LDS read
    float a_ = (float)idx, b_ = 1.0f;
    asm volatile("" : "+v"(a_), "+v"(b_) : "v"(base));
    return make_float2(a_, b_);
#endif
    const lds_f2v v = base[idx];
    return make_float2(v.x, v.y);
}

// one row of [M][2] floats; rowm8 = row base - 8 bytes (bin k lives at byte 8 k of rowm8): a uniform
// (SGPR) row base plus one 32-bit lane offset, immediate offsets per segment
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// A raw buffer descriptor over one output row: base = the row's bin-0 address (uniform), no stride, no bounds
// in the way (2 GB window).  Stores through it take an SGPR descriptor + one 32-bit lane offset + a scalar
// segment offset + an immediate: no per-lane 64-bit address arithmetic at all.
// `present` false: zero records, every store through the descriptor is dropped by the range check (lane offset + scalar offset against
// the records: tools/bufrange.hip) -- a row outside the requested range costs no branch, and the stores of an iteration are
// straight-line code whose count the compiler's vmcnt waits can rely on
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(char *mags, long long row_byte, bool present = true)
{
    const uint32_t olo = __builtin_amdgcn_readfirstlane((uint32_t)row_byte);
    const uint32_t ohi = __builtin_amdgcn_readfirstlane((uint32_t)((unsigned long long)row_byte >> 32));
    const int records = __builtin_amdgcn_readfirstlane(present ? 0x7fffffff : 0);
    return __builtin_amdgcn_make_buffer_rsrc(mags + (long long)(((unsigned long long)ohi << 32) | olo), 0, records, 0x00020000);
}

// the same kind of descriptor over the sample stream from a wave-uniform address on (a frame's first sample)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pcm_rsrc(const float *base)
{
    const unsigned long long a = (unsigned long long)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}

// one row of [M][2] floats; row_byte = byte offset of the row's (absent) bin 0, bin k lives 8 k bytes on
// Cache policy of the magnitude stores (A/B on one device, 1e6 frames): an (l, r) stream re-reads 7/8 of every frame's
// samples through L2, and marking the output non-temporal keeps them there: 6.00 -> 5.57 ms.  A mono stream slides its
// window in registers and re-reads nothing: there the same bit costs 10-40 %, sc1 30 %.
#if defined(SGX_AUX)
constexpr int kAuxNt = SGX_AUX;   // (A/B: bit 0 sc0, bit 1 nt, bit 4 sc1)
#else
constexpr int kAuxNt = 2;   // the `nt` bit of a buffer store
#endif
template <bool DUP>  // DUP: mono, the row holds (m, m); else (va, vb) = (left, right)
__device__ __forceinline__ void store_row(char *mags, long long row_byte, int col, const float (&va)[8], const float (&vb)[8], bool present = true)
{
    const __amdgpu_buffer_rsrc_t r = row_rsrc(mags, row_byte, present);
    const int lane_off = col * 8;
    // k = 0 (DC) is not part of the output (fft.rs:81): thread 0's first store carries a lane offset beyond the descriptor's 2 GB and is
    // dropped by the range check (tools/bufrange.hip) -- eight store instructions in straight-line code for every wave, so that the
    // compiler's count of the stores issued since the prefetch (its vmcnt waits) is exact
    const int lane_off0 = col != 0 ? lane_off : (int)0x80000000;
#pragma unroll
    for (int q3 = 0; q3 < 8; ++q3)
#if SGX_ABL_STORES == 2       // timing only: one store segment of eight
        if (q3 == 1) {
#elif SGX_ABL_STORES == 3     // timing only: every value computed and live, (practically) no store executed
        if (va[q3] == 12345.678f) {
#else
        {
#endif
            const u32x2 d = {__float_as_uint(va[q3]), __float_as_uint(DUP ? va[q3] : vb[q3])};
            __builtin_amdgcn_raw_buffer_store_b64(d, r, (q3 == 0 ? lane_off0 : lane_off) + 2048 * (q3 & 1), 4096 * (q3 >> 1), DUP ? 0 : kAuxNt);
        }
}

// ---- fused pixel column: magnitude_in -> color_for -> put_pixel (simple_spectrogram.rs:141-161) ---------------
// Two passes over LDS, both balanced whatever the rows' sample counts (a row of the log axis averages 1 sample
// at the bottom and 11 at the top: one thread per ROW leaves the wave that owns the top rows with twice the work
// of the others, and walks the sample table in a dependent loop of L1 loads):
//   sample pass  one thread per magnitude_in SAMPLE (2 173 per column at 48 kHz / 1024 rows): coalesced table
//                read, interpolation of BOTH channels (stereo) or BOTH mono columns of the transform -- the
//                column lives in LDS as float2 per bin either way -- result to LDS
//   row pass     one thread per row: sum of its samples in lin_space order (Complex::sum), the divide, dB
//                thresholds, LUT, pixel store (coalesced: consecutive lanes, consecutive image rows).  A row's slots
//                start at an odd distance from the previous row's where the count is even (pad slots, host table
//                only): 32 consecutive rows of 8 samples then touch 32 different bank pairs instead of 4
// The column in LDS (round 4): P[k] = bin k (k = 1 .. 2047, i.e. data[k - 1] of interpolated_frequency_sample.rs), with the end bins
// REPEATED around it -- P[0] = bin 1, P[2048] = P[2049] = bin 2047 -- so that the saturating index arithmetic of :89-105
// (x0 = max(x1 - 1, 0), x2 / x3 = min(x1 + 1 / 2, M - 1)) is a plain read of four CONTIGUOUS slots P[x1 .. x1 + 3] for every sample:
// no clamped variant, no branch per sample, two ds_read2_b64 per cubic sample (one per cosine sample: P[lo + 1], P[lo + 2]).
constexpr int kColSlots = 2050;                                 // slots of the padded column
constexpr int kMaxFusedSamples = kBufComplex - kColSlots;       // float2 per sample behind it (2302)
// which pixel code an instantiation carries: the launch-uniform switches -- the interpolator, the LUT search -- are compile-time,
// each instantiation holds one path's code and live ranges
constexpr int kPixNone = 0, kPixCubic = 1, kPixCosine = 2, kPixGeneric = 3, kPixRowsF16 = 4;   // kPixRowsF16: no pixels either -- half-pair rows (stft4096_wg.hip)
//   // kPixGeneric: interpolator at run time, LUT seed + walk (SGX_FLAG_LUT_WALK / proof failed)

// (The repeats are written by the ONE thread that holds bin 1 / bin 2047, in the one unrolled step where it does -- a test of the
// bin index in every step of every thread kept sixteen compares' worth of values alive and spilled 14 registers.)

template <bool COSINE>
__device__ __forceinline__ float2 interp_sample2(const float2 *P, int i0, float w)
{
    float2 v;
    if (COSINE) {
        // :79-86  data[low] * (1 - o') + data[high] * o',  high = min(low + 1, M - 1)
        const float w1 = 1.0f - w;
        lds_cfloat2 *q = lds_ptr(P + i0);
        const float2 a = lds_read_alone(q, 1), b = lds_read_alone(q, 2);
        v.x = a.x * w1 + b.x * w;
        v.y = a.y * w1 + b.y * w;
    } else {
        // :89-105
        const float mu = w, mu2 = mu * mu, mu3 = mu * mu2;
        lds_cfloat2 *q = lds_ptr(P + i0);
#if SGX_PK_CUBIC
        // both components at once: the same operations in the same order on (x, y) pairs -- v_pk_add_f32 / v_pk_mul_f32 round each half
        // as the scalar instructions do (no contraction: -ffp-contract=off), and the taps arrive from LDS as register pairs
        asm("" : "+v"(q)); const lds_f2v y0 = q[0];
        asm("" : "+v"(q)); const lds_f2v y1 = q[1];
        asm("" : "+v"(q)); const lds_f2v y2 = q[2];
        asm("" : "+v"(q)); const lds_f2v y3 = q[3];
        const lds_f2v a0 = ((y3 - y2) - y0) + y1;
        const lds_f2v a1 = (y0 - y1) - a0;
        const lds_f2v a2 = y2 - y0;
        const lds_f2v r = ((a0 * mu3) + (a1 * mu2)) + ((a2 * mu) + y1);
        v.x = r.x; v.y = r.y;
#else
        const float2 y0 = lds_read_alone(q, 0), y1 = lds_read_alone(q, 1), y2 = lds_read_alone(q, 2), y3 = lds_read_alone(q, 3);
        {
            const float a0 = ((y3.x - y2.x) - y0.x) + y1.x;
            const float a1 = (y0.x - y1.x) - a0;
            const float a2 = y2.x - y0.x;
            v.x = ((a0 * mu3) + (a1 * mu2)) + ((a2 * mu) + y1.x);
        }
        {
            const float a0 = ((y3.y - y2.y) - y0.y) + y1.y;
            const float a1 = (y0.y - y1.y) - a0;
            const float a2 = y2.y - y0.y;
            v.y = ((a0 * mu3) + (a1 * mu2)) + ((a2 * mu) + y1.y);
        }
#endif
    }
    return v;
}

template <bool COSINE>
__device__ __forceinline__ void sample_pass_for(const Params &p, const float2 *P, float2 *vbuf, int tid)
{
    // the table word of the next step is requested before this step's gathers: one L1 latency per step is
    // overlapped instead of exposed (two registers; deeper unrolling costs more registers than these kernels have)
    // (the table through a buffer descriptor: a uniform base + a 32-bit lane offset, no per-lane 64-bit pointer to keep)
    const __amdgpu_buffer_rsrc_t rt = pcm_rsrc(reinterpret_cast<const float *>(p.samples));
    auto item = [&](uint32_t i) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rt, (int)((i < p.n_samples ? i : 0u) * 8u), 0, 0);
        return PackedSample{(int32_t)v.x, __uint_as_float(v.y)};
    };
#if SGX_SAMPLE_PRELOAD
    // the nine steps unrolled (kMaxFusedSamples / 256 rounded up), a thread's table words requested kAhead steps ahead: their L1 / L2
    // latencies overlap each other instead of following one another (all nine at once: 18 registers, 52 bytes of scratch)
    constexpr int kSteps = (kMaxFusedSamples + 255) / 256, kAhead = SGX_SAMPLE_PRELOAD;
    PackedSample se[kSteps];
    asm volatile("" : "+v"(tid));   // (opaque: the unrolled steps' table offsets and slot addresses are not to be hoisted out of the transform loop -- they spill)
#pragma unroll
    for (int k = 0; k < kAhead && k < kSteps; ++k) se[k] = item(tid + 256 * k);
#pragma unroll
    for (int k = 0; k < kSteps; ++k) {
        if (k + kAhead < kSteps) se[k + kAhead] = item(tid + 256 * (k + kAhead));
        const uint32_t s = tid + 256 * k;
        if (s < p.n_samples) vbuf[s] = interp_sample2<COSINE>(P, se[k].i0, se[k].w);
        __builtin_amdgcn_sched_barrier(0);
    }
#else
    uint32_t s = tid;
    PackedSample se = item(s);
    while (s < p.n_samples) {
        const uint32_t s_next = s + 256;
        const PackedSample se_next = item(s_next);
        vbuf[s] = interp_sample2<COSINE>(P, se.i0, se.w);
        se = se_next;
        s = s_next;
    }
#endif
}

// The same pass with the table words requested EARLY (SGX_SAMPLE_EARLY, stft4096_real.hip): a thread's nine words are asked for in front of
// the barrier and the column writes that precede the pass, so that their L1 / L2 latency falls on those instead of on the head of the pass
// (stamped build: the pass was 22 % of config 3's iteration, most of it the wait for the words it had just requested).
constexpr int kSampleSteps = (kMaxFusedSamples + 255) / 256;
#ifndef SGX_SAMPLE_EARLY_N
#define SGX_SAMPLE_EARLY_N 9   // words requested early (in front of the column writes: 17 spilled registers at the 128-register cap; behind each pair of them, as stft4096_real.hip does: none)
#endif
constexpr int kSampleEarly = SGX_SAMPLE_EARLY_N;
struct SampleWords { PackedSample se[kSampleEarly]; };
__device__ __forceinline__ PackedSample sample_word(const __amdgpu_buffer_rsrc_t &rt, const Params &p, uint32_t i)
{
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rt, (int)((i < p.n_samples ? i : 0u) * 8u), 0, 0);
    return PackedSample{(int32_t)v.x, __uint_as_float(v.y)};
}
__device__ __forceinline__ void sample_request(const Params &p, int tid, SampleWords &w)
{
    const __amdgpu_buffer_rsrc_t rt = pcm_rsrc(reinterpret_cast<const float *>(p.samples));
    asm volatile("" : "+v"(tid));   // (opaque: see sample_pass_for)
#pragma unroll
    for (int k = 0; k < kSampleEarly; ++k) w.se[k] = sample_word(rt, p, tid + 256 * k);
}
template <int PIX>
__device__ __forceinline__ void sample_pass_with(const Params &p, const float2 *P, float2 *vbuf, int tid, const SampleWords &w)
{
    const __amdgpu_buffer_rsrc_t rt = pcm_rsrc(reinterpret_cast<const float *>(p.samples));
    asm volatile("" : "+v"(tid));
    PackedSample se[kSampleSteps];
#pragma unroll
    for (int k = 0; k < kSampleSteps; ++k) se[k] = k < kSampleEarly ? w.se[k] : sample_word(rt, p, tid + 256 * k);   // the rest: now, all at once
#pragma unroll
    for (int k = 0; k < kSampleSteps; ++k) {
        const uint32_t s = tid + 256 * k;
        if (s < p.n_samples) {
            if (PIX == kPixCosine || (PIX == kPixGeneric && p.interp == SGX_INTERP_COSINE)) vbuf[s] = interp_sample2<true>(P, se[k].i0, se[k].w);
            else vbuf[s] = interp_sample2<false>(P, se[k].i0, se[k].w);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int PIX>
__device__ __forceinline__ void sample_pass(const Params &p, const float2 *P, float2 *vbuf, int tid)
{
    if (PIX == kPixCosine || (PIX == kPixGeneric && p.interp == SGX_INTERP_COSINE)) sample_pass_for<true>(p, P, vbuf, tid);
    else sample_pass_for<false>(p, P, vbuf, tid);
}

// colorscheme.rs:59-61 as a threshold count: the LUT index is the number of thresholds the power has reached, the
// thresholds being the exact switch points of the host's float32 evaluation (sgx_tables.cpp).  v_log_f32 only SEEDS the
// count.  pal[i] = {the smallest power whose index is i + 1 (NaN for i = 255: no power leaves the last index), RGBA of i}.
//   seed_pm1 (the usual case; the only code of kPixCubic / kPixCosine): the host has checked, threshold by threshold, that the
//     exact value u the seed approximates lies within half an index of the count at every switch point (seed_within_one), so
//     floor(u - 1/2) is the count or one below it: ONE 16-byte LDS access brings that entry's threshold and both candidate
//     colours, one compare picks.  No loop, no second (dependent) LDS access for the colour.  (A NaN power: the seed is 0 and
//     the compare fails -> index 0, as the walk below gives.)
//   otherwise (unreachable levels, SGX_FLAG_LUT_WALK; kPixGeneric): walk from the seed, as the first version of this kernel did.
template <int PIX>
__device__ __forceinline__ uint32_t pixel_from_power(const Params &p, float power, const uint2 *pal)
{
    const float u = fmaf(__builtin_amdgcn_logf(power + 1e-7f), p.guess_a, p.guess_b);
    if (PIX != kPixGeneric || p.seed_pm1) {
        int idx = (int)floorf(u - 0.5f);
        idx = idx < 0 ? 0 : (idx > 254 ? 254 : idx);
        const uint2 e0 = pal[idx], e1 = pal[idx + 1];
        return power >= __uint_as_float(e0.x) ? e1.y : e0.y;      // RGBA, alpha = 1.0 -> 255
    }
    int idx = (int)floorf(u);
    idx = idx < 0 ? 0 : (idx > 255 ? 255 : idx);
    while (idx < 255 && power >= __uint_as_float(pal[idx].x)) ++idx;
    while (idx > 0 && !(power >= __uint_as_float(pal[idx - 1].x))) --idx;
    return pal[idx].y;
}

template <int PIX>
__device__ __forceinline__ uint32_t pixel_for(const Params &p, float l, float r, const uint2 *pal)
{
    return pixel_from_power<PIX>(p, (l * l) + (r * r), pal);      // colorscheme.rs:59
}

bool seed_within_one(const std::vector<float> &lut_thr, double guess_a, double guess_b);

// MONO: .x / .y of a sample are the two columns (frames) of the transform, each a (s, s) pixel; else one (l, r) pixel.
// A thread renders the same four rows tid + 256 i of every column: the loop over i is unrolled, the rows' table words are registers
// by name.  p.single_rows bit i: EVERY row of block i (rows 256 i .. 256 i + 255) is one sample -- 743 of the 1024 rows at 48 kHz,
// blocks 0 and 1 whole -- and the block runs without the sample loop, the count test and the divide.
// (Tried, same-device A/B: the thread's sample-table words and row words requested before the two barriers and the
// loops unrolled (9 samples, 4 rows side by side, thresholds read four at a time): 182 -> 131 M frames/s -- these
// kernels sit at the 128-VGPR cap of four waves per SIMD and every extra live value becomes scratch traffic.)
template <bool MONO, int PIX>
__device__ __forceinline__ void row_pass(const Params &p, const uint32_t (&row_words)[4], const float2 *vbuf, uchar4 *dst_a, uchar4 *dst_b,
                                         bool have_a, bool have_b, const uint2 *pal, int tid)
{
    uint32_t *out_a = reinterpret_cast<uint32_t *>(dst_a), *out_b = reinterpret_cast<uint32_t *>(dst_b);
#pragma unroll 1
    for (int i_row = 0; i_row < 4; ++i_row) {
        const uint32_t py = tid + 256 * i_row;
        if (py >= p.R) break;
        const uint32_t re = i_row == 0 ? row_words[0] : i_row == 1 ? row_words[1] : i_row == 2 ? row_words[2] : row_words[3];
        const uint32_t first = re & 0xffffu, cnt = re >> 16;
        float l, r;
        if (p.single_rows & (1u << i_row)) {          // (launch-uniform)
            const float2 v = vbuf[first];
            l = v.x;                                   // (0 + x and x / 1.0 are x -- but for the sign of a zero, and only l * l + r * r is used)
            r = v.y;
        } else {
            float sl = 0.0f, sr = 0.0f;
#if SGX_ROW_BATCH
            // Complex::sum in lin_space order (interpolated_frequency_sample.rs:66-72), four samples per trip: the four LDS reads are
            // independent (one latency per trip instead of one per sample), the adds stay in order.  The trip count is the block's largest
            // row (launch-uniform: no divergent loop); a lane whose row is shorter reads its own last sample again and adds +0 instead
            // (x + 0 is x; only l * l + r * r is used, so the sign of a zero does not matter).
            const uint32_t trips = (((p.block_max_cnt >> (8 * i_row)) & 0xffu) + 3u) >> 2;
            const uint32_t last = first + cnt - 1;
            for (uint32_t t = 0; t < trips; ++t) {
                const uint32_t i0 = first + 4 * t;
                const float2 v0 = vbuf[i0 < last ? i0 : last], v1 = vbuf[i0 + 1 < last ? i0 + 1 : last];
                const float2 v2 = vbuf[i0 + 2 < last ? i0 + 2 : last], v3 = vbuf[i0 + 3 < last ? i0 + 3 : last];
                const bool k0 = 4 * t < cnt, k1 = 4 * t + 1 < cnt, k2 = 4 * t + 2 < cnt, k3 = 4 * t + 3 < cnt;
                sl = sl + (k0 ? v0.x : 0.0f); sr = sr + (k0 ? v0.y : 0.0f);
                sl = sl + (k1 ? v1.x : 0.0f); sr = sr + (k1 ? v1.y : 0.0f);
                sl = sl + (k2 ? v2.x : 0.0f); sr = sr + (k2 ? v2.y : 0.0f);
                sl = sl + (k3 ? v3.x : 0.0f); sr = sr + (k3 ? v3.y : 0.0f);
            }
#else
            for (uint32_t i = 0; i < cnt; ++i) {
                const float2 v = vbuf[first + i];
                sl = sl + v.x;
                sr = sr + v.y;
            }
#endif
            l = sl;
            r = sr;
            if (cnt > 1) {  // x / 1.0 == x: only rows that average several samples divide (:72)
                const float nf = (float)cnt;
                l = sl / nf;
                r = sr / nf;
            }
        }
        const uint32_t y = p.R - 1 - py;  // simple_spectrogram.rs:150
        if (MONO) {  // mono -> (s, s): both channels carry the same magnitude
            if (have_a) out_a[y] = pixel_for<PIX>(p, l, l, pal);
            if (have_b) out_b[y] = pixel_for<PIX>(p, r, r, pal);
        } else {
            out_a[y] = pixel_for<PIX>(p, l, r, pal);
        }
        // one row at a time: without this fence the scheduler interleaves the four unrolled rows, and the kernels -- at the 128-VGPR cap
        // of four waves per SIMD -- spill 15 to 34 registers
        asm volatile("" ::: "memory");
    }
}

// the same row as IEEE half pairs (round to nearest even): bin k at byte 4 k of rowm4
template <bool DUP>
__device__ __forceinline__ void store_row_f16(char *mags, long long row_byte, int col, const float (&va)[8], const float (&vb)[8], bool present = true)
{
    const __amdgpu_buffer_rsrc_t r = row_rsrc(mags, row_byte, present);
    const int lane_off = col * 4;
    const int lane_off0 = col != 0 ? lane_off : (int)0x80000000;   // k = 0 (DC): dropped by the range check (see store_row)
#pragma unroll
    for (int q3 = 0; q3 < 8; ++q3) {
        const __half2 h = __floats2half2_rn(va[q3], DUP ? va[q3] : vb[q3]);
        __builtin_amdgcn_raw_buffer_store_b32(*reinterpret_cast<const uint32_t *>(&h), r, (q3 == 0 ? lane_off0 : lane_off) + 1024 * (q3 & 3), 4096 * (q3 >> 2), 0);
    }
}

}  // namespace wg
}  // namespace sgx
