// stft4096_wg.hpp -- declarations shared by the two workgroup-per-transform kernels
// (stft4096_wg.hip: scalar codelets; stft4096_wgp.hip: packed (re, im) codelets).
#pragma once
#ifndef SGX_ABL_NSTORE
#define SGX_ABL_NSTORE 8  // ablation builds only: store this many of the 8 row segments
#endif
#include <hip/hip_fp16.h>

#include "sgx_internal.hpp"

namespace sgx {
namespace wg {

constexpr int kW = 2048, kP = 4096, kM = 2047;
// Sliding the sample window in registers (2 new rows per mono transform instead of 9 loads): every
// sample is fetched once per workgroup.  It pins 7 VGPRs across the FFT passes (the scalar kernel
// then spills 3 registers) but the launch is bound by total HBM traffic, and dropping the overlap
// re-reads (3.3 -> 1.0 KB per frame) was worth +5 % (same-device A/B, 1e6 frames).
constexpr bool kSlideWindow = true;
constexpr int kS1 = 272;            // row stride (complex) of the pass-1 -> pass-2 image [q1][t]
constexpr int kS2 = 257;            // row stride (complex) of the pass-2 -> pass-3 image [t0][q1 + 16 q2]
constexpr int kBufComplex = 16 * kS1;  // 4352 complex = 34 816 B (also holds 16*257 and 9*256)
constexpr size_t kLdsBytes = (size_t)(kBufComplex + 256) * sizeof(float2);
constexpr size_t kLdsBytesRender = kLdsBytes + 256 * sizeof(uint2);   // + the palette of pixel_for: [256] thresholds to leave level i, [256] RGBA of level i

// One work item of the fused pixel path's first pass: one magnitude_in sample (interpolated_frequency_sample.rs:66-72).
//   word  bits 0-10   bin index (cubic: floor(index); cosine: low)
//         bit  11     a tap of this sample is clamped at an end of the spectrum (the saturating index arithmetic applies)
//         bits 12-23  the slot of the interpolated value in LDS
//         bit  25     PAD: no sample (keeps a row's slots at an odd stride, see the row table)
struct PackedSample {
    uint32_t word;
    float w;      // cubic: mu;           cosine: o' (the cosine-eased offset)
};
constexpr uint32_t kItemClamped = 1u << 11, kItemPad = 1u << 25;

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
    const uint32_t *rows;      // [n_rows_b] first slot | count << 12 | py << 22
    const PackedSample *samples;   // [n_samples] work items of the first pass
    uint32_t n_samples;        // work items per column (samples + pad slots)
    uint32_t n_rows_b;         // rows of the second pass
    const float *lut_thr;      // [255]
    const uchar4 *lut_rgba;    // [256]
    uint8_t *rgba;             // [F][pairs][R][4]
    uint32_t R, interp;
    float guess_a, guess_b;    // LUT index ~ floor(log2(power + 1e-7) * a + b), then exact fix-up
    uint32_t seed_pm1;         // the host has shown that this seed is never off by more than one (seed_within_one): one compare fixes it
};

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
    uint32_t *d_rows = nullptr;        // packed row table for the fused pixel path (rows of several samples only)
    PackedSample *d_samples = nullptr;
    uint32_t n_samples = 0, n_rows_b = 0;
    bool fusable = false;
    mutable float *d_planes = nullptr;   // more than two channels: (l, r) pair planes of the sample range of a call, grown on demand
    mutable size_t planes_floats = 0;
};

__device__ __forceinline__ void lds_barrier()
{
    // LDS-only workgroup barrier: outstanding global stores are NOT waited for
#ifdef SGX_ABL_NOBARRIER
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// one row of [M][2] floats; rowm8 = row base - 8 bytes (bin k lives at byte 8 k of rowm8): a uniform
// (SGPR) row base plus one 32-bit lane offset, immediate offsets per segment
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// A raw buffer descriptor over one output row: base = the row's bin-0 address (uniform), no stride, no bounds
// in the way (2 GB window).  Stores through it take an SGPR descriptor + one 32-bit lane offset + a scalar
// segment offset + an immediate: no per-lane 64-bit address arithmetic at all.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(char *mags, long long row_byte)
{
    const uint32_t olo = __builtin_amdgcn_readfirstlane((uint32_t)row_byte);
    const uint32_t ohi = __builtin_amdgcn_readfirstlane((uint32_t)((unsigned long long)row_byte >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(mags + (long long)(((unsigned long long)ohi << 32) | olo), 0, 0x7fffffff, 0x00020000);
}

// one row of [M][2] floats; row_byte = byte offset of the row's (absent) bin 0, bin k lives 8 k bytes on
// Cache policy of the magnitude stores (A/B on one device, 1e6 frames): an (l, r) stream re-reads 7/8 of every frame's
// samples through L2, and marking the output non-temporal keeps them there: 6.00 -> 5.57 ms.  A mono stream slides its
// window in registers and re-reads nothing: there the same bit costs 10-40 %, sc1 30 %.
#ifndef SGX_OUT_AUX
#define SGX_OUT_AUX (DUP ? 0 : 2 /* nt */)
#endif
template <bool DUP>  // DUP: mono, the row holds (m, m); else (va, vb) = (left, right)
__device__ __forceinline__ void store_row(char *mags, long long row_byte, int col, const float (&va)[8], const float (&vb)[8])
{
    const __amdgpu_buffer_rsrc_t r = row_rsrc(mags, row_byte);
    const int lane_off = col * 8;
#pragma unroll
    for (int q3 = 0; q3 < 8; ++q3)
        if ((q3 > 0 || col != 0) && q3 < SGX_ABL_NSTORE) {  // k = 0 (DC) is not part of the output (fft.rs:81)
            const u32x2 d = {__float_as_uint(va[q3]), __float_as_uint(DUP ? va[q3] : vb[q3])};
            __builtin_amdgcn_raw_buffer_store_b64(d, r, lane_off + 2048 * (q3 & 1), 4096 * (q3 >> 1), SGX_OUT_AUX);
        }
}

// ---- fused pixel column: magnitude_in -> color_for -> put_pixel (simple_spectrogram.rs:141-161) ---------------
// Two passes over LDS, both balanced whatever the rows' sample counts (a row of the log axis averages 1 sample at the
// bottom and 11 at the top: one thread per ROW through both steps leaves the wave that owns the top rows with twice the
// work of the others, and walks the sample table in a dependent loop of L1 loads):
//   sample pass  one thread per magnitude_in SAMPLE (2 173 per column at 48 kHz / 1024 rows): coalesced table read,
//                interpolation of BOTH channels (stereo) or BOTH mono columns of the transform -- the column lives in LDS
//                as float2 per bin either way -- result to its slot in LDS
//   row pass     one thread per row: sum of its samples in lin_space order (Complex::sum), two requested at a time, the
//                divide, colour, pixel store (coalesced: consecutive lanes, consecutive image rows).  A row's slots start
//                at an odd distance from the previous row's where the count is even (pad slots): 32 consecutive rows of 8
//                samples then touch 32 different bank pairs instead of 4
// (Round 3: the pad slots; SQ_LDS_BANK_CONFLICT before them: 21 % of the LDS-active cycles, profiles/r02_pixel_pipes.json.)
constexpr int kMaxFusedSamples = (kBufComplex - 2048);  // float2 per slot behind the 2048-bin column (2304)

template <bool COSINE, bool INTERIOR>
__device__ __forceinline__ float2 interp_sample2(const float2 *m2, int i0, float w, int last)
{
    float2 v;
    if (COSINE) {
        // :79-86  data[low] * (1 - o') + data[high] * o'
        const int lo = i0, hi = INTERIOR ? lo + 1 : (lo + 1 < last ? lo + 1 : last);
        const float w1 = 1.0f - w;
        const float2 a = m2[lo], b = m2[hi];
        v.x = a.x * w1 + b.x * w;
        v.y = a.y * w1 + b.y * w;
    } else {
        // :89-105
        const int x1 = i0;
        const int x0 = INTERIOR ? x1 - 1 : (x1 > 0 ? x1 - 1 : 0);
        const int x2 = INTERIOR ? x1 + 1 : (x1 + 1 < last ? x1 + 1 : last);
        const int x3 = INTERIOR ? x1 + 2 : (x1 + 2 < last ? x1 + 2 : last);
        const float mu = w, mu2 = mu * mu, mu3 = mu * mu2;
        const float2 y0 = m2[x0], y1 = m2[x1], y2 = m2[x2], y3 = m2[x3];
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
    }
    return v;
}

// colorscheme.rs:59-61 as a threshold count: the LUT level is the number of thresholds the power has reached, the
// thresholds being the exact switch points of the host's float32 evaluation (sgx_tables.cpp).  v_log_f32 only SEEDS the
// count.  pal_thr[i] = the smallest power whose level is i + 1 (NaN for i = 255: no power leaves the last level), pal_rgba[i] =
// the colour of level i (alpha = 1.0 -> 255 included).
//   seed_pm1 (the usual case): the host has checked, threshold by threshold, that the exact value u the seed approximates
//     lies within half an index of the count at every switch point (seed_within_one), so floor(u - 1/2) is the count or
//     one below it: that entry's threshold decides between its colour and the next.  No loop.  (A NaN power: the seed is 0
//     and the compare fails -> level 0, as the walk below gives.)
//   otherwise (unreachable levels, SGX_FLAG_LUT_WALK): walk from the seed, as the first version of this kernel did.
// (Round 3, same-device A/B, profiles/r03_pixel_ab.txt: proving the seed EXACT wherever it is 1 / 256 away from an integer --
// no threshold read for 99 % of the pixels, one 4-byte read at the data-dependent address instead of three -- halves the
// kernel's LDS bank conflicts and costs 3-4 %: five more vector instructions per pixel matter, the conflicts do not.)
__device__ __forceinline__ uint32_t pixel_for(const Params &p, float l, float r, const float *pal_thr, const uint32_t *pal_rgba)
{
    const float power = (l * l) + (r * r);
    const float u = fmaf(__builtin_amdgcn_logf(power + 1e-7f), p.guess_a, p.guess_b);
    if (p.seed_pm1) {
        int idx = (int)floorf(u - 0.5f);
        idx = idx < 0 ? 0 : (idx > 254 ? 254 : idx);
        return power >= pal_thr[idx] ? pal_rgba[idx + 1] : pal_rgba[idx];
    }
    int idx = (int)floorf(u);
    idx = idx < 0 ? 0 : (idx > 255 ? 255 : idx);
    while (idx < 255 && power >= pal_thr[idx]) ++idx;
    while (idx > 0 && !(power >= pal_thr[idx - 1])) --idx;
    return pal_rgba[idx];
}

bool seed_within_one(const std::vector<float> &lut_thr, double guess_a, double guess_b);

// MONO: .x / .y of a sample are the two columns (frames) of the transform, each a (s, s) pixel; else one (l, r) pixel.
template <bool MONO>
__device__ __forceinline__ void put_pixels(const Params &p, float l, float r, uint32_t py, uint32_t *dst_a, uint32_t *dst_b, bool have_a, bool have_b,
                                           const float *pal_thr, const uint32_t *pal_rgba)
{
    const uint32_t y = p.R - 1 - py;  // simple_spectrogram.rs:150
    if (MONO) {  // mono -> (s, s): both channels carry the same magnitude
        const uint32_t ca = pixel_for(p, l, l, pal_thr, pal_rgba), cb = pixel_for(p, r, r, pal_thr, pal_rgba);
        if (have_a) dst_a[y] = ca;
        if (have_b) dst_b[y] = cb;
    } else {
        dst_a[y] = pixel_for(p, l, r, pal_thr, pal_rgba);
    }
}

template <bool COSINE>
__device__ __forceinline__ void sample_pass(const Params &p, const float2 *m2, float2 *vbuf, int tid)
{
    const int last = kM - 1;
    // (No global store may be issued in this loop: vmcnt retires in order, so the wait for a table word requested behind a
    // pixel store is a wait for that store.  Round 3 tried colouring the rows of ONE sample right here, saving their trip
    // through LDS: 8 % slower, SQ_WAIT_ANY +25 %, for exactly that reason.)
    // the table word of the next step is requested before this step's gathers: one L1 latency per step is
    // overlapped instead of exposed (two registers; deeper unrolling costs more registers than these kernels have)
    uint32_t s = tid;
    PackedSample se = p.samples[s < p.n_samples ? s : 0];
    while (s < p.n_samples) {
        const uint32_t s_next = s + 256;
        const PackedSample se_next = p.samples[s_next < p.n_samples ? s_next : 0];
        if (!(se.word & kItemPad)) {
            const int i0 = (int)(se.word & 0x7ffu);
            vbuf[(se.word >> 12) & 0xfffu] = (se.word & kItemClamped) ? interp_sample2<COSINE, false>(m2, i0, se.w, last)
                                                                       : interp_sample2<COSINE, true>(m2, i0, se.w, last);
        }
        se = se_next;
        s = s_next;
    }
}

// (Tried in round 1, same-device A/B: the thread's table words requested before the two barriers and both passes unrolled
// (9 samples, 4 rows side by side, thresholds read four at a time): 182 -> 131 M frames/s -- these kernels sit at the
// 128-VGPR cap of four waves per SIMD and every extra live value becomes scratch traffic.)
template <bool MONO>
__device__ __forceinline__ void row_pass(const Params &p, const uint32_t (&row_words)[4], const float2 *vbuf, uint32_t *dst_a, uint32_t *dst_b,
                                         bool have_a, bool have_b, const float *pal_thr, const uint32_t *pal_rgba, int tid)
{
    int i_row = 0;
    for (uint32_t q = tid; q < p.n_rows_b; q += 256, ++i_row) {
        // a thread renders the same rows of every column: their table words stay in registers (at most 1024 rows)
        const uint32_t re = i_row == 0 ? row_words[0] : i_row == 1 ? row_words[1] : i_row == 2 ? row_words[2] : row_words[3];
        const uint32_t first = re & 0xfffu, cnt = (re >> 12) & 0x3ffu, py = re >> 22;
        const float2 *src = vbuf + first;
        float2 v = src[0];
        float sl = 0.0f + v.x, sr = 0.0f + v.y;  // Complex::sum starts at zero
        uint32_t i = 1;
        for (; i + 1 < cnt; i += 2) {     // two samples requested back to back, added in order
            const float2 v0 = src[i], v1 = src[i + 1];
            sl = (sl + v0.x) + v1.x;
            sr = (sr + v0.y) + v1.y;
        }
        if (i < cnt) {
            v = src[i];
            sl = sl + v.x;
            sr = sr + v.y;
        }
        float l = sl, r = sr;
        if (cnt > 1) {  // x / 1.0 == x: only rows that average several samples divide (:72)
            const float nf = (float)cnt;
            l = sl / nf;
            r = sr / nf;
        }
        put_pixels<MONO>(p, l, r, py, dst_a, dst_b, have_a, have_b, pal_thr, pal_rgba);
    }
}

// the same row as IEEE half pairs (round to nearest even): bin k at byte 4 k of rowm4
template <bool DUP>
__device__ __forceinline__ void store_row_f16(char *mags, long long row_byte, int col, const float (&va)[8], const float (&vb)[8])
{
    const __amdgpu_buffer_rsrc_t r = row_rsrc(mags, row_byte);
    const int lane_off = col * 4;
#pragma unroll
    for (int q3 = 0; q3 < 8; ++q3)
        if (q3 > 0 || col != 0) {
            const __half2 h = __floats2half2_rn(va[q3], DUP ? va[q3] : vb[q3]);
            __builtin_amdgcn_raw_buffer_store_b32(*reinterpret_cast<const uint32_t *>(&h), r, lane_off + 1024 * (q3 & 3), 4096 * (q3 >> 2), 0);
        }
}

}  // namespace wg
}  // namespace sgx
