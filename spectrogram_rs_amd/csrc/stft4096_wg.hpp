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
constexpr size_t kLdsBytesRender = kLdsBytes + 256 * sizeof(float) + 256 * sizeof(uchar4);

struct PackedSample {
    int32_t i0;   // cubic: floor(index); cosine: low
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
    const uint32_t *rows;      // [R]  first | count << 16
    const PackedSample *samples;
    const float *lut_thr;      // [255]
    const uchar4 *lut_rgba;    // [256]
    uint8_t *rgba;             // [F][pairs][R][4]
    uint32_t R, interp;
    float guess_a, guess_b;    // LUT index ~ floor(log2(power + 1e-7) * a + b), then exact fix-up
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
    uint32_t *d_rows = nullptr;        // packed row table for the fused pixel path
    PackedSample *d_samples = nullptr;
    bool fusable = false;
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
template <bool DUP>  // DUP: mono, the row holds (m, m); else (va, vb) = (left, right)
__device__ __forceinline__ void store_row(char *mags, long long row_byte, int col, const float (&va)[8], const float (&vb)[8])
{
    const __amdgpu_buffer_rsrc_t r = row_rsrc(mags, row_byte);
    const int lane_off = col * 8;
#pragma unroll
    for (int q3 = 0; q3 < 8; ++q3)
        if ((q3 > 0 || col != 0) && q3 < SGX_ABL_NSTORE) {  // k = 0 (DC) is not part of the output (fft.rs:81)
            const u32x2 d = {__float_as_uint(va[q3]), __float_as_uint(DUP ? va[q3] : vb[q3])};
            __builtin_amdgcn_raw_buffer_store_b64(d, r, lane_off + 2048 * (q3 & 1), 4096 * (q3 >> 1), 0);
        }
}

// ---- fused pixel column: magnitude_in -> color_for -> put_pixel (simple_spectrogram.rs:141-161) ---------------
// `mc` is one column of magnitudes in LDS: MONO a scalar per bin (l = r), else (l, r) pairs.  Row word:
// first sample | count << 16 | interior << 31, where `interior` says that no tap of the row touches the
// ends of the spectrum, so the taps are the contiguous bins x1-1 .. x1+2 (cubic) / lo, lo+1 (cosine) and
// the saturating index arithmetic of interpolated_frequency_sample.rs:89-105 can be skipped.
template <bool MONO, bool COSINE, bool INTERIOR>
__device__ __forceinline__ void interp_sample(const float *mc, const PackedSample se, int last, float &vl, float &vr)
{
    vr = 0.0f;
    if (COSINE) {
        // :79-86  data[low] * (1 - o') + data[high] * o'
        const int lo = se.i0, hi = INTERIOR ? lo + 1 : (lo + 1 < last ? lo + 1 : last);
        const float w1 = 1.0f - se.w;
        if (MONO) {
            vl = mc[lo] * w1 + mc[hi] * se.w;
        } else {
            const float2 a = reinterpret_cast<const float2 *>(mc)[lo], b = reinterpret_cast<const float2 *>(mc)[hi];
            vl = a.x * w1 + b.x * se.w;
            vr = a.y * w1 + b.y * se.w;
        }
    } else {
        // :89-105
        const int x1 = se.i0;
        const int x0 = INTERIOR ? x1 - 1 : (x1 > 0 ? x1 - 1 : 0);
        const int x2 = INTERIOR ? x1 + 1 : (x1 + 1 < last ? x1 + 1 : last);
        const int x3 = INTERIOR ? x1 + 2 : (x1 + 2 < last ? x1 + 2 : last);
        const float mu = se.w, mu2 = mu * mu, mu3 = mu * mu2;
        if (MONO) {
            const float y0 = mc[x0], y1 = mc[x1], y2 = mc[x2], y3 = mc[x3];
            const float a0 = ((y3 - y2) - y0) + y1;
            const float a1 = (y0 - y1) - a0;
            const float a2 = y2 - y0;
            vl = ((a0 * mu3) + (a1 * mu2)) + ((a2 * mu) + y1);
        } else {
            const float2 *m2 = reinterpret_cast<const float2 *>(mc);
            const float2 y0 = m2[x0], y1 = m2[x1], y2 = m2[x2], y3 = m2[x3];
            {
                const float a0 = ((y3.x - y2.x) - y0.x) + y1.x;
                const float a1 = (y0.x - y1.x) - a0;
                const float a2 = y2.x - y0.x;
                vl = ((a0 * mu3) + (a1 * mu2)) + ((a2 * mu) + y1.x);
            }
            {
                const float a0 = ((y3.y - y2.y) - y0.y) + y1.y;
                const float a1 = (y0.y - y1.y) - a0;
                const float a2 = y2.y - y0.y;
                vr = ((a0 * mu3) + (a1 * mu2)) + ((a2 * mu) + y1.y);
            }
        }
    }
}

// (Tried: dealing rows to the four waves round-robin to even out the sample counts, which grow with the
// row -- the strided pixel stores cost more than the balance gains: 145 -> 133 M frames/s.)
template <bool MONO, bool COSINE>
__device__ __forceinline__ void render_column(const Params &p, const float *mc, uchar4 *dst, const float *thr, const uchar4 *lut, int tid)
{
    const int last = kM - 1;
    for (uint32_t py = tid; py < p.R; py += 256) {
        const uint32_t re = p.rows[py];
        const uint32_t first = re & 0xffffu, cnt = (re >> 16) & 0x7fffu;
        float sl = 0.0f, sr = 0.0f;  // Complex::sum starts at zero
        if (re >> 31) {
            for (uint32_t i = 0; i < cnt; ++i) {
                float vl, vr;
                interp_sample<MONO, COSINE, true>(mc, p.samples[first + i], last, vl, vr);
                sl = sl + vl;
                if (!MONO) sr = sr + vr;
            }
        } else {
            for (uint32_t i = 0; i < cnt; ++i) {
                float vl, vr;
                interp_sample<MONO, COSINE, false>(mc, p.samples[first + i], last, vl, vr);
                sl = sl + vl;
                if (!MONO) sr = sr + vr;
            }
        }
        float l = sl, r = sr;
        if (cnt > 1) {  // x / 1.0 == x: only rows that average several samples divide (:72)
            const float nf = (float)cnt;
            l = sl / nf;
            if (!MONO) r = sr / nf;
        }
        if (MONO) r = l;  // mono -> (s, s): both channels carry the same magnitude
        // colorscheme.rs:59-61 as a threshold count; the log2 only seeds the search
        const float power = (l * l) + (r * r);
        int idx = (int)floorf(fmaf(__builtin_amdgcn_logf(power + 1e-7f), p.guess_a, p.guess_b));
        idx = idx < 0 ? 0 : (idx > 255 ? 255 : idx);
        while (idx < 255 && power >= thr[idx]) ++idx;
        while (idx > 0 && !(power >= thr[idx - 1])) --idx;
        dst[p.R - 1 - py] = lut[idx];  // simple_spectrogram.rs:150; alpha = 1.0 -> 255
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
