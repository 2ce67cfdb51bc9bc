// stft4096_real.hip -- K1R: a mono stream at W = 2048 (H = 256: the tuned form; any other hop too) with EVERY frame its own transform, at the
// price of half a transform.
//
// The reference duplicates a mono sample into (s, s) and runs one 4096-point complex transform per frame
// (audio_input_list_model.rs:67-69, fft.rs:47-99): F = (1 + i) S with S the spectrum of the real windowed frame, and both output
// columns are |S[k]| 2 / W (fft.rs:81-98).  The headline kernel (stft4096_wg.hip) halves that work by packing frames 2j and 2j+1
// into the real and imaginary part of ONE transform -- and pays with a shared rounding floor: the quieter frame of a pair carries
// the louder one's float32 noise (tests/test_gpu_parity.py::test_onsets_*).  This kernel halves the work the other way: the
// 4096-point spectrum of a REAL sequence is a 2048-point complex transform of z[m] = x[2m] + i x[2m+1] plus one butterfly per bin,
//
//     S[k]        = E[k] + w_4096^k O[k],          E[k] = (Z[k] + conj Z[2048-k]) / 2,   O[k] = (Z[k] - conj Z[2048-k]) / 2i
//     S[2048 - k] = conj(E[k] - w_4096^k O[k]),    k = 0 .. 1023   (+ the self-paired k = 1024)
//
// so every frame is transformed on its own -- the reference's dataflow, north_star's tolerance against the frame's OWN peak on any
// input -- for the arithmetic of half a 4096-point transform.  Two frames share a workgroup only as two independent problems.
//
// Shape: K1's, so that K1's measured choices carry over (256 threads, three in-register passes, two LDS exchanges of one padded
// 34 816-byte image, partner exchange, persistent workgroups, four per CU, the same wave priorities).  A workgroup iteration does
// frames fa = first + 2 job (A) and fa + 1 (B), 2048 points each (zero padding: only m < 1024 is non-zero, never materialised):
//
//   sample index  m = t + 256 a            (a < 4 non-zero rows; x[2m], x[2m+1] arrive as ONE 8-byte load per lane)
//   pass 1  thread t, BOTH frames : 8-point DFT over a with 4 non-zero inputs = two 4-point FFTs -> q1; twiddle w_2048^{t q1}
//   pass 2  thread (F, q1, t0)    : t = t0 + 16 t1; 16-point FFT over t1 -> q2; twiddle w_256^{t0 q2} (K1's LDS table)
//   pass 3  thread (F, u = q1 + 8 q2): 16-point FFT over t0 -> q3;  Z_F[k], k = u + 128 q3
//   untangle  Z[k] with Z[2048-k] (thread 128 - u, register 15 - q3, through LDS as K1's split) -> |S[k]|, |S[2048-k]|:
//           every thread stores 16 bins of its frame's row, 8 ascending and 8 descending runs of 512 contiguous bytes per wave
//
// H = 256 = HALF a row of z.  With c[i] = (x[2i], x[2i+1]) the stream as 8-byte columns, frame f reads c[128 f + t + 256 a]: a thread's
// eight rows of a frame pair are R[j] = c[128 fa + t + 128 j], j = 0 .. 7 (even j: frame A, odd j: frame B), and the next pair's are
// R[j + 2]: the window slides in registers.  Of the two new values R[9] = c[128 fa + 1152 + t] is the thread's own load (ONE 8-byte
// load per thread and iteration, requested a whole iteration ahead, in front of the stores); R[8] = c[128 fa + 1024 + t] is what
// thread t - 128 loaded for this iteration (t >= 128) or what thread t + 128 holds as its R[7] (t < 128): every thread publishes
// the one its opposite number needs in 2 KB of LDS beside the partner exchange (same barrier) and reads the other's.  Every sample
// is fetched from memory exactly once.
// (The first version let both frames slide on their own and fetched every sample twice; the output stream keeps evicting the
// input from L2, the second fetch went to HBM, and the launch took exactly the 5.9 % longer that 18 424 B / frame are more than
// 17 400: 3.72 ms per 1e6 frames against the paired kernel's 3.50.)
#include <cmath>
#include <type_traits>

#include "stft4096_wg.hpp"

namespace sgx {

namespace wgr {

using wg::kBufComplex;
using wg::kLdsBytes;
using wg::kLdsBytesRender;
using wg::kM;
using wg::kS1;
using wg::kS2;
using wg::kW;
using wg::lds_barrier;
using wg::lds_read_alone;
using wg::lds_cfloat2;
using wg::lds_ptr;
using wg::pcm_rsrc;
using wg::u32x2;

typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float cl_fma(float a, float c, float u) { return fmaf(a, c, u); }
__device__ __forceinline__ f2v cl_fma(f2v a, float c, f2v u) { return __builtin_elementwise_fma(a, f2v{c, c}, u); }

#include "fft_codelets.inc"

using wg::Params;   // the 4096-point kernels' parameter block: tw1 = [8][256] w_2048^{t q1}, + twu, stream_samples (stft4096_wg.hpp)

__device__ __forceinline__ float2 cmulf(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}

// a raw buffer descriptor over one output row (uniform base); `present` false: zero records, every store through it is dropped by
// the range check -- a frame outside the requested range costs no branch in the store section.  (gfx950 checks lane offset + SCALAR
// offset against the record count -- tools/bufrange.hip, profiles/r05_bufrange.txt: a store with records 0 and soffset 2048 writes
// nothing, a load whose soffset alone passes the records reads 0 -- and test_rows_past_the_requested_count_are_never_written holds
// every two-frames-per-iteration kernel to it with sentinel rows behind and in front of the caller's range.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t out_rsrc(char *base, long long byte, bool present)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)byte);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((unsigned long long)byte >> 32));
    const int records = __builtin_amdgcn_readfirstlane(present ? 0x7fffffff : 0);
    return __builtin_amdgcn_make_buffer_rsrc(base + (long long)(((unsigned long long)hi << 32) | lo), 0, records, 0x00020000);
}

// 8-point forward DFT of (z0, z1, z2, z3, 0, 0, 0, 0): even bins = FFT4(z), odd bins = FFT4(z_a w_8^a); bin q1 in (yr, yi)[q1]
__device__ __forceinline__ void fft8_half_zero(const float (&zr)[4], const float (&zi)[4], float (&yr)[8], float (&yi)[8])
{
    constexpr float kH = 0.70710678118654752440f;
    auto fft4 = [](float a0r, float a0i, float a1r, float a1i, float a2r, float a2i, float a3r, float a3i, float *outr, float *outi) {
        const float s0r = a0r + a2r, s0i = a0i + a2i, s1r = a0r - a2r, s1i = a0i - a2i;
        const float s2r = a1r + a3r, s2i = a1i + a3i, s3r = a1r - a3r, s3i = a1i - a3i;
        outr[0] = s0r + s2r; outi[0] = s0i + s2i;          // bin 0
        outr[4] = s0r - s2r; outi[4] = s0i - s2i;          // bin 2
        outr[2] = s1r + s3i; outi[2] = s1i - s3r;          // bin 1: s1 - i s3
        outr[6] = s1r - s3i; outi[6] = s1i + s3r;          // bin 3: s1 + i s3
    };
    fft4(zr[0], zi[0], zr[1], zi[1], zr[2], zi[2], zr[3], zi[3], yr, yi);                       // even q1 at [0], [2], [4], [6]
    const float y1r = (zr[1] + zi[1]) * kH, y1i = (zi[1] - zr[1]) * kH;                          // z1 (1 - i) / sqrt 2
    const float y3r = (zi[3] - zr[3]) * kH, y3i = -(zr[3] + zi[3]) * kH;                         // z3 (-1 - i) / sqrt 2
    fft4(zr[0], zi[0], y1r, y1i, zi[2], -zr[2], y3r, y3i, yr + 1, yi + 1);                       // odd q1 at [1], [3], [5], [7]
}

constexpr int kRowsF32 = 0, kRowsF16 = 1, kPixels = 2;   // what a launch writes: float rows, half-pair rows, RGBA columns (fused pixel path)

// PIX (kPixels only): the pixel code of the instantiation, wg::kPixCubic / kPixCosine / kPixGeneric (stft4096_wg.hpp)
// SLIDE: H = 256, the window slides in registers (above).  Else: any hop (a frame starts on any sample): the eight columns of the
// next frame pair are requested where the sliding form requests its one, straight into R -- dead since pass 1 -- and every sample is
// fetched 2048 / H times, through L2.
#ifndef SGX_K1R_INTERLEAVE
#define SGX_K1R_INTERLEAVE 1
#endif
#ifndef SGX_SAMPLE_EARLY
#define SGX_SAMPLE_EARLY 1   // fused pixels: the sample-table words requested in front of the column writes (stft4096_wg.hpp: sample_request)
#endif
#ifndef SGX_ADDTID_R
#define SGX_ADDTID_R 1   // (0: the b64 transposes of the non-sliding instantiation, for A/B)
#endif
// TR (the sliding instantiation): the two LDS transposes as real / imaginary planes, written with ds_write_addtid_b32 and read back in
// 16-byte pieces, as in stft4096_wg.hip (there: why pass 1 then runs column t = (tid >> 4) + 16 (tid & 15)).  Image 1: rows 8 F + q1 of 272
// words; image 2: rows q2 of 280 words -- its readers are the threads (F, u = q1 + 8 q2), and with that lane order 280 is the stride whose
// 16-byte reads are conflict-free (tools/lds_b128_conflicts.py).  The imaginary plane starts 4 480 words in, for both.
constexpr int kPlaneIm = 4480;                                        // words
constexpr int kBufComplexTR = kPlaneIm;                               // 2 * 4 480 words = 35 840 B: 1 KB more than the b64 images
static_assert(kBufComplexTR >= kBufComplex && 16 * 280 <= kPlaneIm, "the planes hold both images");
constexpr size_t kLdsBytesR = (size_t)(kBufComplexTR + 256) * sizeof(float2);
constexpr size_t kLdsBytesRenderR = kLdsBytesR + 256 * sizeof(uint2);
static_assert(4 * kLdsBytesRenderR <= 160 * 1024, "four workgroups per CU");
// rows ra and rb (STRIDE words each) of both planes, this wave's 64 words: LDS address = M0 + offset + 4 * lane.  M0 is set inside the
// statement (it cannot be declared clobbered -- a reserved register to clang: stft4096_wg.hip -- so the ISA check also refuses every implicit reader of M0 beside these statements) and one wait state separates a scalar write of M0 from an
// add-TID instruction (tools/isa_check_addtid.py looks at every build).
template <int STRIDE>
__device__ __forceinline__ void addtid_rows(float2 a, float2 b, uint32_t m0_wave, int ra, int rb)
{
    asm volatile("s_mov_b32 m0, %4\n\t"
                 "s_nop 0\n\t"
                 "ds_write_addtid_b32 %0 offset:%5\n\t"
                 "ds_write_addtid_b32 %1 offset:%6\n\t"
                 "ds_write_addtid_b32 %2 offset:%7\n\t"
                 "ds_write_addtid_b32 %3 offset:%8"
                 :
                 : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y), "s"(m0_wave), "i"(4 * STRIDE * ra), "i"(4 * STRIDE * ra + 4 * kPlaneIm), "i"(4 * STRIDE * rb), "i"(4 * STRIDE * rb + 4 * kPlaneIm)
                 : "memory");
}
__device__ __forceinline__ void read_planes(const float4 *rd4, float (&xr)[16], float (&xi)[16])
{
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 r = rd4[c], i = rd4[c + kPlaneIm / 4];
        xr[4 * c] = r.x; xr[4 * c + 1] = r.y; xr[4 * c + 2] = r.z; xr[4 * c + 3] = r.w;
        xi[4 * c] = i.x; xi[4 * c + 1] = i.y; xi[4 * c + 2] = i.z; xi[4 * c + 3] = i.w;
    }
}
__device__ __forceinline__ float lane_xor8(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xf, 0xf, true)); }   // row_ror:8

#if SGX_STAMPS
// diagnostic build only (tools/k1r_phases.py): per-phase wave cycles (s_memtime), summed over all waves and iterations
__device__ unsigned long long g_phase_cycles_r[24];
#define SGX_STAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[i] += now_ - st_last; st_last = now_; }
#else
#define SGX_STAMP(i)
#endif

template <int MODE, int PIX, bool SLIDE>
__global__ void __launch_bounds__(256, 4) stft4096_real_kernel(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *buf = reinterpret_cast<float2 *>(smem_raw);
    constexpr bool TR = SLIDE && SGX_ADDTID_R;
    float2 *tw2 = buf + (TR ? kBufComplexTR : kBufComplex);
    uint2 *pal = reinterpret_cast<uint2 *>(tw2 + 256);          // kPixels only: [256] {threshold, RGBA} (wg::pixel_for)
    constexpr bool F16 = MODE == kRowsF16;

    const int tid = threadIdx.x;
    const int t_p1 = TR ? (tid >> 4) + 16 * (tid & 15) : tid;     // pass-1 column of this thread
    float *plane = reinterpret_cast<float *>(smem_raw);
    const uint32_t m0_wave = __builtin_amdgcn_readfirstlane((uint32_t)(tid >> 6) * 272u);   // TR: byte offset of this wave inside a plane row (64 + 4 words per wave)
    // TR read sides: image 1, thread (g2 = tid >> 4, t0 = tid & 15): the words of writers 16 t0 .. + 15 of row g2; image 2, thread (F, u):
    // the words of writers 16 (8 F + q1) .. + 15 of row q2
    const float4 *rd4_1 = reinterpret_cast<const float4 *>(plane + 272 * (tid >> 4) + 68 * ((tid & 15) >> 2) + 16 * (tid & 3));
    const float4 *rd4_2 = reinterpret_cast<const float4 *>(plane + 280 * ((tid & 127) >> 3) + 68 * ((8 * (tid >> 7) + (tid & 7)) >> 2) + 16 * (tid & 3));
    tw2[tid] = p.tw2[tid];
    uint32_t row_words[4] = {0u, 0u, 0u, 0u};  // kPixels: the table words of this thread's rows tid + 256 i
    if (MODE == kPixels) {
        pal[tid] = make_uint2(__float_as_uint(tid < 255 ? p.lut_thr[tid] : __builtin_nanf("")), *reinterpret_cast<const uint32_t *>(&p.lut_rgba[tid]));
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((uint32_t)tid + 256u * i < p.R) row_words[i] = p.rows[tid + 256 * i];
    }

    // per-thread constants, resident for the life of the (persistent) workgroup
    // the output scale |S| 2 / W rides on the window as 1 / 2048 (the untangle's two halves and 2 / W = 1 / 1024): a power of two
    // commutes with every rounding below
    float win[8];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        win[2 * a] = p.window[2 * t_p1 + 512 * a] * (1.0f / 2048.0f);
        win[2 * a + 1] = p.window[2 * t_p1 + 512 * a + 1] * (1.0f / 2048.0f);
    }
    float2 tw1[8];
#pragma unroll
    for (int q = 1; q < 8; ++q) tw1[q] = p.tw1[q * 256 + t_p1];
    const int F = tid >> 7, u = tid & 127;        // pass-3 / output role: frame of the pair, bins u + 128 q3
    float2 twu[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) twu[q] = p.twu[q * 128 + u];
    const int g2 = tid >> 4, t0_2 = tid & 15;      // pass-2 role: g2 = 8 F + q1
    const int cbase2 = (g2 >> 3) * 128 + (g2 & 7); // its image-2 column is cbase2 + 8 q2
    __syncthreads();

    const unsigned long long job_begin = (unsigned long long)blockIdx.x * p.jobs_per_block;
    unsigned long long job_end = job_begin + p.jobs_per_block;
    if (job_end > p.n_jobs) job_end = p.n_jobs;
    if (job_begin >= job_end) return;

    // Columns from `col0` of the stream on, as a raw buffer (uniform base + one 32-bit lane offset); its record count is what the
    // stream still holds from there, so a column past the end reads as zero: a pair's second frame that the stream does not hold, or
    // the rows requested ahead at the end of a run, need no branch (their results are never stored).
    // (the loads below are 8-byte words at 4-byte aligned addresses when the hop is odd or the stream starts on an odd sample: buffer
    // loads of more than one dword need dword alignment only)
    auto samples_from = [&](unsigned long long first) {                      // first: sample index
        const unsigned long long left = first < p.stream_samples ? (p.stream_samples - first) * 4 : 0;
        const unsigned long long addr = (unsigned long long)(p.pcm + (first < p.stream_samples ? first : 0));
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr), hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
        const int records = __builtin_amdgcn_readfirstlane((int)(left < 0x7fffffffull ? left : 0x7fffffffull));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0, records, 0x00020000);
    };
    auto columns_from = [&](unsigned long long col0) { return samples_from(2 * col0); };
    auto column = [&](__amdgpu_buffer_rsrc_t r, int byte_offset) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, t_p1 * 8, byte_offset, 0);
        return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
    };
    float2 R[8];        // R[j] = c[128 fa + tid + 128 j]: rows a = j / 2 of frame A (even j) and frame B (odd j)
    float2 L;           // the next R[7] = c[128 fa + 1152 + tid]   (SLIDE)
    auto load_pair = [&](unsigned long long fa_) {   // !SLIDE: all eight columns of the pair (fa_, fa_ + 1); any hop
        const __amdgpu_buffer_rsrc_t r0 = samples_from(fa_ * p.H), r1 = samples_from((fa_ + 1) * p.H);
#pragma unroll
        for (int a = 0; a < 4; ++a) { R[2 * a] = column(r0, 2048 * a); R[2 * a + 1] = column(r1, 2048 * a); }
    };
    {
        const __amdgpu_buffer_rsrc_t r0 = columns_from(128 * (p.first_frame + 2 * job_begin));
        if (SLIDE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) R[j] = column(r0, 1024 * j);
            L = column(r0, 9216);         // c[128 fa + 1152 + tid]
        } else {
            load_pair(p.first_frame + 2 * job_begin);
            L = R[7];
        }
        // the window is waited for HERE (an empty asm that reads it), so that the loop header carries no pending load of the entry path:
        // merged with the back edge -- where the same registers are long complete -- it made the compiler wait at the top of EVERY
        // iteration with vmcnt(2), i.e. for the sixteen row stores just issued to be acknowledged by memory
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(R[j].x), "v"(R[j].y));
        asm volatile("" ::"v"(L.x), "v"(L.y));
        // (the resident constants too: the compiler schedules their loads BEHIND the window's, and a constant still pending at the
        // loop entry became a vmcnt(10) .. vmcnt(5) at its first use inside the loop -- in every iteration)
#pragma unroll
        for (int q = 1; q < 8; ++q) asm volatile("" ::"v"(tw1[q].x), "v"(tw1[q].y));
#pragma unroll
        for (int q = 0; q < 8; ++q) asm volatile("" ::"v"(twu[q].x), "v"(twu[q].y), "v"(win[q]));
        // (L as well: in flight at the loop entry it made the wait at its first use -- the exchange write, two thirds into the
        // iteration -- a vmcnt(1): all sixteen row stores of the PREVIOUS iteration acknowledged.  Every later L is complete before the
        // back edge: the copy from Ln behind the stores waits for it with vmcnt(16), the stores still in flight.)
    }
    float2 *xch = buf + 2304;         // [256] the iteration's loads, beside the partner rows (image 2 is dead by then)

    char *out = reinterpret_cast<char *>(p.mags);
    constexpr int kBin = F16 ? 4 : 8;             // bytes per output bin
#if SGX_STAMPS
    unsigned long long st_acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_readcyclecounter();
#endif
    for (unsigned long long job = job_begin; job < job_end; ++job) {
        SGX_STAMP(19)
        const unsigned long long fa = p.first_frame + 2 * job;             // frame A; B = fa + 1
        const unsigned long long la = 2 * job, lb = la + 1;                // their rows in the output
        const bool have_b = lb < p.n_frames;                                // (a B outside the range is computed and dropped)

        // ---- Hann (fft.rs:53-63) and pass 1 for both frames
        float yr[8], yi[8];
        {
            float zr[4], zi[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { zr[a] = R[2 * a].x * win[2 * a]; zi[a] = R[2 * a].y * win[2 * a + 1]; }
            fft8_half_zero(zr, zi, yr, yi);
        }
        float vr[8], vi[8];
        {
            float zr[4], zi[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { zr[a] = R[2 * a + 1].x * win[2 * a]; zi[a] = R[2 * a + 1].y * win[2 * a + 1]; }
            fft8_half_zero(zr, zi, vr, vi);
        }
        SGX_STAMP(0)    // Hann + pass 1 (two frames)
        lds_barrier();  // the previous iteration's partner reads are complete
        SGX_STAMP(1)    // barrier 0
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float2 ya = make_float2(yr[q], yi[q]), yb = make_float2(vr[q], vi[q]);
            if (TR) {
                addtid_rows<272>(q == 0 ? ya : cmulf(ya, tw1[q]), q == 0 ? yb : cmulf(yb, tw1[q]), m0_wave, q, 8 + q);
            } else {
                buf[q * kS1 + tid] = q == 0 ? ya : cmulf(ya, tw1[q]);
                buf[(8 + q) * kS1 + tid] = q == 0 ? yb : cmulf(yb, tw1[q]);
            }
        }
        __builtin_amdgcn_s_setprio(0);  // (wave priorities: stft4096_wg.hip)
        SGX_STAMP(2)    // twiddles + image-1 writes
        lds_barrier();
        SGX_STAMP(3)    // barrier 1

        // ---- pass 2: thread (g2 = 8 F + q1, t0): 16-point FFT over t1, then twiddle w_256^{t0 q2}
        float xr[16], xi[16];
        if (TR) {
            read_planes(rd4_1, xr, xi);
        } else {
#pragma unroll
            for (int t1 = 0; t1 < 16; ++t1) {
                const float2 v = buf[g2 * kS1 + t0_2 + 16 * t1];
                xr[t1] = v.x; xi[t1] = v.y;
            }
        }
        fft16(xr, xi);
        SGX_STAMP(4)    // image-1 reads + FFT16
        lds_barrier();  // everyone has read image 1
        SGX_STAMP(5)    // barrier 2
        if (TR) {
            lds_cfloat2 *tw2p = lds_ptr(tw2 + t0_2);
#pragma unroll
            for (int q2 = 0; q2 < 16; q2 += 2) {
                const int pa = FFT16_OUT[q2], pb = FFT16_OUT[q2 + 1];
                const float2 va = make_float2(xr[pa], xi[pa]), vb = make_float2(xr[pb], xi[pb]);
                addtid_rows<280>(q2 == 0 ? va : cmulf(va, lds_read_alone(tw2p, q2 * 16)), cmulf(vb, lds_read_alone(tw2p, (q2 + 1) * 16)), m0_wave, q2, q2 + 1);
            }
        } else {
#pragma unroll
            for (int q2 = 0; q2 < 16; ++q2) {
                const int pos = FFT16_OUT[q2];
                const float2 v = make_float2(xr[pos], xi[pos]);
                buf[t0_2 * kS2 + cbase2 + 8 * q2] = q2 == 0 ? v : cmulf(v, tw2[q2 * 16 + t0_2]);
            }
        }
        SGX_STAMP(6)    // pass-2 twiddles + image-2 writes
        lds_barrier();
        SGX_STAMP(7)    // barrier 3

        // ---- pass 3: thread (F, u): 16-point FFT over t0 -> Z[u + 128 q3]
        if (TR) {
            read_planes(rd4_2, xr, xi);
        } else {
#pragma unroll
            for (int t0 = 0; t0 < 16; ++t0) {
                const float2 v = buf[t0 * kS2 + tid];
                xr[t0] = v.x; xi[t0] = v.y;
            }
        }
        fft16(xr, xi);

        // ---- rows: the load of the NEXT iteration (its R[9]), ahead of this iteration's stores (vmcnt retires in issue order), into a
        // second register pair -- L is still needed for the exchange.  Unconditional (a conditional request keeps the old value alive
        // around the loop); past the end of the stream it reads zeros.  (Requested at the TOP of the iteration instead -- a whole
        // iteration to return in -- it sits right behind the previous iteration's sixteen stores: 5-7 % slower, same device.)
        float2 Ln = make_float2(0.0f, 0.0f);
        if (SLIDE && MODE != kPixels) Ln = column(columns_from(128 * (fa + 2) + 1152), 0);
        if (!SLIDE) load_pair(fa + 2);     // (R is dead since pass 1; past the end of the stream: zeros, never stored)

        if (MODE == kPixels) __builtin_amdgcn_s_setprio(1);   // the pixel passes are long: 3 only from the row pass (the pixel stores) on
        else __builtin_amdgcn_s_setprio(3);
        SGX_STAMP(8)    // image-2 reads + FFT16 + next column requested
        lds_barrier();  // everyone has read image 2
        SGX_STAMP(9)    // barrier 4
        // partner exchange: publish q3 = 8..15
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pos = FFT16_OUT[8 + j];
            buf[j * 256 + tid] = make_float2(xr[pos], xi[pos]);
        }
        // the next R[6] = c[128 fa + 1024 + tid] of the thread half a row away: its L (lower half publishes) or its R[7] (upper half)
        if (SLIDE && !TR) {
            xch[tid].x = tid < 128 ? L.x : R[7].x;
            xch[tid].y = tid < 128 ? L.y : R[7].y;
        }
        SGX_STAMP(10)   // partner + window-exchange writes
        lds_barrier();
        SGX_STAMP(11)   // barrier 5
        if (SLIDE) {
            // (TR: column t + 128 is lane ^ 8 of the same wave -- no trip through LDS)
            const float2 Y = TR ? make_float2(lane_xor8((tid & 8) ? R[7].x : L.x), lane_xor8((tid & 8) ? R[7].y : L.y)) : xch[(tid + 128) & 255];
            // ---- slide the window by two half rows, here: L is dead from now on (the Hann products of this iteration were taken at its top)
#pragma unroll
            for (int j = 0; j < 6; ++j) R[j] = R[j + 2];
            R[6] = Y;
            R[7] = L;
        }

        // ---- untangle + magnitude: bins k = u + 128 q3 and 2048 - k.  Z[2048 - k] is register 15 - q3 of thread 128 - u (row 7 - q3);
        // thread 0 holds its own partners one row up (register 16 - q3), and spends its q3 = 0 slot -- DC and Nyquist are not outputs
        // (fft.rs:81) -- on the self-paired bin 1024: Z = P = its register 8, twiddle -i (twu[0][0])
        const int pcol = F * 128 + (u == 0 ? 256 : 128 - u);
        float m1[8], m2[8];
        // rows: every bin pair is stored as soon as it is computed (SGX_K1R_INTERLEAVE; stft4096_wg.hip does the same for (l, r) rows):
        // sixteen stores spread over the untangle instead of one burst behind it
        constexpr bool kInterleave = MODE != kPixels && SLIDE && SGX_K1R_INTERLEAVE;   // (same device, 1e6 frames: H 256 3.64 / 3.87 / 3.84 -> 3.56 / 3.82 / 3.81 ms; any other hop -- eight column loads in flight around the stores -- 4.02 -> 4.29: the burst stays there)
        const long long row = (long long)(F == 0 ? la : lb) * (long long)kM * kBin - kBin;      // byte of the (absent) bin 0
        const __amdgpu_buffer_rsrc_t r = out_rsrc(out, MODE != kPixels ? row : 0, MODE != kPixels && (F == 0 || have_b));
        const int l1 = kBin * u, l2 = kBin * (1152 - u);                                        // bins u + 128 q3 ; 2048 - u - 128 q3 = (1152 - u) + 128 (7 - q3)
        auto store_pair = [&](const int q3) {
            // bin k at byte kBin (k - 1).  Thread 0's q3 = 0 slot: bin 1024 from m1, and a second copy of it where its m2 would go.
            const float a = m1[q3], b = (q3 == 0 && u == 0) ? m1[0] : m2[q3];
            const int o1 = (q3 == 0 && u == 0) ? kBin * 1024 : l1;
            const int o2 = (q3 == 0 && u == 0) ? kBin * 128 : l2;                               // 128 + 128 * 7 = 1024
            const int s1 = kBin * 128 * q3, s2 = kBin * 128 * (7 - q3);
            if (F16) {
                const __half2 ha = __floats2half2_rn(a, a), hb = __floats2half2_rn(b, b);
                __builtin_amdgcn_raw_buffer_store_b32(*reinterpret_cast<const uint32_t *>(&ha), r, o1 + (s1 & 2047), s1 & ~2047, 0);
                __builtin_amdgcn_raw_buffer_store_b32(*reinterpret_cast<const uint32_t *>(&hb), r, o2 + (s2 & 2047), s2 & ~2047, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(a), __float_as_uint(a)}, r, o1 + (s1 & 2047), s1 & ~2047, 0);
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(b), __float_as_uint(b)}, r, o2 + (s2 & 2047), s2 & ~2047, 0);
            }
        };
        float2 pvs[8];
        if (kInterleave) {
#pragma unroll
            for (int q3 = 0; q3 < 8; ++q3) pvs[q3] = buf[(7 - q3) * 256 + pcol];
        }
#pragma unroll
        for (int q3 = 0; q3 < 8; ++q3) {
            const int pos = FFT16_OUT[q3];
            float zr_ = xr[pos], zi_ = xi[pos];
            float2 pv = kInterleave ? pvs[q3] : buf[(7 - q3) * 256 + pcol];
            if (q3 == 0) {
                const float2 own = buf[F * 128];              // thread 0 of the frame: its register 8 as published
                const int p8 = FFT16_OUT[8];
                zr_ = u == 0 ? xr[p8] : zr_; zi_ = u == 0 ? xi[p8] : zi_;
                pv.x = u == 0 ? own.x : pv.x; pv.y = u == 0 ? own.y : pv.y;
            }
            const float er = zr_ + pv.x, ei = zi_ - pv.y;    // Z + conj P   = 2 E
            const float orr = zi_ + pv.y, oi = pv.x - zr_;    // -i (Z - conj P) = 2 O
            const float wr = fmaf(twu[q3].x, orr, -(twu[q3].y * oi)), wi = fmaf(twu[q3].x, oi, twu[q3].y * orr);
            const float ar = er + wr, ai = ei + wi, br = er - wr, bi = ei - wi;
            m1[q3] = __builtin_amdgcn_sqrtf(fmaf(ar, ar, ai * ai));   // |S[k]| 2 / W      (the scale rides on the window)
            m2[q3] = __builtin_amdgcn_sqrtf(fmaf(br, br, bi * bi));   // |S[2048 - k]| 2 / W
            if (kInterleave) {
                store_pair(q3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- store row [M][2] (or half pairs): straight-line code: the wait for the prefetched rows below is then vmcnt(stores issued since)
        if (MODE != kPixels) {
            if (!kInterleave) {
#pragma unroll
                for (int q3 = 0; q3 < 8; ++q3) store_pair(q3);
            }
        } else {
            // ---- fused pixel columns (simple_spectrogram.rs:141-161): the pair's magnitudes go to LDS as float2 per bin -- (frame A,
            // frame B), each half written by its own frame's threads -- and K1's two pixel passes render both columns
            float2 *mpair = buf;                           // the padded column [bin] (wg::kColSlots)
            float2 *vbuf = mpair + wg::kColSlots;          // [sample slot]
            float *mcol = reinterpret_cast<float *>(buf) + F;
            lds_barrier();  // partner and exchange reads done: the image can be overwritten
            SGX_STAMP(13)   // (pixels) barrier 6
            // the sample pass's table words (nine per thread, the same in every iteration but eighteen registers nobody has to spare across
            // the transform) are requested as the column writes free m1 / m2, one word behind each pair of writes: their L1 / L2 latency
            // falls on the rest of the writes and on the barrier instead of on the head of the pass
            constexpr bool kEarlyWords = SGX_SAMPLE_EARLY && PIX != wg::kPixGeneric;   // (the generic instantiation -- interpolator and LUT walk at run time -- has no register left: one spill)
            wg::SampleWords sw;
            const __amdgpu_buffer_rsrc_t rt_words = wg::pcm_rsrc(reinterpret_cast<const float *>(p.samples));
            int tid_w = tid;
            asm volatile("" : "+v"(tid_w));   // (opaque: the nine table offsets are not to be hoisted out of the loop over the transforms)
#pragma unroll
            for (int q3 = 0; q3 < 8; ++q3) {
                const bool self = q3 == 0 && u == 0;       // thread 0's bin-1024 slot: m1 twice
                const int k1 = self ? 1024 : u + 128 * q3, k2 = self ? 1024 : 2048 - u - 128 * q3;
                mcol[2 * k1] = m1[q3];
                mcol[2 * k2] = self ? m1[0] : m2[q3];
                if (q3 == 0 && u == 1) {                   // this thread holds bin 1 (m1[0]) and bin 2047 (m2[0]): the repeats around the column
                    mcol[0] = m1[0];
                    mcol[2 * (kM + 1)] = mcol[2 * (kM + 2)] = m2[0];
                }
                if (kEarlyWords) {
                    sw.se[q3] = wg::sample_word(rt_words, p, (uint32_t)tid_w + 256u * q3);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (kEarlyWords) sw.se[8] = wg::sample_word(rt_words, p, (uint32_t)tid_w + 2048u);
            static_assert(wg::kSampleSteps == 9 && wg::kSampleEarly == 9, "one word per column-write step and one behind them");
            SGX_STAMP(14)   // (pixels) column writes
            lds_barrier();
            SGX_STAMP(15)   // (pixels) barrier 7
            if (kEarlyWords) wg::sample_pass_with<PIX>(p, mpair, vbuf, tid, sw);
            else wg::sample_pass<PIX>(p, mpair, vbuf, tid);
            // the fused pixel path requests the next iteration's load HERE, straight into L (dead since the slide): requested in front
            // of the exchange like the rows' it is two more live registers through the sample pass -- two spills, and a spill reload
            // is a vector-memory load the compiler waits for with vmcnt(0), this load included (same device: 3.79 -> 3.68 ms)
            if (SLIDE) L = column(columns_from(128 * (fa + 2) + 1152), 0);
            SGX_STAMP(16)   // (pixels) sample pass
            lds_barrier();
            SGX_STAMP(17)   // (pixels) barrier 8
            uchar4 *rgba = reinterpret_cast<uchar4 *>(p.rgba);
            __builtin_amdgcn_s_setprio(3);
            wg::row_pass<true, PIX>(p, row_words, vbuf, rgba + la * (size_t)p.R, rgba + lb * (size_t)p.R, true, have_b, pal, tid);
        }
        if (SLIDE && MODE != kPixels) {
            // `Ln` is pinned behind the row stores: its copy into L needs the load complete, and scheduled in front of the stores (where
            // the compiler had put it) that is a vmcnt(0) in the middle of the iteration; here it is vmcnt(stores since).  (The load
            // straight into L behind the slide, no second pair: 4 % slower on the same device.)
            asm volatile("" : "+v"(Ln.x), "+v"(Ln.y));
            L = Ln;
        }
        SGX_STAMP(18)   // rows: 16 stores issued + wait for the next column; pixels: the row pass (sums, dB, LUT, pixel stores)
    }
#if SGX_STAMPS
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 20; ++i) atomicAdd(&g_phase_cycles_r[i], st_acc[i]);
        atomicAdd(&g_phase_cycles_r[20], (unsigned long long)(job_end - job_begin));
    }
#endif
}

struct RealTables {
    float2 *d_tw1 = nullptr, *d_tw2 = nullptr, *d_twu = nullptr;
};

}  // namespace wgr

#if SGX_STAMPS
extern "C" __attribute__((visibility("default"))) int sgx_debug_phase_cycles_r(unsigned long long *h_out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(h_out, HIP_SYMBOL(wgr::g_phase_cycles_r), sizeof(unsigned long long) * 24);
    if (e == hipSuccess && reset) {
        unsigned long long zero[24] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(wgr::g_phase_cycles_r), zero, sizeof(zero));
    }
    return e == hipSuccess ? 0 : -1;
}
#endif

hipError_t real4096_init(sgx_ctx *c, void **out)
{
    using namespace wgr;
    (void)c;
    auto *t = new RealTables();
    auto unit = [](unsigned long long e, unsigned long long n) {   // e^{-2 pi i e / n}, exact on the axes
        e %= n;
        if (e == 0) return make_float2(1.0f, 0.0f);
        if (4 * e == n) return make_float2(0.0f, -1.0f);
        if (2 * e == n) return make_float2(-1.0f, 0.0f);
        if (4 * e == 3 * n) return make_float2(0.0f, 1.0f);
        const double ang = -2.0 * M_PI * (double)e / (double)n;
        return make_float2((float)cos(ang), (float)sin(ang));
    };
    std::vector<float2> tw1(8 * 256), tw2(256), twu(8 * 128);
    for (int q = 0; q < 8; ++q)
        for (int tt = 0; tt < 256; ++tt) tw1[q * 256 + tt] = unit((unsigned long long)tt * q, 2048);
    for (int q = 0; q < 16; ++q)
        for (int t0 = 0; t0 < 16; ++t0) tw2[q * 16 + t0] = unit((unsigned long long)t0 * q, 256);
    for (int q3 = 0; q3 < 8; ++q3)
        for (int uu = 0; uu < 128; ++uu) twu[q3 * 128 + uu] = unit((unsigned long long)(uu + 128 * q3), 4096);
    twu[0] = unit(1024, 4096);   // thread 0, q3 = 0: the self-paired bin 1024
    auto up = [](float2 **dst, const std::vector<float2> &v) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), v.size() * sizeof(float2));
        if (e == hipSuccess) e = hipMemcpy(*dst, v.data(), v.size() * sizeof(float2), hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(&t->d_tw1, tw1);
    if (e == hipSuccess) e = up(&t->d_tw2, tw2);
    if (e == hipSuccess) e = up(&t->d_twu, twu);
    if (e != hipSuccess) {
        real4096_destroy(t);
        return e;
    }
    *out = t;
    return hipSuccess;
}

void real4096_destroy(void *tables)
{
    auto *t = static_cast<wgr::RealTables *>(tables);
    if (!t) return;
    if (t->d_tw1) (void)hipFree(t->d_tw1);
    if (t->d_tw2) (void)hipFree(t->d_tw2);
    if (t->d_twu) (void)hipFree(t->d_twu);
    delete t;
}

// the streams this kernel serves: one channel, W = 2048 (H = 256 slides its window in registers)
bool real4096_serves(const sgx_ctx *c, const float *d_pcm, uint32_t channels)
{
    (void)d_pcm;   // any hop, any (4-byte) alignment of the stream: see samples_from
    return channels == 1 && c->d_real && c->W == (uint32_t)wgr::kW;
}

namespace wg {

// p: as launch_wg (stft4096_wg.hip) fills it for a one-channel stream -- stream, window, tw2, output, the pixel tables; the
// transform's own tables and the job split are set here
hipError_t launch_real4096(const sgx_ctx *c, const void *real_tables, Params p, bool out_f16, bool render)
{
    using namespace wgr;
    if (p.n_frames == 0) return hipSuccess;
    const auto *t = static_cast<const RealTables *>(real_tables);
    p.tw1 = t->d_tw1;
    p.tw2 = t->d_tw2;
    p.twu = t->d_twu;
    p.stream_samples = p.total_frames ? (p.total_frames - 1) * (unsigned long long)c->H + (unsigned long long)kW : 0;   // what the stream's frames cover (the caller may hold more)
    p.n_jobs = (p.n_frames + 1) / 2;
    unsigned long long blocks = (unsigned long long)c->n_cu * 4;   // persistent workgroups, four per CU, each a contiguous run of frame pairs
    unsigned long long per = (p.n_jobs + blocks - 1) / blocks;
    if (per < 1) per = 1;
    blocks = (p.n_jobs + per - 1) / per;
    p.jobs_per_block = per;
    const dim3 grid((unsigned)blocks), block(256);
    auto launch = [&](auto slide_c) {
        constexpr bool S_ = decltype(slide_c)::value;
        constexpr size_t lds_rows = (S_ && SGX_ADDTID_R) ? kLdsBytesR : kLdsBytes, lds_render = (S_ && SGX_ADDTID_R) ? kLdsBytesRenderR : kLdsBytesRender;
        if (render) {
            if (!p.seed_pm1) hipLaunchKernelGGL((stft4096_real_kernel<kPixels, kPixGeneric, S_>), grid, block, lds_render, c->stream, p);
            else if (p.interp == SGX_INTERP_COSINE) hipLaunchKernelGGL((stft4096_real_kernel<kPixels, kPixCosine, S_>), grid, block, lds_render, c->stream, p);
            else hipLaunchKernelGGL((stft4096_real_kernel<kPixels, kPixCubic, S_>), grid, block, lds_render, c->stream, p);
        } else if (out_f16) {
            hipLaunchKernelGGL((stft4096_real_kernel<kRowsF16, kPixNone, S_>), grid, block, lds_rows, c->stream, p);
        } else {
            hipLaunchKernelGGL((stft4096_real_kernel<kRowsF32, kPixNone, S_>), grid, block, lds_rows, c->stream, p);
        }
    };
    if (c->H == 256) launch(std::true_type{});
    else launch(std::false_type{});
    return hipGetLastError();
}

}  // namespace wg

}  // namespace sgx
