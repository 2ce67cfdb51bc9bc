// stft4096_wgp.hip -- the workgroup-per-transform 4096-point STFT (see stft4096_wg.hip for the
// algorithm) with its arithmetic on packed (re, im) pairs.
//
// A complex value lives in one aligned VGPR pair, so a complex add is ONE v_pk_add_f32, the
// radix-4 rotations (+-i) are the same instruction with op_sel / neg modifiers, and a twiddle
// multiply is v_pk_mul_f32 + v_pk_fma_f32.  Measured on MI355X (profiles/r01_microbench.txt) a
// packed op costs ~1.5x a scalar op at 4 waves/SIMD while doing the work of two: the FFT passes
// drop from ~690 to ~360 VALU instructions per thread and transform.  hipcc does not fold the
// swap-and-negate of a rotation or of a variable twiddle into operand modifiers (it emits
// v_mov + v_xor pairs), hence the three one-instruction asm helpers below; everything else is
// ordinary vector arithmetic that hipcc selects packed instructions for.
#include "stft4096_wg.hpp"

namespace sgx {

namespace wgp {

using namespace sgx::wg;

typedef float f2v __attribute__((ext_vector_type(2)));

// a - i b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ f2v cx_add_mi(f2v a, f2v b)
{
    f2v r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ f2v cx_add_pi(f2v a, f2v b)
{
    f2v r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// v * w for a twiddle held in registers: t = (-v.y w.y, v.y w.x); r = (v.x w.x, v.x w.y) + t
__device__ __forceinline__ f2v cx_mul_v(f2v v, f2v w)
{
    f2v t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(v), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(v), "v"(w), "v"(t));
    return r;
}
// v * (c + i d) for a literal twiddle (constants land in SGPR pairs)
__device__ __forceinline__ f2v cx_mul_k(f2v v, float c, float d)
{
    const f2v t = f2v{v.y, v.y} * f2v{-d, c};
    return __builtin_elementwise_fma(f2v{v.x, v.x}, f2v{c, d}, t);
}
__device__ __forceinline__ f2v cx_mul_mi(f2v v) { return f2v{v.y, -v.x}; }
__device__ __forceinline__ f2v cx_mul_pi(f2v v) { return f2v{-v.y, v.x}; }

__device__ __forceinline__ float cl_fma(float a, float c, float u) { return fmaf(a, c, u); }
__device__ __forceinline__ f2v cl_fma(f2v a, float c, f2v u) { return __builtin_elementwise_fma(a, f2v{c, c}, u); }
#include "fft_codelets.inc"     // FFT8_OUT / FFT16_OUT (same radices, same output permutation)
#include "fft_codelets_cx.inc"

template <bool MONO, int PAIRING, bool C2, bool RENDER>
__global__ void __launch_bounds__(256, 4) stft4096_wgp_kernel(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *buf = reinterpret_cast<float2 *>(smem_raw);
    float2 *tw2 = buf + kBufComplex;

    uint2 *pal = reinterpret_cast<uint2 *>(tw2 + 256);          // RENDER only: [256] {threshold, RGBA} (pixel_for)

    const int tid = threadIdx.x;
    tw2[tid] = p.tw2[tid];
    uint32_t row_words[4] = {0u, 0u, 0u, 0u};  // RENDER: the table words of this thread's rows tid + 256 i
    if (RENDER) {
        pal[tid] = make_uint2(__float_as_uint(tid < 255 ? p.lut_thr[tid] : __builtin_nanf("")), *reinterpret_cast<const uint32_t *>(&p.lut_rgba[tid]));
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((uint32_t)tid + 256u * i < p.R) row_words[i] = p.rows[tid + 256 * i];
    }

    // per-thread constants, kept in registers for the life of the (persistent) workgroup
    float win[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) win[a] = p.window[tid + 256 * a];
    f2v tw1[16];
#pragma unroll
    for (int q = 1; q < 16; ++q) { const float2 t = p.tw1[q * 256 + tid]; tw1[q] = f2v{t.x, t.y}; }

    const int q1_2 = tid >> 4, t0_2 = tid & 15;                 // pass-2 role
    const float inv_w = 1.0f / (float)kW;                       // (hypot / 2) * (2 / W)
    __syncthreads();

    const unsigned long long job_begin = (unsigned long long)blockIdx.x * p.jobs_per_block;
    unsigned long long job_end = job_begin + p.jobs_per_block;
    if (job_end > p.n_jobs) job_end = p.n_jobs;

    // Software pipeline: the samples of transform j+1 are requested while transform j is still in
    // its FFT passes, i.e. BEFORE j's magnitude stores.  vmcnt retires in issue order, so a load
    // issued after 16-32 stores would have to wait for all of them to reach memory first.
    float sa[(MONO && PAIRING == kPairAdjacentRow) ? 9 : 8], sb[8];
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
#pragma unroll
                    for (int a = 0; a < 7; ++a) sa[a] = sa[a + 2];
                    sa[7] = s0[tid + 256 * 7];
                } else {
#pragma unroll
                    for (int a = 0; a < 8; ++a) sa[a] = s0[tid + 256 * a];
                }
                sa[8] = second ? s0[tid + 256 * 8] : 0.0f;
            } else {
                const float *s1 = second ? p.pcm + fb * p.H : s0;
#pragma unroll
                for (int a = 0; a < 8; ++a) { sa[a] = s0[tid + 256 * a]; sb[a] = s1[tid + 256 * a]; }
            }
        } else {
            const float *s0 = p.pcm + (p.first_frame + job) * p.H * p.C;
            if (C2) {
                // (a sliding register window like the mono one was tried here: stereo input is not traffic-bound and
                // the extra live registers cost more than the saved row loads)
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const float2 v = reinterpret_cast<const float2 *>(s0)[tid + 256 * a];
                    sa[a] = v.x; sb[a] = v.y;
                }
            } else {
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const size_t e = (size_t)(tid + 256 * a) * p.C;
                    sa[a] = s0[e + p.pair_l]; sb[a] = s0[e + p.pair_r];
                }
            }
        }
    };
    if (job_begin < job_end) fetch(job_begin, false);

    for (unsigned long long job = job_begin; job < job_end; ++job) {
        // ---- Hann (fft.rs:53-63) on the prefetched samples
        f2v e[8];
        // local (output) frame indices; for mono f0 may be -1 (the pair's first frame precedes the range)
        const long long f0 = MONO ? (long long)(2 * (p.pair_base + job)) - (long long)p.first_frame : (long long)job;
        const long long f1 = f0 + 1;
        const bool have_first = !MONO || f0 >= 0;
        const bool data_second = !MONO || (unsigned long long)(f1 + (long long)p.first_frame) < p.total_frames;
        const bool have_second = !MONO || f1 < (long long)p.n_frames;  // ... but stored only inside the requested range
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            e[a].x = sa[a] * win[a];
            if (MONO && PAIRING == kPairAdjacentRow) e[a].y = data_second ? sa[a + 1] * win[a] : 0.0f;
            else e[a].y = data_second ? sb[a] * win[a] : 0.0f;
        }
        const int col = tid;                           // pass-3 / output column of this thread
        const int pcol = col == 0 ? 256 : 256 - col;   // partner column (column 0 is its own partner, one row up)

        // ---- pass 1: 16-point DFT over a, inputs a >= 8 are the zero padding:
        //      even q1 = FFT8(z), odd q1 = FFT8(z * w_16^a)
        f2v o[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) o[a] = e[a];
        pretwiddle8c_w16(o);
        fft8c(e);
        fft8c(o);

        lds_barrier();  // the previous transform's partner reads are complete
        f2v *cbuf = reinterpret_cast<f2v *>(buf);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pos = FFT8_OUT[j];
            cbuf[(2 * j) * kS1 + tid] = j == 0 ? e[pos] : cx_mul_v(e[pos], tw1[2 * j]);
            cbuf[(2 * j + 1) * kS1 + tid] = cx_mul_v(o[pos], tw1[2 * j + 1]);
        }
        lds_barrier();

        // ---- pass 2: thread (q1, t0): 16-point FFT over t1, then twiddle w_256^{t0 q2}
        f2v x[16];
#pragma unroll
        for (int t1 = 0; t1 < 16; ++t1) x[t1] = cbuf[q1_2 * kS1 + t0_2 + 16 * t1];
        fft16c(x);
        lds_barrier();  // everyone has read image 1
        const f2v *ctw2 = reinterpret_cast<const f2v *>(tw2);
#pragma unroll
        for (int q2 = 0; q2 < 16; ++q2) {
            const int pos = FFT16_OUT[q2];
            cbuf[t0_2 * kS2 + q1_2 + 16 * q2] = q2 == 0 ? x[pos] : cx_mul_v(x[pos], ctw2[q2 * 16 + t0_2]);
        }
        lds_barrier();

        // ---- pass 3: thread u owns column col = q1 + 16 q2: 16-point FFT over t0 -> bins k = col + 256 q3
#pragma unroll
        for (int t0 = 0; t0 < 16; ++t0) x[t0] = cbuf[t0 * kS2 + col];
        fft16c(x);
        if (job + 1 < job_end) fetch(job + 1, true);  // ahead of this transform's stores (see above)
        lds_barrier();  // everyone has read image 2
        // partner exchange: publish q3 = 8..15 (the bins P-k of the kept half)
#pragma unroll
        for (int j = 0; j < 8; ++j) cbuf[j * 256 + col] = x[FFT16_OUT[8 + j]];
        lds_barrier();

        // ---- split + magnitude (fft.rs:81-98)
        float ml[8], mr[8];
#pragma unroll
        for (int q3 = 0; q3 < 8; ++q3) {
            // F[P-k]: column 256-col holds it as q3' = 15 - q3 (row 7 - q3); column 0 as q3' = 16 - q3
            const f2v b = cbuf[(7 - q3) * 256 + pcol];
            const f2v a = x[FFT16_OUT[q3]];
            const f2v sp = a + f2v{b.x, -b.y};   // a + conj(b) = 2 L^
            const f2v sq = a - f2v{b.x, -b.y};   // a - conj(b) = 2i R^
            const f2v pp = sp * sp, qq = sq * sq;
            ml[q3] = __builtin_amdgcn_sqrtf(pp.x + pp.y) * inv_w;
            mr[q3] = __builtin_amdgcn_sqrtf(qq.x + qq.y) * inv_w;
        }

        if (!RENDER) {
            // ---- store [F][pairs][M][2]: uniform row base (SGPR) + one 32-bit lane offset
            if (p.out_f16) {
                char *base = reinterpret_cast<char *>(p.mags);
                const long long row0 = (long long)(((size_t)(have_first ? f0 : 0) * p.pairs + p.pair) * (size_t)kM * 4) - 4;
                if (MONO) {
                    if (have_first) store_row_f16<true>(base, row0, col, ml, ml);
                    if (have_second) store_row_f16<true>(base, (long long)((f1 * p.pairs + p.pair) * (size_t)kM * 4) - 4, col, mr, mr);
                } else {
                    store_row_f16<false>(base, row0, col, ml, mr);
                }
            } else {
                // byte offset of bin k = 0 of the row (bin k lives 8 k bytes on; k = 0 is never stored)
                char *base = reinterpret_cast<char *>(p.mags);
                const long long row0 = (long long)((((size_t)(have_first ? f0 : 0) * p.pairs + p.pair) * (size_t)kM) * 8) - 8;
                if (MONO) {
                    if (have_first) store_row<true>(base, row0, col, ml, ml);
                    if (have_second) store_row<true>(base, (long long)(((f1 * p.pairs + p.pair) * (size_t)kM) * 8) - 8, col, mr, mr);
                } else {
                    store_row<false>(base, row0, col, ml, mr);
                }
            }
        } else {
            // ---- fused pixel column(s): magnitude_in -> color_for -> put_pixel
            //      (simple_spectrogram.rs:141-161), magnitudes staged in LDS only
            float2 *m2 = reinterpret_cast<float2 *>(buf);  // [bin - 1]: (l, r), or for mono (frame f0, frame f0 + 1)
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
            sample_pass<kPixGeneric>(p, m2, vbuf, tid);   // (the A/B twin keeps the run-time switches)
            lds_barrier();
            uchar4 *rgba = reinterpret_cast<uchar4 *>(p.rgba);
            uchar4 *dst_a = rgba + ((size_t)(have_first ? f0 : 0) * p.pairs + p.pair) * (size_t)p.R;
            uchar4 *dst_b = rgba + ((size_t)f1 * p.pairs + p.pair) * (size_t)p.R;
            row_pass<MONO, kPixGeneric>(p, row_words, vbuf, dst_a, dst_b, have_first, have_second, pal, tid);
        }
    }
}

}  // namespace wgp

namespace {

template <bool RENDER>
hipError_t launch_wgp(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                      size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags, uint8_t *d_rgba, bool out_f16 = false)
{
    using namespace wg;
    using namespace wgp;
    if (n_frames == 0) return hipSuccess;
    const auto *t = static_cast<const WgTables *>(tables);
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
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
        if (RENDER) {
            p.rows = t->d_rows;
            p.samples = t->d_samples;
            p.n_samples = t->n_samples;
            p.lut_thr = c->d_lut_thr;
            p.lut_rgba = c->d_lut_rgba;
            p.rgba = d_rgba;
            p.R = c->R;
            p.interp = c->cfg.interp;
            const double span = (double)c->cfg.max_db - (double)c->cfg.min_db;
            const double n = c->cfg.lut_index_mode == SGX_LUT_ROUND_NM1 ? 255.0 : 256.0;
            p.guess_a = (float)(10.0 * log10(2.0) * n / span);
            p.guess_b = (float)(-(double)c->cfg.min_db * n / span + (c->cfg.lut_index_mode == SGX_LUT_ROUND_NM1 ? 0.5 : 0.0));
            p.seed_pm1 = wg4096_seed_is_within_one(c) ? 1u : 0u;
            p.single_rows = t->single_rows;
        }
        // mono normally rides two frames per transform; SGX_FLAG_INDEPENDENT_FRAMES runs it as (s, s) pairs
        const bool mono = channels == 1 && !(c->cfg.flags & SGX_FLAG_INDEPENDENT_FRAMES);
        p.pair_base = mono ? first_frame / 2 : 0;
        p.n_jobs = mono ? (first_frame + n_frames + 1) / 2 - first_frame / 2 : n_frames;
        unsigned long long blocks = (unsigned long long)n_cu * 4;
        unsigned long long per = (p.n_jobs + blocks - 1) / blocks;
        if (per < 1) per = 1;
        blocks = (p.n_jobs + per - 1) / per;
        p.jobs_per_block = per;
        const dim3 grid((unsigned)blocks), block(256);
        const size_t lds = RENDER ? kLdsBytesRender : kLdsBytes;
        if (mono) {
            if (c->H == 256) hipLaunchKernelGGL((stft4096_wgp_kernel<true, kPairAdjacentRow, false, RENDER>), grid, block, lds, c->stream, p);
            else hipLaunchKernelGGL((stft4096_wgp_kernel<true, kPairAdjacent, false, RENDER>), grid, block, lds, c->stream, p);
        } else if (channels == 2) {
            hipLaunchKernelGGL((stft4096_wgp_kernel<false, kPairAdjacent, true, RENDER>), grid, block, lds, c->stream, p);
        } else {
            hipLaunchKernelGGL((stft4096_wgp_kernel<false, kPairAdjacent, false, RENDER>), grid, block, lds, c->stream, p);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

hipError_t launch_stft_wgp4096(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                               size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags)
{
    return launch_wgp<false>(c, tables, d_pcm, channels, pairs, first_frame, n_frames, total_frames, d_mags, nullptr);
}

hipError_t launch_render_wgp4096(const sgx_ctx *c, const void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                                 size_t first_frame, size_t n_frames, size_t total_frames, uint8_t *d_rgba)
{
    return launch_wgp<true>(c, tables, d_pcm, channels, pairs, first_frame, n_frames, total_frames, nullptr, d_rgba);
}

}  // namespace sgx
