// stft16384_d.hip -- tuned STFT for W = 8192 (P = 16384), third design: ONE 1024-thread workgroup per transform, the transform cut
// into four 4096-point transforms by decimation in TIME, one per lane of a lane quad, recombined across the quad with DPP.
// (BASELINE config 4: 16384-point, hop 512, 8 interleaved channels = 4 (l, r) pairs per hop position.)
//
// Replaces FastFourierTransform::process (fft.rs:43-99) + the hop loop (audio_transform.rs:34-42).
//
// Why.  The second design (stft16384_q.hip) decimates the OUTPUT by four: residues (0, 2) in one pass of a 512-thread workgroup,
// (1, 3) in a second pass over the same samples.  A lane's 16 output bytes per row are one 8-byte bin of each pass, so the even
// bins wait in a parked buffer (1.53 x write amplification, 1.88 x total traffic: profiles/r02_hbm_traffic.json), the samples
// and the window are read twice, and the pass-1 twiddles (a different 128 bytes per lane for each pass) are fetched from L2
// sixteen times per transform BEHIND the previous pass's stores (vmcnt retires in order).  Decimating the INPUT by four instead,
//
//     F[j + 4096 c] = sum_{r < 4} (-i)^{r c} w_16384^{r j} G_r[j],     G_r[j] = sum_{m < 2048} z[4 m + r] w_4096^{m j},
//     z[n] = (l[n] + i r[n]) hann[n]   (n >= 8192 is the zero padding, fft.rs:65-69: only m < 2048 is non-zero),
//
// every G_r is EXACTLY the transform the headline kernel computes (4096 points, 2048 of them non-zero: stft4096_wg.hip), all four
// are in flight at once in the four lanes of a quad (lane tid = 4 col + r plays thread col of that kernel for residue r), and the
// quad holds the four consecutive outputs... no: the four outputs j + 4096 c of one j -- kept bins and their partners P - k alike.
// Consequences:
//   * a lane reads the samples n = tid + 1024 a, a < 8: every sample once per transform, 1024 consecutive samples per instruction;
//   * one pass per transform: the pass-1 twiddles w_16384^{q1 tid} (the 4096-point kernel's w_4096^{col q1} times the first part
//     of the recombination twiddle) and the Hann factors stay in registers for the life of the workgroup: no table traffic at all;
//   * nothing is parked: after the recombination and the partner exchange every lane holds 8 finished (l, r) bins of one row;
//   * 139 KB of LDS for the four interleaved images (element (index, r) at 4 index + r: every access pattern of the 4096-point
//     kernel stays conflict-free with four times the stride), so ONE workgroup per CU, 16 waves = 4 per SIMD.
//
//   sample index  n = tid + 1024 a      (a < 8; m = col + 256 a, r = tid & 3)
//   pass 1  lane (col, r)    : 16-point DFT over a (8 non-zero inputs = two 8-point FFTs) -> q1; twiddle w_16384^{q1 tid}
//   pass 2  lane (q1, t0, r) : col = t0 + 16 t1; 16-point FFT over t1 -> q2; twiddle w_1024^{q2 (4 t0 + r)}   (LDS, 8 KB)
//   pass 3  lane (q1, q2, r), u = q1 + 16 q2 : 16-point FFT over t0 -> q3: G_r[j] w_16384^{r (j mod 256)}, j = u + 256 q3;
//           twiddle w_64^{r q3} (LDS) completes w_16384^{r j}.  WAVE q1 does both: pass 2 and pass 3 of one q1 exchange through row
//           q1 of the image, which no other wave touches in between -- no workgroup barrier from the end of pass 1 to the publish
//           (round 3: -6 % of the launch; until then wave q2 ran pass 3 for all q1, two more barriers).  A wave therefore holds bins
//           16 apart; rows leave through the LDS staging area anyway, so only LDS index maps changed (partner slots at a stride of 68
//           per q2, staged row slots swizzled: stage_slot)
//   combine radix-4 across the quad (two DPP stages): lane L ends with F[j + 4096 c]: c = {0, 2, 1, 3}[L] in its registers q3 < 8
//           and 3 - c in its registers q3 >= 8 -- so every lane holds 8 kept bins (k < 8192) and 8 partners, and the partner of
//           its kept (u, q3) is register 15 - q3 of the SAME lane of quad 256 - u: the 4096-point kernel's exchange, unchanged
//   split   (fft.rs:81-98) 8 bins per lane: k = {0, 6144, 4096, 2048}[L] + u + 256 qq
#include <type_traits>

#include "sgx_internal.hpp"

namespace sgx {

namespace d16k {

typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float cl_fma(float a, float c, float u) { return fmaf(a, c, u); }
__device__ __forceinline__ f2v cl_fma(f2v a, float c, f2v u) { return __builtin_elementwise_fma(a, f2v{c, c}, u); }

#include "fft_codelets.inc"

constexpr int kW = 8192, kP = 16384, kM = 8191;
constexpr uint32_t kXcdHint = 8;         // XCDs of an MI355X (SPX): workgroup i runs on XCD i % 8.  Used for job locality only
constexpr int kS1 = 272;                 // row stride of the pass-1 -> pass-2 image [q1][col]   (as stft4096_wg.hpp)
// (the pass-2 -> pass-3 image lives in the rows of the first: element (q2, t0, r) of row q1 at 68 q2 + 4 t0 + r)
constexpr int kImg = 16 * kS1;           // indices per residue (4352)
constexpr int kBufComplex = 4 * kImg;    // 17 408 complex = 139 264 B, the four residues interleaved: (index, r) at 4 index + r
constexpr size_t kLdsBytes = (size_t)(kBufComplex + 1024 + 64) * sizeof(float2);   // + tw2 [q2][t0][r], tw3 [q3][r]

struct Params {
    const float *pcm;        // MONO: [n] floats; DIRECT: the interleaved stream [n][C]; else per-pair planes of (l, r): plane p starts at pcm + p * plane_floats
    size_t plane_floats;
    uint32_t stride_floats;  // DIRECT: floats from one sample of a pair to the next (the stream's channel count)
    long long sample_base;   // absolute sample index of pcm[0] (the de-interleaved workspace holds a sub-range)
    const float2 *T1;        // [16][1024]  w_16384^{q1 tid} at [q1][tid]
    const float2 *tw2;       // [16][16][4] w_1024^{q2 (4 t0 + r)} at [q2][t0][r]
    const float2 *tw3;       // [16][4]     w_64^{r q3} at [q3][r]
    const float *win8;       // [2][1024][4] hann[tid + 1024 a] / W at [a / 4][tid][a % 4]   (fft.rs:61; the scale (hypot / 2) (2 / W) = 2^-13 rides along)
    float *mags;
    unsigned long long first_frame, n_frames, total_frames, pair_base, n_jobs, jobs_per_xcd;
    uint32_t xcds;           // kXcdHint when the grid is a multiple of it, else 1 (plain round-robin)
    uint32_t H, pairs;
};

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ float2 cmulf(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
// An 8-byte LDS read that stays one (stft4096_wg.hpp: a merged ds_read2_b64 / ds_read2st64_b64 holds the LDS pipe for 8 cycles, two
// ds_read_b64 for 2 each): the empty statement makes the base a new value for every read.
#ifndef SGX_NO_READ2
#define SGX_NO_READ2 55   // (every site but the recombination twiddles: there the separate reads cost two spilled registers)
#endif
typedef const f2v __attribute__((address_space(3))) lds_cfloat2;
__device__ __forceinline__ lds_cfloat2 *lds_ptr(const float2 *p) { return (lds_cfloat2 *)p; }
template <int SITE>   // (one bit of SGX_NO_READ2 per read site, for A/B)
__device__ __forceinline__ float2 lds_read_alone(lds_cfloat2 *&base, int idx)
{
    if ((SGX_NO_READ2 >> SITE) & 1) asm("" : "+v"(base));
    const f2v v = base[idx];
    return make_float2(v.x, v.y);
}
// the value of another lane of the quad
template <int PERM>
__device__ __forceinline__ float quad(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), PERM, 0xf, 0xf, true));
}
constexpr int kSwap2 = 0x4E;   // quad_perm [2, 3, 0, 1]
constexpr int kSwap1 = 0xB1;   // quad_perm [1, 0, 3, 2]

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// every global access goes through a raw buffer descriptor: a wave-uniform base (4 SGPRs) + ONE 32-bit lane offset + a scalar
// offset (no per-lane 64-bit pointers: they are what spills first, and a spill reload is a vector-memory load)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void *base)
{
    const unsigned long long a = (unsigned long long)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
#ifndef D_OUT_AUX
#define D_OUT_AUX 2   // nt: a write-once stream (same-device A/B: 0.80 -> 0.73 ms per 20 000 transforms)
#endif
// A 16-byte buffer store followed at once by a vector write of one of its data registers: the compiler's hazard table asks for
// wait states only when the store's scalar offset is an immediate; with the offset in an SGPR it inserts none, and on this device
// the word then stored was the NEW value (seen as LDS addresses inside stored rows: the address arithmetic of the next piece
// reuses the data registers of the piece just stored).  Two wait states after every such store.
__device__ __forceinline__ void store16(u32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int aux_is_compile_time_below)
{
    (void)aux_is_compile_time_below;
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, D_OUT_AUX);
    asm volatile("s_nop 2" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));   // (reads the data registers: the scheduler cannot move their rewrite in front of the wait states)
}
#ifdef D_ABL_NOSTORE
#define D_STORE_OK(v) ((v) == 12345.678f)   // ablation builds: (practically) never true, but the value stays live
#else
#define D_STORE_OK(v) true
#endif

// The staged row (round 4): EVEN bins (0-based) in one 32 KB region, ODD bins in a second one kRegion slots on, bin b at index b >> 1 of
// its region -- so that the two bins a lane takes back as one 16-byte store come from the same index of the two regions (one
// ds_read2st64_b64) whatever the index map -- and the index swizzled for the WRITES: a wave holds bins u = q1 + 16 q2 of ONE parity in
// four parts of the row 2048 bins apart, i.e. sixteen lanes = (q2 & 3) x (part): index bits 3, 4 and 10, 11.  Bit 4 folded onto bit 2
// and bits 10, 11 onto bits 0, 1 give the sixteen lanes sixteen different bank pairs; thirty-two consecutive readers permute among
// themselves.  (Round 3 kept neighbouring bins in neighbouring slots for a 16-byte read-back and swizzled around that: with one parity per
// wave that can do no better than 2-way on the writes; it did 3-way, the read-back 2-way -- the two sources, with the odd special case of
// wave 0's partner reads, of the 12 % of LDS cycles SQ_LDS_BANK_CONFLICT counted; the three FFT exchanges and the partner exchange are
// conflict-free under the bank rules of profiles/r04_k1_wg5_bound.txt.)
constexpr int kRegion = 4096;   // float2 slots per parity region: 32 768 B = 64 x 512 B (the unit of ds_read2st64_b64's offsets)
__device__ __forceinline__ int stage_index(int idx) { return idx ^ (((idx >> 4) & 1) << 2) ^ ((idx >> 10) & 3); }
__device__ __forceinline__ int stage_slot(int b) { return (b & 1) * kRegion + stage_index(b >> 1); }

// a mono sample range as one (s, s) plane: what the reference's capture callback does to a mono device (audio_input_list_model.rs:67-69)
__global__ void __launch_bounds__(256) duplicate_mono_kernel(const float *pcm, float *plane, size_t first, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float s = pcm[first + i];
        reinterpret_cast<float2 *>(plane)[i] = make_float2(s, s);
    }
}

#if SGX_STAMPS
// diagnostic build only (tools/k16_phases.py): per-phase wave cycles (s_memtime), summed over all waves and iterations
__device__ unsigned long long g_phase_cycles16[24];
#define SGX_STAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[i] += now_ - st_last; st_last = now_; }
#else
#define SGX_STAMP(i)
#endif

// DIRECT: more than two interleaved channels, pair p = channels (2 p, 2 p + 1), read where they lie: 8-byte loads at a stride of C floats.
// A wave's load then touches 64 x 4 C bytes instead of 512 contiguous ones (+4 % on this kernel at C = 8: 16 lines per load for 4), but
// the four pairs of a hop position run on CUs of one XCD at the same time and share the lines in its L2 -- and the pass that used to split
// the stream into (l, r) planes first (0.125 x the algorithmic bytes written and read back, 4.6 % of the launch, a workspace of the
// stream's size grown on the first call) is not needed: HBM traffic 1.15 -> 1.02 x algorithmic, nothing allocated on the call.  Same
// device, planes -> direct: 2.41-2.46 -> 2.47-2.50 ms per 80 000 transforms, 12.00-12.13 -> 11.87-12.06 ms at BASELINE config 4's own
// 400 000 (there the 1.6 GB of planes no longer stay in the memory-side cache).  The default; SGX_FLAG_CHANNEL_PLANES is the A/B.
template <bool MONO, bool DIRECT = false>
__global__ void __launch_bounds__(1024, 1) stft16384_d_kernel(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *buf = reinterpret_cast<float2 *>(smem_raw);
    float2 *tw2 = buf + kBufComplex;
    float2 *tw3 = tw2 + 1024;

    const int tid = threadIdx.x;
    tw2[tid] = p.tw2[tid];
    if (tid < 64) tw3[tid] = p.tw3[tid];

    // Per-lane constants: the fifteen pass-1 twiddles and the eight Hann factors stay in registers for the life of the (persistent)
    // workgroup -- no table traffic per transform.  (An earlier version re-requested the twiddles of q1 >= 8 and the Hann factors
    // with every transform's samples because, resident, they had been spilled; since the prefetched samples are consumed at the
    // END of the iteration that requests them -- `take` below -- the registers are there: 128 VGPRs, no scratch.)
    float2 tw1[16];
#pragma unroll
    for (int q = 1; q < 16; ++q) tw1[q] = p.T1[q * 1024 + tid];
    float win[8];
    {
        const __amdgpu_buffer_rsrc_t rw = uniform_rsrc(p.win8);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rw, 16 * tid, g * (1024 * 16), 0);
            win[4 * g] = __uint_as_float(w.x); win[4 * g + 1] = __uint_as_float(w.y);
            win[4 * g + 2] = __uint_as_float(w.z); win[4 * g + 3] = __uint_as_float(w.w);
        }
    }
    const int L = tid & 3;
    // signs of the two recombination stages (see `combine` below) and the lanes whose value is turned by -i in between
    const float sA = L < 2 ? 1.0f : -1.0f, sB = (L & 1) ? -1.0f : 1.0f;
    const bool rot_lo = L == 3, rot_hi = L == 1, odd = (L & 1) != 0;
    __syncthreads();

    // Software pipeline: the samples of the NEXT transform are requested before this transform's stores: vmcnt retires in issue
    // order, so a load issued behind the stores would wait for every one of them to be acknowledged.
    float pl[8], pr[8];
    struct JobIn { const float *base; bool data_second; };   // base: wave-uniform
    // (hop, pair) of a job: job = hop * pairs + pair.  Carried along the loop -- one 64-bit division at the start; `job / p.pairs`
    // inside `job_in` and at the loop head was two software divisions, ~220 scalar instructions, per transform and wave
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
    auto prefetch = [&](const JobIn &j) {
        const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(j.base);
        const int sec = j.data_second ? second_off : 0;
        const int lane_bytes = DIRECT ? (int)(4u * p.stride_floats) * tid : 8 * tid;                  // one sample of the pair per lane
        const int row_bytes = DIRECT ? (int)(4096u * p.stride_floats) : 8192;                        // 1024 samples on
#pragma unroll
        for (int a = 0; a < 8; ++a) {
#ifdef D_ABL_NOLOAD
            pl[a] = (float)(a + 1); pr[a] = (float)tid;
            (void)rs; (void)sec;
#else
            if (MONO) {
                pl[a] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, 4 * tid, 4096 * a, 0));
                pr[a] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, 4 * tid, 4096 * a + sec, 0));
            } else {
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_bytes, row_bytes * a, 0);
                pl[a] = __uint_as_float(v.x); pr[a] = __uint_as_float(v.y);
            }
#endif
        }
    };

    // The prefetched values are CONSUMED (Hann, fft.rs:53-63: the product is what the next transform starts from) at the END of the
    // iteration that requested them, behind its stores, in straight-line code: the compiler then waits for them with
    // `vmcnt(stores issued since)`.  Consumed at the head of the next iteration, the wait sits at the loop header, where the
    // entry path is merged in and the compiler falls back to vmcnt(0) -- every transform then waits for the previous one's stores
    // to be acknowledged (measured: 0.93 ms per 20 000 transforms against 0.61 ms without the stores).
    float er[8], ei[8];
    auto take = [&](bool data_second) {
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            er[a] = pl[a] * win[a];
            ei[a] = (MONO && !data_second) ? 0.0f : pr[a] * win[a];
        }
        // ... and PINNED there: the products are plain arithmetic, and the compiler sank them into the head of the next iteration,
        // behind its row stores (`flush`, under branches it cannot count through) -- where the wait for the samples became vmcnt(5) ..
        // vmcnt(0): every transform waited for the previous one's stores after all (round 4, read off the ISA)
#pragma unroll
        for (int a = 0; a < 8; ++a) asm volatile("" : "+v"(er[a]), "+v"(ei[a]));
    };
    // Job order.  Jobs are hop-major (the pairs of one hop position, then the next hop) and consecutive hop positions share
    // 15/16 of their samples.  Workgroup i runs on XCD i % 8 (MI355X in SPX mode: kXcdHint), each XCD has its own L2: every XCD
    // takes one contiguous eighth of the jobs and deals it round-robin to its workgroups, so that at any time the 32 CUs of an XCD
    // work on ~8 neighbouring hop positions and a sample is fetched from the fabric once.  The XCD count is a LOCALITY hint only: on
    // a part with another count (or a CPX partition) every job is still done exactly once, the samples just miss L2 more often.  (Measured, FETCH_SIZE per hop position: a contiguous run of
    // jobs per WORKGROUP 262 KB -- every load a miss: 32 CUs x 4 pairs x 64 KB of window do not fit a 4 MB L2 --, plain
    // round-robin over all workgroups 87 KB -- every XCD fetches every sample --; the time is the same.)
    const unsigned long long nx = p.xcds, xcd = blockIdx.x % nx, local = blockIdx.x / nx;
    const unsigned long long job_step = gridDim.x / nx;
    const unsigned long long job_begin = xcd * p.jobs_per_xcd + local;
    const unsigned long long job_end = (xcd + 1) * p.jobs_per_xcd < p.n_jobs ? (xcd + 1) * p.jobs_per_xcd : p.n_jobs;
    unsigned long long hop_c = MONO ? 0 : job_begin / p.pairs;                           // (hop, pair) of the current job
    uint32_t pair_c = MONO ? 0 : (uint32_t)(job_begin - hop_c * p.pairs);
    const unsigned long long step_hops = MONO ? 0 : job_step / p.pairs;                  // ... and of one step of the loop
    const uint32_t step_pairs = MONO ? 0 : (uint32_t)(job_step - step_hops * p.pairs);
    if (job_begin < job_end) {
        const JobIn first = job_in(job_begin, hop_c, pair_c);
        prefetch(first);
        take(first.data_second);
    }
    // The finished row of a transform waits in LDS (`stage`) and is read back and stored by the NEXT iteration, behind its first
    // barrier: the wait for the staging writes and the latency of the read-back then fall on that barrier (which the iteration
    // needs anyway) and on the twiddle multiplies of the first half of its image, instead of on a barrier of their own at the
    // end of the transform (same-device A/B: profiles/r03_k16_ablation.txt).
    constexpr int kStage = 8 * 4 * kS1;   // behind rows 0..7 of the first image (and the partner slots): 8704
    static_assert(kStage + 2 * kRegion <= kBufComplex, "the staged row must fit behind the partner slots");
    struct Pending { long long f0, f1; uint32_t pair; bool have_first, have_second, valid; } prev{0, 0, 0, false, false, false};
    auto flush = [&](const Pending &o) {
        lds_cfloat2 *stage = lds_ptr(buf + kStage);
        // rows: bin k lives at byte 8 (k - 1) of its row
        const __amdgpu_buffer_rsrc_t r0 = uniform_rsrc(p.mags + (((size_t)(o.have_first ? o.f0 : 0) * p.pairs + o.pair) * (size_t)kM) * 2);
        const __amdgpu_buffer_rsrc_t r1 = uniform_rsrc(p.mags + (((size_t)o.f1 * p.pairs + o.pair) * (size_t)kM) * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 1024 * i;           // bins 2 c, 2 c + 1 (0-based); the last piece of the row holds one bin
            lds_cfloat2 *sc = stage + stage_index(c);
            const float2 ev = lds_read_alone<4>(sc, 0), od = lds_read_alone<4>(sc, kRegion);   // bins 2 c and 2 c + 1
            const float4 v = make_float4(ev.x, ev.y, od.x, od.y);
            if (!D_STORE_OK(v.x)) continue;
            const bool whole = c != 4095;
            if (MONO) {
                if (o.have_first) {
                    if (whole) store16(u32x4{__float_as_uint(v.x), __float_as_uint(v.x), __float_as_uint(v.z), __float_as_uint(v.z)}, r0, 16 * tid, 16384 * i, 0);
                    else __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v.x), __float_as_uint(v.x)}, r0, 16 * tid, 16384 * i, D_OUT_AUX);
                }
                if (o.have_second) {
                    if (whole) store16(u32x4{__float_as_uint(v.y), __float_as_uint(v.y), __float_as_uint(v.w), __float_as_uint(v.w)}, r1, 16 * tid, 16384 * i, 0);
                    else __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v.y), __float_as_uint(v.y)}, r1, 16 * tid, 16384 * i, D_OUT_AUX);
                }
            } else {
                if (whole) store16(u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, r0, 16 * tid, 16384 * i, 0);
                else __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v.x), __float_as_uint(v.y)}, r0, 16 * tid, 16384 * i, D_OUT_AUX);
            }
        }
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

        // ---- Hann (fft.rs:53-63) on the prefetched samples; pass 1: 16-point DFT over a, inputs a >= 8 are the zero padding:
        //      even q1 = FFT8(z), odd q1 = FFT8(z * w_16^a)                                        (as stft4096_wg.hip)
        float orr[8], oi[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) { orr[a] = er[a]; oi[a] = ei[a]; }
        pretwiddle8_w16(orr, oi);
        fft8(er, ei);
        fft8(orr, oi);
        SGX_STAMP(0)    // job bookkeeping + pass-1 arithmetic (two FFT8)
        lds_barrier();  // the previous transform's partner reads are complete, its staged row is complete
        SGX_STAMP(1)    // barrier A
        if (prev.valid) flush(prev);
        SGX_STAMP(2)    // flush: staged row read back, the first 16-byte store issued
        {
            float2 *w1 = buf + tid;   // element (index, r) at 4 index + r = 4 (q kS1 + col) + r = 4 q kS1 + tid
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j == 4) {   // rows 8..15 of the image lie over the staged row: everyone has read it back
                    SGX_STAMP(3)
                    lds_barrier();
                    SGX_STAMP(4)
                }
                const int pos = FFT8_OUT[j];
                const float2 ve = make_float2(er[pos], ei[pos]);
                const float2 vo = make_float2(orr[pos], oi[pos]);
                w1[4 * (2 * j) * kS1] = j == 0 ? ve : cmulf(ve, tw1[2 * j]);
                w1[4 * (2 * j + 1) * kS1] = cmulf(vo, tw1[2 * j + 1]);
            }
        }
        SGX_STAMP(5)    // image-1 writes, second half (3: first half; 4: barrier B)
        lds_barrier();
        SGX_STAMP(6)    // barrier C

        // ---- pass 2: lane (q1, t0, r), col = t0 + 16 t1: FFT16 over t1 -> q2, twiddle w_1024^{q2 (4 t0 + r)}
        float xr[16], xi[16];
        {
            const int q1_2 = tid >> 6, low = tid & 63;   // low = 4 t0 + r
            lds_cfloat2 *r2 = lds_ptr(buf + 4 * q1_2 * kS1 + low);
#pragma unroll
            for (int t1 = 0; t1 < 16; ++t1) {
                const float2 v = lds_read_alone<0>(r2, 64 * t1);
                xr[t1] = v.x; xi[t1] = v.y;
            }
            fft16(xr, xi);
            SGX_STAMP(7)    // image-1 reads + FFT16
            // Row q1 of the image belongs to wave q1 alone from here to the end of pass 3: it read all of it above, writes the pass-2
            // results back into it (element (q2, t0, r) at 68 q2 + 4 t0 + r: 16 x 68 = the row's 1088 slots) and reads them again as
            // the pass-3 thread (q1, q2, r) -- LDS instructions of one wave execute in order, so neither the overwrite nor the read-back
            // needs a workgroup barrier, and the sixteen waves drift apart through two of the three FFT passes.
            float2 *w2 = buf + 4 * q1_2 * kS1 + low;
            lds_cfloat2 *tw = lds_ptr(tw2 + low);
#pragma unroll
            for (int q2 = 0; q2 < 16; ++q2) {
                const int pos = FFT16_OUT[q2];
                const float2 v = make_float2(xr[pos], xi[pos]);
                w2[68 * q2] = q2 == 0 ? v : cmulf(v, lds_read_alone<1>(tw, 64 * q2));
            }
        }
        SGX_STAMP(8)    // pass-2 twiddles + writes (to completion)

        // ---- pass 3: lane (q1, q2, r) = (wave, (tid >> 2) & 15, tid & 3), i.e. u = q1 + 16 q2: FFT16 over t0 -> q3
        {
            lds_cfloat2 *r3 = lds_ptr(buf + 4 * (tid >> 6) * kS1 + 68 * ((tid >> 2) & 15) + (tid & 3));
#pragma unroll
            for (int t0 = 0; t0 < 16; ++t0) {
                const float2 v = lds_read_alone<2>(r3, 4 * t0);
                xr[t0] = v.x; xi[t0] = v.y;
            }
        }
        fft16(xr, xi);
        SGX_STAMP(9)    // pass-3 reads + FFT16
        // the next transform's samples: ahead of this transform's stores (vmcnt retires in order), and as early as the registers
        // allow -- here, before the recombination (same-device A/B against "after the publish": +3 %; after the recombination the
        // compiler spills 14 registers)
        if (more) prefetch(nxt);
        // ---- the rest of the recombination twiddle, w_64^{r q3}, and the radix-4 recombination across the quad.
        //      Lane r holds X_r.  Registers q3 < 8 ("lo"):  stage A: lanes 0, 1: X_r + X_{r+2}; lanes 2, 3: X_{r-2} - X_r
        //        -> E0, E1, O0, O1; lane 3 turns O1 by -i; stage B: lanes 0, 2: own + partner; lanes 1, 3: partner - own
        //        -> F[j], F[j + 2*4096], F[j + 4096], F[j + 3*4096] in lanes 0, 1, 2, 3.
        //      Registers q3 >= 8 ("hi"): the same with the roles of the lane halves exchanged (stage A: lanes 0, 1: own - partner;
        //        lanes 2, 3: own + partner -> O0, O1, E0, E1; lane 1 turns O1; stage B: lanes 0, 2: own - partner; lanes 1, 3: own +
        //        partner) -> c = 3, 1, 2, 0 in lanes 0, 1, 2, 3: every lane ends with kept bins in one half and partners in the other.
        lds_cfloat2 *tw3p = lds_ptr(tw3 + L);
#pragma unroll
        for (int q3 = 0; q3 < 16; ++q3) {
            const int pos = FFT16_OUT[q3];
            float ar = xr[pos], ai = xi[pos];
            if (q3 > 0) {
                const float2 t = lds_read_alone<3>(tw3p, 4 * q3);
                const float br = fmaf(ar, t.x, -(ai * t.y)), bi = fmaf(ar, t.y, ai * t.x);
                ar = br; ai = bi;
            }
            const float pr_ = quad<kSwap2>(ar), pi_ = quad<kSwap2>(ai);
            float cr, ci;
            if (q3 < 8) { cr = fmaf(ar, sA, pr_); ci = fmaf(ai, sA, pi_); }
            else { cr = fmaf(pr_, -sA, ar); ci = fmaf(pi_, -sA, ai); }
            const bool rot = q3 < 8 ? rot_lo : rot_hi;
            const float dr = rot ? ci : cr, di = rot ? -cr : ci;
            const float qr_ = quad<kSwap1>(dr), qi_ = quad<kSwap1>(di);
            if (q3 < 8) { xr[pos] = fmaf(dr, sB, qr_); xi[pos] = fmaf(di, sB, qi_); }
            else { xr[pos] = fmaf(qr_, -sB, dr); xi[pos] = fmaf(qi_, -sB, di); }
        }
        // odd lanes hold their kept bins in the hi registers: exchange the halves, so that every lane keeps 0..7 and publishes 8..15
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const int a = FFT16_OUT[h], b = FFT16_OUT[8 + h];
            const float ar = xr[a], ai = xi[a], br = xr[b], bi = xi[b];
            xr[a] = odd ? br : ar; xi[a] = odd ? bi : ai;
            xr[b] = odd ? ar : br; xi[b] = odd ? ai : bi;
        }
        SGX_STAMP(10)   // prefetch requests + recombination (twiddles, two DPP stages)
        lds_barrier();  // everyone has read image 2
        SGX_STAMP(11)   // barrier D
        {
            // slot (h, u, L) at 1088 h + 4 (u & 15) + 68 (u >> 4) + L: the lanes of a wave hold u = q1 + 16 q2 for one q1 -- at a stride of
            // 68 complex per q2 they fall into different banks, here and in the partner reads below
            float2 *wp = buf + 4 * (tid >> 6) + 68 * ((tid >> 2) & 15) + L;
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const int pos = FFT16_OUT[8 + h];
                wp[1088 * h] = make_float2(xr[pos], xi[pos]);
            }
        }
        SGX_STAMP(12)   // partner writes
        lds_barrier();
        SGX_STAMP(13)   // barrier E

        // ---- split + magnitude (fft.rs:81-98).  Kept (u, qq) of lane L pairs with slot h = 7 - qq of lane L of quad 256 - u;
        //      quad 0 is its own mirror one row up (h = 8 - qq), and its qq = 0 partners are single values of other lanes of the
        //      quad: lane 2 (k = 4096) <- lane 3's slot 0; lanes 1, 3 (k = 6144, 2048) <- slot 0 of lanes 2, 0
        const int u = (tid >> 6) + 16 * ((tid >> 2) & 15);
        const int up = 256 - u, pslot = 4 * (up & 15) + 68 * ((up >> 4) & 15) + L;   // (u = 0: unused)
        lds_cfloat2 *pp = lds_ptr(buf + (u == 0 ? 1088 + L : pslot));                          // + 1088 (7 - qq), qq >= 1
        const float2 *pp0 = buf + (u == 0 ? (L == 2 ? 3 : (L == 1 ? 2 : 0)) : 1088 * 7 + pslot);   // qq = 0
        const int kbase = (L == 0 ? 0 : (L == 1 ? 6144 : (L == 2 ? 4096 : 2048))) + u;        // bin k = kbase + 256 qq
        // The row is put together in LDS and leaves as 16 bytes per lane, consecutive lanes consecutive addresses: a wave of this
        // kernel holds 16 consecutive bins of four distant parts of the row, and storing them as they lie -- 8 bytes per lane,
        // four 128-byte runs per instruction, 128 instructions per transform -- cost a third of the launch (ablation: 0.90 ms per
        // 20 000 transforms against 0.61 ms without the stores; profiles/r03_k16_ablation.txt).  Staging area: behind the
        // partner slots and the first eight rows of the next image (no barrier between the partner reads and these writes);
        // bin k at stage_slot(k - 1).  The row is read back and stored by the next iteration (`flush`).
        float2 *stage = buf + kStage;
        float2 *st1 = stage + stage_slot(kbase + 255);
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) {
            const int pos = FFT16_OUT[qq];
            const float2 pv = qq == 0 ? pp0[0] : lds_read_alone<5>(pp, 1088 * (7 - qq));
            const float ar = xr[pos], ai = xi[pos];
            const float sr_ = ar + pv.x, si_ = ai - pv.y;   // a + conj(b) = 2 L^
            const float dr_ = ar - pv.x, di_ = ai + pv.y;   // a - conj(b) = 2i R^
            const float ml = __builtin_amdgcn_sqrtf(fmaf(sr_, sr_, si_ * si_));  // the scale 1 / W rides on the window
            const float mr = __builtin_amdgcn_sqrtf(fmaf(dr_, dr_, di_ * di_));
            const bool dc = qq == 0 && tid == 0;   // k = 0 (DC) is not an output (fft.rs:81)
            // 0-based bin b = kbase + 256 qq - 1.  For qq >= 1 the swizzle bits of stage_slot (bit 5 and bits 11, 12 of b) do not depend on
            // qq, and it only touches the low three index bits: slot(qq) = slot(1) + 128 (qq - 1) -- one address, seven immediates, instead
            // of eight swizzles (~70 vector instructions per transform)
            if (qq == 0) { if (!dc) stage[stage_slot(kbase - 1)] = make_float2(ml, mr); }
            else st1[128 * (qq - 1)] = make_float2(ml, mr);
        }
        SGX_STAMP(14)   // partner reads + split + staging writes
        prev = Pending{f0, f1, pair, have_first, have_second, true};
        if (more) take(nxt.data_second);
        SGX_STAMP(15)   // wait for the next samples + Hann
    }
#if SGX_STAMPS
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 20; ++i) atomicAdd(&g_phase_cycles16[i], st_acc[i]);
        atomicAdd(&g_phase_cycles16[20], st_iters);
    }
#endif
    if (prev.valid) {   // the last transform's row
        lds_barrier();
        flush(prev);
    }
}

struct TablesD {
    float2 *d_T1 = nullptr, *d_tw2 = nullptr, *d_tw3 = nullptr;
    float *d_win8 = nullptr;
    float *d_planes = nullptr;   // de-interleave workspace, grown on demand
    size_t planes_floats = 0;
};

}  // namespace d16k

#if SGX_STAMPS
extern "C" __attribute__((visibility("default"))) int sgx_debug_phase_cycles16(unsigned long long *h_out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(h_out, HIP_SYMBOL(d16k::g_phase_cycles16), sizeof(unsigned long long) * 24);
    if (e == hipSuccess && reset) {
        unsigned long long zero[24] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(d16k::g_phase_cycles16), zero, sizeof(zero));
    }
    return e == hipSuccess ? 0 : -1;
}
#endif

bool d16384_supported(const sgx_ctx *c)
{
    // (l, r) pairs are moved as 8-byte words: the stream must be mono or have an even channel count
    return c->W == d16k::kW && (c->C == 1 || (c->C & 1) == 0) && c->lds_optin >= d16k::kLdsBytes;
}

hipError_t d16384_init(sgx_ctx *c, void **out)
{
    using namespace d16k;
    auto *t = new TablesD();
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
    std::vector<float2> T1(16 * 1024), tw2(1024), tw3(64);
    std::vector<float> win8((size_t)kW);
    for (int q = 0; q < 16; ++q)
        for (int tid = 0; tid < 1024; ++tid) T1[q * 1024 + tid] = unit((unsigned long long)q * tid, kP);
    for (int q2 = 0; q2 < 16; ++q2)
        for (int t0 = 0; t0 < 16; ++t0)
            for (int r = 0; r < 4; ++r) tw2[(q2 * 16 + t0) * 4 + r] = unit((unsigned long long)q2 * (4 * t0 + r), 1024);
    for (int q3 = 0; q3 < 16; ++q3)
        for (int r = 0; r < 4; ++r) tw3[q3 * 4 + r] = unit((unsigned long long)r * q3, 64);
    // the scale (hypot / 2) * (2 / W) = 1 / W is a power of two and commutes with every rounding
    for (int a = 0; a < 8; ++a)
        for (int tid = 0; tid < 1024; ++tid) win8[((size_t)(a >> 2) * 1024 + tid) * 4 + (a & 3)] = c->tab.window[tid + 1024 * a] * (1.0f / (float)kW);
    auto up = [](float2 **dst, const std::vector<float2> &v) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), v.size() * sizeof(float2));
        if (e == hipSuccess) e = hipMemcpy(*dst, v.data(), v.size() * sizeof(float2), hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(&t->d_T1, T1);
    if (e == hipSuccess) e = up(&t->d_tw2, tw2);
    if (e == hipSuccess) e = up(&t->d_tw3, tw3);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&t->d_win8), win8.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(t->d_win8, win8.data(), win8.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft16384_d_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft16384_d_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e != hipSuccess) {
        d16384_destroy(t);
        return e;
    }
    *out = t;
    return hipSuccess;
}

void d16384_destroy(void *tables)
{
    auto *t = static_cast<d16k::TablesD *>(tables);
    if (!t) return;
    if (t->d_T1) (void)hipFree(t->d_T1);
    if (t->d_tw2) (void)hipFree(t->d_tw2);
    if (t->d_tw3) (void)hipFree(t->d_tw3);
    if (t->d_win8) (void)hipFree(t->d_win8);
    if (t->d_planes) (void)hipFree(t->d_planes);
    delete t;
}

hipError_t launch_stft_d16384(const sgx_ctx *c, void *tables, const float *d_pcm, uint32_t channels, uint32_t pairs,
                              size_t first_frame, size_t n_frames, size_t total_frames, float *d_mags)
{
    using namespace d16k;
    if (n_frames == 0) return hipSuccess;
    auto *t = static_cast<TablesD *>(tables);
    Params p{};
    p.T1 = t->d_T1;
    p.tw2 = t->d_tw2;
    p.tw3 = t->d_tw3;
    p.win8 = t->d_win8;
    p.mags = d_mags;
    p.first_frame = first_frame;
    p.n_frames = n_frames;
    p.total_frames = total_frames;
    p.H = c->H;
    p.pairs = pairs;
    const bool mono = channels == 1 && !(c->cfg.flags & SGX_FLAG_INDEPENDENT_FRAMES);
    // a mono stream whose frames are not paired (the default): every frame the (s, s) transform of the reference
    // (audio_input_list_model.rs:67-69) -- the sample range duplicated into one (s, s) plane, then the two-channel kernel
    const bool dup = channels == 1 && !mono;
    p.pair_base = mono ? first_frame / 2 : 0;
    p.n_jobs = mono ? (first_frame + n_frames + 1) / 2 - first_frame / 2 : (unsigned long long)n_frames * pairs;
    // more than two channels: every pair read where it lies (no workspace, no second kernel, HBM traffic 1.02 x algorithmic), or -- under
    // SGX_FLAG_CHANNEL_PLANES, the default of rounds 3-5 -- the sample range of the call split into per-pair (l, r) planes first
    const bool direct = channels > 2 && !(c->cfg.flags & SGX_FLAG_CHANNEL_PLANES);
    if ((channels > 2 && !direct) || dup) {
        // per-pair planes of the sample range these frames read: [first_frame H, (first_frame + n - 1) H + W)
        const size_t first_sample = first_frame * (size_t)c->H;
        const size_t n_samp = (n_frames - 1) * (size_t)c->H + kW;
        const size_t plane = (2 * n_samp + 63) & ~(size_t)63;  // floats per plane
        if (plane * pairs > t->planes_floats) {
            hipError_t e = hipStreamSynchronize(c->stream);  // a previous launch may still read the old planes
            if (e != hipSuccess) return e;
            if (t->d_planes) { (void)hipFree(t->d_planes); t->d_planes = nullptr; t->planes_floats = 0; }
            e = hipMalloc(reinterpret_cast<void **>(&t->d_planes), plane * pairs * sizeof(float));
            if (e != hipSuccess) return e;
            t->planes_floats = plane * pairs;
        }
        hipError_t e = hipSuccess;
        if (dup) {
            const unsigned blocks = (unsigned)std::min<size_t>((n_samp + 255) / 256, (size_t)c->n_cu * 16);
            hipLaunchKernelGGL(d16k::duplicate_mono_kernel, dim3(blocks), dim3(256), 0, c->stream, d_pcm, t->d_planes, first_sample, n_samp);
            e = hipGetLastError();
        } else {
            e = launch_deinterleave_pairs(c, d_pcm, t->d_planes, plane, first_sample, n_samp, channels, pairs);
        }
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
    // persistent workgroups, one per CU (145 KB of LDS); jobs are dealt round-robin in output-row order
    unsigned long long blocks = (unsigned long long)c->n_cu;
    if (blocks > p.n_jobs) blocks = p.n_jobs;
    p.xcds = (blocks % kXcdHint == 0 && p.n_jobs >= kXcdHint * blocks) ? kXcdHint : 1u;   // (short launches: plain round-robin keeps every workgroup busy)
    const unsigned long long group = mono ? 1 : pairs;                   // a hop position's pairs stay together
    p.jobs_per_xcd = ((p.n_jobs + p.xcds - 1) / p.xcds + group - 1) / group * group;
    const dim3 grid((unsigned)blocks), block(1024);
    if (mono) hipLaunchKernelGGL((stft16384_d_kernel<true>), grid, block, kLdsBytes, c->stream, p);
    else if (direct) hipLaunchKernelGGL((stft16384_d_kernel<false, true>), grid, block, kLdsBytes, c->stream, p);
    else hipLaunchKernelGGL((stft16384_d_kernel<false>), grid, block, kLdsBytes, c->stream, p);
    return hipGetLastError();
}

}  // namespace sgx
