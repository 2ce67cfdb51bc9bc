// overlap_lds_valu.hip -- can a SIMD issue vector arithmetic while its LDS traffic is in flight?
// Eight waves per workgroup, one workgroup per CU (two waves per SIMD: wave w runs on SIMD w % 4).  Waves 0-3 run a loop of independent
// v_fma_f32 (or idle), waves 4-7 a loop of conflict-free ds_write_b64 / ds_read_b64 (or idle).  Three runs per LDS instruction: arithmetic
// alone, LDS alone, both together.  If the two pipes overlap, "together" takes max(alone, alone); if they share an issue resource, the sum.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/overlap_lds_valu tools/overlap_lds_valu.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

template <int LDS_OP>   // 0: ds_write_b64, 1: ds_read_b64
__global__ void __launch_bounds__(1024) mix_kernel(unsigned long long *cycles, float *sink, int valu_iters, int lds_iters, int valu_waves_per_simd, int lds_waves_per_simd)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x / 64, lane = threadIdx.x & 63;
    const int slot = wave / 4;                       // waves 4 s .. 4 s + 3 are the s-th wave of SIMDs 0..3
    const bool is_valu = slot < valu_waves_per_simd;
    const bool is_lds = !is_valu && slot < valu_waves_per_simd + lds_waves_per_simd;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (float)(i + lane);
    float x = 1.0001f, y = 1e-3f;
    f2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = f2{(float)lane, (float)i};
    unsigned addr = wave * 64 * 8 * 8 + lane * 8;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (is_valu) {
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
        }
    } else if (is_lds) {
        for (int it = 0; it < lds_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (LDS_OP == 0) asm volatile("ds_write_b64 %1, %0 offset:%2" ::"v"(v[i]), "v"(addr), "n"(0) : "memory");
                else asm volatile("ds_read_b64 %0, %1" : "=v"(v[i]) : "v"(addr));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0) cycles[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int LDS_OP>
static void run(const char *name, unsigned long long *d_cyc, float *d_sink, int vw, int lw, int valu_iters, int lds_iters)
{
    const int blocks = 256, threads = 64 * 4 * (vw + lw);
    std::vector<unsigned long long> h(blocks * threads / 64);
    const size_t lds = threads * 64 + 4096;
    mix_kernel<LDS_OP><<<blocks, threads, lds>>>(d_cyc, d_sink, 4, 4, vw, lw);
    mix_kernel<LDS_OP><<<blocks, threads, lds>>>(d_cyc, d_sink, valu_iters, lds_iters, vw, lw);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double sv = 0, sl = 0;
    int nv = 0, nl = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < threads / 64; ++w) {
            const double c = (double)h[b * (threads / 64) + w];
            if (w / 4 < vw) { sv += c; ++nv; } else { sl += c; ++nl; }
        }
    printf("%-14s valu waves/SIMD %d, lds waves/SIMD %d:", name, vw, lw);
    if (nv && valu_iters) printf("  arithmetic waves %9.0f clocks (%5.2f per v_fma per wave)", sv / nv, sv / nv / (valu_iters * 16.0));
    if (nl && lds_iters) printf("  LDS waves %9.0f clocks (%5.2f per instruction per CU)", sl / nl, sl / nl / (lds_iters * 8.0) / (4.0 * lw));
    printf("\n");
}

int main()
{
    unsigned long long *d_cyc;
    float *d_sink;
    CK(hipMalloc(&d_cyc, 8 * 256 * 16));
    CK(hipMalloc(&d_sink, 64));
    const int VI = 4000, LI = 4000;
    for (int vw : {1, 2}) {
        run<0>("alone", d_cyc, d_sink, vw, 0, VI, 0);
        for (int lw : {1, 2}) {
            run<0>("ds_write_b64", d_cyc, d_sink, 0, lw, 0, LI);
            run<0>("ds_write_b64", d_cyc, d_sink, vw, lw, VI, LI);
            run<1>("ds_read_b64", d_cyc, d_sink, 0, lw, 0, LI);
            run<1>("ds_read_b64", d_cyc, d_sink, vw, lw, VI, LI);
        }
    }
    return 0;
}
