// valu_cadence.hip -- how often can ONE wave issue a vector instruction on gfx950, and what does a dependent instruction wait for?
// A long unrolled run (256 instructions per loop trip: loop overhead < 2 %) of v_fma_f32 / v_add_f32 / v_pk_add_f32 over `nacc` independent accumulators
// (dependency distance = nacc instructions), at 1, 2, 3, 4 waves per SIMD; s_memtime around the run; one workgroup per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_cadence tools/valu_cadence.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP, int NACC>
__global__ void run(unsigned long long *cycles, float *sink, int trips)
{
    f2 acc[NACC];
    f2 x = {1.0001f + threadIdx.x * 1e-7f, 0.9999f}, y = {1e-3f, 2e-3f};
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f2{(float)i, (float)(i + 1)};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < trips; ++it) {
#pragma unroll
        for (int r = 0; r < 256 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(x.x), "v"(y.x));
                if (OP == 1) asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[i].x) : "v"(x.x));
                if (OP == 2) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(x));
                if (OP == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
            }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y;
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP, int NACC>
void one(const char *name, int n_cu, unsigned long long *d_cyc, float *d_sink, double tick_per_clk)
{
    const int trips = 2000;
    for (int wps : {1, 2, 3, 4}) {
        const int threads = 256 * wps;
        hipLaunchKernelGGL((run<OP, NACC>), dim3(n_cu), dim3(threads), 0, 0, d_cyc, d_sink, trips);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(n_cu * (threads / 64));
        CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (auto v : h) sum += (double)v;
        const double ticks = sum / h.size() / ((double)trips * 256);
        printf("%-14s accumulators %2d  waves/SIMD %d: %.2f clocks per instruction and wave = one per %.2f clocks per SIMD\n", name, NACC, wps,
               ticks / tick_per_clk, ticks / tick_per_clk / wps);
    }
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int wall_khz = 0;
    CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    // s_memtime counts shader clocks on gfx950 (tools/microbench.hip reads 2.1 per v_fma_f32 and SIMD at four waves per SIMD: the pipe's 2)
    const double tick_per_clk = 1.0;
    printf("device %s  CUs %d  shader clock %d kHz  wall clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, wall_khz);
    unsigned long long *d_cyc;
    float *d_sink;
    CK(hipMalloc(&d_cyc, 8 * 1024 * 64));
    CK(hipMalloc(&d_sink, 4));
    const int n = prop.multiProcessorCount;
    one<0, 16>("v_fma_f32", n, d_cyc, d_sink, tick_per_clk);
    one<0, 4>("v_fma_f32", n, d_cyc, d_sink, tick_per_clk);
    one<0, 2>("v_fma_f32", n, d_cyc, d_sink, tick_per_clk);
    one<0, 1>("v_fma_f32", n, d_cyc, d_sink, tick_per_clk);
    one<1, 16>("v_add_f32", n, d_cyc, d_sink, tick_per_clk);
    one<2, 16>("v_pk_add_f32", n, d_cyc, d_sink, tick_per_clk);
    one<2, 2>("v_pk_add_f32", n, d_cyc, d_sink, tick_per_clk);
    one<3, 16>("v_pk_fma_f32", n, d_cyc, d_sink, tick_per_clk);
    return 0;
}
