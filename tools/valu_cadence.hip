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
                if (OP == 4) asm volatile("v_mul_f32 %0, 0x3f3504f3, %0" : "+v"(acc[i].x));                  // a 32-bit literal: an 8-byte instruction
                if (OP == 5) asm volatile("v_fmac_f32 %0, 0x3f3504f3, %1" : "+v"(acc[i].x) : "v"(x.x));
                if (OP == 6) asm volatile("v_sqrt_f32 %0, %0" : "+v"(acc[i].x));
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

// Register-file banks (register number mod 4).  16 independent destinations v16 .. v31 per group, the same source registers in every instruction.
#define BANK_CASE(NAME, INSTR)                                                                                                                      \
    __global__ void NAME(unsigned long long *cycles, float *sink, int trips)                                                                        \
    {                                                                                                                                               \
        asm volatile("v_mov_b32 v40, 1.0\n\tv_mov_b32 v41, 1.0\n\tv_mov_b32 v42, 1.0\n\tv_mov_b32 v43, 1.0\n\tv_mov_b32 v44, 1.0\n\tv_mov_b32 v48, 1.0" ::         \
                         : "v40", "v41", "v42", "v43", "v44", "v48");                                                                               \
        __syncthreads();                                                                                                                            \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                                                 \
        for (int it = 0; it < trips; ++it) {                                                                                                        \
            _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                                                          \
                asm volatile(INSTR(16) INSTR(17) INSTR(18) INSTR(19) INSTR(20) INSTR(21) INSTR(22) INSTR(23) INSTR(24) INSTR(25) INSTR(26) INSTR(27) \
                                 INSTR(28) INSTR(29) INSTR(30) INSTR(31) ::                                                                          \
                                 : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");  \
        }                                                                                                                                           \
        asm volatile("s_nop 0" ::: "memory");                                                                                                       \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                                                 \
        if (t1 == 1) sink[0] = 1.0f;                                                                                                                \
        if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;                                           \
    }
#define I_FMA_3BANKS(d) "v_fma_f32 v" #d ", v40, v41, v42\n\t"
#define I_FMA_1BANK(d) "v_fma_f32 v" #d ", v40, v44, v48\n\t"
#define I_FMA_2OF3(d) "v_fma_f32 v" #d ", v40, v44, v41\n\t"
#define I_ADD_2BANKS(d) "v_add_f32 v" #d ", v40, v41\n\t"
#define I_ADD_1BANK(d) "v_add_f32 v" #d ", v40, v44\n\t"
#define I_ADD_SAMEREG(d) "v_add_f32 v" #d ", v40, v40\n\t"
#define I_FMAC_2BANKS(d) "v_fmac_f32 v" #d ", v41, v42\n\t"
#define I_FMAC_1BANK(d) "v_fmac_f32 v" #d ", v40, v44\n\t"
#define I_MULK_1(d) "v_mul_f32 v" #d ", 0x3f3504f3, v40\n\t"
BANK_CASE(k_fma_3banks, I_FMA_3BANKS)
BANK_CASE(k_fma_1bank, I_FMA_1BANK)
BANK_CASE(k_fma_2of3, I_FMA_2OF3)
BANK_CASE(k_add_2banks, I_ADD_2BANKS)
BANK_CASE(k_add_1bank, I_ADD_1BANK)
BANK_CASE(k_add_samereg, I_ADD_SAMEREG)
BANK_CASE(k_fmac_2banks, I_FMAC_2BANKS)
BANK_CASE(k_fmac_1bank, I_FMAC_1BANK)
BANK_CASE(k_mulk, I_MULK_1)

void banks(const char *what, void (*kern)(unsigned long long *, float *, int), int n_cu, unsigned long long *d_cyc, float *d_sink)
{
    const int trips = 2000;
    for (int wps : {1, 2, 4}) {
        const int threads = 256 * wps;
        hipLaunchKernelGGL(kern, dim3(n_cu), dim3(threads), 0, 0, d_cyc, d_sink, trips);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(n_cu * (threads / 64));
        CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (auto v : h) sum += (double)v;
        const double clk = sum / h.size() / ((double)trips * 256);
        printf("%-72s waves/SIMD %d: %.2f clocks per instruction and wave = one per %.2f clocks per SIMD\n", what, wps, clk, clk / wps);
    }
}

// the same run with a LONG straight-line loop body (BODY instructions per trip: 8-byte v_fma_f32 or 4-byte v_add_f32 encodings): does a wave's
// instruction fetch keep up when eight waves of a CU walk through tens of kilobytes of code per trip?
template <int OP, int BODY>
__global__ void run_long(unsigned long long *cycles, float *sink, int trips)
{
    f2 acc[16];
    f2 x = {1.0001f + threadIdx.x * 1e-7f, 0.9999f}, y = {1e-3f, 2e-3f};
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f2{(float)i, (float)(i + 1)};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < trips; ++it) {
#pragma unroll
        for (int r = 0; r < BODY / 16; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(x.x), "v"(y.x));
                if (OP == 1) asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[i].x) : "v"(x.x));
            }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP, int BODY>
void long_body(const char *name, int n_cu, unsigned long long *d_cyc, float *d_sink)
{
    const int trips = 512 * 1024 / BODY;
    for (int wps : {1, 2, 4}) {
        const int threads = 256 * wps;
        hipLaunchKernelGGL((run_long<OP, BODY>), dim3(n_cu), dim3(threads), 0, 0, d_cyc, d_sink, trips);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(n_cu * (threads / 64));
        CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (auto v : h) sum += (double)v;
        const double clk = sum / h.size() / ((double)trips * BODY);
        printf("%-12s loop body %5d instructions (%3d KB)  waves/SIMD %d: %.2f clocks per instruction and wave = one per %.2f clocks per SIMD\n", name, BODY,
               BODY * (OP == 0 ? 8 : 4) / 1024, wps, clk, clk / wps);
    }
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
    one<4, 16>("v_mul literal", n, d_cyc, d_sink, tick_per_clk);
    one<5, 16>("v_fmac literal", n, d_cyc, d_sink, tick_per_clk);
    one<6, 16>("v_sqrt_f32", n, d_cyc, d_sink, tick_per_clk);
    long_body<0, 256>("v_fma_f32", n, d_cyc, d_sink);
    long_body<0, 2048>("v_fma_f32", n, d_cyc, d_sink);
    long_body<0, 4096>("v_fma_f32", n, d_cyc, d_sink);
    long_body<0, 16384>("v_fma_f32", n, d_cyc, d_sink);
    long_body<1, 4096>("v_add_f32", n, d_cyc, d_sink);
    long_body<1, 16384>("v_add_f32", n, d_cyc, d_sink);
    banks("v_fma_f32 d, v40, v41, v42   (three banks)", k_fma_3banks, n, d_cyc, d_sink);
    banks("v_fma_f32 d, v40, v44, v41   (two sources of one bank)", k_fma_2of3, n, d_cyc, d_sink);
    banks("v_fma_f32 d, v40, v44, v48   (three sources of one bank)", k_fma_1bank, n, d_cyc, d_sink);
    banks("v_add_f32 d, v40, v41        (two banks)", k_add_2banks, n, d_cyc, d_sink);
    banks("v_add_f32 d, v40, v44        (one bank)", k_add_1bank, n, d_cyc, d_sink);
    banks("v_add_f32 d, v40, v40        (the same register twice)", k_add_samereg, n, d_cyc, d_sink);
    banks("v_fmac_f32 d, v41, v42       (d = v16 .. v31: every bank in turn)", k_fmac_2banks, n, d_cyc, d_sink);
    banks("v_fmac_f32 d, v40, v44       (sources of one bank)", k_fmac_1bank, n, d_cyc, d_sink);
    banks("v_mul_f32 d, literal, v40", k_mulk, n, d_cyc, d_sink);
    return 0;
}
