// microbench.hip -- gfx950 instruction-rate probes that size the tuned STFT kernel:
//   VALU issue rates (scalar vs packed f32) at 1/2/4 waves per SIMD, LDS read/write/bpermute
//   rates, and streaming-store bandwidth for the [frame][2047][2] float output layout.
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench tools/microbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

enum { OP_FMA, OP_ADD, OP_MUL, OP_PKFMA, OP_PKADD, OP_PKMUL, OP_N };
static const char *op_names[] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32"};

template <int OP>
__global__ void valu_kernel(unsigned long long *cycles, float *sink, int iters)
{
    f2 acc[16];
    f2 x = {1.0001f + threadIdx.x * 1e-7f, 0.9999f};
    f2 y = {1e-3f, 2e-3f};
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f2{(float)i, (float)(i + 1)};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(x.x), "v"(y.x));
            if (OP == OP_ADD) asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[i].x) : "v"(x.x));
            if (OP == OP_MUL) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(acc[i].x) : "v"(x.x));
            if (OP == OP_PKFMA) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
            if (OP == OP_PKADD) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(x));
            if (OP == OP_PKMUL) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(x));
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

enum { L_R32, L_R64, L_R128, L_W32, L_W64, L_W128, L_BPERM, L_N };
static const char *lds_names[] = {"ds_read_b32", "ds_read_b64", "ds_read_b128", "ds_write_b32", "ds_write_b64", "ds_write_b128", "ds_bpermute_b32"};

template <int OP>
__global__ void lds_kernel(unsigned long long *cycles, float *sink, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x / 64, lane = threadIdx.x & 63;
    const int width = (OP == L_R32 || OP == L_W32 || OP == L_BPERM) ? 4 : (OP == L_R64 || OP == L_W64) ? 8 : 16;
    unsigned addr = wave * 64 * 16 * 2 + lane * width;  // conflict-free, contiguous per wave
    for (int i = threadIdx.x; i < (int)(blockDim.x * 8); i += blockDim.x) ((float *)smem)[i] = (float)i;
    __syncthreads();
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = f4{(float)lane, 1.f, 2.f, 3.f};
    unsigned perm = ((63 - lane) * 4);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == L_R32) asm volatile("ds_read_b32 %0, %1" : "=v"(v[i].x) : "v"(addr));
            if (OP == L_R64) asm volatile("ds_read_b64 %0, %1" : "=v"(*(f2 *)&v[i]) : "v"(addr));
            if (OP == L_R128) asm volatile("ds_read_b128 %0, %1" : "=v"(v[i]) : "v"(addr));
            if (OP == L_W32) asm volatile("ds_write_b32 %1, %0" ::"v"(v[i].x), "v"(addr) : "memory");
            if (OP == L_W64) asm volatile("ds_write_b64 %1, %0" ::"v"(*(f2 *)&v[i]), "v"(addr) : "memory");
            if (OP == L_W128) asm volatile("ds_write_b128 %1, %0" ::"v"(v[i]), "v"(addr) : "memory");
            if (OP == L_BPERM) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(v[i].x) : "v"(perm), "v"(v[i].y));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0) cycles[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

// streaming stores shaped like the STFT output: rows of ROWB bytes, each wave writes one row with
// 8-byte (float2) or 16-byte (float4) stores per lane
template <int VEC>
__global__ void store_kernel(float *out, size_t rows, int row_floats)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const size_t nw = (size_t)gridDim.x * (blockDim.x / 64);
    for (size_t r = wave; r < rows; r += nw) {
        float *row = out + r * row_floats;
        if (VEC == 2) {
            for (int j = lane * 2; j + 1 < row_floats; j += 128) *(float2 *)(row + j) = make_float2((float)r, (float)j);
        } else {
            for (int j = lane * 4; j + 3 < row_floats; j += 256) *(float4 *)(row + j) = make_float4((float)r, (float)j, 0.f, 1.f);
        }
    }
}

// the STFT kernel's exact store pattern and nothing else: persistent 256-thread workgroups, each
// streaming through its own contiguous run of 16 376-byte rows, thread t writing float2 at bins
// t + 256 q (q = 0..7) of two rows per iteration
__global__ void stft_store_pattern_kernel(float *out, size_t rows_total, size_t rows_per_block)
{
    const int tid = threadIdx.x;
    size_t r0 = (size_t)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block;
    if (r1 > rows_total) r1 = rows_total;
    for (size_t r = r0; r + 1 < r1; r += 2) {
        char *row0 = reinterpret_cast<char *>(out) + r * 16376 - 8;
        char *row1 = row0 + 16376;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q > 0 || tid != 0) *reinterpret_cast<float2 *>(row0 + 2048 * q + tid * 8) = make_float2((float)r, (float)q);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q > 0 || tid != 0) *reinterpret_cast<float2 *>(row1 + 2048 * q + tid * 8) = make_float2((float)r, (float)q);
    }
}

__global__ void copy_kernel(const float4 *in, float4 *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

template <int OP>
void run_valu(unsigned long long *d_cyc, float *d_sink, int waves_per_simd)
{
    const int iters = 2000, threads = 256 * waves_per_simd, blocks = 256;
    std::vector<unsigned long long> h(blocks * threads / 64);
    valu_kernel<OP><<<blocks, threads>>>(d_cyc, d_sink, 10);
    valu_kernel<OP><<<blocks, threads>>>(d_cyc, d_sink, iters);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double sum = 0;
    for (auto c : h) sum += (double)c;
    double avg = sum / h.size();
    double per_instr_wave = avg / (iters * 16.0);
    printf("VALU %-14s waves/SIMD=%d  cycles/instr/wave=%.2f  => SIMD issues one per %.2f cycles\n", op_names[OP],
           waves_per_simd, per_instr_wave, per_instr_wave / waves_per_simd);
}

template <int OP>
void run_lds(unsigned long long *d_cyc, float *d_sink, int waves_per_simd)
{
    const int iters = 2000, threads = 256 * waves_per_simd, blocks = 256;
    std::vector<unsigned long long> h(blocks * threads / 64);
    size_t lds = threads * 32 + 4096;
    lds_kernel<OP><<<blocks, threads, lds>>>(d_cyc, d_sink, 10);
    lds_kernel<OP><<<blocks, threads, lds>>>(d_cyc, d_sink, iters);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double sum = 0;
    for (auto c : h) sum += (double)c;
    double avg = sum / h.size();
    const int width = (OP == L_R32 || OP == L_W32 || OP == L_BPERM) ? 4 : (OP == L_R64 || OP == L_W64) ? 8 : 16;
    double per_instr_cu = avg / (iters * 8.0) / (4 * waves_per_simd);
    printf("LDS  %-16s waves/SIMD=%d  cycles/instr/CU=%.2f  => %.1f B/clk/CU\n", lds_names[OP], waves_per_simd,
           per_instr_cu, 64.0 * width / per_instr_cu);
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz  LDS/block=%zu\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate,
           prop.sharedMemPerBlock);
    unsigned long long *d_cyc;
    float *d_sink;
    CK(hipMalloc(&d_cyc, 8 * 256 * 16));
    CK(hipMalloc(&d_sink, 64));
    for (int w : {1, 2, 4}) {
        run_valu<OP_FMA>(d_cyc, d_sink, w);
        run_valu<OP_ADD>(d_cyc, d_sink, w);
        run_valu<OP_MUL>(d_cyc, d_sink, w);
        run_valu<OP_PKFMA>(d_cyc, d_sink, w);
        run_valu<OP_PKADD>(d_cyc, d_sink, w);
        run_valu<OP_PKMUL>(d_cyc, d_sink, w);
    }
    for (int w : {1, 2}) {
        run_lds<L_R32>(d_cyc, d_sink, w);
        run_lds<L_R64>(d_cyc, d_sink, w);
        run_lds<L_R128>(d_cyc, d_sink, w);
        run_lds<L_W32>(d_cyc, d_sink, w);
        run_lds<L_W64>(d_cyc, d_sink, w);
        run_lds<L_W128>(d_cyc, d_sink, w);
        run_lds<L_BPERM>(d_cyc, d_sink, w);
    }
    // store bandwidth
    {
        const size_t rows = 262144;
        float *buf;
        CK(hipMalloc(&buf, rows * 4096 * sizeof(float)));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        struct { const char *name; int vec; int row_floats; } cases[] = {
            {"float2 rows of 16376 B (STFT layout)", 2, 4094}, {"float2 rows of 16384 B", 2, 4096}, {"float4 rows of 16384 B", 4, 4096}};
        for (auto &c : cases) {
            for (int blocks : {1024, 2048, 4096}) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipEventRecord(e0));
                    if (c.vec == 2) store_kernel<2><<<blocks, 256>>>(buf, rows, c.row_floats);
                    else store_kernel<4><<<blocks, 256>>>(buf, rows, c.row_floats);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                printf("STORE %-40s blocks=%d  %.3f ms  %.2f TB/s\n", c.name, blocks, best,
                       rows * (double)c.row_floats * 4 / best / 1e9);
            }
        }
        {
            const size_t rows_total = 1000000;
            float *big;
            CK(hipMalloc(&big, rows_total * 16376 + 64));
            for (int blocks : {1024, 2048}) {
                float best = 1e9f;
                const size_t per = ((rows_total + blocks - 1) / blocks + 1) & ~(size_t)1;
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipEventRecord(e0));
                    stft_store_pattern_kernel<<<blocks, 256>>>(big, rows_total, per);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                printf("STORE stft pattern (1e6 rows of 16376 B, persistent WGs) blocks=%d  %.3f ms  %.2f TB/s\n", blocks, best,
                       rows_total * 16376.0 / best / 1e9);
            }
            CK(hipFree(big));
        }
        float *src;
        CK(hipMalloc(&src, rows * 4096 * sizeof(float)));
        CK(hipMemset(src, 1, rows * 4096 * sizeof(float)));
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            copy_kernel<<<4096, 256>>>((const float4 *)src, (float4 *)buf, rows * 1024);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("COPY float4 4 GiB: %.3f ms  read+write %.2f TB/s\n", best, 2.0 * rows * 4096.0 * 4 / best / 1e9);
    }
    return 0;
}
