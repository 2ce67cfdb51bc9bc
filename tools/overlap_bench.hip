// overlap_bench.hip -- does a stream of 8-byte-per-lane row stores overlap with independent VALU work on
// MI355X?  Persistent 256-thread workgroups (4 per CU), each iteration = NV fma instructions per thread
// followed by the STFT kernel's store pattern (two 16 376-byte rows).  Development aid for DESIGN.md section 4.
//   hipcc -O3 --offload-arch=gfx950 tools/overlap_bench.hip -o tools/bin/overlap_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int NV, int NSTORE, int WIDE>
__global__ void __launch_bounds__(256) overlap_kernel(float *out, size_t rows_total, size_t rows_per_block, float seed, size_t pitch)
{
    const int tid = threadIdx.x;
    size_t r0 = (size_t)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block;
    if (r1 > rows_total) r1 = rows_total;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = seed + (float)(tid + i);
    for (size_t r = r0; r + 1 < r1; r += 2) {
#pragma unroll
        for (int k = 0; k < NV / 8; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(acc[i], 1.0000001f, 0.5f);
        char *row0 = reinterpret_cast<char *>(out) + r * pitch + (pitch == 16384 ? 0 : -8);
        char *row1 = row0 + pitch;
        if (WIDE == 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if ((q > 0 || tid != 0) && q < NSTORE) *reinterpret_cast<float2 *>(row0 + 2048 * q + tid * 8) = make_float2(acc[q], acc[q]);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if ((q > 0 || tid != 0) && q < NSTORE) *reinterpret_cast<float2 *>(row1 + 2048 * q + tid * 8) = make_float2(acc[7 - q], acc[q]);
        } else {
            // the same bytes as 16-byte stores: thread t writes bins 2t, 2t+1 of segment q (4 segments of 4 KB per row)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < NSTORE) *reinterpret_cast<float4 *>(row0 + 8 + 4096 * q + tid * 16) = make_float4(acc[q], acc[q], acc[q + 4], acc[q + 4]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < NSTORE) *reinterpret_cast<float4 *>(row1 + 8 + 4096 * q + tid * 16) = make_float4(acc[7 - q], acc[q], acc[q], acc[q]);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7] == 12345.678f) out[tid] = acc[0];
}

template <int NV, int NSTORE, int WIDE>
void run(float *big, size_t rows_total, int blocks, hipEvent_t e0, hipEvent_t e1, size_t pitch = 16376)
{
    const size_t per = ((rows_total + blocks - 1) / blocks + 1) & ~(size_t)1;
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        overlap_kernel<NV, NSTORE, WIDE><<<blocks, 256>>>(big, rows_total, per, 1.0f, pitch);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("pitch=%zu NV=%4d fma/thread/iter  stores=%d/8 segments x %2d B/lane  blocks=%d : %.3f ms  (%.2f TB/s of stores)\n", pitch, NV, NSTORE * (WIDE == 8 ? 1 : 2), WIDE,
           blocks, best, rows_total * 16376.0 * NSTORE * (WIDE == 8 ? 1 : 2) / 8 / best / 1e9);
}

int main()
{
    const size_t rows_total = 1000000;
    float *big;
    CK(hipMalloc(&big, rows_total * 16384 + 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int blocks : {1024, 512, 2048}) {
        run<0, 8, 8>(big, rows_total, blocks, e0, e1);
        run<400, 0, 8>(big, rows_total, blocks, e0, e1);
        run<400, 8, 8>(big, rows_total, blocks, e0, e1);
        run<800, 0, 8>(big, rows_total, blocks, e0, e1);
        run<800, 1, 8>(big, rows_total, blocks, e0, e1);
        run<800, 4, 8>(big, rows_total, blocks, e0, e1);
        run<800, 8, 8>(big, rows_total, blocks, e0, e1);
        run<1600, 0, 8>(big, rows_total, blocks, e0, e1);
        run<1600, 8, 8>(big, rows_total, blocks, e0, e1);
        run<0, 4, 16>(big, rows_total, blocks, e0, e1);
        run<800, 4, 16>(big, rows_total, blocks, e0, e1);
        run<1600, 4, 16>(big, rows_total, blocks, e0, e1);
        run<0, 8, 8>(big, rows_total, blocks, e0, e1, 16384);
        run<800, 8, 8>(big, rows_total, blocks, e0, e1, 16384);
        run<1600, 8, 8>(big, rows_total, blocks, e0, e1, 16384);
        run<0, 4, 16>(big, rows_total, blocks, e0, e1, 16384);
        run<800, 4, 16>(big, rows_total, blocks, e0, e1, 16384);
        run<1600, 4, 16>(big, rows_total, blocks, e0, e1, 16384);
    }
    return 0;
}
