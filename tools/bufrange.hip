// tools/bufrange.hip -- which offsets does gfx950's raw-buffer range check see?  (stft4096_real.hip drops the absent
// row of an odd launch through a zero-record descriptor while its stores carry a scalar offset.)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/bufrange tools/bufrange.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(void *base, int records)
{
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, records, 0x00020000);
}

// out[0..]: results of loads; mem: 4096 words, word i = 1000 + i
__global__ void probe(uint32_t *mem, uint32_t *out, uint32_t *storebuf)
{
    const int lane = threadIdx.x;
    // 1. records = 256 bytes (64 words).  voffset = lane * 4 (in range), soffset = 1024 (beyond the records)
    out[lane] = __builtin_amdgcn_raw_buffer_load_b32(rsrc(mem, 256), lane * 4, 1024, 0);
    // 2. records = 256, voffset = lane * 4 + 1024 (beyond), soffset 0
    out[64 + lane] = __builtin_amdgcn_raw_buffer_load_b32(rsrc(mem, 256), lane * 4 + 1024, 0, 0);
    // 3. records = 2048, voffset = lane * 4, soffset = 1024 (sum in range)
    out[128 + lane] = __builtin_amdgcn_raw_buffer_load_b32(rsrc(mem, 2048), lane * 4, 1024, 0);
    // 4. records = 1024 + 128: voffset + soffset in range for lanes < 32 only
    out[192 + lane] = __builtin_amdgcn_raw_buffer_load_b32(rsrc(mem, 1024 + 128), lane * 4, 1024, 0);
    // stores: 5. records = 0, voffset = lane * 4, soffset = 2048
    __builtin_amdgcn_raw_buffer_store_b32(7u, rsrc(storebuf, 0), lane * 4, 2048, 0);
    // 6. records = 0, soffset = 0
    __builtin_amdgcn_raw_buffer_store_b32(8u, rsrc(storebuf, 0), lane * 4 + 256, 0, 0);
    // 7. records = 0x7fffffff, voffset with bit 31 set, soffset = 2048
    __builtin_amdgcn_raw_buffer_store_b32(9u, rsrc(storebuf, 0x7fffffff), (int)(0x80000000u | (lane * 4 + 512)), 2048, 0);
    // 8. records = 256, voffset in range, soffset 2048 -> lands at 2048 + lane*4 if soffset is not checked
    __builtin_amdgcn_raw_buffer_store_b32(10u, rsrc(storebuf, 256), lane * 4, 4096, 0);
}

int main()
{
    uint32_t *mem, *out, *sb;
    hipMalloc(&mem, 4096 * 4); hipMalloc(&out, 256 * 4); hipMalloc(&sb, 4096 * 4);
    std::vector<uint32_t> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = 1000 + i;
    hipMemcpy(mem, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipMemset(sb, 0, 4096 * 4); hipMemset(out, 0xff, 256 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mem, out, sb);
    hipDeviceSynchronize();
    std::vector<uint32_t> o(256), s(4096);
    hipMemcpy(o.data(), out, 256 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(s.data(), sb, 4096 * 4, hipMemcpyDeviceToHost);
    printf("1 load records 256, voffset in, soffset 1024 out : lane0 %u lane63 %u  (0 = dropped; 1256 = read past the records)\n", o[0], o[63]);
    printf("2 load records 256, voffset out, soffset 0       : lane0 %u lane63 %u\n", o[64], o[127]);
    printf("3 load records 2048, sum in range                : lane0 %u lane63 %u  (expect 1256, 1319)\n", o[128], o[191]);
    printf("4 load records 1152, sum in range for lanes < 32 : lane31 %u lane32 %u\n", o[192 + 31], o[192 + 32]);
    int n7 = 0, n8 = 0, n9 = 0, n10 = 0;
    for (int i = 0; i < 4096; ++i) { n7 += s[i] == 7; n8 += s[i] == 8; n9 += s[i] == 9; n10 += s[i] == 10; }
    printf("5 store records 0, soffset 2048: %d words written (0 = dropped)\n", n7);
    printf("6 store records 0, soffset 0   : %d words written\n", n8);
    printf("7 store voffset bit 31, soffset 2048, records 0x7fffffff: %d words written\n", n9);
    printf("8 store records 256, voffset in, soffset 4096: %d words written (64 = the scalar offset is not range-checked)\n", n10);
    return 0;
}
