// deinterleave.hip -- an interleaved multi-channel stream [n][C] split into per-pair (l, r) planes: what the 4096-point kernel runs on when
// a stream has more than two channels (stft4096_wg.hip; the reference itself takes one or two channels, audio_input_list_model.rs:67-75:
// channel pairs (2p, 2p + 1) are this library's extension).  The 16384-point kernel reads its pairs where they
// lie (stft16384_w.hip) and does not come through here.
#include <algorithm>

#include "sgx_internal.hpp"

namespace sgx {

namespace {

// [n][C] interleaved -> per-pair planes of (l, r): plane p holds samples [first, first + n) of channels (2p, 2p + 1)
__global__ void __launch_bounds__(256) deinterleave_pairs_kernel(const float *pcm, float *planes, size_t plane_floats,
                                                                 size_t first, size_t n, uint32_t C, uint32_t pairs)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float2 *row = reinterpret_cast<const float2 *>(pcm + (first + i) * C);
        for (uint32_t pr = 0; pr < pairs; ++pr)
            reinterpret_cast<float2 *>(planes + (size_t)pr * plane_floats)[i] = row[pr];
    }
}

// the same, two samples per thread and 16 bytes per access: C a multiple of 4, the stream and the planes 16-byte aligned
__global__ void __launch_bounds__(256) deinterleave_pairs_wide_kernel(const float *pcm, float *planes, size_t plane_floats,
                                                                      size_t first, size_t n_half, uint32_t C, uint32_t pairs)
{
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_half; j += (size_t)gridDim.x * blockDim.x) {
        const float4 *s0 = reinterpret_cast<const float4 *>(pcm + (first + 2 * j) * C), *s1 = reinterpret_cast<const float4 *>(pcm + (first + 2 * j + 1) * C);
        for (uint32_t q = 0; q < pairs / 2; ++q) {
            const float4 a = s0[q], b = s1[q];
            reinterpret_cast<float4 *>(planes + (size_t)(2 * q) * plane_floats)[j] = make_float4(a.x, a.y, b.x, b.y);
            reinterpret_cast<float4 *>(planes + (size_t)(2 * q + 1) * plane_floats)[j] = make_float4(a.z, a.w, b.z, b.w);
        }
    }
}

}  // namespace

// (l, r) pair planes of an interleaved multi-channel stream, plain order: plane p = samples [first, first + n) of channels 2p, 2p + 1
hipError_t launch_deinterleave_pairs(const sgx_ctx *c, const float *d_pcm, float *d_planes, size_t plane_floats, size_t first_sample, size_t n_samples,
                                     uint32_t channels, uint32_t pairs)
{
    const int n_cu = c->n_cu;
    const bool wide = channels % 4 == 0 && reinterpret_cast<uintptr_t>(d_pcm) % 16 == 0 && reinterpret_cast<uintptr_t>(d_planes) % 16 == 0 &&
                      plane_floats % 4 == 0 && (first_sample * channels) % 4 == 0 && n_samples >= 2;
    size_t done = 0;
    if (wide) {
        const size_t n_half = n_samples / 2;
        const unsigned blocks = (unsigned)std::min<size_t>((n_half + 255) / 256, (size_t)n_cu * 16);
        hipLaunchKernelGGL(deinterleave_pairs_wide_kernel, dim3(blocks), dim3(256), 0, c->stream, d_pcm, d_planes, plane_floats, first_sample, n_half,
                           channels, pairs);
        done = 2 * n_half;
    }
    if (done < n_samples) {   // everything, or the odd sample at the end
        const size_t rest = n_samples - done;
        const unsigned blocks = (unsigned)std::min<size_t>((rest + 255) / 256, (size_t)n_cu * 16);
        hipLaunchKernelGGL(deinterleave_pairs_kernel, dim3(blocks), dim3(256), 0, c->stream, d_pcm, d_planes + 2 * done, plane_floats,
                           first_sample + done, rest, channels, pairs);
    }
    return hipGetLastError();
}

}  // namespace sgx
