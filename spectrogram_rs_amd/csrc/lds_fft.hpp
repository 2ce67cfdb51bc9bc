// lds_fft.hpp -- in-place power-of-two FFT stages over an array that lives in LDS, radix 4 with one radix-2
// stage where the length needs it (shared by the chirp-z kernel; the arithmetic of one stage is two radix-2
// stages merged: half the LDS round trips and barriers of a radix-2 ladder).
//
// Forward = decimation in frequency, natural order in, DIGIT-reversed order out: after a stage on blocks of
// length B = 4h, sub-block q (q = 0..3) of each block holds the length-h problem of the bins k = q (mod 4).
// pos_of() says where bin k ends up; the inverse (decimation in time, conjugate twiddles) undoes exactly that
// order, so a pointwise multiply in between needs its table stored at pos_of(k) and nothing is ever permuted.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace sgx {
namespace ldsfft {

// position of bin k after the forward stages: radix-4 digits first (least significant digit of k = largest stride),
// then the single radix-2 digit when log2(N) is odd
__host__ __device__ inline uint32_t pos_of(uint32_t k, uint32_t logN)
{
    uint32_t pos = 0, len = 1u << logN;
    for (uint32_t i = 0; i < logN / 2; ++i) {
        len >>= 2;
        pos += (k & 3u) * len;
        k >>= 2;
    }
    if (logN & 1u) pos += k & 1u;
    return pos;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b)  // a * conj(b)
{
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -(a.x * b.y)));
}

// forward FFTs of every aligned block of length 2^logB in s[0 .. 2^logArr) (logB = logArr: one transform);
// tw[i] = e^{-2 pi i / T}, i < T = 2^logTw >= 2^logB (the full circle); ends with a barrier
__device__ __forceinline__ void forward_dif(float2 *s, uint32_t logArr, uint32_t logB, const float2 *tw, uint32_t logTw, uint32_t tid,
                                            uint32_t nt)
{
    const uint32_t N = 1u << logArr, logN = logB;
    uint32_t lh = logN - 2;  // log2 of the quarter block
    for (uint32_t st = 0; st < logN / 2; ++st, lh -= 2) {
        const uint32_t h = 1u << lh, step = (1u << logTw) >> (lh + 2);  // w_B^j = tw[j * T / B]
        for (uint32_t b = tid; b < (N >> 2); b += nt) {
            const uint32_t grp = b >> lh, j = b & (h - 1);
            const uint32_t i0 = (grp << (lh + 2)) + j;
            const float2 a0 = s[i0], a1 = s[i0 + h], a2 = s[i0 + 2 * h], a3 = s[i0 + 3 * h];
            const float2 b0 = make_float2(a0.x + a2.x, a0.y + a2.y), b1 = make_float2(a0.x - a2.x, a0.y - a2.y);
            const float2 b2 = make_float2(a1.x + a3.x, a1.y + a3.y);
            const float2 b3 = make_float2(a1.y - a3.y, a3.x - a1.x);  // -i (a1 - a3)
            const float2 y0 = make_float2(b0.x + b2.x, b0.y + b2.y), y2 = make_float2(b0.x - b2.x, b0.y - b2.y);
            const float2 y1 = make_float2(b1.x + b3.x, b1.y + b3.y), y3 = make_float2(b1.x - b3.x, b1.y - b3.y);
            s[i0] = y0;
            if (j == 0) {
                s[i0 + h] = y1; s[i0 + 2 * h] = y2; s[i0 + 3 * h] = y3;
            } else {
                s[i0 + h] = cmul(y1, tw[j * step]);
                s[i0 + 2 * h] = cmul(y2, tw[2 * j * step]);
                s[i0 + 3 * h] = cmul(y3, tw[3 * j * step]);
            }
        }
        __syncthreads();
    }
    if (logN & 1u) {
        for (uint32_t b = tid; b < (N >> 1); b += nt) {
            const float2 u = s[2 * b], v = s[2 * b + 1];
            s[2 * b] = make_float2(u.x + v.x, u.y + v.y);
            s[2 * b + 1] = make_float2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
    }
}

// unnormalised inverse FFT of the digit-reversed array forward_dif leaves: natural order out; ends with a barrier
__device__ __forceinline__ void forward_dif(float2 *s, uint32_t logN, const float2 *tw, uint32_t tid, uint32_t nt)
{
    forward_dif(s, logN, logN, tw, logN, tid, nt);
}

__device__ __forceinline__ void inverse_dit(float2 *s, uint32_t logN, const float2 *tw, uint32_t tid, uint32_t nt)
{
    const uint32_t N = 1u << logN;
    if (logN & 1u) {
        for (uint32_t b = tid; b < (N >> 1); b += nt) {
            const float2 u = s[2 * b], v = s[2 * b + 1];
            s[2 * b] = make_float2(u.x + v.x, u.y + v.y);
            s[2 * b + 1] = make_float2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
    }
    uint32_t lh = logN & 1u;  // quarter block: 1 (or 2 after the radix-2 stage), then x4 per stage
    for (uint32_t st = 0; st < logN / 2; ++st, lh += 2) {
        const uint32_t h = 1u << lh, step = N >> (lh + 2);
        for (uint32_t b = tid; b < (N >> 2); b += nt) {
            const uint32_t grp = b >> lh, j = b & (h - 1);
            const uint32_t i0 = (grp << (lh + 2)) + j;
            const float2 x0 = s[i0];
            float2 t1 = s[i0 + h], t2 = s[i0 + 2 * h], t3 = s[i0 + 3 * h];
            if (j != 0) {
                t1 = cmul_conj(t1, tw[j * step]);
                t2 = cmul_conj(t2, tw[2 * j * step]);
                t3 = cmul_conj(t3, tw[3 * j * step]);
            }
            const float2 c0 = make_float2(x0.x + t2.x, x0.y + t2.y), c1 = make_float2(x0.x - t2.x, x0.y - t2.y);
            const float2 c2 = make_float2(t1.x + t3.x, t1.y + t3.y);
            const float2 c3 = make_float2(t3.y - t1.y, t1.x - t3.x);  // +i (t1 - t3)
            s[i0] = make_float2(c0.x + c2.x, c0.y + c2.y);
            s[i0 + h] = make_float2(c1.x + c3.x, c1.y + c3.y);
            s[i0 + 2 * h] = make_float2(c0.x - c2.x, c0.y - c2.y);
            s[i0 + 3 * h] = make_float2(c1.x - c3.x, c1.y - c3.y);
        }
        __syncthreads();
    }
}

}  // namespace ldsfft
}  // namespace sgx
