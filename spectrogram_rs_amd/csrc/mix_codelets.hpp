// mix_codelets.hpp -- in-register forward DFTs of 2, 3, 4, 5 and 7 points and of their products RA * RB (natural order in and out):
// the butterflies of stft_mixed.hip (any smooth length) and stft4800_wg.hip (the application's 4800 points).
#pragma once

#include <hip/hip_runtime.h>

namespace sgx {

namespace mix {

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// a - i b and a + i b
__device__ __forceinline__ float2 sub_i(float2 a, float2 b) { return make_float2(a.x + b.y, a.y - b.x); }
__device__ __forceinline__ float2 add_i(float2 a, float2 b) { return make_float2(a.x - b.y, a.y + b.x); }
__device__ __forceinline__ float2 scale2(float2 a, float c) { return make_float2(a.x * c, a.y * c); }

// forward r-point DFTs (kernel e^{-2 pi i q k / r}), in place on x[0 .. r)
__device__ __forceinline__ void dft2(float2 *x)
{
    const float2 a = x[0], b = x[1];
    x[0] = cadd(a, b);
    x[1] = csub(a, b);
}
__device__ __forceinline__ void dft3(float2 *x)
{
    const float s = 0.86602540378443864676f;  // sin(2 pi / 3)
    const float2 t = cadd(x[1], x[2]), d = scale2(csub(x[1], x[2]), s);
    const float2 a = make_float2(fmaf(t.x, -0.5f, x[0].x), fmaf(t.y, -0.5f, x[0].y));
    x[0] = cadd(x[0], t);
    x[1] = sub_i(a, d);
    x[2] = add_i(a, d);
}
__device__ __forceinline__ void dft4(float2 *x)
{
    const float2 b0 = cadd(x[0], x[2]), b1 = csub(x[0], x[2]), b2 = cadd(x[1], x[3]), d = csub(x[1], x[3]);
    x[0] = cadd(b0, b2);
    x[2] = csub(b0, b2);
    x[1] = sub_i(b1, d);
    x[3] = add_i(b1, d);
}
__device__ __forceinline__ void dft5(float2 *x)
{
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;  // cos(2 pi / 5), cos(4 pi / 5)
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;   // sin(2 pi / 5), sin(4 pi / 5)
    const float2 t1 = cadd(x[1], x[4]), t2 = cadd(x[2], x[3]), t3 = csub(x[1], x[4]), t4 = csub(x[2], x[3]);
    const float2 a1 = make_float2(fmaf(t2.x, c2, fmaf(t1.x, c1, x[0].x)), fmaf(t2.y, c2, fmaf(t1.y, c1, x[0].y)));
    const float2 a2 = make_float2(fmaf(t2.x, c1, fmaf(t1.x, c2, x[0].x)), fmaf(t2.y, c1, fmaf(t1.y, c2, x[0].y)));
    const float2 b1 = make_float2(fmaf(t4.x, s2, t3.x * s1), fmaf(t4.y, s2, t3.y * s1));
    const float2 b2 = make_float2(fmaf(t4.x, -s1, t3.x * s2), fmaf(t4.y, -s1, t3.y * s2));
    x[0] = cadd(x[0], cadd(t1, t2));
    x[1] = sub_i(a1, b1);
    x[4] = add_i(a1, b1);
    x[2] = sub_i(a2, b2);
    x[3] = add_i(a2, b2);
}
__device__ __forceinline__ void dft7(float2 *x)
{
    const float c1 = 0.62348980185873353053f, c2 = -0.22252093395631440429f, c3 = -0.90096886790241912624f;  // cos(2 pi k / 7)
    const float s1 = 0.78183148246802980871f, s2 = 0.97492791218182360702f, s3 = 0.43388373911755812048f;    // sin(2 pi k / 7)
    const float2 t1 = cadd(x[1], x[6]), t2 = cadd(x[2], x[5]), t3 = cadd(x[3], x[4]);
    const float2 u1 = csub(x[1], x[6]), u2 = csub(x[2], x[5]), u3 = csub(x[3], x[4]);
    // a_k = x0 + sum_q cos(2 pi q k / 7) t_q ; b_k = sum_q sin(2 pi q k / 7) u_q   (q k mod 7 folded to 1..3 with sign)
    const float2 a1 = make_float2(fmaf(t3.x, c3, fmaf(t2.x, c2, fmaf(t1.x, c1, x[0].x))), fmaf(t3.y, c3, fmaf(t2.y, c2, fmaf(t1.y, c1, x[0].y))));
    const float2 a2 = make_float2(fmaf(t3.x, c1, fmaf(t2.x, c3, fmaf(t1.x, c2, x[0].x))), fmaf(t3.y, c1, fmaf(t2.y, c3, fmaf(t1.y, c2, x[0].y))));
    const float2 a3 = make_float2(fmaf(t3.x, c2, fmaf(t2.x, c1, fmaf(t1.x, c3, x[0].x))), fmaf(t3.y, c2, fmaf(t2.y, c1, fmaf(t1.y, c3, x[0].y))));
    const float2 b1 = make_float2(fmaf(u3.x, s3, fmaf(u2.x, s2, u1.x * s1)), fmaf(u3.y, s3, fmaf(u2.y, s2, u1.y * s1)));
    const float2 b2 = make_float2(fmaf(u3.x, -s1, fmaf(u2.x, -s3, u1.x * s2)), fmaf(u3.y, -s1, fmaf(u2.y, -s3, u1.y * s2)));
    const float2 b3 = make_float2(fmaf(u3.x, s2, fmaf(u2.x, -s1, u1.x * s3)), fmaf(u3.y, s2, fmaf(u2.y, -s1, u1.y * s3)));
    x[0] = cadd(x[0], cadd(t1, cadd(t2, t3)));
    x[1] = sub_i(a1, b1);
    x[6] = add_i(a1, b1);
    x[2] = sub_i(a2, b2);
    x[5] = add_i(a2, b2);
    x[3] = sub_i(a3, b3);
    x[4] = add_i(a3, b3);
}

template <int N>
__device__ __forceinline__ void dft_prime(float2 *x)
{
    if (N == 2) dft2(x);
    else if (N == 3) dft3(x);
    else if (N == 4) dft4(x);
    else if (N == 5) dft5(x);
    else dft7(x);
}

#include "mix_consts.inc"

// forward R-point DFT, R = RA * RB, natural order in and out: input q = RB qa + qb, output k = ka + RA kb
template <int RA, int RB>
__device__ __forceinline__ void dft_composite(float2 (&x)[RA * RB])
{
    if constexpr (RB == 1) {
        dft_prime<RA>(x);
    } else {
        constexpr int R = RA * RB;
        float2 t[RB][RA];
#pragma unroll
        for (int qb = 0; qb < RB; ++qb) {
#pragma unroll
            for (int qa = 0; qa < RA; ++qa) t[qb][qa] = x[RB * qa + qb];
            dft_prime<RA>(t[qb]);
        }
#pragma unroll
        for (int ka = 0; ka < RA; ++ka) {
            float2 c[RB];
            c[0] = t[0][ka];
#pragma unroll
            for (int qb = 1; qb < RB; ++qb) c[qb] = ka == 0 ? t[qb][0] : cmul(t[qb][ka], cw<R>((qb * ka) % R));
            dft_prime<RB>(c);
#pragma unroll
            for (int kb = 0; kb < RB; ++kb) x[ka + RA * kb] = c[kb];
        }
    }
}

}  // namespace mix

}  // namespace sgx
