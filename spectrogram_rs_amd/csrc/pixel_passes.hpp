// pixel_passes.hpp -- the pixel stage on a column of magnitudes that lies in LDS (magnitude_in -> color_for -> put_pixel,
// simple_spectrogram.rs:141-161), shared by the fused kernels whose parameter blocks carry the generic tables (RowEntry, SampleEntry,
// the {threshold, RGBA} palette with the seed proof): stft_mixed.hip and stft4800_wg.hip.
//   m[0 .. M)   the column: float2 per bin, (l, r) of a stereo frame or (frame a, frame b) of a mono frame pair (two_columns)
//   vbuf        room for p.n_samples float2 behind it
// P needs: n_samples, samples, interp, rows, R, pairs, n_frames, rgba, pal, guess_a, guess_b.  The caller has a barrier between its
// last write of m and this call; the function ends without one.
#pragma once

#include "sgx_internal.hpp"

namespace sgx {

template <uint32_t NT, typename P>
__device__ __forceinline__ void pixel_passes(const P &p, const float2 *m, float2 *vbuf, uint32_t M, bool two_columns, uint32_t pair,
                                             long long row_a, long long row_b, uint32_t tid)
{
    const int32_t last = (int32_t)M - 1;
    // ---- sample pass (interpolated_frequency_sample.rs:79-105)
    for (uint32_t sidx = tid; sidx < p.n_samples; sidx += NT) {
        const SampleEntry se = p.samples[sidx];
        float2 v;
        if (p.interp == SGX_INTERP_COSINE) {
            const float2 a = m[se.i0], b = m[se.i1];
            v.x = a.x * se.w1 + b.x * se.w2;
            v.y = a.y * se.w1 + b.y * se.w2;
        } else {
            const int32_t x1 = se.i0;
            const int32_t x0 = x1 > 0 ? x1 - 1 : 0;
            const int32_t x2 = x1 + 1 < last ? x1 + 1 : last;
            const int32_t x3 = x1 + 2 < last ? x1 + 2 : last;
            const float2 y0 = m[x0], y1 = m[x1], y2 = m[x2], y3 = m[x3];
            const float mu = se.w0, mu2 = se.w1, mu3 = se.w2;
            v = cubic_pair(y0, y1, y2, y3, mu, mu2, mu3);
        }
        vbuf[sidx] = v;
    }
    __syncthreads();
    // ---- row pass (:60-75 the mean; colorscheme.rs:59-61,67-70; simple_spectrogram.rs:150-160)
    const bool st_a = row_a >= 0 && (unsigned long long)row_a < p.n_frames;
    const bool st_b = two_columns && row_b >= 0 && (unsigned long long)row_b < p.n_frames;
    uint32_t *dst_a = reinterpret_cast<uint32_t *>(p.rgba) + ((size_t)(st_a ? row_a : 0) * p.pairs + pair) * p.R;
    uint32_t *dst_b = reinterpret_cast<uint32_t *>(p.rgba) + ((size_t)(st_b ? row_b : 0) * p.pairs + pair) * p.R;
    auto pixel = [&](float l, float r) -> uint32_t {
        const float power = (l * l) + (r * r);
        const float u = fmaf(__builtin_amdgcn_logf(power + 1e-7f), p.guess_a, p.guess_b);
        int idx = (int)floorf(u - 0.5f);
        idx = idx < 0 ? 0 : (idx > 254 ? 254 : idx);
        const uint4 e = *reinterpret_cast<const uint4 *>(p.pal + idx);   // {thr(idx), rgba(idx), thr(idx + 1), rgba(idx + 1)}
        return power >= __uint_as_float(e.x) ? e.w : e.y;
    };
    for (uint32_t py = tid; py < p.R; py += NT) {
        const RowEntry row = p.rows[py];
        float sl = 0.0f, sr = 0.0f;  // Complex::sum starts at zero
        for (uint32_t i = 0; i < row.count; ++i) {
            const float2 v = vbuf[row.first + i];
            sl = sl + v.x;
            sr = sr + v.y;
        }
        float l = sl, r = sr;
        if (row.count > 1) {  // x / 1.0 == x: only rows that average several samples divide (:72)
            l = sl / row.count_f;
            r = sr / row.count_f;
        }
        const uint32_t y = p.R - 1 - py;  // simple_spectrogram.rs:150
        if (two_columns) {  // mono -> (s, s): both channels carry the same magnitude
            if (st_a) dst_a[y] = pixel(l, l);
            if (st_b) dst_b[y] = pixel(r, r);
        } else if (st_a) {
            dst_a[y] = pixel(l, r);
        }
    }
}

}  // namespace sgx
