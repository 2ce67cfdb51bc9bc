// sgx_view.hip -- the default widget's variant of the pixel path (GPUSpectrogram, src/widgets/gpu_spectrogram.rs):
// frames are kept as rows of an F16F16 texture used as a ring (VIEWPORT_FRAMES x (W - 1) texels, :21,218-226), and a
// fragment program turns the ring into the picture -- log-frequency lookup, dB, pan, 32 x 32 palette texture
// (:150-186).  Here the texture is a device buffer, the upload loop (:255-275) is a device-to-device row copy, and the
// fragment program is one kernel over the viewport.
//
// This is NOT the parity target: OpenGL leaves the filtering arithmetic to the
// implementation.  What is restated is the program text and the sampler state the reference sets:
//   fft texture     F16F16, wrap REPEAT, min / mag LINEAR (the mipmaps it allocates are never selected by LINEAR)   :218-226,282-285
//   palette texture RGBA32F 32 x 32 = ColorScheme::lookup_table(32), wrap CLAMP (to edge), LINEAR                    :230-238,286-289
//   GL_LINEAR       sample point s * size - 0.5, the two nearest texels per axis, weights = the fractional part
// in float32, with the shadowed locals of quirk Q9 (32 and 22030 Hz whatever the uniforms say).  NaN coordinates
// (pan = 0 / 0 at silence) are undefined in GL; here they sample coordinate 0.
#include <hip/hip_fp16.h>

#include <cmath>
#include <cstdio>
#include <new>

#include "sgx_internal.hpp"

struct sgx_view {
    sgx_ctx *ctx = nullptr;
    uint32_t rows = 0;           // VIEWPORT_FRAMES (texture height)
    uint32_t offset = 0;         // gpu_spectrogram.rs:67,274
    __half2 *d_ring = nullptr;   // [rows][M] (l, r) half pairs
    float4 *d_palette = nullptr; // [32][32] RGBA32F: texel (x = j, y = i) = lookup_table(32)[i][j]
    unsigned long long palette_gen = ~0ull;
};

namespace sgx {

struct ViewParams {
    const __half2 *ring;
    const float4 *palette;
    float4 *out;
    uint32_t M, rows, offset, width, height;
    float min_db, max_db;
};

__device__ __forceinline__ float2 texel(const ViewParams &p, int x, int y)
{
    const __half2 h = p.ring[(size_t)y * p.M + x];
    return make_float2(__low2float(h), __high2float(h));
}

// GL_LINEAR with wrap REPEAT on both axes; s, t in texture coordinates
__device__ __forceinline__ float2 sample_fft(const ViewParams &p, float s, float t)
{
    const float x = s * (float)p.M - 0.5f, y = t * (float)p.rows - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    const float ax = x - fx0, ay = y - fy0;
    auto wrap = [](float f, int n) { int i = (int)fmodf(f, (float)n); return i < 0 ? i + n : i; };
    const int x0 = wrap(fx0, (int)p.M), x1 = wrap(fx0 + 1.0f, (int)p.M);
    const int y0 = wrap(fy0, (int)p.rows), y1 = wrap(fy0 + 1.0f, (int)p.rows);
    const float2 a = texel(p, x0, y0), b = texel(p, x1, y0), c = texel(p, x0, y1), d = texel(p, x1, y1);
    const float lx0 = a.x + (b.x - a.x) * ax, ly0 = a.y + (b.y - a.y) * ax;
    const float lx1 = c.x + (d.x - c.x) * ax, ly1 = c.y + (d.y - c.y) * ax;
    return make_float2(lx0 + (lx1 - lx0) * ay, ly0 + (ly1 - ly0) * ay);
}

// GL_LINEAR with wrap CLAMP_TO_EDGE, 32 x 32
__device__ __forceinline__ float4 sample_palette(const ViewParams &p, float s, float t)
{
    if (!(s == s)) s = 0.0f;
    if (!(t == t)) t = 0.0f;
    const float x = s * 32.0f - 0.5f, y = t * 32.0f - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    const float ax = x - fx0, ay = y - fy0;
    auto cl = [](float f) { return f < 0.0f ? 0 : (f > 31.0f ? 31 : (int)f); };
    const int x0 = cl(fx0), x1 = cl(fx0 + 1.0f), y0 = cl(fy0), y1 = cl(fy0 + 1.0f);
    const float4 a = p.palette[y0 * 32 + x0], b = p.palette[y0 * 32 + x1], c = p.palette[y1 * 32 + x0], d = p.palette[y1 * 32 + x1];
    auto mix = [](float u, float v, float w) { return u + (v - u) * w; };
    return make_float4(mix(mix(a.x, b.x, ax), mix(c.x, d.x, ax), ay), mix(mix(a.y, b.y, ax), mix(c.y, d.y, ax), ay),
                       mix(mix(a.z, b.z, ax), mix(c.z, d.z, ax), ay), mix(mix(a.w, b.w, ax), mix(c.w, d.w, ax), ay));
}

// the fragment program of gpu_spectrogram.rs:150-186, one thread per fragment; out[py][px], py = 0 at the BOTTOM (GL)
__global__ void __launch_bounds__(256) view_fragment_kernel(ViewParams p)
{
    const uint32_t px = blockIdx.x * blockDim.x + threadIdx.x, py = blockIdx.y;
    if (px >= p.width) return;
    const float uvx = ((float)px + 0.5f) / (float)p.width, uvy = ((float)py + 0.5f) / (float)p.height;
    const float min_frequency = 32.0f, max_frequency = 22030.0f;   // the shader's locals shadow its uniforms (quirk Q9)
    const float log_min = logf(min_frequency), log_max = logf(max_frequency);
    const float log_frequency = uvy * (log_max - log_min) + log_min;
    const float log_frequency_mapped = expf(log_frequency) / max_frequency;
    const float time = (uvx * (float)p.rows + (float)p.offset) / (float)p.rows;
    const float2 magnitude = sample_fft(p, log_frequency_mapped, time);   // texture(fft, coord.yx)
    const float power = magnitude.x * magnitude.x + magnitude.y * magnitude.y;
    const float magnitude_log = 10.0f * logf(power + 1e-7f) / logf(10.0f);
    const float magnitude_db = (magnitude_log - p.min_db) / (p.max_db - p.min_db);
    const float pan = magnitude.y / (magnitude.x + magnitude.y);
    p.out[(size_t)py * p.width + px] = sample_palette(p, pan, magnitude_db);
}

}  // namespace sgx

namespace {

int view_fail_hip(sgx_view *v, hipError_t e, const char *what)
{
    char buf[512];
    std::snprintf(buf, sizeof(buf), "%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
    if (v && v->ctx) v->ctx->err = buf;
    return SGX_ERR_HIP;
}

#define VIEW_HIP(v, call)                                              \
    do {                                                               \
        hipError_t e__ = (call);                                       \
        if (e__ != hipSuccess) return view_fail_hip((v), e__, #call);  \
    } while (0)

}  // namespace

namespace sgx {
void detach_views(sgx_ctx *c)
{
    for (sgx_view *v : c->views) v->ctx = nullptr;
    c->views.clear();
}
sgx_ctx *view_context(const sgx_view *v) { return v ? v->ctx : nullptr; }
}  // namespace sgx

extern "C" {

int sgx_view_create(sgx_ctx *c, uint32_t viewport_frames, sgx_view **out)
{
    if (out) *out = nullptr;
    if (!c || !out || viewport_frames < 2) return SGX_ERR_INVALID_ARG;
    sgx_view *v = new (std::nothrow) sgx_view();
    if (!v) return SGX_ERR_NOMEM;
    v->ctx = c;
    v->rows = viewport_frames;
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&v->d_ring), (size_t)v->rows * c->M * sizeof(__half2));
    if (e == hipSuccess) e = hipMemsetAsync(v->d_ring, 0, (size_t)v->rows * c->M * sizeof(__half2), c->stream);   // a fresh texture: zeros
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&v->d_palette), 32 * 32 * sizeof(float4));
    if (e != hipSuccess) {
        const int rc = view_fail_hip(v, e, "sgx_view_create");
        sgx_view_destroy(v);
        return rc;
    }
    c->views.push_back(v);   // sgx_destroy(ctx) detaches the views still alive: their calls then fail instead of reading freed memory
    *out = v;
    return SGX_OK;
}

void sgx_view_destroy(sgx_view *v)
{
    if (!v) return;
    if (v->ctx) {
        (void)hipSetDevice(v->ctx->device);
        (void)hipStreamSynchronize(v->ctx->stream);
        auto &vs = v->ctx->views;
        for (size_t i = 0; i < vs.size(); ++i)
            if (vs[i] == v) { vs.erase(vs.begin() + (long)i); break; }
    }
    if (v->d_ring) (void)hipFree(v->d_ring);
    if (v->d_palette) (void)hipFree(v->d_palette);
    delete v;
}

int sgx_view_write_rows(sgx_view *v, const void *d_rows_f16, size_t n_rows, uint32_t *offset_out)
{
    if (!v || !v->ctx) return SGX_ERR_INVALID_ARG;   // (no context: it was destroyed before this view)
    sgx_ctx *c = v->ctx;
    if (n_rows && !d_rows_f16) { c->err = "sgx_view_write_rows: null buffer"; return SGX_ERR_INVALID_ARG; }
    VIEW_HIP(v, hipSetDevice(c->device));
    const size_t row_bytes = (size_t)c->M * sizeof(__half2);
    const char *src = static_cast<const char *>(d_rows_f16);
    // gpu_spectrogram.rs:255-275: fill up to the top of the texture, wrap, go on
    while (n_rows > 0) {
        const size_t room = v->rows - v->offset, take = n_rows < room ? n_rows : room;
        VIEW_HIP(v, hipMemcpyAsync(reinterpret_cast<char *>(v->d_ring) + (size_t)v->offset * row_bytes, src, take * row_bytes,
                                   hipMemcpyDeviceToDevice, c->stream));
        v->offset = (uint32_t)((v->offset + take) % v->rows);
        src += take * row_bytes;
        n_rows -= take;
    }
    if (offset_out) *offset_out = v->offset;
    return SGX_OK;
}

int sgx_view_draw(sgx_view *v, uint32_t width, uint32_t height, float *d_rgba_f32)
{
    if (!v || !v->ctx) return SGX_ERR_INVALID_ARG;
    sgx_ctx *c = v->ctx;
    if (!d_rgba_f32 || width == 0 || height == 0 || height > 65535) { c->err = "sgx_view_draw: bad argument"; return SGX_ERR_INVALID_ARG; }
    VIEW_HIP(v, hipSetDevice(c->device));
    if (v->palette_gen != c->palette_gen) {
        // the palette texture is rebuilt whenever the colour scheme changed (set_palette, gpu_spectrogram.rs:329-333)
        std::vector<float> lut(32 * 32 * 4);
        const int rc = sgx_lookup_table(c, 32, lut.data());
        if (rc != SGX_OK) return rc;
        VIEW_HIP(v, hipStreamSynchronize(c->stream));
        VIEW_HIP(v, hipMemcpy(v->d_palette, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice));
        v->palette_gen = c->palette_gen;
    }
    sgx::ViewParams p{};
    p.ring = v->d_ring;
    p.palette = v->d_palette;
    p.out = reinterpret_cast<float4 *>(d_rgba_f32);
    p.M = c->M;
    p.rows = v->rows;
    p.offset = v->offset;
    p.width = width;
    p.height = height;
    p.min_db = c->cfg.min_db;
    p.max_db = c->cfg.max_db;
    hipLaunchKernelGGL(sgx::view_fragment_kernel, dim3((width + 255) / 256, height), dim3(256), 0, c->stream, p);
    VIEW_HIP(v, hipGetLastError());
    return SGX_OK;
}

uint32_t sgx_view_offset(const sgx_view *v) { return v ? v->offset : 0; }

}  // extern "C"
