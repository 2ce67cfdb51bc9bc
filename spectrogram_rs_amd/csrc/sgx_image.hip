// sgx_image.hip -- the CPU pixel path's image ring (SimpleSpectrogram, src/widgets/simple_spectrogram.rs) on the device.
//
// The reference keeps a row-major 1024 x 1024 RGBA Pixbuf (:89-94), writes every new frame as ONE pixel column at x = offset,
// row height - 1 - py (:140-161), advances offset = (px + 1) % width (:164) and composes the scrolling picture from the two
// sub-images [offset, width) and [0, offset) (:181-209).  The engine's pixel kernels produce columns -- [frames][R][4] bytes, one
// column contiguous, already in image-row order (row 0 = top) -- so the ring is a transposing scatter:
//
//     image[row][(offset + i) % width] = column i, row `row`;   offset += n (mod width);   later columns win when n > width
//
// through a 32 x 32 tile of pixels in LDS (columns are read along their rows, the image is written along its rows).
#include <cstdio>
#include <new>

#include "sgx_internal.hpp"

struct sgx_image {
    sgx_ctx *ctx = nullptr;
    uint32_t width = 0, height = 0;   // height = the context's rows R (one column = one frame)
    uint32_t offset = 0;              // simple_spectrogram.rs:164
    uchar4 *d_pixels = nullptr;       // [height][width] RGBA, rowstride 4 * width (gdk-pixbuf's layout for this size)
};

namespace sgx {

// columns [first, first + n) of `cols` ([..][height] pixels) go to x = (x0 + i) % width
__global__ void __launch_bounds__(256) image_scatter_kernel(const uchar4 *cols, uchar4 *image, uint32_t n, uint32_t width, uint32_t height, uint32_t x0)
{
    __shared__ uint32_t tile[32][33];
    const uint32_t c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8 threads
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t c = c0 + ty + 8 * k, r = r0 + tx;             // read: consecutive lanes, consecutive rows of one column
        if (c < n && r < height) tile[ty + 8 * k][tx] = reinterpret_cast<const uint32_t *>(cols)[(size_t)c * height + r];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t r = r0 + ty + 8 * k, c = c0 + tx;             // write: consecutive lanes, consecutive x of one image row
        if (c < n && r < height) reinterpret_cast<uint32_t *>(image)[(size_t)r * width + (x0 + c) % width] = tile[tx][ty + 8 * k];
    }
}

// out[row][x] = image[row][(x + offset) % width]: the picture the two append_scaled_texture calls compose (:181-209)
__global__ void __launch_bounds__(256) image_scrolled_kernel(const uchar4 *image, uchar4 *out, uint32_t width, uint32_t height, uint32_t offset)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= width) return;
    for (uint32_t row = blockIdx.y; row < height; row += gridDim.y)      // (65 536 rows are allowed, 65 535 blocks along y are)
        out[(size_t)row * width + x] = image[(size_t)row * width + (x + offset) % width];
}

void detach_images(sgx_ctx *c)
{
    for (sgx_image *im : c->images) im->ctx = nullptr;
    c->images.clear();
}
sgx_ctx *image_context(const sgx_image *im) { return im ? im->ctx : nullptr; }

}  // namespace sgx

namespace {

int image_fail(sgx_image *im, int rc, const char *msg)
{
    if (im && im->ctx) im->ctx->err = msg;
    return rc;
}
int image_fail_hip(sgx_image *im, hipError_t e, const char *what)
{
    char buf[512];
    std::snprintf(buf, sizeof(buf), "%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
    if (im && im->ctx) im->ctx->err = buf;
    return SGX_ERR_HIP;
}
#define IMAGE_HIP(im, call)                                             \
    do {                                                                \
        hipError_t e__ = (call);                                        \
        if (e__ != hipSuccess) return image_fail_hip((im), e__, #call); \
    } while (0)

}  // namespace

extern "C" {

int sgx_image_create(sgx_ctx *c, uint32_t width, sgx_image **out)
{
    if (out) *out = nullptr;
    if (!c || !out || width == 0) return SGX_ERR_INVALID_ARG;
    sgx_image *im = new (std::nothrow) sgx_image();
    if (!im) return SGX_ERR_NOMEM;
    im->ctx = c;
    im->width = width;
    im->height = c->R;
    const size_t bytes = (size_t)im->width * im->height * sizeof(uchar4);
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&im->d_pixels), bytes);
    if (e == hipSuccess) e = hipMemsetAsync(im->d_pixels, 0, bytes, c->stream);
    if (e != hipSuccess) {
        const int rc = image_fail_hip(im, e, "sgx_image_create");
        sgx_image_destroy(im);
        return rc;
    }
    c->images.push_back(im);   // sgx_destroy(ctx) detaches the images still alive, as it does the views
    *out = im;
    return SGX_OK;
}

void sgx_image_destroy(sgx_image *im)
{
    if (!im) return;
    if (im->ctx) {
        (void)hipSetDevice(im->ctx->device);
        (void)hipStreamSynchronize(im->ctx->stream);
        auto &v = im->ctx->images;
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i] == im) { v.erase(v.begin() + (long)i); break; }
    }
    if (im->d_pixels) (void)hipFree(im->d_pixels);
    delete im;
}

int sgx_image_write_columns(sgx_image *im, const uint8_t *d_rgba, size_t n_columns, uint32_t *offset_out)
{
    if (!im || !im->ctx) return SGX_ERR_INVALID_ARG;   // (no context: it was destroyed before this image)
    sgx_ctx *c = im->ctx;
    if (n_columns && !d_rgba) return image_fail(im, SGX_ERR_INVALID_ARG, "sgx_image_write_columns: null buffer");
    IMAGE_HIP(im, hipSetDevice(c->device));
    // more columns than the image is wide: the ring laps itself and only the last `width` survive (the reference writes them one by
    // one, :140-164, the later over the earlier)
    size_t first = n_columns > im->width ? n_columns - im->width : 0;
    const uint32_t x_first = (uint32_t)((im->offset + first) % im->width);
    const uint32_t n = (uint32_t)(n_columns - first);
    if (n) {
        const dim3 grid((n + 31) / 32, (im->height + 31) / 32);
        hipLaunchKernelGGL(sgx::image_scatter_kernel, grid, dim3(256), 0, c->stream,
                           reinterpret_cast<const uchar4 *>(d_rgba) + first * im->height, im->d_pixels, n, im->width, im->height, x_first);
        IMAGE_HIP(im, hipGetLastError());
    }
    im->offset = (uint32_t)((im->offset + n_columns) % im->width);
    if (offset_out) *offset_out = im->offset;
    return SGX_OK;
}

uint32_t sgx_image_offset(const sgx_image *im) { return im ? im->offset : 0; }
uint32_t sgx_image_width(const sgx_image *im) { return im ? im->width : 0; }
uint32_t sgx_image_height(const sgx_image *im) { return im ? im->height : 0; }

int sgx_image_read(sgx_image *im, int scrolled, uint8_t *d_out)
{
    if (!im || !im->ctx) return SGX_ERR_INVALID_ARG;
    sgx_ctx *c = im->ctx;
    if (!d_out) return image_fail(im, SGX_ERR_INVALID_ARG, "sgx_image_read: null buffer");
    IMAGE_HIP(im, hipSetDevice(c->device));
    if (!scrolled || im->offset == 0) {
        IMAGE_HIP(im, hipMemcpyAsync(d_out, im->d_pixels, (size_t)im->width * im->height * sizeof(uchar4), hipMemcpyDeviceToDevice, c->stream));
    } else {
        hipLaunchKernelGGL(sgx::image_scrolled_kernel, dim3((im->width + 255) / 256, im->height < 65535u ? im->height : 65535u), dim3(256), 0, c->stream,
                           im->d_pixels, reinterpret_cast<uchar4 *>(d_out), im->width, im->height, im->offset);
        IMAGE_HIP(im, hipGetLastError());
    }
    return SGX_OK;
}

const uint8_t *sgx_image_pixels(const sgx_image *im) { return im ? reinterpret_cast<const uint8_t *>(im->d_pixels) : nullptr; }

}  // extern "C"
