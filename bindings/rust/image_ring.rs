// src/widgets/image_ring.rs -- SimpleSpectrogram's Pixbuf as a device-resident ring.
//
// Replaces `buffer: Pixbuf` + `offset: Cell<usize>` (simple_spectrogram.rs:60-66, 89-94), the put_pixel loop with its offset
// update (:140-164) and the two new_subpixbuf calls that compose the scrolling picture (:181-209).
use std::os::raw::c_int;

use crate::devices::live_ring::SgxLive;
use crate::fourier::sgx_sys::SgxCtx;

#[repr(C)] pub struct SgxImage { _private: [u8; 0] }

extern "C" {
    pub fn sgx_image_create(ctx: *mut SgxCtx, width: u32, out: *mut *mut SgxImage) -> c_int;
    pub fn sgx_image_destroy(image: *mut SgxImage);
    pub fn sgx_image_write_columns(image: *mut SgxImage, d_rgba: *const u8, n_columns: usize, offset_out: *mut u32) -> c_int;
    pub fn sgx_image_offset(image: *const SgxImage) -> u32;
    pub fn sgx_image_width(image: *const SgxImage) -> u32;
    pub fn sgx_image_height(image: *const SgxImage) -> u32;
    pub fn sgx_live_tick_image(live: *mut SgxLive, image: *mut SgxImage, max_frames: usize, n_frames: *mut usize) -> c_int;
    pub fn sgx_image_read(image: *mut SgxImage, scrolled: c_int, d_out: *mut u8) -> c_int;
    pub fn sgx_image_pixels(image: *const SgxImage) -> *const u8;
}

pub struct ImageRing { raw: *mut SgxImage, pub width: u32, pub height: u32 }

impl ImageRing {
    /// `Pixbuf::new(Colorspace::Rgb, true, 8, TEXTURE_WIDTH, TEXTURE_HEIGHT)` (:89-94); the height is the context's row count
    pub fn new(ctx: *mut SgxCtx, width: u32) -> Self {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { sgx_image_create(ctx, width, &mut raw) };
        assert_eq!(rc, 0);
        // `buffer.width()` / `buffer.height()` (:150, :186-187): read back, so that a caller sizes `scrolled_into`'s buffer from the image
        Self { raw, width: unsafe { sgx_image_width(raw) }, height: unsafe { sgx_image_height(raw) } }
    }

    /// the body of `for frequency_sample in self.fft.borrow_mut().process() { ... }` (:136-165) for one GUI tick:
    /// every complete frame of the capture ring becomes one pixel column at `offset`, device to device
    pub fn tick(&self, live: *mut SgxLive) -> usize {
        let mut got = 0usize;
        let rc = unsafe { sgx_live_tick_image(live, self.raw, self.width as usize, &mut got) };
        assert_eq!(rc, 0);
        got
    }

    /// `self.offset.get()` (:164, :181)
    pub fn offset(&self) -> usize { unsafe { sgx_image_offset(self.raw) as usize } }

    /// device pointer to [height][width][4] bytes, rowstride 4 * width: what a GL / Vulkan interop texture (or one
    /// hipMemcpyDtoH into the Pixbuf's own pixels) takes; `scrolled_into` gives the picture of :181-209 in one piece instead
    pub fn pixels(&self) -> *const u8 { unsafe { sgx_image_pixels(self.raw) } }
    /// `d_out`: a device buffer of `height * width * 4` bytes
    pub fn scrolled_into(&self, d_out: *mut u8) { assert_eq!(unsafe { sgx_image_read(self.raw, 1, d_out) }, 0); }
}

impl Drop for ImageRing {
    fn drop(&mut self) { unsafe { sgx_image_destroy(self.raw) } }
}
