// src/devices/live_ring.rs -- the capture ring with its consumed side on the GPU.
//
// Replaces the pair (HeapProd<(f32, f32)>, HeapCons<(f32, f32)>) made at audio_input_list_model.rs:30 and
// the hop loop that AudioStreamTransform::process runs over the consumer half (audio_transform.rs:34-42).
use std::os::raw::{c_int, c_longlong, c_void};
use std::sync::Arc;

use crate::fourier::sgx_sys::SgxCtx;
use crate::fourier::StereoMagnitude;

#[repr(C)] pub struct SgxLive { _private: [u8; 0] }
#[repr(C)] pub struct SgxView { _private: [u8; 0] }   // GPUSpectrogram's F16F16 ring texture on the device (sgx_view_create)
pub const SGX_LIVE_MAGS: c_int = 0;
pub const SGX_LIVE_MAGS_F16: c_int = 1;
pub const SGX_LIVE_RGBA: c_int = 2;
pub const SGX_LIVE_REFERENCE_SKIP: u32 = 1;

extern "C" {
    pub fn sgx_live_create(ctx: *mut SgxCtx, capacity_pairs: usize, flags: u32, out: *mut *mut SgxLive) -> c_int;
    pub fn sgx_live_destroy(live: *mut SgxLive);
    pub fn sgx_live_push(live: *mut SgxLive, h_samples: *const f32, n_values: usize, channels: u32) -> c_longlong;
    pub fn sgx_live_occupied(live: *const SgxLive) -> usize;
    pub fn sgx_live_tick(live: *mut SgxLive, what: c_int, h_out: *mut c_void, max_frames: usize, n_frames: *mut usize) -> c_int;
    pub fn sgx_live_tick_view(live: *mut SgxLive, view: *mut SgxView, max_frames: usize, n_frames: *mut usize) -> c_int;
    pub fn sgx_spectrum_levels(ctx: *mut SgxCtx, d_column: *const f32, n_bars: u32, h_levels: *mut f64) -> c_int;
}

pub struct LiveRing { raw: *mut SgxLive, m: usize, rows: usize }
unsafe impl Send for LiveRing {}   // one producer thread + one consumer thread, as HeapRb
unsafe impl Sync for LiveRing {}

impl LiveRing {
    pub fn new(ctx: *mut SgxCtx, m: usize, rows: usize) -> Arc<Self> {
        let mut raw = std::ptr::null_mut();
        // 4096 pairs and the terminating skip: what the reference does today
        let rc = unsafe { sgx_live_create(ctx, 4096, SGX_LIVE_REFERENCE_SKIP, &mut raw) };
        assert_eq!(rc, 0);
        Arc::new(Self { raw, m, rows })
    }

    /// the body of the cpal callback (audio_input_list_model.rs:63-75)
    pub fn push(&self, data: &[f32], channels: u16) {
        let rc = unsafe { sgx_live_push(self.raw, data.as_ptr(), data.len(), channels as u32) };
        if rc == -2 { eprintln!("{}-channel input not supported!", channels); }
    }

    /// the frames AudioStreamTransform::process would yield this tick
    pub fn frames(&self) -> Vec<Vec<StereoMagnitude>> {
        let max = 32;
        let mut flat = vec![(0f32, 0f32); max * self.m];
        let mut got = 0usize;
        let rc = unsafe { sgx_live_tick(self.raw, SGX_LIVE_MAGS, flat.as_mut_ptr() as *mut c_void, max, &mut got) };
        assert_eq!(rc, 0);
        flat.truncate(got * self.m);
        flat.chunks(self.m).map(|c| c.to_vec()).collect()
    }

    /// GPUSpectrogram::render's upload loop (gpu_spectrogram.rs:255-275): this tick's frames as half-pair rows straight into the
    /// device-resident ring texture; returns the block size the reference adds to `texture_offset`
    pub fn upload_into(&self, view: *mut SgxView) -> usize {
        let mut got = 0usize;
        let rc = unsafe { sgx_live_tick_view(self.raw, view, 2048, &mut got) };
        assert_eq!(rc, 0);
        got
    }

    /// the pixel columns SimpleSpectrogram::snapshot would put_pixel this tick: [frames][rows][4]
    pub fn columns(&self, out: &mut [u8]) -> usize {
        let mut got = 0usize;
        let rc = unsafe { sgx_live_tick(self.raw, SGX_LIVE_RGBA, out.as_mut_ptr() as *mut c_void, out.len() / (self.rows * 4), &mut got) };
        assert_eq!(rc, 0);
        got
    }
}

impl Drop for LiveRing {
    fn drop(&mut self) { unsafe { sgx_live_destroy(self.raw) } }
}
