// src/fourier/sgx_sys.rs -- FFI declarations, a mirror of include/sgx.h (the parts the reference's path needs)
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct SgxConfig {
    pub struct_size: u32,
    pub sample_rate: f32, pub period: f32, pub stride: f32,
    pub window_samples: u32, pub hop_samples: u32, pub channels: u32, pub rows: u32,
    pub f_min: f64, pub f_max: f64,
    pub min_db: f32, pub max_db: f32,
    pub interp: u32, pub lut_index_mode: u32,
    pub device: i32, pub flags: u32,
}
#[repr(C)] pub struct SgxCtx { _private: [u8; 0] }

extern "C" {
    pub fn sgx_config_init(cfg: *mut SgxConfig) -> c_int;
    pub fn sgx_create(cfg: *const SgxConfig, out: *mut *mut SgxCtx) -> c_int;
    pub fn sgx_destroy(ctx: *mut SgxCtx);
    pub fn sgx_last_error(ctx: *const SgxCtx) -> *const c_char;
    pub fn sgx_num_frames(ctx: *const SgxCtx, n_samples: usize) -> usize;
    pub fn sgx_process_one(ctx: *mut SgxCtx, h_lr: *const f32, n_avail: usize, h_out: *mut f32) -> c_int;
    pub fn sgx_stft_batch(ctx: *mut SgxCtx, d_pcm: *const f32, n_samples: usize, first_frame: usize,
                          max_frames: usize, d_mags: *mut f32, n_out: *mut usize) -> c_int;
    pub fn sgx_stft_batch_f16(ctx: *mut SgxCtx, d_pcm: *const f32, n_samples: usize, first_frame: usize,
                              max_frames: usize, d_mags_f16: *mut c_void, n_out: *mut usize) -> c_int;   // F16F16 ring rows
    pub fn sgx_render_batch(ctx: *mut SgxCtx, d_pcm: *const f32, n_samples: usize, first_frame: usize,
                            max_frames: usize, d_rgba: *mut u8, n_out: *mut usize) -> c_int;
    pub fn sgx_magnitude_in(ctx: *mut SgxCtx, d_mags: *const f32, n_columns: usize, h_ranges: *const f32,
                            n_ranges: u32, d_out: *mut f32) -> c_int;              // FrequencySample::magnitude_in
    pub fn sgx_set_gradient(ctx: *mut SgxCtx, h_rgb: *const u8, n: u32, stereo: c_int) -> c_int;
    pub fn sgx_set_gradient_fn(ctx: *mut SgxCtx, eval: extern "C" fn(f64, *mut u8, *mut c_void), user: *mut c_void,
                               stereo: c_int) -> c_int;
    pub fn sgx_set_builtin_scheme(ctx: *mut SgxCtx, name: *const c_char, stereo: c_int) -> c_int;
    pub fn sgx_lookup_table(ctx: *mut SgxCtx, resolution: u32, h_out: *mut f32) -> c_int;
    pub fn sgx_sync(ctx: *mut SgxCtx) -> c_int;
}
extern "C" {   // from libamdhip64, for staging buffers
    pub fn hipMalloc(p: *mut *mut c_void, bytes: usize) -> c_int;
    pub fn hipFree(p: *mut c_void) -> c_int;
    pub fn hipMemcpy(dst: *mut c_void, src: *const c_void, bytes: usize, kind: c_int) -> c_int;
}
pub const HIP_MEMCPY_HOST_TO_DEVICE: c_int = 1;
pub const HIP_MEMCPY_DEVICE_TO_HOST: c_int = 2;
// SgxConfig::flags (include/sgx.h).  A mono device (audio_input_list_model.rs:67-69): `channels = 1` and no flag -- every frame is its
// own transform, the reference's dataflow (as a real-input transform at W 2048, the application's 2400 / 2205 and every other window but W 8192 and the smallest).
pub const SGX_FLAG_PAIRED_FRAMES: u32 = 1024;   // opt-in: two frames per transform (half the work; tolerance against the pair's peak)
pub const SGX_FLAG_COMPLEX_MONO: u32 = 512;     // A/B: the literal (s, s) transform per frame where a real-input kernel would run
