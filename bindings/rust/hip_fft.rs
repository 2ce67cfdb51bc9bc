// src/fourier/hip_fft.rs -- `impl AudioTransform`, replaces FastFourierTransform (fft.rs:11-99) one for one
use std::ffi::CStr;

use crate::fourier::audio_transform::AudioTransform;
use crate::fourier::sgx_sys::*;
use crate::fourier::{Frequency, Period, StereoMagnitude};

pub struct HipFastFourierTransform { pub(crate) ctx: *mut SgxCtx, sample_rate: Frequency, period: Period }

impl HipFastFourierTransform {
    /// same signature as FastFourierTransform::new (fft.rs:18)
    pub fn new(sample_rate: Frequency, period: Period) -> Self { Self::with_stride(sample_rate, period, 0.0) }

    /// ... plus the hop of the stream wrapper, for the batched entry points (audio_transform.rs:35)
    pub fn with_stride(sample_rate: Frequency, period: Period, stride: Period) -> Self {
        let mut cfg = unsafe { std::mem::zeroed::<SgxConfig>() };
        unsafe { sgx_config_init(&mut cfg) };
        cfg.sample_rate = sample_rate; cfg.period = period; cfg.stride = stride;
        cfg.window_samples = 0;            // W = (period * sample_rate) as usize, computed by the library (fft.rs:19)
        cfg.hop_samples = if stride > 0.0 { 0 } else { 1 };   // H = (stride * sample_rate) as usize, ditto; unused by process()
        cfg.channels = 2;                  // process() receives (l, r) pairs
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { sgx_create(&cfg, &mut ctx) };
        assert!(rc == 0, "{}", unsafe { CStr::from_ptr(sgx_last_error(std::ptr::null())) }.to_string_lossy());
        Self { ctx, sample_rate, period }
    }
    pub fn num_output_frequencies(&self) -> usize { self.num_input_samples() - 1 }   // fft.rs:33
    pub fn period(&self) -> Period { self.period }
}

impl AudioTransform for HipFastFourierTransform {
    type Output = Vec<StereoMagnitude>;
    fn sample_rate(&self) -> Frequency { self.sample_rate }
    fn num_input_samples(&self) -> usize { (self.period * self.sample_rate) as usize }   // fft.rs:41

    fn process<'a>(&mut self, samples: impl IntoIterator<Item = &'a StereoMagnitude>) -> Option<Self::Output> {
        let w = self.num_input_samples();
        let lr: Vec<f32> = samples.into_iter().take(w).flat_map(|(l, r)| [*l, *r]).collect();   // fft.rs:48-57
        let mut out = vec![(0f32, 0f32); w - 1];
        match unsafe { sgx_process_one(self.ctx, lr.as_ptr(), lr.len() / 2, out.as_mut_ptr() as *mut f32) } {
            1 => Some(out),
            0 => None,                       // fewer than W samples (fft.rs:72)
            _ => panic!("sgx_process_one: {}", unsafe { CStr::from_ptr(sgx_last_error(self.ctx)) }.to_string_lossy()),
        }
    }
}
impl Drop for HipFastFourierTransform { fn drop(&mut self) { unsafe { sgx_destroy(self.ctx) } } }
