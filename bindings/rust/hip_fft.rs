// Source of the binding shown in INTEGRATION.md (not compiled in this image: no Rust toolchain).
// src/fourier/hip_fft.rs
pub struct HipFastFourierTransform { ctx: *mut SgxCtx, sample_rate: Frequency, period: Period }

impl HipFastFourierTransform {
    /// same signature as FastFourierTransform::new (fft.rs:18)
    pub fn new(sample_rate: Frequency, period: Period) -> Self {
        let mut cfg = unsafe { std::mem::zeroed::<SgxConfig>() };
        unsafe { sgx_config_init(&mut cfg) };
        cfg.sample_rate = sample_rate; cfg.period = period;
        cfg.window_samples = 0;            // W = (period * sample_rate) as usize, computed by the library (fft.rs:19)
        cfg.channels = 2;                  // process() receives (l, r) pairs
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { sgx_create(&cfg, &mut ctx) };
        assert!(rc == 0, "{}", unsafe { std::ffi::CStr::from_ptr(sgx_last_error(std::ptr::null())) }.to_string_lossy());
        Self { ctx, sample_rate, period }
    }
    pub fn num_output_frequencies(&self) -> usize { self.num_input_samples() - 1 }   // fft.rs:33
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
            _ => panic!("sgx_process_one failed"),
        }
    }
}
impl Drop for HipFastFourierTransform { fn drop(&mut self) { unsafe { sgx_destroy(self.ctx) } } }
