// Source of the binding shown in INTEGRATION.md (not compiled in this image: no Rust toolchain).
impl AudioStreamTransform<HipFastFourierTransform> {
    pub fn process_batched(&mut self) -> Vec<Vec<StereoMagnitude>> {
        let h = (self.stride * self.transform.sample_rate()) as usize;            // audio_transform.rs:35
        let w = self.transform.num_input_samples();
        let n = self.input_stream.occupied_len();
        let frames = if n < w { 0 } else { (n - w) / h + 1 };
        let lr: Vec<f32> = self.input_stream.iter().take((frames.max(1) - 1) * h + w).flat_map(|(l, r)| [*l, *r]).collect();
        // hipMemcpy lr -> d_pcm; sgx_stft_batch(ctx, d_pcm, lr.len()/2, 0, frames, d_mags, &mut got); copy back
        self.input_stream.skip((frames + 1) * h);   // the reference also skips on its terminating read (:37-41)
        /* split d_mags into `frames` Vec<(f32, f32)> of W-1 entries */
        todo!()
    }
}
