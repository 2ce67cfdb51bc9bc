// src/fourier/stream_batched.rs -- the hop loop of AudioStreamTransform::process (audio_transform.rs:34-42) as ONE
// launch for every frame the ring currently holds, instead of one transform call per frame
use std::ffi::CStr;
use std::os::raw::c_void;

use ringbuf::traits::{Consumer, Observer};

use crate::fourier::audio_transform::{AudioStreamTransform, AudioTransform};
use crate::fourier::hip_fft::HipFastFourierTransform;
use crate::fourier::sgx_sys::*;
use crate::fourier::{Period, StereoMagnitude};

/// Device staging for one tick: interleaved (l, r) in, [frames][W-1][2] out; grown on demand, kept.
/// The context is rebuilt when `stride` changes: it is a public, mutable field of the reference's wrapper.
pub struct BatchState { hop_ctx: Option<(Period, HipFastFourierTransform)>, d_pcm: *mut c_void, pcm_bytes: usize, d_mags: *mut c_void, mags_bytes: usize }

impl Default for BatchState {
    fn default() -> Self { Self { hop_ctx: None, d_pcm: std::ptr::null_mut(), pcm_bytes: 0, d_mags: std::ptr::null_mut(), mags_bytes: 0 } }
}
impl Drop for BatchState {
    fn drop(&mut self) { unsafe { if !self.d_pcm.is_null() { hipFree(self.d_pcm); } if !self.d_mags.is_null() { hipFree(self.d_mags); } } }
}

fn grow(p: &mut *mut c_void, have: &mut usize, need: usize) {
    if need <= *have { return; }
    unsafe {
        if !p.is_null() { hipFree(*p); }
        assert_eq!(hipMalloc(p, need), 0, "hipMalloc({need})");
    }
    *have = need;
}

impl AudioStreamTransform<HipFastFourierTransform> {
    /// What `process()` yields this tick, computed in one launch.  Same frames (frame t = pairs [t H, t H + W) of the
    /// ring), same skip -- (frames + 1) * H, because the reference also skips on its terminating short read (:37-41).
    pub fn process_batched(&mut self, st: &mut BatchState) -> Vec<Vec<StereoMagnitude>> {
        let sr = self.transform.sample_rate();
        let h = (self.stride * sr) as usize;                                       // audio_transform.rs:35
        let w = self.transform.num_input_samples();
        let m = w - 1;
        let n = self.input_stream.occupied_len();
        let frames = if n < w || h == 0 { 0 } else { (n - w) / h + 1 };
        let mut out = Vec::with_capacity(frames);
        if frames > 0 {
            if st.hop_ctx.as_ref().map(|(s, _)| *s != self.stride).unwrap_or(true) {
                st.hop_ctx = Some((self.stride, HipFastFourierTransform::with_stride(sr, self.transform.period(), self.stride)));
            }
            let ctx = st.hop_ctx.as_ref().unwrap().1.ctx;
            let pairs = (frames - 1) * h + w;
            let lr: Vec<f32> = self.input_stream.iter().take(pairs).flat_map(|(l, r)| [*l, *r]).collect();   // non-consuming peek
            let (pcm_bytes, mags_bytes) = (lr.len() * 4, frames * m * 2 * 4);
            grow(&mut st.d_pcm, &mut st.pcm_bytes, pcm_bytes);
            grow(&mut st.d_mags, &mut st.mags_bytes, mags_bytes);
            let mut flat = vec![(0f32, 0f32); frames * m];
            let mut got = 0usize;
            unsafe {
                assert_eq!(hipMemcpy(st.d_pcm, lr.as_ptr() as *const c_void, pcm_bytes, HIP_MEMCPY_HOST_TO_DEVICE), 0);
                let rc = sgx_stft_batch(ctx, st.d_pcm as *const f32, pairs, 0, frames, st.d_mags as *mut f32, &mut got);
                assert!(rc == 0 && got == frames, "sgx_stft_batch: {}", CStr::from_ptr(sgx_last_error(ctx)).to_string_lossy());
                assert_eq!(sgx_sync(ctx), 0);
                assert_eq!(hipMemcpy(flat.as_mut_ptr() as *mut c_void, st.d_mags, mags_bytes, HIP_MEMCPY_DEVICE_TO_HOST), 0);
            }
            out.extend(flat.chunks(m).map(|c| c.to_vec()));      // `frames` vectors of W - 1 (left, right) magnitudes
        }
        self.input_stream.skip((frames + 1) * h);
        out
    }
}
