// src/colorscheme_hip.rs -- ColorScheme::new_mono / new_stereo (colorscheme.rs:24-39) for ANY colorous gradient:
// colorous' own eval_continuous is handed over as a callback; the engine locates the byte switch points of the colour
// function once (on the host) and renders with thresholds, so no gradient table of the library is involved
use std::os::raw::c_void;

use crate::fourier::sgx_sys::{sgx_set_gradient_fn, SgxCtx};

extern "C" fn eval(t: f64, rgb: *mut u8, user: *mut c_void) {
    let g = unsafe { &*(user as *const colorous::Gradient) };
    let c = g.eval_continuous(t);
    unsafe { *rgb = c.r; *rgb.add(1) = c.g; *rgb.add(2) = c.b; }
}

/// e.g. `set_colorous_gradient(ctx, &colorous::RED_YELLOW_BLUE, true)` for the first entry of default_color_schemes
pub fn set_colorous_gradient(ctx: *mut SgxCtx, gradient: &'static colorous::Gradient, stereo: bool) {
    let rc = unsafe { sgx_set_gradient_fn(ctx, eval, gradient as *const _ as *mut c_void, stereo as i32) };
    assert_eq!(rc, 0);
}
