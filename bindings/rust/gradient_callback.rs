// Source of the binding shown in INTEGRATION.md (not compiled in this image: no Rust toolchain).
extern "C" fn eval(t: f64, rgb: *mut u8, user: *mut c_void) {
    let g = unsafe { &*(user as *const colorous::Gradient) };
    let c = g.eval_continuous(t);
    unsafe { *rgb = c.r; *rgb.add(1) = c.g; *rgb.add(2) = c.b; }
}
// ColorScheme::new_stereo(RED_YELLOW_BLUE, ...):
unsafe { sgx_set_gradient_fn(ctx, eval, &colorous::RED_YELLOW_BLUE as *const _ as *mut c_void, 1) };
