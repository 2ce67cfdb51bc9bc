// build.rs -- link the reference against libsgx.so (which itself links libamdhip64.so)
fn main() {
    println!("cargo:rustc-link-search=native={}", std::env::var("SGX_LIB_DIR").unwrap());
    println!("cargo:rustc-link-lib=dylib=sgx");
    println!("cargo:rustc-link-lib=dylib=amdhip64");      // hipMalloc / hipMemcpy for the staging buffers of stream_batched.rs
}
