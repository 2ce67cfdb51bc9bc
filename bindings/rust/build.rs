// Source of the binding shown in INTEGRATION.md (not compiled in this image: no Rust toolchain).
// build.rs
fn main() {
    println!("cargo:rustc-link-search=native={}", std::env::var("SGX_LIB_DIR").unwrap());
    println!("cargo:rustc-link-lib=dylib=sgx");           // libsgx.so (links libamdhip64.so.7)
}
