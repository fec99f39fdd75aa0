// Mirrors vma/build.rs: the native library is built by `make -C renderer_amd/csrc` (hipcc), this
// script only tells cargo where it is. MIP_LIB_DIR overrides the default location.
fn main() {
    let dir = std::env::var("MIP_LIB_DIR").unwrap_or_else(|_| "../../../renderer_amd/lib".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=mi_instance_pipeline");
    println!("cargo:rerun-if-changed=../../../include/mi_instance_pipeline.h");
}
