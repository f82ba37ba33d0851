// Links librenderbaby_hip.so (built by `make` in the renderbaby-hip repository).
// Set RENDERBABY_HIP_LIB_DIR to the directory that holds it.
fn main() {
    if let Ok(dir) = std::env::var("RENDERBABY_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=renderbaby_hip");
    println!("cargo:rerun-if-env-changed=RENDERBABY_HIP_LIB_DIR");
}
