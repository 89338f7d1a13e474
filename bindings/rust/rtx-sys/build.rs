// Links librtx.so.  RTX_LIB_DIR names the directory that holds it (the
// raytracing_amd/ directory of the backend's checkout after `make`).
fn main() {
    println!("cargo:rerun-if-env-changed=RTX_LIB_DIR");
    if let Ok(dir) = std::env::var("RTX_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=rtx");
}
