// Link against libx3hip.so built by `python x3-rust_amd/build.py` (set X3HIP_LIB_DIR to its directory).
fn main() {
    let dir = std::env::var("X3HIP_LIB_DIR").unwrap_or_else(|_| "../lib".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=x3hip");
    println!("cargo:rerun-if-env-changed=X3HIP_LIB_DIR");
}
