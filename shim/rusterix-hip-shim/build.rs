fn main() {
    let dir = std::env::var("RXR_LIB_DIR").unwrap_or_else(|_| "../../rusterix_amd/csrc".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rxr_hip");
    println!("cargo:rerun-if-env-changed=RXR_LIB_DIR");
}
