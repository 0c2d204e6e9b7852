// Looks for libsrack_hip.so in $SRACK_HIP_LIB_DIR (default: ../../s-rack_amd relative to this crate).
fn main() {
    let dir = std::env::var("SRACK_HIP_LIB_DIR").unwrap_or_else(|_| format!("{}/../../s-rack_amd", env!("CARGO_MANIFEST_DIR")));
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=srack_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=SRACK_HIP_LIB_DIR");
}
