// Links libfawkes_hip.so (built by `make -C fawkes-crypto_amd/csrc`).  FAWKES_HIP_LIB_DIR = directory that holds it.
fn main() {
    let dir = std::env::var("FAWKES_HIP_LIB_DIR").unwrap_or_else(|_| "../fawkes-crypto_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=fawkes_hip");
    println!("cargo:rerun-if-env-changed=FAWKES_HIP_LIB_DIR");
}
