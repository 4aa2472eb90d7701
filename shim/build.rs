// build.rs of the portrayer crate with the `hip` feature: links libportrayer_hip.so (include/portrayer_hip.h).
// PORTRAYER_HIP_DIR = the directory that holds libportrayer_hip.so (this repository's portrayer_amd/).
fn main() {
    println!("cargo:rerun-if-env-changed=PORTRAYER_HIP_DIR");
    if std::env::var("CARGO_FEATURE_HIP").is_err() {
        return;
    }
    let dir = std::env::var("PORTRAYER_HIP_DIR")
        .expect("the `hip` feature needs PORTRAYER_HIP_DIR = the directory of libportrayer_hip.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=portrayer_hip");
    // the examples are run from the crate root: find the library without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
}
