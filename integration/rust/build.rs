// build.rs — links the nrays crate against libnrays_hip.so (the C ABI of include/nrays_abi.h).
//
// NRAYS_HIP_LIB_DIR names the directory that holds libnrays_hip.so (default: the in-tree build of this repository,
// ../../nrays_amd/lib relative to the crate root once integration/rust/ has been copied into the nrays checkout's
// sibling).  The library's own RUNPATH finds /opt/rocm/lib (libamdhip64.so.7, librccl.so.1).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("NRAYS_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../nrays_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=nrays_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=NRAYS_HIP_LIB_DIR");
}
