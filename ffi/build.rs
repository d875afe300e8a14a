// Links libplonk_hip.so (built in-tree by `python -m distributed_plonk_amd.build`).  PLONK_HIP_LIB_DIR overrides the location.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("PLONK_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("distributed_plonk_amd").join("lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=plonk_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=PLONK_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=plonk_hip.rs");
}
