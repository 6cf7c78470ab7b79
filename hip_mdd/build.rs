// Links libddo_hip.so (built by `make -C ddo_amd/csrc` into ddo_amd/_build; DDO_HIP_LIB_DIR overrides the directory) and
// records its directory as an rpath so that `cargo test` finds it at run time.
use std::path::PathBuf;

fn main() {
    let dir = std::env::var("DDO_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("ddo_amd").join("_build")
    });
    let dir = dir.canonicalize().unwrap_or(dir);
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=ddo_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=DDO_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../include/ddo_hip.h");
}
