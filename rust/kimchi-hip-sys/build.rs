// Locates libkimchi_hip.so.  KIMCHI_HIP_LIB_DIR points at the directory that holds it (the repository's
// `proof_systems_amd/` after `python -c "import __graft_entry__ as g; g.build()"`); the HIP runtime comes from ROCM_PATH.
use std::env;

fn main() {
    let dir = env::var("KIMCHI_HIP_LIB_DIR").unwrap_or_else(|_| format!("{}/../../proof_systems_amd", env!("CARGO_MANIFEST_DIR")));
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-lib=dylib=kimchi_hip");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{rocm}/lib");
    println!("cargo:rerun-if-env-changed=KIMCHI_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
}
