// Links libaesw.so (built by `python -c "import __graft_entry__ as g; g.build()"` or the hipcc line in
// halo2-aes_amd/_build.py: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared ...) and the HIP runtime.
//   AESW_LIB_DIR  directory that holds libaesw.so      (default: ../../halo2-aes_amd, relative to this crate)
//   ROCM_PATH     ROCm installation                     (default: /opt/rocm)
use std::env;
use std::path::PathBuf;

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let lib_dir = env::var("AESW_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| manifest.join("..").join("..").join("halo2-aes_amd"));
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=aesw");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    // the library is found at run time next to where it was linked from
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rerun-if-env-changed=AESW_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    println!("cargo:rerun-if-changed=build.rs");
}
