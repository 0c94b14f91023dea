// Links libzkm_hip.so (built by `python -m ziren_amd.build`, i.e. hipcc --offload-arch=gfx950) and tells the shim where the
// ahead-of-time compiled per-chip quotient kernels live (ziren_amd/_jit/manifest.json, kept by ziren_amd/codegen.py, which __graft_entry__.build() runs over every recorded chip).
fn main() {
    let dir = std::env::var("ZKM_HIP_LIB_DIR").unwrap_or_else(|_| "../../ziren_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=zkm_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rustc-env=ZKM_HIP_KERNEL_DIR={dir}/_jit");
    println!("cargo:rerun-if-env-changed=ZKM_HIP_LIB_DIR");
}
