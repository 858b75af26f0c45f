// Link against libzkcp_amd.so, built in-tree by `python contangle-zkcp_amd/build.py` (hipcc --offload-arch=gfx950).
fn main() {
    if let Ok(dir) = std::env::var("ZKCP_AMD_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=zkcp_amd");
    // DeviceBuf (src/lib.rs) calls hipMalloc / hipMemcpy / hipFree directly
    println!("cargo:rustc-link-search=native={}", std::env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into()) + "/lib");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rerun-if-env-changed=ZKCP_AMD_LIB_DIR");
}
