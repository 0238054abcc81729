// UNVERIFIED (never compiled here).  Links the C-ABI library built by `make -C fft_wgpu_amd/csrc`.
// FFT_WGPU_AMD_LIB_DIR = directory holding libfft_wgpu_amd.so (default: ../fft_wgpu_amd relative to this crate).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("FFT_WGPU_AMD_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("fft_wgpu_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=fft_wgpu_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=FFT_WGPU_AMD_LIB_DIR");
}
