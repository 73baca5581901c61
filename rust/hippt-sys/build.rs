// Builds libhippt.so with hipcc for gfx950 (the same recipe as pathtracer_amd/csrc/Makefile) and links it.
// UNVERIFIED IN THIS ENVIRONMENT (no Rust toolchain in the build image).
use std::{env, path::PathBuf, process::Command};

fn main() {
    let root = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../..");
    let csrc = root.join("pathtracer_amd/csrc");
    let status = Command::new("make")
        .arg("-C").arg(&csrc)
        .arg("ARCH=gfx950")
        .status()
        .expect("failed to run make (needs hipcc from ROCm >= 7.0)");
    assert!(status.success(), "building libhippt.so failed");
    let lib = root.join("pathtracer_amd/lib");
    println!("cargo:rustc-link-search=native={}", lib.display());
    println!("cargo:rustc-link-lib=dylib=hippt");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib.display());
    for f in ["pt_kernels.hip", "pt_device.h", "pt_trace.h", "hippt_api.hip", "hippt_multi.hip", "host_scene.cpp"] {
        println!("cargo:rerun-if-changed={}", csrc.join(f).display());
    }
    println!("cargo:rerun-if-changed={}", root.join("include/hippt.h").display());
}
