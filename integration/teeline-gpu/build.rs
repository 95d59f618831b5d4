// Links libteeline_gpu.so.  TEELINE_GPU_LIB_DIR = directory holding the library (the build tree's teeline_amd/, or an
// install prefix's lib/); an rpath is emitted so the binary finds it at run time without LD_LIBRARY_PATH.
use std::env;

fn main() {
    println!("cargo:rerun-if-env-changed=TEELINE_GPU_LIB_DIR");
    if let Ok(dir) = env::var("TEELINE_GPU_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=teeline_gpu");
}
