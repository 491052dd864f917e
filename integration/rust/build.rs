// build.rs of the reference crate (new file): link libnvr.so.  NVR_LIB_DIR = the directory that holds it (nano-vllm-rs_amd/).
fn main() {
    let dir = std::env::var("NVR_LIB_DIR").expect("NVR_LIB_DIR: path to nano-vllm-rs_amd/ (libnvr.so)");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=nvr");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=NVR_LIB_DIR");
}
