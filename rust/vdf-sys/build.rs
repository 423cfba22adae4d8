// Links libvdf_hip.so.  VDF_LIB_DIR = directory holding the library (vid_dup_finder_lib_amd/ of the engine checkout).
fn main() {
    let dir = std::env::var("VDF_LIB_DIR").expect("set VDF_LIB_DIR to the directory that holds libvdf_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=vdf_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=VDF_LIB_DIR");
}
