// Plain dynamic linking against the in-tree build of the library (city-rollup_amd/libcityprover_hip.so).
// CITYPROVER_LIB_DIR names the directory that holds it; the worker then runs with that directory on
// LD_LIBRARY_PATH (or the binary gets an rpath through RUSTFLAGS="-C link-arg=-Wl,-rpath,<dir>").
fn main() {
    println!("cargo:rerun-if-env-changed=CITYPROVER_LIB_DIR");
    let dir = std::env::var("CITYPROVER_LIB_DIR")
        .expect("set CITYPROVER_LIB_DIR to the directory that holds libcityprover_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=cityprover_hip");
}
