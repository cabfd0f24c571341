// The reference's bitpolymul-sys compiles its C sources with `cc` (crates/bitpolymul-sys/build.rs:3-19). libflute_sot is
// HIP code built by `make -C flute_amd/csrc` (hipcc --offload-arch=gfx950), so this build script only links it.
fn main() {
    let dir = std::env::var("FLUTE_SOT_LIB_DIR")
        .expect("set FLUTE_SOT_LIB_DIR to the directory that holds libflute_sot.so (flute_amd/ in the engine's tree)");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=flute_sot");
    println!("cargo:rerun-if-env-changed=FLUTE_SOT_LIB_DIR");
}
