// Link against libsirius_amd.so: SIRIUS_AMD_LIB_DIR = the directory holding it (sirius_amd/csrc of this repository after
// `python -c "import __graft_entry__ as g; g.build()"`).
fn main() {
    let dir = std::env::var("SIRIUS_AMD_LIB_DIR").expect("set SIRIUS_AMD_LIB_DIR to the directory of libsirius_amd.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=sirius_amd");
    println!("cargo:rerun-if-env-changed=SIRIUS_AMD_LIB_DIR");
}
