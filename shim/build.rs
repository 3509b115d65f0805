// Links libakp.so (built in-tree by `make -C crypto_primitives_amd/csrc`).  AKP_LIB_DIR overrides the location.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("AKP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../crypto_primitives_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=akp");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=AKP_LIB_DIR");
    println!("cargo:rerun-if-changed=../include/akp.h");
}
