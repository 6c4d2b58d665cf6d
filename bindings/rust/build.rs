// Links libresampler_amd.so (set RESAMPLER_AMD_LIB_DIR to the directory holding it).
fn main() {
    if let Ok(dir) = std::env::var("RESAMPLER_AMD_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
    }
    println!("cargo:rustc-link-lib=dylib=resampler_amd");
}
