//! Seam in `vid_dup_finder_app/src/video_hash_filesystem_cache/video_hash_filesystem_cache.rs:76-139`
//! (`validate_or_create_metadata_file`) and `cache_metadata.rs:45-168`: the sidecar `<stem>.metadata.txt` through the library, so that a
//! reader of an app-written cache and the app agree byte for byte on its name, its text and the first-mismatch messages
//! (`csrc/cache_metadata.cpp`; tests/test_cache_ingest.py holds the cases).  Uncompiled here (no Rust toolchain in the engine repository).
//! Only the three pure functions move; file I/O, `create_dir_all`, the `process::exit(1)` on "cache exists but metadata is absent" stay
//! where they are in the app.
use std::ffi::OsString;
use std::os::unix::ffi::{OsStrExt, OsStringExt};
use std::path::{Path, PathBuf};

use vdf_sys::*;

fn text_of(buf: &[u8]) -> String {
    String::from_utf8_lossy(&buf[..buf.iter().position(|&b| b == 0).unwrap_or(buf.len())]).into_owned()
}

/// `cache_path.file_stem()` + `with_file_name("{stem}.metadata.txt")` (:93-104); None where the original raises EINVAL (no file name).
pub fn metadata_path(cache_path: &Path) -> Option<PathBuf> {
    let raw = cache_path.as_os_str().as_bytes();
    let mut buf = vec![0u8; raw.len() + 32];
    let mut n = 0usize;
    let rc = unsafe { vdf_cache_metadata_path(raw.as_ptr() as *const _, raw.len(), buf.as_mut_ptr() as *mut _, buf.len(), &mut n) };
    if rc != VDF_OK { return None; }
    buf.truncate(n);
    Some(PathBuf::from(OsString::from_vec(buf)))
}

/// `VdfCacheMetadata::new(cropdetect, skip_forward_amount).to_disk_fmt()` (cache_metadata.rs:54-89): what `create_metadata_file` writes.
/// crop: VDF_CROPDETECT_NONE / _LETTERBOX / _MOTION (definitions.rs:46-54).
pub fn new_disk_text(crop: i32, skip_forward_amount: f64) -> Option<String> {
    let mut m = vdf_cache_metadata::default();
    if unsafe { vdf_cache_metadata_new(crop, skip_forward_amount, &mut m) } != VDF_OK { return None; }
    let mut buf = vec![0u8; 512];
    let mut n = 0usize;
    if unsafe { vdf_cache_metadata_format(&m, buf.as_mut_ptr() as *mut _, buf.len(), &mut n) } != VDF_OK { return None; }
    buf.truncate(n);
    String::from_utf8(buf).ok()
}

/// `VdfCacheMetadata::try_parse(&content)?.validate(cropdetect, skip_forward_amount)` (:125-136): Err carries the app's own message
/// ("Could not parse crop. Got letterbox", "skip_forward_amount mismatch: Act: 15.0, Exp: 0.5", ...), ready for
/// `VdfCacheError::MetadataValidationError`.
pub fn parse_and_validate(content: &str, crop: i32, skip_forward_amount: f64) -> Result<(), String> {
    let mut m = vdf_cache_metadata::default();
    let mut err = vec![0u8; 1024];
    let rc = unsafe { vdf_cache_metadata_parse(content.as_ptr() as *const _, content.len(), &mut m, err.as_mut_ptr() as *mut _, err.len()) };
    if rc != VDF_OK { return Err(text_of(&err)); }
    let rc = unsafe { vdf_cache_metadata_validate(&m, crop, skip_forward_amount, err.as_mut_ptr() as *mut _, err.len()) };
    if rc != VDF_OK { return Err(text_of(&err)); }
    Ok(())
}
