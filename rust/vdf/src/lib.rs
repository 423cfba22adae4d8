//! Glue shared by the seams: the process-wide context and error text.  Drop next to video_hash.rs / search.rs
//! (e.g. as `vid_dup_finder_lib/src/video_hashing/gpu.rs`).  Uncompiled in the engine repository (no Rust toolchain there).
use std::ffi::CStr;
use std::os::raw::c_int;
use std::sync::OnceLock;

use vdf_sys::*;

struct Ctx(*mut vdf_ctx);
// The library serialises calls on a context with its own mutex (include/vdf.h, "Thread safety").
unsafe impl Send for Ctx {}
unsafe impl Sync for Ctx {}

static CTX: OnceLock<Option<Ctx>> = OnceLock::new();

/// The shared context, or None when no usable GPU exists (callers then take the crate's CPU path).
/// VDF_DEVICES="0,1,2,3" makes one context over several GPUs (vdf_ctx_create_multi); default: device 0.
pub(crate) fn ctx() -> Option<*mut vdf_ctx> {
    CTX.get_or_init(|| {
        let devices: Vec<c_int> = std::env::var("VDF_DEVICES")
            .ok()
            .map(|s| s.split(',').filter_map(|t| t.trim().parse().ok()).collect())
            .filter(|v: &Vec<c_int>| !v.is_empty())
            .unwrap_or_else(|| vec![0]);
        let mut p: *mut vdf_ctx = std::ptr::null_mut();
        let rc = unsafe { vdf_ctx_create_multi(devices.as_ptr(), devices.len() as c_int, &mut p) };
        if rc == VDF_OK { Some(Ctx(p)) } else { None }
    })
    .as_ref()
    .map(|c| c.0)
}

pub(crate) fn last_error(ctx: *const vdf_ctx) -> String {
    unsafe {
        let p = vdf_last_error(ctx);
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}
