//! Seam for the app's parallel hashing (`vid_dup_finder_app/src/video_hash_filesystem_cache/video_hash_filesystem_cache.rs:237-257`:
//! `par_bridge().for_each(.. VideoHashBuilder::hash ..)` -> `video_hash_builder.rs:214-223` `gen_hash`): every rayon worker submits
//! its clip and blocks like `from_frames` does, while the clips of concurrent workers share one batched launch
//! (`csrc/hash_queue.cpp`; tests/test_gpu_hash_queue.py compares every clip with the oracle).  One queue per frame size.  Uncompiled here.
use std::collections::HashMap;
use std::sync::Mutex;

use vdf_sys::*;

struct Queue(*mut vdf_hash_queue);
unsafe impl Send for Queue {}
unsafe impl Sync for Queue {}
impl Drop for Queue {
    fn drop(&mut self) {
        unsafe { vdf_hash_queue_destroy(self.0) }
    }
}

/// Queues by (w, h, letterbox).  `submit` may be called from any number of threads.
pub struct HashQueues {
    ctx: *mut vdf_ctx,
    queues: Mutex<HashMap<(u32, u32, bool), std::sync::Arc<Queue>>>,
}
unsafe impl Send for HashQueues {}
unsafe impl Sync for HashQueues {}

impl HashQueues {
    pub fn new(ctx: *mut vdf_ctx) -> Self { Self { ctx, queues: Mutex::new(HashMap::new()) } }

    /// frames: 16 packed w x h gray frames.  Returns (hash words, crop l / r / t / b) or the vdf status.
    pub fn submit(&self, frames: &[u8], w: u32, h: u32, letterbox: bool) -> Result<([u64; 16], [u32; 4]), i32> {
        assert_eq!(frames.len(), 16 * (w * h) as usize);
        let q = {
            let mut map = self.queues.lock().unwrap();
            if let Some(q) = map.get(&(w, h, letterbox)) { q.clone() } else {
                let mut p: *mut vdf_hash_queue = std::ptr::null_mut();
                // up to 256 clips per launch; the first caller of a batch waits at most 2 ms for company
                let rc = unsafe { vdf_hash_queue_create(self.ctx, w, h, 256, 2000, letterbox as i32, &mut p) };
                if rc != VDF_OK { return Err(rc); }
                let q = std::sync::Arc::new(Queue(p));
                map.insert((w, h, letterbox), q.clone());
                q
            }
        };
        let (mut hash, mut crop) = ([0u64; 16], [0u32; 4]);
        let rc = unsafe { vdf_hash_queue_submit(q.0, frames.as_ptr(), hash.as_mut_ptr(), crop.as_mut_ptr()) };
        if rc == VDF_OK { Ok((hash, crop)) } else { Err(rc) }
    }
}
