//! Seam 1: `VideoHash::from_frames` (vid_dup_finder_lib/src/video_hashing/video_hash.rs:45-73).
//! Same signature, same two NotEnoughFrames paths (:53 empty iterator, dct_3d.rs:47-52 fewer than 16 frames); the
//! per-frame crop_resize_buf (:59), Dct3d::from_images (:61) and the bit-pack loop (:64-68) run on the GPU.
//! Paste as the body of from_frames inside `impl VideoHash` (it uses the private fields).
use std::path::PathBuf;

use image::GrayImage;
use vdf_sys::*;

use super::gpu::{ctx, last_error};
use crate::definitions::{DCT_SIZE, HASH_WORDS};
use crate::Error::{self, NotEnoughFrames};

impl super::VideoHash {
    pub(crate) fn from_frames(
        frames: impl Clone + IntoIterator<Item = GrayImage>,
        src_path: PathBuf,
        duration: u32,
    ) -> Result<Self, Error> {
        let frames: Vec<GrayImage> = frames.into_iter().take(DCT_SIZE as usize).collect();
        let first = frames.first().ok_or(NotEnoughFrames)?; // :53
        if frames.len() < DCT_SIZE as usize {
            return Err(NotEnoughFrames); // dct_3d.rs:47-52
        }
        let (w, h) = first.dimensions();
        // dct_3d.rs:31-38 asserts equal frame sizes after the resize; the builder checks it before (video_hash_builder.rs:169-186)
        if frames.iter().any(|f| f.dimensions() != (w, h)) {
            return Err(Error::VidProc("frames differ in size".into()));
        }
        let Some(ctx) = ctx() else {
            return Self::from_frames_cpu(frames, src_path, duration); // the crate's original body, kept under this name
        };
        let mut packed = Vec::with_capacity(frames.len() * (w * h) as usize);
        for f in &frames {
            packed.extend_from_slice(f.as_raw());
        }
        let mut hash = [0usize; HASH_WORDS as usize]; // [usize; 16] == [u64; 16] on 64-bit targets
        let rc = unsafe {
            vdf_hash_frames_u8(
                ctx, packed.as_ptr(), 1, DCT_SIZE, w, h, (w * h) as usize, (DCT_SIZE * w * h) as usize,
                hash.as_mut_ptr() as *mut u64, std::ptr::null_mut(),
            )
        };
        match rc {
            VDF_OK => Ok(Self { hash, src_path, duration }),
            VDF_E_NOT_ENOUGH_FRAMES => Err(NotEnoughFrames),
            _ => Err(Error::VidProc(last_error(ctx))),
        }
    }
}
