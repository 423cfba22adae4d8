//! Seam in `vid_dup_finder_lib/src/video_hashing/video_hash_builder.rs:188-223` (`gen_hash` with `Cropdetect::Letterbox`): detection
//! (frames 0 and 8, `vid_dup_finder_common/src/video_frames_gray.rs:38-128,201-210`), the crop (`crop.rs:53-68,92-103`) and
//! `VideoHash::from_frames` in one call on the uncropped frames; the crop box is read in place on the device.  Uncompiled here.
use std::path::PathBuf;

use image::GrayImage;
use vdf_sys::*;

use super::gpu::{ctx, last_error};
use crate::definitions::{DCT_SIZE, HASH_WORDS};
use crate::Error::{self, NotEnoughFrames};

impl super::VideoHash {
    /// The hash of the UNCROPPED frames' letterbox crop, and the crop as (left, right, top, bottom) pixels removed (crop.rs:3-10).
    /// Paste inside `impl VideoHash` (it uses the private fields); `gen_hash` calls it instead of detect_crop + cropped + from_frames.
    pub(crate) fn from_frames_letterbox(frames: &[GrayImage], src_path: PathBuf, duration: u32) -> Result<(Self, [u32; 4]), Error> {
        let first = frames.first().ok_or(NotEnoughFrames)?;
        if frames.len() < DCT_SIZE as usize {
            return Err(NotEnoughFrames);
        }
        let (w, h) = first.dimensions();
        let ctx = ctx().ok_or_else(|| Error::VidProc("no usable GPU".into()))?; // the caller then takes the crate's own three steps
        let mut packed = Vec::with_capacity(DCT_SIZE as usize * (w * h) as usize);
        for f in &frames[..DCT_SIZE as usize] {
            packed.extend_from_slice(f.as_raw());
        }
        let mut hash = [0usize; HASH_WORDS as usize];
        let mut crop = [0u32; 4];
        let rc = unsafe {
            vdf_hash_frames_u8_letterbox(
                ctx, packed.as_ptr(), 1, DCT_SIZE as u32, w, h, (w * h) as usize, (DCT_SIZE as usize) * (w * h) as usize,
                hash.as_mut_ptr() as *mut u64, crop.as_mut_ptr(), std::ptr::null_mut(),
            )
        };
        match rc {
            VDF_OK => Ok((Self { hash, src_path, duration }, crop)),
            VDF_E_NOT_ENOUGH_FRAMES => Err(NotEnoughFrames),
            _ => Err(Error::VidProc(last_error(ctx))),
        }
    }
}
