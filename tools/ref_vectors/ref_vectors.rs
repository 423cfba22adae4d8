//! Reference-vector dump for the MI355X-native engine (vid_dup_finder_lib_amd).  NOT part of the crate and not compiled by
//! that repository (its image has no Rust toolchain): drop this file into
//! `vid_dup_finder_lib/src/video_hashing/video_hash/ref_vectors.rs`, add `#[cfg(test)] mod ref_vectors;` at the end of
//! `video_hash.rs`, and run
//!   VDF_VECTORS_DIR=<engine repo>/tools/ref_vectors cargo test -p vid_dup_finder_lib \
//!       --features test-util,app_only_fns ref_vectors -- --nocapture
//! It feeds the engine's committed fixture inputs (inputs/*.bin, see export_inputs.py) through the crate's own code:
//!   * `VideoHash::from_frames` (video_hash.rs:45-73)            -> the 16 hash words of every clip,
//!   * `crop_resize_buf` exactly as from_frames calls it (:57-59) -> the sixteen 16x16 frames (fast_image_resize output),
//!   * `raw_dct_ops::dct_3d` on the cube `Dct3d::from_images` builds (dct_3d.rs:21-51) -> the 1000 consumed coefficients,
//!   * `search` / `search_with_references` (video_dup_finder.rs:7-46) -> MatchGroups as index lists,
//! and writes outputs/ref_hash_out.bin and outputs/ref_search_out.bin (formats: import_outputs.py).
//! Being a child module of `video_hash` it may use the `pub(crate)` constructor and the private fields.

use std::{
    env,
    fs::{self, File},
    io::{BufReader, BufWriter, Read, Write},
    num::NonZeroU32,
    path::{Path, PathBuf},
};

use bitvec::prelude::*;
use image::GrayImage;
use ndarray::Array3;
use vid_dup_finder_common::{crop_resize_buf, Crop};

use super::VideoHash;
use crate::definitions::{DCT_SIZE, HASH_SIZE, HASH_WORDS};
use crate::video_hashing::raw_dct_ops::dct_3d;
use crate::{search, search_with_references, MatchGroup};

fn rd_u32(r: &mut impl Read) -> u32 {
    let mut b = [0u8; 4];
    r.read_exact(&mut b).expect("short input file");
    u32::from_le_bytes(b)
}

fn rd_u64s(r: &mut impl Read, n: usize) -> Vec<u64> {
    let mut raw = vec![0u8; n * 8];
    r.read_exact(&mut raw).expect("short input file");
    raw.chunks_exact(8).map(|c| u64::from_le_bytes(c.try_into().unwrap())).collect()
}

fn rd_u32s(r: &mut impl Read, n: usize) -> Vec<u32> {
    let mut raw = vec![0u8; n * 4];
    r.read_exact(&mut raw).expect("short input file");
    raw.chunks_exact(4).map(|c| u32::from_le_bytes(c.try_into().unwrap())).collect()
}

fn dump_hashes(dir: &Path) {
    let mut r = BufReader::new(File::open(dir.join("inputs/hash_inputs.bin")).expect("run export_inputs.py first"));
    let mut w = BufWriter::new(File::create(dir.join("outputs/ref_hash_out.bin")).unwrap());
    let n_cases = rd_u32(&mut r);
    w.write_all(&n_cases.to_le_bytes()).unwrap();
    let dct_size = NonZeroU32::try_from(DCT_SIZE).unwrap();
    for _ in 0..n_cases {
        let name_len = rd_u32(&mut r) as usize;
        let mut name = vec![0u8; name_len];
        r.read_exact(&mut name).unwrap();
        let (n_clips, n_frames, h, wd) = (rd_u32(&mut r), rd_u32(&mut r), rd_u32(&mut r), rd_u32(&mut r));
        w.write_all(&(name_len as u32).to_le_bytes()).unwrap();
        w.write_all(&name).unwrap();
        w.write_all(&n_clips.to_le_bytes()).unwrap();
        for _ in 0..n_clips {
            let frames: Vec<GrayImage> = (0..n_frames)
                .map(|_| {
                    let mut buf = vec![0u8; (h * wd) as usize];
                    r.read_exact(&mut buf).unwrap();
                    GrayImage::from_raw(wd, h, buf).unwrap()
                })
                .collect();

            // the product: VideoHash::from_frames
            let vh = VideoHash::from_frames(frames.clone(), PathBuf::from("clip"), 0).expect("clips hold >= 16 frames");
            for word in vh.hash.iter() {
                w.write_all(&(*word as u64).to_le_bytes()).unwrap();
            }

            // its two stages, separately: resize exactly as from_frames does it (whole-image crop) ...
            let no_crop = Crop::from_edge_offsets((wd, h), 0, 0, 0, 0);
            let small: Vec<GrayImage> = frames
                .iter()
                .take(DCT_SIZE as usize)
                .map(|f| crop_resize_buf(f, dct_size, dct_size, no_crop))
                .collect();
            for img in &small {
                assert_eq!((img.width(), img.height()), (DCT_SIZE, DCT_SIZE));
                w.write_all(img.as_raw()).unwrap();
            }
            // ... and the cube of Dct3d::from_images (m[frame][col][row] = pix - 128) through dct_3d
            let n = DCT_SIZE as usize;
            let mut cube = Array3::<f64>::zeros((n, n, n));
            for (t, img) in small.iter().enumerate() {
                for (col, row, pix) in img.enumerate_pixels() {
                    cube[[t, col as usize, row as usize]] = f64::from(pix.0[0]) - 128.0;
                }
            }
            let d = dct_3d(&cube);
            let hs = HASH_SIZE as usize;
            for kt in 0..hs {
                for kx in 0..hs {
                    for ky in 0..hs {
                        w.write_all(&d[[kt, kx, ky]].to_le_bytes()).unwrap();
                    }
                }
            }
        }
    }
}

fn index_of(p: &Path) -> u32 {
    p.to_str().unwrap()[1..].parse().unwrap()
}

fn write_groups(w: &mut impl Write, groups: &[MatchGroup]) {
    w.write_all(&(groups.len() as u32).to_le_bytes()).unwrap();
    for g in groups {
        let r: i64 = g.reference().map_or(-1, |p| i64::from(index_of(p)));
        w.write_all(&r.to_le_bytes()).unwrap();
        let members: Vec<u32> = g.duplicates().map(index_of).collect();
        w.write_all(&(members.len() as u32).to_le_bytes()).unwrap();
        for m in members {
            w.write_all(&m.to_le_bytes()).unwrap();
        }
    }
}

fn dump_search(dir: &Path) {
    assert_eq!(usize::BITS, 64, "hash words are [usize; 16]: run on a 64-bit target");
    let mut r = BufReader::new(File::open(dir.join("inputs/search_inputs.bin")).expect("run export_inputs.py first"));
    let mut w = BufWriter::new(File::create(dir.join("outputs/ref_search_out.bin")).unwrap());
    let mut read_set = |prefix: &str| -> Vec<VideoHash> {
        let n = rd_u32(&mut r) as usize;
        let words = rd_u64s(&mut r, n * HASH_WORDS as usize);
        let durs = rd_u32s(&mut r, n);
        (0..n)
            .map(|i| {
                let mut arr = [0usize; HASH_WORDS as usize];
                for (k, a) in arr.iter_mut().enumerate() {
                    *a = words[i * HASH_WORDS as usize + k] as usize;
                }
                // zero-padded index as the path: Search::sort's (duration, path) key then orders equal durations by
                // index, which is the order the fixture arrays are already in
                VideoHash::from_components(PathBuf::from(format!("{prefix}{i:08}")), BitArray::new(arr), durs[i])
            })
            .collect()
    };
    let cands = read_set("c");
    let refs = read_set("r");
    write_groups(&mut w, &search(cands.clone(), 0.35));
    write_groups(&mut w, &search(cands.clone(), 0.1));
    write_groups(&mut w, &search_with_references(refs, cands, 0.35));
}

#[test]
fn ref_vectors() {
    let dir = PathBuf::from(env::var("VDF_VECTORS_DIR").expect("set VDF_VECTORS_DIR to <engine repo>/tools/ref_vectors"));
    fs::create_dir_all(dir.join("outputs")).unwrap();
    dump_hashes(&dir);
    dump_search(&dir);
    println!("wrote {}/outputs/ref_hash_out.bin and ref_search_out.bin", dir.display());
}
