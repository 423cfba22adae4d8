//! Resize-table dump for the MI355X-native engine (vid_dup_finder_lib_amd).  NOT part of the crate and not compiled by that
//! repository (its image has no Rust toolchain): drop this file into `vid_dup_finder_common/src/ref_resize_tables.rs`, add
//! `#[cfg(test)] mod ref_resize_tables;` at the end of `vid_dup_finder_common/src/lib.rs`, and run
//!   VDF_VECTORS_DIR=<engine repo>/tools/ref_vectors cargo test -p vid_dup_finder_common ref_resize_tables -- --nocapture
//! (vid_dup_finder_common is the package that depends on fast_image_resize directly: Cargo.toml:18.)
//!
//! fast_image_resize keeps its coefficient tables private, so they are read through its public API, one axis size n at a time
//! (inputs/resize_probe_inputs.bin, see export_inputs.py), in a form that lets the engine's tests say WHICH stage disagrees:
//!   * the f64 Lanczos3 weights and their bounds: an n x 1 image of PixelType::F32 holding 1.0 at column x, resized to 16 x 1
//!     with the crate's default algorithm (Convolution(Lanczos3), what resize_gray.rs:34-47 uses), returns weight[o][x] as f32 -
//!     zero exactly outside output o's bounds;
//!   * the i16 quantisation (precision rule, rounding of each weight, rounding and clamping of the sums) of the U8 path, through the
//!     product's own call `crop_resize_buf` (resize_gray.rs:11-54): constant images of every grey level, step edges at every
//!     position, and the random rows of the inputs file - each as a horizontal probe (n x 16 image of sixteen equal rows -> 16 x 16:
//!     no vertical resize) and as a vertical one (16 x n image of equal columns).
//! Output outputs/ref_resize_tables.bin (format: import_outputs.py).

use std::{
    env,
    fs::{self, File},
    io::{BufReader, BufWriter, Read, Write},
    num::NonZeroU32,
    path::PathBuf,
};

use fast_image_resize as fr;
use image::GrayImage;

use crate::{crop_resize_buf, Crop};

fn rd_u32(r: &mut impl Read) -> u32 {
    let mut b = [0u8; 4];
    r.read_exact(&mut b).expect("short input file");
    u32::from_le_bytes(b)
}

/// 16 output values of the horizontal probe `row` (n pixels): the first row of the 16 x 16 result (all rows are equal).
fn probe_h(row: &[u8]) -> Vec<u8> {
    let n = row.len() as u32;
    let mut buf = Vec::with_capacity(row.len() * 16);
    for _ in 0..16 {
        buf.extend_from_slice(row);
    }
    let img = GrayImage::from_raw(n, 16, buf).unwrap();
    let d = NonZeroU32::new(16).unwrap();
    let out = crop_resize_buf(&img, d, d, Crop::from_edge_offsets((n, 16), 0, 0, 0, 0));
    out.as_raw()[..16].to_vec()
}

/// The same values as a column of a 16 x n image: the first column of the 16 x 16 result.
fn probe_v(col: &[u8]) -> Vec<u8> {
    let n = col.len() as u32;
    let mut buf = Vec::with_capacity(col.len() * 16);
    for v in col {
        buf.extend(std::iter::repeat(*v).take(16));
    }
    let img = GrayImage::from_raw(16, n, buf).unwrap();
    let d = NonZeroU32::new(16).unwrap();
    let out = crop_resize_buf(&img, d, d, Crop::from_edge_offsets((16, n), 0, 0, 0, 0));
    (0..16).map(|y| out.as_raw()[y * 16]).collect()
}

#[test]
fn ref_resize_tables() {
    let dir = PathBuf::from(env::var("VDF_VECTORS_DIR").expect("set VDF_VECTORS_DIR to <engine repo>/tools/ref_vectors"));
    fs::create_dir_all(dir.join("outputs")).unwrap();
    let mut r = BufReader::new(File::open(dir.join("inputs/resize_probe_inputs.bin")).expect("run export_inputs.py first"));
    let mut w = BufWriter::new(File::create(dir.join("outputs/ref_resize_tables.bin")).unwrap());
    let n_sizes = rd_u32(&mut r);
    w.write_all(&n_sizes.to_le_bytes()).unwrap();
    for _ in 0..n_sizes {
        let n = rd_u32(&mut r);
        let n_rand = rd_u32(&mut r);
        let mut rand_rows = vec![0u8; (n * n_rand) as usize];
        r.read_exact(&mut rand_rows).unwrap();
        w.write_all(&n.to_le_bytes()).unwrap();
        w.write_all(&n_rand.to_le_bytes()).unwrap();

        // f32 impulse responses: weight[o][x] for every source column x
        let mut resizer = fr::Resizer::new();
        for x in 0..n {
            let mut src = fr::images::Image::new(n, 1, fr::PixelType::F32);
            src.buffer_mut()[(x * 4) as usize..(x * 4 + 4) as usize].copy_from_slice(&1.0f32.to_le_bytes());
            let mut dst = fr::images::Image::new(16, 1, fr::PixelType::F32);
            resizer.resize(&src, &mut dst, Some(&fr::ResizeOptions::new())).unwrap();
            w.write_all(dst.buffer()).unwrap(); // 16 f32, native = little endian on the targets this runs on
        }
        // U8 path: constants, step edges (255 left of k, 0 from k on), random rows; horizontal then vertical
        for c in 0..=255u8 {
            w.write_all(&probe_h(&vec![c; n as usize])).unwrap();
        }
        for c in 0..=255u8 {
            w.write_all(&probe_v(&vec![c; n as usize])).unwrap();
        }
        for k in 0..=n {
            let row: Vec<u8> = (0..n).map(|x| if x < k { 255 } else { 0 }).collect();
            w.write_all(&probe_h(&row)).unwrap();
        }
        for k in 0..=n {
            let col: Vec<u8> = (0..n).map(|x| if x < k { 255 } else { 0 }).collect();
            w.write_all(&probe_v(&col)).unwrap();
        }
        for i in 0..n_rand {
            w.write_all(&probe_h(&rand_rows[(i * n) as usize..((i + 1) * n) as usize])).unwrap();
        }
        for i in 0..n_rand {
            w.write_all(&probe_v(&rand_rows[(i * n) as usize..((i + 1) * n) as usize])).unwrap();
        }
    }
    println!("wrote {}/outputs/ref_resize_tables.bin", dir.display());
}
