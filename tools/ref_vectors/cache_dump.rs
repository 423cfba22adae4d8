//! Hash-cache dump for the MI355X-native engine (vid_dup_finder_lib_amd).  NOT part of the app and not compiled by that
//! repository (its image has no Rust toolchain): drop this file into
//! `vid_dup_finder_app/src/video_hash_filesystem_cache/generic_filesystem_cache/processing_fs_cache/ref_cache_dump.rs`,
//! add `#[cfg(test)] mod ref_cache_dump;` at the end of `processing_fs_cache.rs`, and run
//!   VDF_VECTORS_DIR=<engine repo>/tools/ref_vectors cargo test -p vid_dup_finder ref_cache_dump -- --nocapture
//! It reads inputs/cache_inputs.bin (export_inputs.py) and writes outputs/ref_cache.bin with the APP'S OWN cache writer:
//! `BaseFsCache::insert` + `BaseFsCache::save` (base_fs_cache.rs:56-165: bincode 2 `config::standard()` of
//! `HashMap<PathBuf, MtimeCacheEntry<Result<VideoHash, Error>>>`, tmp file + rename), i.e. exactly the bytes a user's cache
//! file holds.  `vdf_cache_decode` (csrc/cache_format.cpp) must read that file back into the entries of cache_cases.py
//! (tests/test_cache_format.py::test_app_written_cache_decodes).
//! Being a child module of `processing_fs_cache` it may name the private `MtimeCacheEntry` and the `BaseFsCache` imported there.
//! `VideoHash` has no public constructor from words; it derives `Deserialize` (video_hash.rs:26), so the entries are built
//! through serde_json (a dependency of the app) from {"hash": [16 words], "src_path": .., "duration": ..}.

use std::{
    env,
    fs::{self, File},
    io::{BufReader, Read},
    path::PathBuf,
    time::{Duration, UNIX_EPOCH},
};

use vid_dup_finder_lib::{Error, VideoHash};

use super::{BaseFsCache, MtimeCacheEntry};

fn rd_u32(r: &mut impl Read) -> u32 {
    let mut b = [0u8; 4];
    r.read_exact(&mut b).expect("short input file");
    u32::from_le_bytes(b)
}

fn rd_u64(r: &mut impl Read) -> u64 {
    let mut b = [0u8; 8];
    r.read_exact(&mut b).expect("short input file");
    u64::from_le_bytes(b)
}

fn rd_string(r: &mut impl Read) -> String {
    let n = rd_u32(r) as usize;
    let mut b = vec![0u8; n];
    r.read_exact(&mut b).expect("short input file");
    String::from_utf8(b).expect("UTF-8")
}

#[test]
fn ref_cache_dump() {
    let dir = PathBuf::from(env::var("VDF_VECTORS_DIR").expect("set VDF_VECTORS_DIR to <engine repo>/tools/ref_vectors"));
    fs::create_dir_all(dir.join("outputs")).unwrap();
    let out_path = dir.join("outputs/ref_cache.bin");
    let _ = fs::remove_file(&out_path);
    let mut r = BufReader::new(File::open(dir.join("inputs/cache_inputs.bin")).expect("run export_inputs.py first"));
    let cache: BaseFsCache<MtimeCacheEntry<Result<VideoHash, Error>>> = BaseFsCache::new(u32::MAX, out_path.clone()).unwrap();
    let n = rd_u32(&mut r);
    for _ in 0..n {
        let kind = rd_u32(&mut r);
        let words: Vec<u64> = (0..16).map(|_| rd_u64(&mut r)).collect();
        let duration = rd_u32(&mut r);
        let secs = rd_u64(&mut r);
        let nanos = rd_u32(&mut r);
        let path = rd_string(&mut r);
        let msg = rd_string(&mut r);
        let value: Result<VideoHash, Error> = match kind {
            0 => Ok(serde_json::from_value(serde_json::json!({"hash": words, "src_path": path, "duration": duration}))
                .expect("VideoHash from its serde shape")),
            1 => Err(Error::NotVideo),
            2 => Err(Error::VidProc(msg)),
            3 => Err(Error::NotEnoughFrames),
            k => panic!("unknown entry kind {k}"),
        };
        let entry = MtimeCacheEntry { cache_mtime: UNIX_EPOCH + Duration::new(secs, nanos), value };
        cache.insert(PathBuf::from(path), entry).unwrap();
    }
    cache.save().unwrap();
    println!("wrote {} ({} entries)", out_path.display(), n);
}
