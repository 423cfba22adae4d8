//! Seam in `vid_dup_finder_app/src/app/app_fns.rs:428-482` (`search_disk`): from the loaded hash cache to MatchGroups in ONE library
//! call.  Uncompiled in the engine repository (no Rust toolchain there); the C side of every call below is exercised by
//! tests/test_cache_ingest.py through ctypes and measured by bench.py's `cache_ingest` leg (profiles/r05_cache_ingest.txt:
//! 10 M entries - decode 31-35 ms, PathBuf ranks 0.26-0.29 s, upload + device sort 34 ms, then the search itself).
//!
//! What the original does per search: `all_cached_paths()` -> two filename filters -> `cache.fetch(p)` (a clone of one `VideoHash`
//! per selected path) -> `search()` / `search_with_references()`, whose `Search::new` sorts by `(duration, src_path.to_owned())`
//! (search_algorithm.rs:31-34,55-61: one PathBuf clone per key evaluation).  Here the cache file's bytes are decoded straight into
//! SoA arrays (`vdf_cache_decode`: no per-entry object), the filters run over the path blob and yield INDEX lists, and
//! `vdf_search_cache_entries` does the rest: PathBuf ranks on all host threads, upload, Search::sort on the device, the search,
//! and the map back to entry indices.  Paths are materialised for the grouped members only.
use std::ffi::OsStr;
use std::os::unix::ffi::OsStrExt;
use std::path::{Path, PathBuf};
use std::ptr::null_mut;

use vdf_sys::*;

/// The decoded cache: owner of a `vdf_cache_soa` (freed on drop).  Entries that held `Err(..)` are counted in `n_err` and absent.
pub struct CacheSoa(pub vdf_cache_soa);

impl Drop for CacheSoa {
    fn drop(&mut self) {
        unsafe { vdf_cache_free(&mut self.0) }
    }
}

impl CacheSoa {
    /// base_fs_cache.rs:167-223 (`load_cache_from_disk`, bincode backend): the file's bytes -> arrays.  Malformed bytes are the
    /// original's `Deserialization` error.
    pub fn from_bytes(bytes: &[u8]) -> Result<Self, String> {
        let mut soa: vdf_cache_soa = unsafe { std::mem::zeroed() };
        let rc = unsafe { vdf_cache_decode(bytes.as_ptr(), bytes.len(), &mut soa) };
        if rc == VDF_OK { Ok(CacheSoa(soa)) } else { Err(format!("cache file does not deserialize (vdf status {rc})")) }
    }
    pub fn len(&self) -> usize { self.0.n_ok as usize }
    pub fn path(&self, i: usize) -> &Path {
        unsafe {
            let (a, b) = (*self.0.path_offsets.add(i) as usize, *self.0.path_offsets.add(i + 1) as usize);
            Path::new(OsStr::from_bytes(std::slice::from_raw_parts(self.0.paths.add(a) as *const u8, b - a)))
        }
    }
}

/// `search_disk`'s middle (app_fns.rs:443-482): `includes_cand` / `includes_ref` are the two filename filters
/// (`create_cands_filename_filter(cfg).includes`, `create_refs_filename_filter(cfg).includes`).  Returns what the original's
/// `matchset` holds: (reference path or None, duplicate paths) per group, groups and members in the crate's order.
/// Err = the library could not run (no GPU, out of memory): the caller falls back to the original body - `search*` itself cannot
/// fail in the crate (search_algorithm.rs:89-91), and it still cannot.
pub fn search_cache(
    ctx: *mut vdf_ctx, cache: &CacheSoa, tolerance: f64, includes_cand: impl Fn(&Path) -> bool, includes_ref: impl Fn(&Path) -> bool,
) -> Result<Vec<(Option<PathBuf>, Vec<PathBuf>)>, String> {
    let n = cache.len();
    let cand: Vec<u64> = (0..n).filter(|&i| includes_cand(cache.path(i))).map(|i| i as u64).collect();
    let refs: Vec<u64> = (0..n).filter(|&i| includes_ref(cache.path(i))).map(|i| i as u64).collect();
    if cand.is_empty() {
        return Ok(vec![]); // "No files were found at the paths given by --files" (app_fns.rs:465-467)
    }
    let mut g = vdf_groups { n_groups: 0, offsets: null_mut(), members: null_mut(), ref_index: null_mut() };
    let mut timing = vdf_cache_search_timing::default();
    let rc = unsafe {
        vdf_search_cache_entries(
            ctx, cache.0.hashes, cache.0.durations, cache.0.path_offsets, cache.0.paths, n,
            cand.as_ptr(), cand.len(), if refs.is_empty() { std::ptr::null() } else { refs.as_ptr() }, refs.len(),
            vdf_tolerance_int(tolerance), &mut g, &mut timing,
        )
    };
    if rc != VDF_OK {
        return Err(format!("vdf_search_cache_entries failed ({rc})"));
    }
    let out = (0..g.n_groups as usize)
        .map(|k| unsafe {
            let (a, b) = (*g.offsets.add(k) as usize, *g.offsets.add(k + 1) as usize);
            let dups = (a..b).map(|q| cache.path(*g.members.add(q) as usize).to_path_buf()).collect();
            let r = *g.ref_index.add(k);
            (if r >= 0 { Some(cache.path(r as usize).to_path_buf()) } else { None }, dups)
        })
        .collect();
    unsafe { vdf_groups_free(&mut g) };
    Ok(out)
}

/// `SearchOutput::sort`'s `Sorting::Distance` key (search_output.rs:43-60): per group the largest pairwise Hamming distance over
/// `contained_paths()` - one launch for all groups (`vdf_groups_max_distance`) instead of a `cache.fetch` per member and a
/// `tuple_combinations` loop per group.  `groups` = the CSR arrays of entry indices a search just returned.
pub fn distance_keys(ctx: *mut vdf_ctx, cache: &CacheSoa, groups: &vdf_groups) -> Result<Vec<u32>, String> {
    let mut keys = vec![0u32; groups.n_groups as usize];
    let rc = unsafe { vdf_groups_max_distance(ctx, cache.0.hashes, cache.len(), cache.0.hashes, cache.len(), groups, keys.as_mut_ptr()) };
    if rc == VDF_OK { Ok(keys) } else { Err(format!("vdf_groups_max_distance failed ({rc})")) }
}
