//! Seams 2 and 3: `Search::search_self` (search_algorithm.rs:81-171) and the per-reference loop of
//! `search_with_references` (video_dup_finder.rs:25-45 -> search_one / duration_slice, search_algorithm.rs:63-77,173-185).
//!
//! `Search::sort` (:55-61) keeps its meaning and may keep its body; `sort_through_the_engine` below is the drop-in for large sets
//! (a `PathBuf` clone per comparison costs seconds at a million entries - beside a 0.1 s search).  Each replacement extracts SoA arrays in the sorted order,
//! makes ONE call, and maps the returned indices back to paths.  Group and member order come back exactly as the original
//! builds them (search(): hits ascending then the target, groups in descending target order; references: input order,
//! members ascending), so video_dup_finder.rs:7-13 and MatchGroup are untouched.
//!
//! The reference's search cannot fail (:89-91).  Neither can this: on ANY non-OK status the crate's original loop runs
//! (kept as `search_self_cpu` / `search_one`), after the library's message has been logged.
use std::path::PathBuf;
use std::ptr::null_mut;

use vdf_sys::*;

use super::gpu::{ctx, last_error};
use super::search_algorithm::Search;
use crate::VideoHash;

fn soa(entries: impl Iterator<Item = impl std::borrow::Borrow<VideoHash>>) -> (Vec<u64>, Vec<u32>) {
    let mut words = Vec::new();
    let mut durs = Vec::new();
    for e in entries {
        let h = e.borrow();
        words.extend(h.hash.iter().map(|w| *w as u64));
        durs.push(h.duration());
    }
    (words, durs)
}

/// CSR -> Vec of index lists; frees the library's payload.
unsafe fn take_groups(g: &mut vdf_groups) -> Vec<(i64, Vec<usize>)> {
    let out = (0..g.n_groups as usize)
        .map(|i| {
            let (a, b) = (*g.offsets.add(i) as usize, *g.offsets.add(i + 1) as usize);
            (*g.ref_index.add(i), (a..b).map(|k| *g.members.add(k) as usize).collect())
        })
        .collect();
    vdf_groups_free(g);
    out
}

impl Search {
    /// Optional replacement for the body of `sort` (:55-61): the same stable order by (duration, src_path) - `PathBuf` order, the path
    /// half on the device for plain paths - from ONE call over the entries' durations and path bytes (vdf_sort_order_paths), applied as
    /// a permutation.  Falls back to the crate's own `sort_by_key` on any non-OK status or without a context.
    #[cfg(unix)]
    pub fn sort_through_the_engine(&mut self) {
        use std::os::unix::ffi::OsStrExt;
        let n = self.entries.len();
        if n >= 2048 {
            if let Some(ctx) = ctx() {
                let durs: Vec<u32> = self.entries.iter().map(|e| e.value.duration()).collect();
                let mut offs: Vec<u64> = Vec::with_capacity(n + 1);
                let mut blob: Vec<u8> = Vec::new();
                offs.push(0);
                for e in &self.entries {
                    blob.extend_from_slice(e.value.src_path().as_os_str().as_bytes());
                    offs.push(blob.len() as u64);
                }
                let mut order = vec![0u32; n];
                let rc = unsafe {
                    vdf_sort_order_paths(ctx, durs.as_ptr(), offs.as_ptr(), blob.as_ptr() as *const _, n, order.as_mut_ptr(), null_mut())
                };
                if rc == VDF_OK {
                    let mut taken: Vec<Option<_>> = std::mem::take(&mut self.entries).into_iter().map(Some).collect();
                    self.entries = order.iter().map(|&k| taken[k as usize].take().expect("a permutation")).collect();
                    return;
                }
                log::warn!(target: "gpu_search", "vdf_sort_order_paths failed ({rc}): {}; sorting on the host", last_error(ctx));
            }
        }
        self.sort(); // the crate's ORIGINAL body
    }

    /// Replaces the body of search_self (:81-171).  `self.entries` are already sorted (seed() sorts, :31-34).
    pub fn search_self(&mut self, tolerance: f64) -> Vec<Vec<PathBuf>> {
        if self.entries.is_empty() {
            return vec![]; // :89-91
        }
        if let Some(ctx) = ctx() {
            let (words, durs) = soa(self.entries.iter().map(|e| &e.value));
            let mut g = vdf_groups { n_groups: 0, offsets: null_mut(), members: null_mut(), ref_index: null_mut() };
            let rc = unsafe { vdf_search_self(ctx, words.as_ptr(), durs.as_ptr(), durs.len(), vdf_tolerance_int(tolerance), &mut g) };
            if rc == VDF_OK {
                return unsafe { take_groups(&mut g) }
                    .into_iter()
                    .map(|(_, members)| members.into_iter().map(|k| self.entries[k].value.src_path().to_path_buf()).collect())
                    .collect();
            }
            log::warn!(target: "gpu_search", "vdf_search_self failed ({rc}): {}; using the CPU loop", last_error(ctx));
        }
        self.search_self_cpu(tolerance) // the crate's original loop, renamed
    }

    /// Replaces the `for ref_hash in refs` loop of search_with_references (video_dup_finder.rs:25-45): all references in one
    /// call.  Returns (reference index, duplicate paths in sorted-candidate order) for every reference with >= 1 match, in
    /// reference input order - what the loop feeds to MatchGroup::new_with_reference (:38-43).
    pub fn search_all_references(&mut self, refs: &[VideoHash], tolerance: f64) -> Vec<(usize, Vec<PathBuf>)> {
        if self.entries.is_empty() || refs.is_empty() {
            return vec![];
        }
        if let Some(ctx) = ctx() {
            let (cw, cd) = soa(self.entries.iter().map(|e| &e.value));
            let (rw, rd) = soa(refs.iter());
            let mut g = vdf_groups { n_groups: 0, offsets: null_mut(), members: null_mut(), ref_index: null_mut() };
            let rc = unsafe {
                vdf_search_refs(ctx, cw.as_ptr(), cd.as_ptr(), cd.len(), rw.as_ptr(), rd.as_ptr(), rd.len(), vdf_tolerance_int(tolerance), &mut g)
            };
            if rc == VDF_OK {
                return unsafe { take_groups(&mut g) }
                    .into_iter()
                    .map(|(r, members)| (r as usize, members.into_iter().map(|k| self.entries[k].value.src_path().to_path_buf()).collect()))
                    .collect();
            }
            log::warn!(target: "gpu_search", "vdf_search_refs failed ({rc}): {}; using the CPU loop", last_error(ctx));
        }
        // the crate's original per-reference path (search_with_references(&[&ref], tolerance, false), :28-29)
        refs.iter()
            .enumerate()
            .filter_map(|(i, r)| {
                let mut res = self.search_with_references(&[r], tolerance, false);
                let paths: Vec<PathBuf> = res.drain(..).flatten().collect();
                (!paths.is_empty()).then_some((i, paths))
            })
            .collect()
    }
}
