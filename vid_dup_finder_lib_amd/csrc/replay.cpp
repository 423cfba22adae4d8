// Host-side pieces of the search path that are inherently sequential or trivial:
//   - the greedy, order-dependent consumption of Search::search_self
//     (vid_dup_finder_lib/src/video_hashing/search_algorithm.rs:131-170) replayed over the
//     thresholded adjacency the GPU produced;
//   - MatchGroup assembly for search_with_references (video_dup_finder.rs:25-45);
//   - scalar helpers (hamming_distance, tolerance conversion, window pair counts).
// No GPU call in this file.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/vdf.h"

namespace {

uint32_t sat_u32(double x)
{  // Rust `f64 as u32`
    if (!(x == x)) return 0u;
    if (x <= 0.0) return 0u;
    if (x >= 4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)x;
}

// Append groups (given as member lists) to a vdf_groups, growing its arrays.
int groups_append(vdf_groups *g, const std::vector<uint64_t> &offs /* relative, starts at 0 */,
                  const std::vector<uint64_t> &mem, const std::vector<int64_t> &refs)
{
    const uint64_t add_groups = offs.size() - 1;
    const uint64_t old_groups = g->n_groups;
    const uint64_t old_members = (g->offsets && old_groups) ? g->offsets[old_groups] : 0;
    const uint64_t new_groups = old_groups + add_groups;
    uint64_t *no = (uint64_t *)std::realloc(g->offsets, (new_groups + 1) * sizeof(uint64_t));
    if (!no) return VDF_E_OOM;
    g->offsets = no;
    if (old_groups == 0) g->offsets[0] = 0;
    uint64_t *nm = (uint64_t *)std::realloc(g->members, std::max<uint64_t>(old_members + mem.size(), 1) * sizeof(uint64_t));
    if (!nm) return VDF_E_OOM;
    g->members = nm;
    int64_t *nr = (int64_t *)std::realloc(g->ref_index, std::max<uint64_t>(new_groups, 1) * sizeof(int64_t));
    if (!nr) return VDF_E_OOM;
    g->ref_index = nr;
    if (!mem.empty()) std::memcpy(g->members + old_members, mem.data(), mem.size() * sizeof(uint64_t));
    for (uint64_t i = 0; i < add_groups; i++) {
        g->offsets[old_groups + i + 1] = old_members + offs[i + 1];
        g->ref_index[old_groups + i] = refs[i];
    }
    g->n_groups = new_groups;
    return VDF_OK;
}

}  // namespace

extern "C" {

const char *vdf_version(void) { return "vid_dup_finder_lib_amd 0.1.0 (gfx950)"; }

uint32_t vdf_hamming_u1024(const uint64_t *a, const uint64_t *b)
{
    uint32_t acc = 0;
    for (int i = 0; i < VDF_HASH_WORDS; i++) acc += (uint32_t)__builtin_popcountll(a[i] ^ b[i]);
    return acc;
}

uint32_t vdf_tolerance_int(double tolerance) { return sat_u32(tolerance * 1000.0); }

uint64_t vdf_count_pairs_self(const uint32_t *dur, size_t n)
{
    uint64_t pairs = 0;
    size_t rhs = 0;
    for (size_t i = 0; i < n; i++) {
        const uint32_t thresh = sat_u32((double)dur[i] * 1.1);
        if (rhs < i + 1) rhs = i + 1;
        while (rhs < n && dur[rhs] <= thresh) rhs++;
        pairs += rhs - (i + 1);
    }
    return pairs;
}

uint64_t vdf_count_pairs_refs(const uint32_t *cand, size_t n_cand, const uint32_t *ref, size_t n_ref)
{
    uint64_t pairs = 0;
    for (size_t r = 0; r < n_ref; r++) {
        const uint32_t lo_d = sat_u32((double)ref[r] * 0.95), hi_d = sat_u32((double)ref[r] * 1.05);
        const uint32_t *lo = std::lower_bound(cand, cand + n_cand, lo_d);  // first !(d < lo_d)
        const uint32_t *hi = std::upper_bound(cand, cand + n_cand, hi_d);  // first !(d <= hi_d)
        if (hi > lo) pairs += (uint64_t)(hi - lo);
    }
    return pairs;
}

void vdf_groups_free(vdf_groups *g)
{
    if (!g) return;
    std::free(g->offsets);
    std::free(g->members);
    std::free(g->ref_index);
    std::memset(g, 0, sizeof *g);
}

int vdf_replay_self(size_t n, const vdf_hit *hits, uint64_t n_hits, uint32_t row_begin, uint32_t row_end,
                    uint8_t *matched_io, vdf_groups *out)
{
    if (!out || (n_hits && !hits)) return VDF_E_INVAL;
    std::vector<uint8_t> local;
    uint8_t *matched = matched_io;
    if (!matched) { local.assign(n, 0); matched = local.data(); }
    std::vector<uint64_t> offs(1, 0), mem;
    std::vector<int64_t> refs;
    uint64_t k = 0;
    while (k < n_hits) {
        const uint32_t i = hits[k].row;
        uint64_t e = k;
        while (e < n_hits && hits[e].row == i) e++;
        if (i >= row_begin && i < row_end && i < n && !matched[i]) {
            matched[i] = 1;  // target.matched = true, search_algorithm.rs:147
            const size_t first = mem.size();
            for (uint64_t q = k; q < e; q++) {
                const uint32_t j = hits[q].col;
                if (j >= n) return VDF_E_INVAL;
                if (!matched[j]) {  // :152-155
                    mem.push_back(j);
                    matched[j] = 1;
                }
            }
            if (mem.size() != first) {
                mem.push_back(i);  // target last, :159
                offs.push_back(mem.size());
                refs.push_back(-1);
            }
        }
        k = e;
    }
    return groups_append(out, offs, mem, refs);
}

int vdf_groups_finish_self(vdf_groups *g)
{  // ret.reverse(), search_algorithm.rs:167: reverse the group order, keep member order
    if (!g) return VDF_E_INVAL;
    if (g->n_groups == 0) {
        if (!g->offsets) {
            g->offsets = (uint64_t *)std::calloc(1, sizeof(uint64_t));
            if (!g->offsets) return VDF_E_OOM;
        }
        return VDF_OK;
    }
    const uint64_t ng = g->n_groups, nm = g->offsets[ng];
    std::vector<uint64_t> no(ng + 1), nmem(nm);
    uint64_t pos = 0;
    no[0] = 0;
    for (uint64_t i = 0; i < ng; i++) {
        const uint64_t src = ng - 1 - i;
        const uint64_t len = g->offsets[src + 1] - g->offsets[src];
        std::memcpy(nmem.data() + pos, g->members + g->offsets[src], len * sizeof(uint64_t));
        pos += len;
        no[i + 1] = pos;
    }
    std::memcpy(g->offsets, no.data(), (ng + 1) * sizeof(uint64_t));
    std::memcpy(g->members, nmem.data(), nm * sizeof(uint64_t));
    std::reverse(g->ref_index, g->ref_index + ng);
    return VDF_OK;
}

int vdf_groups_from_ref_hits(const vdf_hit *hits, uint64_t n_hits, vdf_groups *out)
{
    if (!out || (n_hits && !hits)) return VDF_E_INVAL;
    std::vector<uint64_t> offs(1, 0), mem;
    std::vector<int64_t> refs;
    mem.reserve(n_hits);
    uint64_t k = 0;
    while (k < n_hits) {
        const uint32_t r = hits[k].row;
        while (k < n_hits && hits[k].row == r) mem.push_back(hits[k++].col);
        offs.push_back(mem.size());
        refs.push_back((int64_t)r);
    }
    int rc = groups_append(out, offs, mem, refs);
    if (rc == VDF_OK && !out->offsets) {
        out->offsets = (uint64_t *)std::calloc(1, sizeof(uint64_t));
        if (!out->offsets) rc = VDF_E_OOM;
    }
    return rc;
}

}  // extern "C"
