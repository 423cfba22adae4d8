// vdf.hpp -- C++ host mirror of the reference crate's public surface for the hot path, over the C ABI
// (include/vdf.h).  Header-only; link with libvdf_hip.so.
//
// The reference is Rust and this image has no Rust toolchain, so this is the compiled-language host side above
// the boundary: same names, argument meaning and error behaviour as
//   vid_dup_finder_lib/src/lib.rs:132-140            VideoHash, MatchGroup, search, search_with_references, Error
//   vid_dup_finder_lib/src/video_hashing/video_hash.rs:26-32,45-73,176-192
//   vid_dup_finder_lib/src/video_hashing/video_dup_finder.rs:7-46
//   vid_dup_finder_lib/src/video_hashing/matches/match_group.rs:10-105
// What stays on the host (it needs paths): Search::sort (search_algorithm.rs:55-61) including Rust's
// component-wise PathBuf ordering; MatchGroup assembly from the index lists the library returns.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <numeric>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/vdf.h"

namespace vdf {

// ---- Error (video_hashing/mod.rs:17-28) -------------------------------------------------------------------
struct Error : std::runtime_error {
    enum Kind { NotVideo, VidProc, NotEnoughFrames, Device } kind;
    Error(Kind k, const std::string &m) : std::runtime_error(m), kind(k) {}
    static Error not_enough_frames() { return Error(NotEnoughFrames, "Could not extract enough frames"); }
    static Error vid_proc(const std::string &m) { return Error(VidProc, "Video processing error: " + m); }
};
struct TooFewEntries : std::runtime_error {  // match_group.rs:15-16
    TooFewEntries() : std::runtime_error("too few entries") {}
};

// ---- a context: one GPU, or ONE context over several GPUs of the node ---------------------------------------
class Context {
public:
    explicit Context(int device = 0)
    {
        vdf_ctx *c = nullptr;
        if (vdf_ctx_create(device, &c) != VDF_OK) throw Error(Error::Device, vdf_last_error(nullptr));
        ctx_.reset(c, vdf_ctx_destroy);
    }
    // search() / search_with_references() / hashing then use every listed GPU from the one call (vdf_ctx_create_multi)
    explicit Context(const std::vector<int> &devices)
    {
        vdf_ctx *c = nullptr;
        if (vdf_ctx_create_multi(devices.data(), (int)devices.size(), &c) != VDF_OK) throw Error(Error::Device, vdf_last_error(nullptr));
        ctx_.reset(c, vdf_ctx_destroy);
    }
    int device_count() const { return vdf_ctx_device_count(ctx_.get()); }
    vdf_ctx *get() const { return ctx_.get(); }
    // VDF_DEVICES="0,1,2,3": the default context spans those GPUs (one process, vdf_ctx_create_multi); default: device 0
    static Context &default_context()
    {
        static Context c = [] {
            std::vector<int> devs;
            if (const char *e = std::getenv("VDF_DEVICES")) {
                const char *p = e;
                while (*p) {
                    char *end = nullptr;
                    const long v = std::strtol(p, &end, 10);
                    if (end == p) break;
                    devs.push_back((int)v);
                    p = (*end == ',') ? end + 1 : end;
                }
            }
            return devs.empty() ? Context(0) : Context(devs);
        }();
        return c;
    }

private:
    std::shared_ptr<vdf_ctx> ctx_;
};

// ---- Rust `PathBuf: Ord`: std::path compares Components, not bytes ------------------------------------------
// RootDir < CurDir < ParentDir < Normal(bytes); repeated '/' and inner "." are not components.
struct PathKey {
    std::vector<std::pair<int, std::string>> comps;
    explicit PathKey(const std::string &p)
    {
        if (!p.empty() && p[0] == '/') comps.emplace_back(1, "");
        else if (p == "." || p.rfind("./", 0) == 0) comps.emplace_back(2, "");
        size_t i = 0;
        while (i <= p.size()) {
            size_t j = p.find('/', i);
            if (j == std::string::npos) j = p.size();
            const std::string part = p.substr(i, j - i);
            if (!part.empty() && part != ".") comps.emplace_back(part == ".." ? 3 : 4, part == ".." ? "" : part);
            i = j + 1;
        }
    }
    bool operator<(const PathKey &o) const { return comps < o.comps; }
    bool operator==(const PathKey &o) const { return comps == o.comps; }
};

// ---- Crop (vid_dup_finder_common/src/crop.rs:3-10) ---------------------------------------------------------------
// The crop box of a frame as edge offsets: what the letterbox detection yields and crop_resize_buf takes; one row {left, right, top,
// bottom} of the C ABI's out_crops.  Member order = the derive's comparison order.
struct Crop {
    std::pair<uint32_t, uint32_t> orig_res{0, 0};
    uint32_t left = 0, right = 0, top = 0, bottom = 0;

    // crop.rs:13-30 (the reference asserts; a box that leaves no pixel throws here)
    static Crop from_edge_offsets(std::pair<uint32_t, uint32_t> res, uint32_t l, uint32_t r, uint32_t t, uint32_t b)
    {
        if ((uint64_t)l + r >= res.first || (uint64_t)t + b >= res.second) throw std::invalid_argument("crop box leaves no pixels");
        return Crop{res, l, r, t, b};
    }
    // crop.rs:32-50
    static Crop from_topleft_and_dims(std::pair<uint32_t, uint32_t> res, uint32_t x, uint32_t y, uint32_t w, uint32_t h)
    {
        if ((uint64_t)x + w > res.first || (uint64_t)y + h > res.second) throw std::invalid_argument("box outside the frame");
        return Crop{res, x, res.first - w - x, y, res.second - h - y};
    }
    static Crop from_abi(std::pair<uint32_t, uint32_t> res, const uint32_t *box) { return from_edge_offsets(res, box[0], box[1], box[2], box[3]); }
    // crop.rs:53-68: per-edge minimum (unites the crops of the probed frames)
    Crop unite(const Crop &o) const
    {
        return from_edge_offsets(orig_res, std::min(left, o.left), std::min(right, o.right), std::min(top, o.top), std::min(bottom, o.bottom));
    }
    // crop.rs:92-103: {x, y, width, height}
    std::array<uint32_t, 4> as_view_args() const
    {
        if ((uint64_t)left + right > orig_res.first || (uint64_t)top + bottom > orig_res.second) throw std::overflow_error("crop offsets exceed the frame");
        return {left, top, orig_res.first - (left + right), orig_res.second - (top + bottom)};
    }
    uint32_t width() const { return orig_res.first - (left + right); }
    uint32_t height() const { return orig_res.second - (top + bottom); }
    uint32_t area() const { return width() * height(); }
    bool is_uncropped() const { return left == 0 && right == 0 && top == 0 && bottom == 0; }
    bool operator==(const Crop &o) const { return orig_res == o.orig_res && left == o.left && right == o.right && top == o.top && bottom == o.bottom; }
    bool operator<(const Crop &o) const
    {
        return std::tie(orig_res, left, right, top, bottom) < std::tie(o.orig_res, o.left, o.right, o.top, o.bottom);
    }
};

// ---- VideoHash (video_hash.rs:26-32) ----------------------------------------------------------------------
class VideoHash {
public:
    VideoHash() : hash_{}, duration_(0) {}  // Default, video_hash.rs:34-42
    VideoHash(const std::array<uint64_t, VDF_HASH_WORDS> &h, std::string src_path, uint32_t duration)
        : hash_(h), src_path_(std::move(src_path)), duration_(duration) {}

    // video_hash.rs:45-73.  frames: equal-size gray u8 frames (row-major, w x h); fewer than 16 (or none) ->
    // NotEnoughFrames; only the first 16 are used (dct_3d.rs:25).
    static VideoHash from_frames(const std::vector<const uint8_t *> &frames, uint32_t w, uint32_t h,
                                 const std::string &src_path, uint32_t duration, Context *ctx_opt = nullptr)
    {
        if (frames.size() < VDF_DCT_SIZE) throw Error::not_enough_frames();
        Context &ctx = ctx_opt ? *ctx_opt : Context::default_context();
        std::vector<uint8_t> packed((size_t)VDF_DCT_SIZE * w * h);
        for (size_t f = 0; f < VDF_DCT_SIZE; f++) std::copy(frames[f], frames[f] + (size_t)w * h, packed.begin() + f * (size_t)w * h);
        std::array<uint64_t, VDF_HASH_WORDS> words{};
        const int rc = vdf_hash_frames_u8(ctx.get(), packed.data(), 1, VDF_DCT_SIZE, w, h, (size_t)w * h,
                                          (size_t)w * h * VDF_DCT_SIZE, words.data(), nullptr);
        if (rc == VDF_E_NOT_ENOUGH_FRAMES) throw Error::not_enough_frames();
        if (rc == VDF_E_BAD_DIMS) throw Error::vid_proc(vdf_last_error(ctx.get()));
        if (rc != VDF_OK) throw Error(Error::Device, vdf_last_error(ctx.get()));
        return VideoHash(words, src_path, duration);
    }

    const std::string &src_path() const { return src_path_; }
    uint32_t duration() const { return duration_; }
    uint32_t hamming_distance(const VideoHash &o) const { return vdf_hamming_u1024(hash_.data(), o.hash_.data()); }
    double normalized_hamming_distance(const VideoHash &o) const { return hamming_distance(o) / 1000.0; }
    const std::array<uint64_t, VDF_HASH_WORDS> &words() const { return hash_; }

    VideoHash with_duration(uint32_t d) const { return VideoHash(hash_, src_path_, d); }
    VideoHash with_src_path(const std::string &p) const { return VideoHash(hash_, p, duration_); }
    static VideoHash empty_hash(const std::string &p) { return VideoHash({}, p, 0); }
    static VideoHash full_hash(const std::string &p)
    {
        std::array<uint64_t, VDF_HASH_WORDS> h;
        h.fill(~0ull);
        return VideoHash(h, p, 0);
    }
    bool operator==(const VideoHash &o) const { return hash_ == o.hash_ && PathKey(src_path_) == PathKey(o.src_path_) && duration_ == o.duration_; }

private:
    std::array<uint64_t, VDF_HASH_WORDS> hash_;
    std::string src_path_;
    uint32_t duration_;
};

// ---- MatchGroup (matches/match_group.rs) ---------------------------------------------------------------------
class MatchGroup {
public:
    static MatchGroup make(std::vector<std::string> entries)
    {
        if (entries.size() < 2) throw TooFewEntries();
        return MatchGroup(std::nullopt, std::move(entries));
    }
    static MatchGroup make_with_reference(std::string reference, std::vector<std::string> entries)
    {
        if (entries.empty()) throw TooFewEntries();
        return MatchGroup(std::move(reference), std::move(entries));
    }
    size_t len() const { return duplicates_.size(); }
    const std::optional<std::string> &reference() const { return reference_; }
    const std::vector<std::string> &duplicates() const { return duplicates_; }
    std::vector<std::string> contained_paths() const
    {  // duplicates, then the reference (match_group.rs:68-81)
        std::vector<std::string> v = duplicates_;
        if (reference_) v.push_back(*reference_);
        return v;
    }
    std::vector<MatchGroup> dup_combinations() const
    {  // match_group.rs:87-105
        std::vector<MatchGroup> out;
        if (reference_) {
            for (const auto &d : duplicates_) out.push_back(make_with_reference(*reference_, {d}));
        } else {
            for (size_t i = 0; i < duplicates_.size(); i++)
                for (size_t j = i + 1; j < duplicates_.size(); j++) out.push_back(make({duplicates_[i], duplicates_[j]}));
        }
        return out;
    }

private:
    MatchGroup(std::optional<std::string> r, std::vector<std::string> d) : reference_(std::move(r)), duplicates_(std::move(d)) {}
    std::optional<std::string> reference_;
    std::vector<std::string> duplicates_;
};

namespace detail {
// Search::sort, search_algorithm.rs:55-61: stable by (duration, src_path) with Path ordering - on the host, a PathKey per entry (the
// readable form; what sort_order() below falls back to for a handful of hashes or without a context).
inline std::vector<size_t> sort_order_host(const std::vector<VideoHash> &h)
{
    std::vector<PathKey> keys;
    keys.reserve(h.size());
    for (const auto &x : h) keys.emplace_back(x.src_path());
    std::vector<size_t> idx(h.size());
    std::iota(idx.begin(), idx.end(), size_t{0});
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
        if (h[a].duration() != h[b].duration()) return h[a].duration() < h[b].duration();
        return keys[a] < keys[b];
    });
    return idx;
}
// The same order through the engine (vdf_sort_order_paths: the path half on the device for plain paths, the library's multi-threaded
// component comparator otherwise).  A million PathKeys and their comparisons cost seconds on the host - beside a 0.1 s search.
inline std::vector<size_t> sort_order(const std::vector<VideoHash> &h, Context *ctx = nullptr)
{
    if (h.size() < 2048 || !ctx) return sort_order_host(h);
    std::vector<uint32_t> dur(h.size());
    std::vector<uint64_t> offs(h.size() + 1, 0);
    for (size_t i = 0; i < h.size(); i++) { dur[i] = h[i].duration(); offs[i + 1] = offs[i] + h[i].src_path().size(); }
    std::string blob;
    blob.reserve((size_t)offs.back());
    for (const auto &x : h) blob += x.src_path();
    std::vector<uint32_t> order32(h.size());
    if (vdf_sort_order_paths(ctx->get(), dur.data(), offs.data(), blob.data(), h.size(), order32.data(), nullptr) != VDF_OK)
        return sort_order_host(h);  // (more than 2^32 - 1 entries, a device error: the host's order is the same one)
    return std::vector<size_t>(order32.begin(), order32.end());
}
inline void to_soa(const std::vector<VideoHash> &h, const std::vector<size_t> &order, std::vector<uint64_t> &words,
                   std::vector<uint32_t> &dur)
{
    words.resize(order.size() * VDF_HASH_WORDS);
    dur.resize(order.size());
    for (size_t k = 0; k < order.size(); k++) {
        std::copy(h[order[k]].words().begin(), h[order[k]].words().end(), words.begin() + k * VDF_HASH_WORDS);
        dur[k] = h[order[k]].duration();
    }
}
struct Groups {
    vdf_groups g{};
    ~Groups() { vdf_groups_free(&g); }
};
}  // namespace detail

// video_dup_finder.rs:7-13
inline std::vector<MatchGroup> search(const std::vector<VideoHash> &hashes, double tolerance,
                                      Context &ctx = Context::default_context())
{
    std::vector<MatchGroup> out;
    if (hashes.empty()) return out;  // search_algorithm.rs:89-91
    const auto order = detail::sort_order(hashes, &ctx);
    std::vector<uint64_t> words;
    std::vector<uint32_t> dur;
    detail::to_soa(hashes, order, words, dur);
    detail::Groups gr;
    if (vdf_search_self(ctx.get(), words.data(), dur.data(), dur.size(), vdf_tolerance_int(tolerance), &gr.g) != VDF_OK)
        throw Error(Error::Device, vdf_last_error(ctx.get()));
    for (uint64_t g = 0; g < gr.g.n_groups; g++) {
        std::vector<std::string> paths;
        for (uint64_t k = gr.g.offsets[g]; k < gr.g.offsets[g + 1]; k++) paths.push_back(hashes[order[gr.g.members[k]]].src_path());
        if (paths.size() >= 2) out.push_back(MatchGroup::make(std::move(paths)));  // filter_map(.. .ok())
    }
    return out;
}

// video_dup_finder.rs:19-46
inline std::vector<MatchGroup> search_with_references(const std::vector<VideoHash> &ref_hashes,
                                                      const std::vector<VideoHash> &new_hashes, double tolerance,
                                                      Context &ctx = Context::default_context())
{
    std::vector<MatchGroup> out;
    if (ref_hashes.empty() || new_hashes.empty()) return out;
    const auto order = detail::sort_order(new_hashes, &ctx);
    std::vector<uint64_t> words, rwords;
    std::vector<uint32_t> dur, rdur;
    detail::to_soa(new_hashes, order, words, dur);
    std::vector<size_t> ident(ref_hashes.size());
    std::iota(ident.begin(), ident.end(), size_t{0});
    detail::to_soa(ref_hashes, ident, rwords, rdur);
    detail::Groups gr;
    if (vdf_search_refs(ctx.get(), words.data(), dur.data(), dur.size(), rwords.data(), rdur.data(), rdur.size(),
                        vdf_tolerance_int(tolerance), &gr.g) != VDF_OK)
        throw Error(Error::Device, vdf_last_error(ctx.get()));
    for (uint64_t g = 0; g < gr.g.n_groups; g++) {
        std::vector<std::string> paths;
        for (uint64_t k = gr.g.offsets[g]; k < gr.g.offsets[g + 1]; k++) paths.push_back(new_hashes[order[gr.g.members[k]]].src_path());
        out.push_back(MatchGroup::make_with_reference(ref_hashes[(size_t)gr.g.ref_index[g]].src_path(), std::move(paths)));
    }
    return out;
}

// ---- the app's side of the path: its hash cache, the sidecar, search_disk, the distance key -------------------
// vid_dup_finder_app/src/video_hash_filesystem_cache/generic_filesystem_cache/base_fs_cache.rs:167-223 (load), cache_metadata.rs:45-168,
// video_hash_filesystem_cache.rs:76-139 (sidecar), app/app_fns.rs:428-482 (search_disk), app/search_output.rs:43-60 (Sorting::Distance).

// The loaded cache as arrays: no object per entry.  Entries that held Err(..) are counted (n_err) and absent.
class Cache {
public:
    // load_cache_from_disk (bincode backend): malformed bytes are the original's Deserialization error
    static Cache from_bytes(const void *data, size_t len)
    {
        Cache c;
        if (vdf_cache_decode(static_cast<const uint8_t *>(data), len, &c.soa_) != VDF_OK) throw Error(Error::VidProc, "cache file does not deserialize");
        return c;
    }
    Cache(Cache &&o) noexcept : soa_(o.soa_) { o.soa_ = vdf_cache_soa{}; }
    Cache &operator=(Cache &&o) noexcept { if (this != &o) { vdf_cache_free(&soa_); soa_ = o.soa_; o.soa_ = vdf_cache_soa{}; } return *this; }
    Cache(const Cache &) = delete;
    Cache &operator=(const Cache &) = delete;
    ~Cache() { vdf_cache_free(&soa_); }
    size_t len() const { return (size_t)soa_.n_ok; }
    uint64_t n_err() const { return soa_.n_err; }
    std::string path(size_t i) const { return std::string(soa_.paths + soa_.path_offsets[i], (size_t)(soa_.path_offsets[i + 1] - soa_.path_offsets[i])); }
    uint32_t duration(size_t i) const { return soa_.durations[i]; }
    const vdf_cache_soa &soa() const { return soa_; }

private:
    Cache() = default;
    vdf_cache_soa soa_{};
};

// search_disk's middle: `includes_cand` / `includes_ref` are the --files / --with-refs filename filters; no reference passes its filter =>
// find-all search (app_fns.rs:474-478).  One library call: PathBuf ranks, upload, Search::sort on the device, the search, map back.
// keys (optional): SearchOutput::sort's Sorting::Distance key of every group, from the same call's groups (vdf_groups_max_distance).
template <class FC, class FR>
inline std::vector<MatchGroup> search_cache(const Cache &cache, double tolerance, FC includes_cand, FR includes_ref,
                                            std::vector<uint32_t> *keys = nullptr, Context &ctx = Context::default_context())
{
    std::vector<uint64_t> cand, refs;
    for (size_t i = 0; i < cache.len(); i++) {
        const std::string p = cache.path(i);
        if (includes_cand(p)) cand.push_back(i);
        if (includes_ref(p)) refs.push_back(i);
    }
    std::vector<MatchGroup> out;
    if (cand.empty()) return out;  // "No files were found at the paths given by --files"
    const vdf_cache_soa &a = cache.soa();
    detail::Groups gr;
    if (vdf_search_cache_entries(ctx.get(), a.hashes, a.durations, a.path_offsets, a.paths, cache.len(), cand.data(), cand.size(),
                                 refs.empty() ? nullptr : refs.data(), refs.size(), vdf_tolerance_int(tolerance), &gr.g, nullptr) != VDF_OK)
        throw Error(Error::Device, vdf_last_error(ctx.get()));
    std::vector<uint32_t> all_keys((size_t)gr.g.n_groups, 0u);
    if (keys && gr.g.n_groups &&
        vdf_groups_max_distance(ctx.get(), a.hashes, cache.len(), a.hashes, cache.len(), &gr.g, all_keys.data()) != VDF_OK)
        throw Error(Error::Device, vdf_last_error(ctx.get()));
    if (keys) keys->clear();
    for (uint64_t g = 0; g < gr.g.n_groups; g++) {
        std::vector<std::string> paths;
        for (uint64_t k = gr.g.offsets[g]; k < gr.g.offsets[g + 1]; k++) paths.push_back(cache.path((size_t)gr.g.members[k]));
        if (gr.g.ref_index[g] >= 0) out.push_back(MatchGroup::make_with_reference(cache.path((size_t)gr.g.ref_index[g]), std::move(paths)));
        else if (paths.size() >= 2) out.push_back(MatchGroup::make(std::move(paths)));
        else continue;
        if (keys) keys->push_back(all_keys[(size_t)g]);
    }
    return out;
}

// VdfCacheMetadata (cache_metadata.rs:45-51) and the sidecar's name; MetadataError carries the app's own message
struct MetadataError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct CacheMetadata {
    vdf_cache_metadata m{};
    static CacheMetadata make(int32_t crop, double skip_forward_amount)  // VdfCacheMetadata::new: Unix, FfmpegBackend, version 1
    {
        CacheMetadata x;
        if (vdf_cache_metadata_new(crop, skip_forward_amount, &x.m) != VDF_OK) throw MetadataError("bad cropdetect");
        return x;
    }
    std::string to_disk_fmt() const
    {
        char buf[512];
        size_t n = 0;
        if (vdf_cache_metadata_format(&m, buf, sizeof buf, &n) != VDF_OK) throw MetadataError("metadata fields out of range");
        return std::string(buf, n);
    }
    static CacheMetadata try_parse(const std::string &text)
    {
        CacheMetadata x;
        char err[1024] = {0};
        if (vdf_cache_metadata_parse(text.data(), text.size(), &x.m, err, sizeof err) != VDF_OK) throw MetadataError(err);
        return x;
    }
    void validate(int32_t exp_crop, double exp_skip_forward_amount) const
    {
        char err[1024] = {0};
        if (vdf_cache_metadata_validate(&m, exp_crop, exp_skip_forward_amount, err, sizeof err) != VDF_OK) throw MetadataError(err);
    }
};
inline std::optional<std::string> metadata_path(const std::string &cache_path)  // file_stem + with_file_name; nullopt = no file name (EINVAL in the app)
{
    std::string buf(cache_path.size() + 32, '\0');
    size_t n = 0;
    if (vdf_cache_metadata_path(cache_path.data(), cache_path.size(), buf.data(), buf.size(), &n) != VDF_OK) return std::nullopt;
    buf.resize(n);
    return buf;
}

}  // namespace vdf
