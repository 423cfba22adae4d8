// Driver for the C++ host mirror (vid_dup_finder_lib_amd/host/vdf.hpp): reads like the reference's own tests.
//   selftest                      no GPU: Path ordering, MatchGroup contract, Hamming axioms
//   search <file> <tolerance>     GPU: prints one group per line (duplicate paths, tab separated)
//   refs <file> <n_ref> <tol>     GPU: first n_ref entries are the references
//   cache <cache.bin> <tol> <cand prefix> <ref prefix | ->   GPU: the app's search_disk on a cache FILE (bincode bytes): entries whose path starts
//                                 with the prefixes are the candidates / references; prints "key<TAB>reference or -<TAB>duplicates..."
// File format: u64 n, then n x {16 x u64 hash, u32 duration, u32 path_len, path bytes}.
#include <cassert>
#include <chrono>
#include <random>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "../../vid_dup_finder_lib_amd/host/vdf.hpp"

static std::vector<vdf::VideoHash> read_file(const char *path)
{
    std::ifstream f(path, std::ios::binary);
    uint64_t n = 0;
    f.read((char *)&n, 8);
    std::vector<vdf::VideoHash> out;
    for (uint64_t i = 0; i < n; i++) {
        std::array<uint64_t, 16> h;
        uint32_t d, len;
        f.read((char *)h.data(), 128);
        f.read((char *)&d, 4);
        f.read((char *)&len, 4);
        std::string p(len, '\0');
        f.read(p.data(), len);
        out.emplace_back(h, p, d);
    }
    return out;
}

static void print_groups(const std::vector<vdf::MatchGroup> &gs)
{
    for (const auto &g : gs) {
        std::cout << (g.reference() ? *g.reference() : std::string("-"));
        for (const auto &p : g.duplicates()) std::cout << '\t' << p;
        std::cout << '\n';
    }
}

static int selftest()
{
    using vdf::PathKey;
    assert(PathKey("a/b") < PathKey("a.b"));               // component-wise, not bytewise
    assert(PathKey("a//b/") == PathKey("a/b") && PathKey("a/./b") == PathKey("a/b"));
    assert(PathKey("/a") < PathKey("a") && PathKey("./a") < PathKey("../a") && PathKey("../a") < PathKey("a"));
    // crop.rs:204-271 (the reference's as_view_args / from_topleft_and_dims cases) and the per-edge minimum of :53-68
    using vdf::Crop;
    using A4 = std::array<uint32_t, 4>;
    assert((Crop::from_edge_offsets({100, 100}, 0, 0, 0, 0).as_view_args() == A4{0, 0, 100, 100}));
    assert((Crop::from_edge_offsets({100, 100}, 1, 0, 0, 0).as_view_args() == A4{1, 0, 99, 100}));
    assert((Crop::from_edge_offsets({100, 100}, 0, 1, 0, 0).as_view_args() == A4{0, 0, 99, 100}));
    assert((Crop::from_edge_offsets({100, 100}, 0, 0, 1, 0).as_view_args() == A4{0, 1, 100, 99}));
    assert((Crop::from_edge_offsets({100, 100}, 0, 0, 0, 1).as_view_args() == A4{0, 0, 100, 99}));
    assert((Crop::from_edge_offsets({100, 100}, 25, 25, 25, 25).as_view_args() == A4{25, 25, 50, 50}));
    assert((Crop::from_edge_offsets({768, 432}, 96, 96, 0, 0).as_view_args() == A4{96, 0, 576, 432}));
    assert((Crop::from_topleft_and_dims({100, 100}, 11, 12, 13, 14).as_view_args() == A4{11, 12, 13, 14}));
    assert(Crop::from_edge_offsets({3, 3}, 2, 0, 2, 0) == Crop::from_topleft_and_dims({3, 3}, 2, 2, 1, 1));
    assert(Crop::from_edge_offsets({100, 80}, 10, 0, 5, 7).unite(Crop::from_edge_offsets({100, 80}, 3, 4, 9, 2)) == Crop::from_edge_offsets({100, 80}, 3, 0, 5, 2));
    {
        const uint32_t row[4] = {1, 1, 1, 2};  // video_frames_gray.rs:444-459 as the C ABI returns it
        const Crop c = Crop::from_abi({5, 6}, row);
        assert(c.width() == 3 && c.height() == 3 && c.area() == 9 && !c.is_uncropped());
    }
    bool none_left = false;
    try { Crop::from_edge_offsets({100, 100}, 50, 50, 50, 50); } catch (const std::invalid_argument &) { none_left = true; }
    assert(none_left);
    // matches/match_group.rs
    bool threw = false;
    try { vdf::MatchGroup::make({"a"}); } catch (const vdf::TooFewEntries &) { threw = true; }
    assert(threw);
    threw = false;
    try { vdf::MatchGroup::make_with_reference("r", {}); } catch (const vdf::TooFewEntries &) { threw = true; }
    assert(threw);
    auto g = vdf::MatchGroup::make({"a", "b", "c"});
    assert(g.len() == 3 && !g.reference() && g.dup_combinations().size() == 3);
    auto r = vdf::MatchGroup::make_with_reference("ref", {"x", "y"});
    assert(r.contained_paths() == (std::vector<std::string>{"x", "y", "ref"}) && r.dup_combinations().size() == 2);
    // video_hash.rs:325-371
    auto e = vdf::VideoHash::empty_hash(""), f = vdf::VideoHash::full_hash("");
    assert(e.hamming_distance(e) == 0 && f.hamming_distance(f) == 0 && e.hamming_distance(f) == 1024);
    assert(vdf::VideoHash() == vdf::VideoHash::empty_hash(""));
    assert(vdf_tolerance_int(0.35) == 350 && vdf_tolerance_int(0.3) == 300);
    // too few frames is rejected before any device work (video_hash.rs:53,61)
    threw = false;
    try { vdf::VideoHash::from_frames({}, 16, 16, "p", 1); } catch (const vdf::Error &err) { threw = err.kind == vdf::Error::NotEnoughFrames; }
    assert(threw);
    // the sidecar (cache_metadata.rs; video_hash_filesystem_cache.rs:76-139): host only
    auto md = vdf::CacheMetadata::make(VDF_CROPDETECT_LETTERBOX, 15.0);
    assert(md.to_disk_fmt() == "Unix,FfmpegBackend,Letterbox,15,1");
    vdf::CacheMetadata::try_parse(md.to_disk_fmt()).validate(VDF_CROPDETECT_LETTERBOX, 15.0);
    std::string msg;
    try { vdf::CacheMetadata::try_parse("Unix,FfmpegBackend,None,15,1").validate(VDF_CROPDETECT_LETTERBOX, 15.0); } catch (const vdf::MetadataError &err) { msg = err.what(); }
    assert(msg == "crop mismatch: Act: None, Exp: Letterbox");
    msg.clear();
    try { vdf::CacheMetadata::try_parse("Unix,FfmpegBackend,letterbox,15,1"); } catch (const vdf::MetadataError &err) { msg = err.what(); }
    assert(msg.find("Could not parse crop") == 0);
    assert(*vdf::metadata_path("/home/u/.cache/vdf/cache.bin") == "/home/u/.cache/vdf/cache.metadata.txt" && *vdf::metadata_path("d//c.tar.gz") == "d/c.tar.metadata.txt");
    assert(!vdf::metadata_path("..") && !vdf::metadata_path("/"));
    // a cache file's bytes -> arrays (host only): count 1, key "a/b", mtime (7 s, 9 ns), Ok, 16 words, src_path "a/b", duration 42
    std::string bytes("\x01\x03" "a/b" "\x07\x09\x00", 8);
    for (int i = 0; i < 16; i++) bytes += (char)(i + 1);
    bytes += std::string("\x03" "a/b" "\x2a", 5);
    auto cache = vdf::Cache::from_bytes(bytes.data(), bytes.size());
    assert(cache.len() == 1 && cache.path(0) == "a/b" && cache.duration(0) == 42 && cache.soa().hashes[15] == 16 && cache.soa().mtime_secs[0] == 7);
    threw = false;
    try { vdf::Cache::from_bytes(bytes.data(), bytes.size() - 1); } catch (const vdf::Error &) { threw = true; }
    assert(threw);
    std::puts("selftest ok");
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && !std::strcmp(argv[1], "selftest")) return selftest();
    if (argc >= 4 && !std::strcmp(argv[1], "search")) {
        print_groups(vdf::search(read_file(argv[2]), std::atof(argv[3])));
        return 0;
    }
    if (argc >= 5 && !std::strcmp(argv[1], "refs")) {
        auto all = read_file(argv[2]);
        const size_t nr = std::strtoul(argv[3], nullptr, 10);
        std::vector<vdf::VideoHash> refs(all.begin(), all.begin() + nr), news(all.begin() + nr, all.end());
        print_groups(vdf::search_with_references(refs, news, std::atof(argv[4])));
        return 0;
    }
    if (argc >= 6 && !std::strcmp(argv[1], "cache")) {
        std::ifstream f(argv[2], std::ios::binary);
        const std::string bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        const auto cache = vdf::Cache::from_bytes(bytes.data(), bytes.size());
        const std::string cp = argv[4], rp = argv[5];
        std::vector<uint32_t> keys;
        const auto groups = vdf::search_cache(
            cache, std::atof(argv[3]), [&](const std::string &p) { return p.rfind(cp, 0) == 0; },
            [&](const std::string &p) { return rp != "-" && p.rfind(rp, 0) == 0; }, &keys);
        for (size_t g = 0; g < groups.size(); g++) {
            std::cout << keys[g] << '\t' << (groups[g].reference() ? *groups[g].reference() : std::string("-"));
            for (const auto &p : groups[g].duplicates()) std::cout << '\t' << p;
            std::cout << '\n';
        }
        return 0;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "bench")) {  // what vdf::search() costs around the GPU call at n hashes (plain paths of a media library)
        const size_t n = std::strtoull(argv[2], nullptr, 10);
        std::mt19937_64 rng(7);
        std::vector<vdf::VideoHash> hs;
        hs.reserve(n);
        for (size_t i = 0; i < n; i++) {
            std::array<uint64_t, 16> h;
            for (auto &w : h) w = rng();
            h[15] &= (1ull << 40) - 1;
            char path[96];
            const uint64_t id = rng() % (4 * n + 1);
            std::snprintf(path, sizeof path, "/srv/media/lib_%02u/show_%04u/season_%02u/clip_%08llu.mkv", (unsigned)(id % 100), (unsigned)(id / 40 % 10000),
                          (unsigned)(id / 8 % 5), (unsigned long long)id);
            hs.emplace_back(h, path, (uint32_t)(5 + rng() % 7200));
        }
        if (n >= 2) hs[1] = vdf::VideoHash(hs[0].words(), "/srv/media/copy.mkv", hs[0].duration());  // one pair to find
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        vdf::search(std::vector<vdf::VideoHash>(hs.begin(), hs.begin() + std::min<size_t>(n, 1000)), 0.35);  // context, code objects
        const auto t0 = now();
        const auto order_host = vdf::detail::sort_order_host(hs);
        const auto t1 = now();
        const auto order = vdf::detail::sort_order(hs, &vdf::Context::default_context());
        const auto t2 = now();
        const auto groups = vdf::search(hs, 0.35);
        const auto t3 = now();
        std::printf("n=%zu: Search::sort on the host (PathKey) %.1f ms, through vdf_sort_order_paths %.1f ms (same order: %s), whole vdf::search %.1f ms, %zu groups\n", n,
                    ms(t0, t1), ms(t1, t2), order == order_host ? "yes" : "NO", ms(t2, t3), groups.size());
        return order == order_host ? 0 : 1;
    }
    std::fprintf(stderr, "usage: selftest | search <file> <tol> | refs <file> <n_ref> <tol> | cache <cache.bin> <tol> <cand prefix> <ref prefix | ->\n");
    return 2;
}
