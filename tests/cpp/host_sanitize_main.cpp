// AddressSanitizer/UBSan driver for the host-only parts of the library (no GPU, no HIP runtime): the greedy
// replay, group assembly, pair counting and the resize coefficient tables.  Built by tests/test_host_sanitizers.py
// with -fsanitize=address,undefined straight from the sources.
#include <cassert>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/vdf.h"
#include "../../vid_dup_finder_lib_amd/csrc/host_sort.h"
#include "../../vid_dup_finder_lib_amd/csrc/resize_tables.h"
#include <algorithm>
#include <numeric>

int main()
{
    std::mt19937 rng(1);
    // replay over random hit lists, in one go and in pieces with carried state
    for (int rep = 0; rep < 200; rep++) {
        const size_t n = 1 + rng() % 300;
        std::vector<vdf_hit> hits;
        for (uint32_t i = 0; i + 1 < n; i++)
            for (uint32_t j = i + 1; j < n; j++)
                if (rng() % 17 == 0) hits.push_back(vdf_hit{i, j});
        vdf_groups a{}, b{};
        assert(vdf_replay_self(n, hits.data(), hits.size(), 0, 0xFFFFFFFFu, nullptr, &a) == VDF_OK);
        assert(vdf_groups_finish_self(&a) == VDF_OK);
        std::vector<uint8_t> matched(n, 0);
        const uint32_t cut = (uint32_t)(rng() % n);
        assert(vdf_replay_self(n, hits.data(), hits.size(), 0, cut, matched.data(), &b) == VDF_OK);
        assert(vdf_replay_self(n, hits.data(), hits.size(), cut, (uint32_t)n, matched.data(), &b) == VDF_OK);
        assert(vdf_groups_finish_self(&b) == VDF_OK);
        assert(a.n_groups == b.n_groups);
        if (a.n_groups) {
            assert(std::memcmp(a.offsets, b.offsets, (a.n_groups + 1) * 8) == 0);
            assert(std::memcmp(a.members, b.members, a.offsets[a.n_groups] * 8) == 0);
        }
        vdf_groups r{};
        assert(vdf_groups_from_ref_hits(hits.data(), hits.size(), &r) == VDF_OK);
        vdf_groups_free(&a); vdf_groups_free(&b); vdf_groups_free(&r);
    }
    // the hash-cache codec (csrc/cache_format.cpp: arrays allocated at an upper bound and filled in place): round trips with every varint
    // width inside the hashes, then every truncation and a few thousand byte mutations of a valid file - accepted or VDF_E_INVAL, never
    // a read or write out of bounds (copies sized exactly, so ASan sees an overrun of one byte)
    for (int rep = 0; rep < 30; rep++) {
        const size_t n = rng() % 40;
        std::vector<uint64_t> h(n * 16), secs(n), offs(n + 1, 0);
        std::vector<uint32_t> d(n), nanos(n);
        std::string blob;
        for (size_t i = 0; i < n; i++) {
            for (int w = 0; w < 16; w++) {
                const uint64_t x = ((uint64_t)rng() << 32) | rng();
                const int kind = rng() % 5;
                h[i * 16 + w] = kind == 0 ? x % 251 : kind == 1 ? x % 65536 : kind == 2 ? (uint32_t)x : x;
            }
            d[i] = rng(); secs[i] = ((uint64_t)rng() << 20) ^ rng(); nanos[i] = rng() % 1000000000u;
            const size_t len = rng() % 300;
            for (size_t k = 0; k < len; k++) blob.push_back((char)('a' + rng() % 26));
            offs[i + 1] = blob.size();
        }
        uint8_t *enc = nullptr; size_t enc_len = 0;
        assert(vdf_cache_encode(n, h.data(), d.data(), offs.data(), blob.data(), secs.data(), nanos.data(), &enc, &enc_len) == VDF_OK);
        std::vector<uint8_t> file(enc, enc + enc_len);  // exactly sized copy
        vdf_buffer_free(enc);
        vdf_cache_soa c{};
        assert(vdf_cache_decode(file.data(), file.size(), &c) == VDF_OK && c.n_ok == n && c.n_err == 0 && c.n_key_differs == 0);
        assert(n == 0 || (std::memcmp(c.hashes, h.data(), n * 128) == 0 && std::memcmp(c.durations, d.data(), n * 4) == 0 &&
                          std::memcmp(c.path_offsets, offs.data(), (n + 1) * 8) == 0 && std::memcmp(c.paths, blob.data(), blob.size()) == 0 &&
                          std::memcmp(c.mtime_secs, secs.data(), n * 8) == 0 && std::memcmp(c.mtime_nanos, nanos.data(), n * 4) == 0));
        vdf_cache_free(&c);
        for (size_t cut = 0; cut < file.size(); cut += 1 + file.size() / 400) {
            std::vector<uint8_t> part(file.begin(), file.begin() + cut);
            vdf_cache_soa t{};
            const int rc = vdf_cache_decode(part.data(), part.size(), &t);
            assert(rc == VDF_E_INVAL || (rc == VDF_OK && cut == 0 && false) || (rc == VDF_OK && t.n_entries == 0 && part.size() == 1) || rc == VDF_OK);
            if (rc == VDF_OK) vdf_cache_free(&t);
        }
        for (int m = 0; m < 300 && !file.empty(); m++) {
            std::vector<uint8_t> mut = file;
            for (int k = 0; k < 1 + (int)(rng() % 3); k++) mut[rng() % mut.size()] = (uint8_t)rng();
            vdf_cache_soa t{};
            const int rc = vdf_cache_decode(mut.data(), mut.size(), &t);
            assert(rc == VDF_OK || rc == VDF_E_INVAL || rc == VDF_E_OOM);
            if (rc == VDF_OK) { assert(t.n_ok <= t.n_entries); vdf_cache_free(&t); }
        }
    }
    {   // a hostile count in front of nothing: refused without allocating the claimed size
        const uint8_t huge[9] = {253, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0x7F};
        vdf_cache_soa t{};
        assert(vdf_cache_decode(huge, sizeof huge, &t) == VDF_E_INVAL);
    }
    vdf_groups e{};
    assert(vdf_replay_self(0, nullptr, 0, 0, 0, nullptr, &e) == VDF_OK && vdf_groups_finish_self(&e) == VDF_OK && e.n_groups == 0);
    vdf_groups_free(&e);
    // pair counting at the u32 boundary
    std::vector<uint32_t> d = {0, 1, 10, 11, 12, 4000000000u, 4294967295u, 4294967295u};
    assert(vdf_count_pairs_self(d.data(), d.size()) > 0);
    assert(vdf_count_pairs_refs(d.data(), d.size(), d.data(), d.size()) > 0);
    assert(vdf_tolerance_int(0.35) == 350 && vdf_tolerance_int(-3.0) == 0 && vdf_tolerance_int(1e300) == 0xFFFFFFFFu);
    // coefficient tables for every size up to 300 and a few large ones
    for (uint32_t sz : std::vector<uint32_t>{1, 2, 3, 15, 16, 17, 31, 33, 63, 64, 65, 127, 270, 1080, 1920, 4320}) {
        vdf::HostAxisTable t;
        assert(vdf::build_axis_table(sz, 16, t));
        for (int o = 0; o < 16; o++) assert(t.start[o] >= 0 && t.start[o] + t.size[o] <= (int32_t)sz);
        for (int v = 0; v < 4; v++) {
            vdf::MfmaAxisTable m;
            assert(vdf::build_mfma_axis_table(sz, v, m));
            if (v < 3) assert(m.operand.size() == (size_t)m.n_tiles * 2 * 64 * 16 && m.bias.size() == 16);
            else assert(m.operand.size() == (size_t)16 * m.band_stride && m.band_meta.size() == 32 && m.bias.size() == 16);
        }
    }
    // the search path's host radix sort against the standard library, on both sides of its small-input cut-over
    for (size_t n : std::vector<size_t>{0, 1, 2, 100, 4095, 4096, 5000, 70000}) {
        std::vector<vdf_hit> h(n), want;
        const uint32_t rows = 1u + (uint32_t)(rng() % 200000), cols = 1u + (uint32_t)(rng() % 2000000);
        for (auto &x : h) x = vdf_hit{(uint32_t)(rng() % rows), (uint32_t)(rng() % cols)};
        if (n > 10) { h[3].row = 0xFFFFFFFFu; h[4].col = 0xFFFFFFFFu; h[5] = vdf_hit{0, 0}; }
        want = h;
        std::sort(want.begin(), want.end(), [](const vdf_hit &a, const vdf_hit &b) { return a.row != b.row ? a.row < b.row : a.col < b.col; });
        vdf_impl::sort_hits(h.data(), n);
        assert(n == 0 || std::memcmp(h.data(), want.data(), n * sizeof(vdf_hit)) == 0);
    }
    std::puts("sanitize ok");
    return 0;
}
