// Sanitizer driver for the host-only, MULTI-THREADED and text-parsing parts of the library (no GPU, no HIP runtime): the speculative
// multi-threaded cache decoder (csrc/cache_format.cpp: ranges cut where an Ok entry's byte pattern resynchronises), the sidecar parser
// (csrc/cache_metadata.cpp) and the PathBuf ranker (csrc/path_order.cpp: a sample sort over host threads).  Built twice by
// tests/test_host_sanitizers.py: -fsanitize=address,undefined and -fsanitize=thread.
#include <algorithm>
#include <cassert>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "../../include/vdf.h"

static std::mt19937 rng(7);

static std::string random_path(bool plain)
{
    static const char *pieces[] = {"a", "b", "ab", "a.b", "a b", "videos", "x", "Z", "\x01", "\xff\xfe", "~", "-", "0"};
    std::string p;
    if (rng() % 3) p += "/";
    const int nc = rng() % 6;
    for (int c = 0; c < nc; c++) {
        if (c) p += "/";
        const int kind = plain ? 0 : (int)(rng() % 9);
        if (kind == 1) p += ".";
        else if (kind == 2) p += "..";
        else if (kind == 3) p += "";  // an empty component: "a//b"
        else {
            p += pieces[rng() % (sizeof pieces / sizeof *pieces)];
            if (rng() % 4 == 0) p += std::to_string(rng() % 50);
        }
    }
    if (!plain && rng() % 7 == 0) p += "/";
    if (!plain && rng() % 11 == 0 && !p.empty()) p[rng() % p.size()] = '\0';
    if (rng() % 40 == 0) p += std::string(1 + rng() % 2000, 'q');  // longer than the device route's 1024 bytes
    return p;
}

static void check_decode_mt()
{
    for (int rep = 0; rep < 12; rep++) {
        const size_t n = rep == 0 ? 0 : 1 + rng() % 3000;
        std::vector<uint64_t> h(n * 16), secs(n), offs(n + 1, 0);
        std::vector<uint32_t> d(n), nanos(n);
        std::string blob;
        for (size_t i = 0; i < n; i++) {
            // most entries look like real hashes (sixteen 9-byte varints: where the decoder cuts its ranges); some hold every other varint
            // width, which a cut must not land in
            const bool real = rng() % 5 != 0;
            for (int w = 0; w < 16; w++) {
                const uint64_t x = ((uint64_t)rng() << 32) | rng();
                const int kind = real ? 4 : (int)(rng() % 5);
                h[i * 16 + w] = kind == 0 ? x % 251 : kind == 1 ? x % 65536 : kind == 2 ? (uint32_t)x : (x | (1ull << 63));
            }
            d[i] = rng() % 8 ? rng() % 7200 : rng();
            secs[i] = ((uint64_t)rng() << 20) ^ rng();
            nanos[i] = rng() % 1000000000u;
            blob += random_path(rng() % 2 == 0);
            offs[i + 1] = blob.size();
        }
        uint8_t *enc = nullptr;
        size_t enc_len = 0;
        assert(vdf_cache_encode(n, h.data(), d.data(), offs.data(), blob.data(), secs.data(), nanos.data(), &enc, &enc_len) == VDF_OK);
        const std::vector<uint8_t> file(enc, enc + enc_len);  // exactly sized: ASan sees an overrun of one byte
        vdf_buffer_free(enc);
        vdf_cache_soa one{};
        assert(vdf_cache_decode_mt(file.data(), file.size(), 1, &one) == VDF_OK && one.n_ok == n);
        const unsigned long long fb0 = vdf_cache_decode_fallbacks();
        for (int nt : {2, 3, 5, 8, 16}) {
            vdf_cache_soa c{};
            assert(vdf_cache_decode_mt(file.data(), file.size(), nt, &c) == VDF_OK);
            assert(c.n_entries == one.n_entries && c.n_ok == n && c.n_err == 0 && c.n_key_differs == 0);
            if (n) {
                assert(std::memcmp(c.hashes, h.data(), n * 128) == 0 && std::memcmp(c.durations, d.data(), n * 4) == 0);
                assert(std::memcmp(c.path_offsets, offs.data(), (n + 1) * 8) == 0 && std::memcmp(c.paths, blob.data(), blob.size()) == 0);
                assert(std::memcmp(c.mtime_secs, secs.data(), n * 8) == 0 && std::memcmp(c.mtime_nanos, nanos.data(), n * 4) == 0);
            }
            vdf_cache_free(&c);
        }
        if (n > 500) assert(vdf_cache_decode_fallbacks() == fb0);  // a valid file's ranges meet: the threads really ran side by side
        // damaged files on several threads: the verdict (and for accepted files the content) is the one-thread decoder's
        for (int m = 0; m < 60 && !file.empty(); m++) {
            std::vector<uint8_t> mut = file;
            if (m % 3 == 0) mut.resize(rng() % mut.size());
            else
                for (int k = 0; k < 1 + (int)(rng() % 3); k++) mut[rng() % mut.size()] = (uint8_t)rng();
            vdf_cache_soa a{}, b{};
            const int ra = vdf_cache_decode_mt(mut.data(), mut.size(), 1, &a);
            const int rb = vdf_cache_decode_mt(mut.data(), mut.size(), 2 + (int)(rng() % 7), &b);
            assert(ra == VDF_OK || ra == VDF_E_INVAL || ra == VDF_E_OOM);
            assert(ra == rb || ra == VDF_E_OOM || rb == VDF_E_OOM);
            if (ra == VDF_OK && rb == VDF_OK) {
                assert(a.n_entries == b.n_entries && a.n_ok == b.n_ok && a.n_err == b.n_err && a.n_key_differs == b.n_key_differs);
                if (a.n_ok) {
                    assert(std::memcmp(a.hashes, b.hashes, a.n_ok * 128) == 0 && std::memcmp(a.durations, b.durations, a.n_ok * 4) == 0);
                    assert(std::memcmp(a.path_offsets, b.path_offsets, (a.n_ok + 1) * 8) == 0);
                    assert(std::memcmp(a.paths, b.paths, a.path_offsets[a.n_ok]) == 0);
                }
            }
            if (ra == VDF_OK) vdf_cache_free(&a);
            if (rb == VDF_OK) vdf_cache_free(&b);
        }
        vdf_cache_free(&one);
    }
}

static void check_metadata()
{
    vdf_cache_metadata m{};
    assert(vdf_cache_metadata_new(VDF_CROPDETECT_LETTERBOX, 15.0, &m) == VDF_OK);
    char text[256];
    size_t len = 0;
    assert(vdf_cache_metadata_format(&m, text, sizeof text, &len) == VDF_OK && len > 0 && len < sizeof text);
    for (size_t cap = 0; cap <= len + 1; cap++) {  // every capacity around the text's length, on an exactly sized heap buffer
        std::vector<char> buf(cap);
        size_t l2 = 0;
        const int rc = vdf_cache_metadata_format(&m, buf.data(), cap, &l2);
        assert(rc == VDF_OK || rc == VDF_E_OVERFLOW);
        if (rc == VDF_OK) assert(l2 == len && std::memcmp(buf.data(), text, len) == 0);
    }
    vdf_cache_metadata p{};
    char err[64];
    assert(vdf_cache_metadata_parse(text, len, &p, err, sizeof err) == VDF_OK);
    assert(p.crop == m.crop && p.skip_forward_amount == m.skip_forward_amount && p.cache_version == m.cache_version);
    assert(vdf_cache_metadata_validate(&p, VDF_CROPDETECT_LETTERBOX, 15.0, err, sizeof err) == VDF_OK);
    assert(vdf_cache_metadata_validate(&p, VDF_CROPDETECT_NONE, 15.0, err, sizeof err) == VDF_E_INVAL);
    assert(vdf_cache_metadata_validate(&p, VDF_CROPDETECT_NONE, 15.0, nullptr, 0) == VDF_E_INVAL);
    const std::string good(text, len);
    for (int rep = 0; rep < 4000; rep++) {  // truncations, mutations, insertions: parsed or refused, never out of bounds (text is NOT NUL terminated)
        std::string s = good;
        const int kind = rng() % 4;
        if (kind == 0) s.resize(rng() % (s.size() + 1));
        else if (kind == 1) for (int k = 0; k < 1 + (int)(rng() % 3); k++) s[rng() % s.size()] = (char)rng();
        else if (kind == 2) s.insert(rng() % (s.size() + 1), std::string(1 + rng() % 40, ",0e9-+. x"[rng() % 9]));
        else s = std::string(rng() % 300, (char)rng());
        const std::vector<char> exact(s.begin(), s.end());
        const size_t ecap = rng() % 3 == 0 ? rng() % 8 : sizeof err;
        std::vector<char> e(ecap);
        vdf_cache_metadata q{};
        const int rc = vdf_cache_metadata_parse(exact.data(), exact.size(), &q, ecap ? e.data() : nullptr, ecap);
        assert(rc == VDF_OK || rc == VDF_E_INVAL);
        if (rc == VDF_OK) {
            std::vector<char> ev(ecap);
            (void)vdf_cache_metadata_validate(&q, (int32_t)(rng() % 3), 15.0, ecap ? ev.data() : nullptr, ecap);
        }
    }
    for (int rep = 0; rep < 2000; rep++) {  // sidecar paths
        std::string c = random_path(false);
        if (rng() % 2) c += "." + std::string(rng() % 5, 'e');
        const std::vector<char> exact(c.begin(), c.end());
        const size_t cap = rng() % 2 ? c.size() + 32 : rng() % (c.size() + 20);
        std::vector<char> buf(cap);
        size_t l2 = 0;
        const int rc = vdf_cache_metadata_path(exact.data(), exact.size(), buf.data(), cap, &l2);
        assert(rc == VDF_OK || rc == VDF_E_INVAL || rc == VDF_E_OVERFLOW);
        if (rc == VDF_OK) assert(l2 <= cap);
    }
}

static void check_path_ranks()
{
    for (int rep = 0; rep < 10; rep++) {
        const size_t n = rep == 0 ? 0 : rep == 1 ? 1 : 1 + rng() % 20000;
        std::vector<std::string> paths(n);
        std::string blob;
        std::vector<uint64_t> offs(n + 1, 0);
        for (size_t i = 0; i < n; i++) {
            paths[i] = (i && rng() % 5 == 0) ? paths[rng() % i] : random_path(rep % 2 == 0);  // duplicates share a rank
            blob += paths[i];
            offs[i + 1] = blob.size();
        }
        const std::vector<char> exact(blob.begin(), blob.end());
        std::vector<size_t> order(n);
        std::iota(order.begin(), order.end(), (size_t)0);
        auto cmp = [&](size_t a, size_t b) { return vdf_path_compare(paths[a].data(), paths[a].size(), paths[b].data(), paths[b].size()); };
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cmp(a, b) < 0; });
        std::vector<uint32_t> want(n, 0);
        uint32_t distinct = 0;
        for (size_t k = 0; k < n; k++) {
            if (k && cmp(order[k - 1], order[k]) != 0) distinct++;
            want[order[k]] = distinct;
        }
        for (int nt : {1, 2, 4, 7, 0}) {
            std::vector<uint32_t> got(n, 0xFFFFFFFFu);
            assert(vdf_path_ranks(exact.data(), offs.data(), n, got.data(), nt) == VDF_OK);
            assert(got == want);
        }
    }
}

int main()
{
    check_decode_mt();
    check_metadata();
    check_path_ranks();
    std::printf("decoder fell back to one range %llu times\n", vdf_cache_decode_fallbacks());
    std::puts("sanitize mt ok");
    return 0;
}
