// On-disk hash cache of the reference app  <->  SoA arrays.  Host only, no GPU call.
//
// The app stores  HashMap<PathBuf, MtimeCacheEntry<Result<VideoHash, Error>>>  with
// bincode::serde::encode_into_std_write(.., bincode::config::standard())
//   vid_dup_finder_app/src/video_hash_filesystem_cache/generic_filesystem_cache/base_fs_cache.rs:26,106-118,192-204
//   .../processing_fs_cache.rs:23-27 (MtimeCacheEntry { cache_mtime: SystemTime, value })
//   .../generic_cache_if.rs:23 (T = Result<VideoHash, Error>)
//   vid_dup_finder_lib/src/video_hashing/video_hash.rs:26-32 (VideoHash { hash: [usize;16], src_path, duration })
//   vid_dup_finder_lib/src/video_hashing/mod.rs:17-28 (Error { NotVideo, VidProc(String), NotEnoughFrames })
// bincode 2 "standard" = little endian + varint: u < 251 -> 1 byte; 251 + u16; 252 + u32; 253 + u64.  serde shapes:
// map = len + (key, value)*; PathBuf / String = len + UTF-8 bytes; SystemTime = { secs_since_epoch: u64,
// nanos_since_epoch: u32 }; Result = variant index (u32: 0 Ok, 1 Err) + payload; [usize;16] = 16 values, no length;
// unit / newtype enum variants = index (+ payload).
// The decoder goes straight to the arrays the search ABI takes (hashes, durations, path blob): no per-entry objects.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "../../include/vdf.h"

namespace {

struct Reader {
    const uint8_t *p, *end;
    bool ok = true;
    bool need(size_t n) { if ((size_t)(end - p) < n) { ok = false; return false; } return true; }
    uint64_t varint()
    {
        if (!need(1)) return 0;
        const uint8_t b = *p++;
        if (b < 251) return b;
        const int n = b == 251 ? 2 : b == 252 ? 4 : b == 253 ? 8 : -1;
        if (n < 0 || !need((size_t)n)) { ok = false; return 0; }
        uint64_t v = 0;
        std::memcpy(&v, p, (size_t)n);  // little endian on both sides
        p += n;
        return v;
    }
    // the 16 words of a hash: random bits, so nearly every word is the 9-byte form - one bounds check for the common case
    bool hash_words(uint64_t *w)
    {
        if ((size_t)(end - p) >= 9 * VDF_HASH_WORDS) {
            const uint8_t *q = p;
            bool all9 = true;
            for (int i = 0; i < VDF_HASH_WORDS; i++) all9 = all9 && q[9 * i] == 253;
            if (all9) {
                for (int i = 0; i < VDF_HASH_WORDS; i++) std::memcpy(&w[i], q + 9 * i + 1, 8);
                p += 9 * VDF_HASH_WORDS;
                return true;
            }
        }
        for (int i = 0; i < VDF_HASH_WORDS; i++) w[i] = varint();
        return ok;
    }
    bool str(const uint8_t **s, uint64_t *len)
    {
        *len = varint();
        if (!ok || !need((size_t)*len)) { ok = false; return false; }
        *s = p;
        p += *len;
        return true;
    }
};

// output bytes appended through a raw cursor (the encoder reserves the exact upper bound first)
inline uint8_t *put_varint(uint8_t *o, uint64_t v)
{
    if (v < 251) { *o++ = (uint8_t)v; return o; }
    int n;
    if (v < (1ull << 16)) { *o++ = 251; n = 2; }
    else if (v < (1ull << 32)) { *o++ = 252; n = 4; }
    else { *o++ = 253; n = 8; }
    std::memcpy(o, &v, (size_t)n);
    return o + n;
}

inline uint8_t *put_str(uint8_t *o, const char *s, uint64_t len)
{
    o = put_varint(o, len);
    std::memcpy(o, s, (size_t)len);
    return o + len;
}

// The entries of [r.p, stop): counted (COPY = false) or written to the arrays from position (ok_base, blob_base) on.  Returns false on
// malformed input or when an entry runs past `stop` (a range handed to a worker thread must end exactly on an entry boundary).
struct RangeCounts { uint64_t entries = 0, n_ok = 0, n_err = 0, n_key_differs = 0, blob = 0; };

template <bool COPY>
static bool run_entries(Reader &r, const uint8_t *stop, uint64_t max_entries, RangeCounts &c, vdf_cache_soa *out, uint64_t ok_base,
                        uint64_t blob_base, uint64_t cap)
{
    uint64_t n_ok = 0, blob_len = 0;
    while (r.p < stop && c.entries < max_entries) {
        const uint8_t *key; uint64_t klen;
        if (!r.str(&key, &klen)) return false;                  // map key: PathBuf
        const uint64_t secs = r.varint();                       // MtimeCacheEntry.cache_mtime
        const uint64_t nanos = r.varint();
        const uint64_t variant = r.varint();                    // Result<VideoHash, Error>
        if (!r.ok || nanos > 0xFFFFFFFFull) return false;
        if (variant == 0) {
            if (ok_base + n_ok >= cap) return false;            // more Ok entries than the file can hold
            uint64_t scratch[VDF_HASH_WORDS];
            if (!r.hash_words(COPY ? out->hashes + (ok_base + n_ok) * VDF_HASH_WORDS : scratch)) return false;
            const uint8_t *sp; uint64_t slen;
            if (!r.str(&sp, &slen)) return false;               // VideoHash.src_path
            const uint64_t dur = r.varint();
            if (!r.ok || dur > 0xFFFFFFFFull) return false;
            if (COPY) {
                const uint64_t at = ok_base + n_ok;
                out->durations[at] = (uint32_t)dur;
                std::memcpy(out->paths + blob_base + blob_len, sp, (size_t)slen);
                out->path_offsets[at + 1] = blob_base + blob_len + slen;
                out->mtime_secs[at] = secs;
                out->mtime_nanos[at] = (uint32_t)nanos;
            } else if (slen != klen || std::memcmp(sp, key, (size_t)slen) != 0) {
                c.n_key_differs++;
            }
            blob_len += slen;
            n_ok++;
        } else if (variant == 1) {
            const uint64_t ev = r.varint();                     // Error: 0 NotVideo, 1 VidProc(String), 2 NotEnoughFrames
            if (!r.ok || ev > 2) return false;
            if (ev == 1) { const uint8_t *m; uint64_t ml; if (!r.str(&m, &ml)) return false; }
            c.n_err++;
        } else {
            return false;
        }
        c.entries++;
    }
    c.n_ok = n_ok;
    c.blob = blob_len;
    return r.p <= stop;
}

// A position inside [from, end) where an entry STARTS, found without parsing from the front: the 16 words of an Ok entry's hash are
// (for all but ~2^-32 of real hashes) sixteen 9-byte varints - the byte 253 at stride 9, behind the one-byte variant 0 - a pattern that
// does not occur by chance; from there the rest of that entry (src_path, duration) leads to the start of the next one.  nullptr = none
// found.  A false positive (the pattern inside a path, say) is caught by the caller: the previous range then does not end on it.
static const uint8_t *find_entry_start(const uint8_t *from, const uint8_t *end)
{
    for (const uint8_t *p = from + 1; p + 9 * VDF_HASH_WORDS <= end; p++) {
        if (*p != 253 || p[-1] != 0) continue;
        bool all = true;
        for (int i = 1; i < VDF_HASH_WORDS && all; i++) all = p[9 * i] == 253;
        if (!all) continue;
        Reader r{p + 9 * VDF_HASH_WORDS, end};
        const uint8_t *sp; uint64_t slen;
        if (!r.str(&sp, &slen)) continue;
        (void)r.varint();
        if (!r.ok) continue;
        return r.p;
    }
    return nullptr;
}

std::atomic<unsigned long long> g_decode_fallbacks{0};

// The decoder's output arrays: first touched by the copy pass, i.e. one page fault per 4 KB of a few hundred MB (10 M entries: 1.5 GB)
// - which costs more than parsing the bytes.  Large arrays are 2 MB-aligned and ask for transparent huge pages (the usual system setting
// is "madvise"); freed with free() like the small ones.
void *big_alloc(size_t bytes)
{
    constexpr size_t kHuge = 2u << 20;
    if (bytes < 4 * kHuge) return std::malloc(std::max<size_t>(bytes, 1));
    void *p = nullptr;
    const size_t rounded = (bytes + kHuge - 1) / kHuge * kHuge;
    if (posix_memalign(&p, kHuge, rounded) != 0) return std::malloc(bytes);
    (void)madvise(p, rounded, MADV_HUGEPAGE);  // advisory: failure changes nothing but the fault count
    return p;
}

}  // namespace

extern "C" {

unsigned long long vdf_cache_decode_fallbacks(void) { return g_decode_fallbacks.load(); }


void vdf_cache_free(vdf_cache_soa *c)
{
    if (!c) return;
    std::free(c->hashes); std::free(c->durations); std::free(c->path_offsets); std::free(c->paths);
    std::free(c->mtime_secs); std::free(c->mtime_nanos);
    std::memset(c, 0, sizeof *c);
}

int vdf_cache_decode_mt(const uint8_t *data, size_t len, int n_threads, vdf_cache_soa *out)
{
    if (!out || (len && !data)) return VDF_E_INVAL;
    std::memset(out, 0, sizeof *out);
    Reader r{data, data + len};
    const uint64_t n = r.varint();
    if (!r.ok) return VDF_E_INVAL;
    // Every array is allocated once at its upper bound and filled in place: an Ok entry takes at least 22 bytes of input (two
    // one-byte strings, mtime, variant, 16 words, duration), so a hostile count cannot make the allocations larger than the file.
    const uint64_t cap = std::min<uint64_t>(n, len / 22 + 1);
    out->hashes = (uint64_t *)big_alloc((size_t)cap * VDF_HASH_WORDS * sizeof(uint64_t));
    out->durations = (uint32_t *)big_alloc((size_t)cap * sizeof(uint32_t));
    out->path_offsets = (uint64_t *)big_alloc(((size_t)cap + 1) * sizeof(uint64_t));
    out->paths = (char *)big_alloc(std::max<size_t>(len, 1));  // the paths are a subset of the file's bytes
    out->mtime_secs = (uint64_t *)big_alloc((size_t)cap * sizeof(uint64_t));
    out->mtime_nanos = (uint32_t *)big_alloc((size_t)cap * sizeof(uint32_t));
    if (!out->hashes || !out->durations || !out->path_offsets || !out->paths || !out->mtime_secs || !out->mtime_nanos) {
        vdf_cache_free(out);
        return VDF_E_OOM;
    }
    auto bad = [&]() { vdf_cache_free(out); return (int)VDF_E_INVAL; };
    out->path_offsets[0] = 0;
    const uint8_t *body = r.p, *end = data + len;
    // ---- ranges: one for a small file; else cut where find_entry_start resynchronises (a 1 M-entry cache decodes in 0.11 s on one
    // thread - 1.9 GB/s - so a 10 M-entry cache would spend 1.2 s here in front of a 10 s search)
    // Automatic thread count (n_threads = 0): one thread per 8 MB, at most 32 and at most the host's - a 2 MB cache is not worth a thread
    // start, and beyond ~32 ranges the page faults of the output arrays, not the parsing, set the pace.
    unsigned nt = n_threads > 0 ? (unsigned)n_threads : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    const size_t kMinRange = n_threads > 0 ? 4096 : (8u << 20);
    nt = (unsigned)std::min<size_t>(nt, std::max<size_t>((size_t)(end - body) / kMinRange, 1));
    std::vector<const uint8_t *> cut;
    std::vector<RangeCounts> cnt;
    std::vector<uint8_t> ok;
    try {
        cut.push_back(body);
        const size_t range_len = (size_t)(end - body) / nt;
        for (unsigned k = 1; k < nt; k++) {
            const uint8_t *guess = body + range_len * k;
            if (guess <= cut.back()) continue;
            // a scan that finds nothing within one range length (a stretch of Err entries, hashes whose words are not all 9-byte varints)
            // gives up: later guesses would only scan a suffix of the same bytes again
            const uint8_t *q = find_entry_start(guess, std::min(end, guess + range_len + 4096));
            if (!q) break;
            if (q > cut.back() && q < end) cut.push_back(q);
        }
        cut.push_back(end);
        cnt.resize(cut.size() - 1);
        ok.assign(cut.size() - 1, 0);
    } catch (const std::bad_alloc &) {
        vdf_cache_free(out);
        return VDF_E_OOM;
    }
    const size_t n_ranges = cut.size() - 1;
    // Threads may fail to start (a pids cgroup, RLIMIT_NPROC): what was started is joined, the ranges that got no thread run here -
    // nothing may be thrown through the C ABI.
    auto for_ranges = [&](const std::function<void(size_t)> &f) {
        if (n_ranges == 1) { f(0); return; }
        std::vector<std::thread> th;
        size_t started = 0;
        try {
            th.reserve(n_ranges);
            for (; started < n_ranges; started++) th.emplace_back(f, started);
        } catch (...) {
        }
        for (size_t k = started; k < n_ranges; k++) f(k);
        for (auto &t : th) t.join();
    };
    // pass 1: every range is parsed and counted; it must end exactly where the next one begins
    for_ranges([&](size_t k) {
        Reader rr{cut[k], end};
        ok[k] = run_entries<false>(rr, cut[k + 1], n, cnt[k], out, 0, 0, cap) && rr.p == cut[k + 1];
    });
    bool all_ok = true;
    uint64_t total = 0;
    for (size_t k = 0; k < n_ranges; k++) { all_ok = all_ok && ok[k]; total += cnt[k].entries; }
    if (n_ranges > 1 && (!all_ok || total != n)) {
        // a cut was not an entry boundary after all (or the file is malformed): one range, front to back - the answer does not
        // depend on the speculation
        vdf_cache_free(out);
        g_decode_fallbacks++;
        return vdf_cache_decode_mt(data, len, 1, out);
    }
    if (!all_ok || total != n) return bad();  // malformed, trailing bytes, or fewer entries than the count says
    // pass 2: fill the arrays, every range from its own position
    std::vector<uint64_t> ok_base, blob_base;
    try {
        ok_base.assign(n_ranges + 1, 0);
        blob_base.assign(n_ranges + 1, 0);
    } catch (const std::bad_alloc &) {
        vdf_cache_free(out);
        return VDF_E_OOM;
    }
    for (size_t k = 0; k < n_ranges; k++) { ok_base[k + 1] = ok_base[k] + cnt[k].n_ok; blob_base[k + 1] = blob_base[k] + cnt[k].blob; }
    if (ok_base[n_ranges] > cap) return bad();
    for_ranges([&](size_t k) {
        Reader rr{cut[k], end};
        RangeCounts c2;
        ok[k] = run_entries<true>(rr, cut[k + 1], n, c2, out, ok_base[k], blob_base[k], cap);
    });
    for (size_t k = 0; k < n_ranges; k++)
        if (!ok[k]) return bad();
    out->n_entries = n;
    out->n_ok = ok_base[n_ranges];
    for (size_t k = 0; k < n_ranges; k++) { out->n_err += cnt[k].n_err; out->n_key_differs += cnt[k].n_key_differs; }
    return VDF_OK;
}

int vdf_cache_decode(const uint8_t *data, size_t len, vdf_cache_soa *out) { return vdf_cache_decode_mt(data, len, 0, out); }

int vdf_cache_encode(uint64_t n, const uint64_t *hashes, const uint32_t *durations, const uint64_t *path_offsets,
                     const char *paths, const uint64_t *mtime_secs, const uint32_t *mtime_nanos, uint8_t **out_data,
                     size_t *out_len)
{
    if (!out_data || !out_len || (n && (!hashes || !durations || !path_offsets || !paths))) return VDF_E_INVAL;
    // exact upper bound: count + per entry 2 x (9 + path) + 9 + 5 + 1 + 16 x 9 + 5
    const uint64_t path_bytes = n ? path_offsets[n] - path_offsets[0] : 0;
    const size_t bound = 9 + (size_t)n * (2 * 9 + 9 + 5 + 1 + 9 * VDF_HASH_WORDS + 5) + 2 * (size_t)path_bytes;
    uint8_t *buf = (uint8_t *)std::malloc(bound);
    if (!buf) return VDF_E_OOM;
    uint8_t *o = put_varint(buf, n);
    for (uint64_t e = 0; e < n; e++) {
        const char *s = paths + path_offsets[e];
        const uint64_t slen = path_offsets[e + 1] - path_offsets[e];
        o = put_str(o, s, slen);                                        // key
        o = put_varint(o, mtime_secs ? mtime_secs[e] : 0);
        o = put_varint(o, mtime_nanos ? mtime_nanos[e] : 0);
        o = put_varint(o, 0);                                           // Ok
        for (int i = 0; i < VDF_HASH_WORDS; i++) o = put_varint(o, hashes[e * VDF_HASH_WORDS + i]);
        o = put_str(o, s, slen);                                        // VideoHash.src_path
        o = put_varint(o, durations[e]);
    }
    *out_data = buf;
    *out_len = (size_t)(o - buf);
    return VDF_OK;
}

void vdf_buffer_free(void *p) { std::free(p); }

}  // extern "C"
