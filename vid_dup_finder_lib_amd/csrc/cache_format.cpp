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
#include <cstdlib>
#include <cstring>

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

}  // namespace

extern "C" {

void vdf_cache_free(vdf_cache_soa *c)
{
    if (!c) return;
    std::free(c->hashes); std::free(c->durations); std::free(c->path_offsets); std::free(c->paths);
    std::free(c->mtime_secs); std::free(c->mtime_nanos);
    std::memset(c, 0, sizeof *c);
}

int vdf_cache_decode(const uint8_t *data, size_t len, vdf_cache_soa *out)
{
    if (!out || (len && !data)) return VDF_E_INVAL;
    std::memset(out, 0, sizeof *out);
    Reader r{data, data + len};
    const uint64_t n = r.varint();
    if (!r.ok) return VDF_E_INVAL;
    // Every array is allocated once at its upper bound and filled in place: an Ok entry takes at least 22 bytes of input (two
    // one-byte strings, mtime, variant, 16 words, duration), so a hostile count cannot make the allocations larger than the file.
    const uint64_t cap = std::min<uint64_t>(n, len / 22 + 1);
    out->hashes = (uint64_t *)std::malloc((size_t)cap * VDF_HASH_WORDS * sizeof(uint64_t));
    out->durations = (uint32_t *)std::malloc((size_t)cap * sizeof(uint32_t));
    out->path_offsets = (uint64_t *)std::malloc(((size_t)cap + 1) * sizeof(uint64_t));
    out->paths = (char *)std::malloc(std::max<size_t>(len, 1));  // the paths are a subset of the file's bytes
    out->mtime_secs = (uint64_t *)std::malloc((size_t)cap * sizeof(uint64_t));
    out->mtime_nanos = (uint32_t *)std::malloc((size_t)cap * sizeof(uint32_t));
    if (!out->hashes || !out->durations || !out->path_offsets || !out->paths || !out->mtime_secs || !out->mtime_nanos) {
        vdf_cache_free(out);
        return VDF_E_OOM;
    }
    auto bad = [&]() { vdf_cache_free(out); return (int)VDF_E_INVAL; };
    uint64_t n_ok = 0, n_err = 0, n_key_differs = 0, blob_len = 0;
    out->path_offsets[0] = 0;
    for (uint64_t e = 0; e < n; e++) {
        const uint8_t *key; uint64_t klen;
        if (!r.str(&key, &klen)) return bad();                  // map key: PathBuf
        const uint64_t secs = r.varint();                       // MtimeCacheEntry.cache_mtime
        const uint64_t nanos = r.varint();
        const uint64_t variant = r.varint();                    // Result<VideoHash, Error>
        if (!r.ok || nanos > 0xFFFFFFFFull) return bad();
        if (variant == 0) {
            if (n_ok >= cap) return bad();                      // more Ok entries than the file can hold
            if (!r.hash_words(out->hashes + n_ok * VDF_HASH_WORDS)) return bad();
            const uint8_t *sp; uint64_t slen;
            if (!r.str(&sp, &slen)) return bad();               // VideoHash.src_path
            const uint64_t dur = r.varint();
            if (!r.ok || dur > 0xFFFFFFFFull) return bad();
            out->durations[n_ok] = (uint32_t)dur;
            std::memcpy(out->paths + blob_len, sp, (size_t)slen);
            blob_len += slen;
            out->path_offsets[n_ok + 1] = blob_len;
            out->mtime_secs[n_ok] = secs;
            out->mtime_nanos[n_ok] = (uint32_t)nanos;
            if (slen != klen || std::memcmp(sp, key, (size_t)slen) != 0) n_key_differs++;
            n_ok++;
        } else if (variant == 1) {
            const uint64_t ev = r.varint();                     // Error: 0 NotVideo, 1 VidProc(String), 2 NotEnoughFrames
            if (!r.ok || ev > 2) return bad();
            if (ev == 1) { const uint8_t *m; uint64_t ml; if (!r.str(&m, &ml)) return bad(); }
            n_err++;
        } else {
            return bad();
        }
    }
    if (r.p != r.end) return bad();  // trailing bytes
    out->n_entries = n;
    out->n_ok = n_ok;
    out->n_err = n_err;
    out->n_key_differs = n_key_differs;
    return VDF_OK;
}

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
