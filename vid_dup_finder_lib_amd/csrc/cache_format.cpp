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
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

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
        for (int i = 0; i < n; i++) v |= (uint64_t)p[i] << (8 * i);
        p += n;
        return v;
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

void put_varint(std::vector<uint8_t> &o, uint64_t v)
{
    if (v < 251) { o.push_back((uint8_t)v); return; }
    int n;
    if (v < (1ull << 16)) { o.push_back(251); n = 2; }
    else if (v < (1ull << 32)) { o.push_back(252); n = 4; }
    else { o.push_back(253); n = 8; }
    for (int i = 0; i < n; i++) o.push_back((uint8_t)(v >> (8 * i)));
}

void put_str(std::vector<uint8_t> &o, const char *s, uint64_t len)
{
    put_varint(o, len);
    o.insert(o.end(), (const uint8_t *)s, (const uint8_t *)s + len);
}

template <class T> T *dup_array(const std::vector<T> &v)
{
    T *p = (T *)std::malloc(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
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
    std::vector<uint64_t> hashes, offs(1, 0), msecs;
    std::vector<uint32_t> durs, mnanos;
    std::vector<char> blob;
    uint64_t n_err = 0, n_key_differs = 0;
    for (uint64_t e = 0; e < n; e++) {
        const uint8_t *key; uint64_t klen;
        if (!r.str(&key, &klen)) return VDF_E_INVAL;            // map key: PathBuf
        const uint64_t secs = r.varint();                       // MtimeCacheEntry.cache_mtime
        const uint64_t nanos = r.varint();
        const uint64_t variant = r.varint();                    // Result<VideoHash, Error>
        if (!r.ok || nanos > 0xFFFFFFFFull) return VDF_E_INVAL;
        if (variant == 0) {
            uint64_t words[VDF_HASH_WORDS];
            for (int i = 0; i < VDF_HASH_WORDS; i++) words[i] = r.varint();
            const uint8_t *sp; uint64_t slen;
            if (!r.ok || !r.str(&sp, &slen)) return VDF_E_INVAL;  // VideoHash.src_path
            const uint64_t dur = r.varint();
            if (!r.ok || dur > 0xFFFFFFFFull) return VDF_E_INVAL;
            hashes.insert(hashes.end(), words, words + VDF_HASH_WORDS);
            durs.push_back((uint32_t)dur);
            blob.insert(blob.end(), (const char *)sp, (const char *)sp + slen);
            offs.push_back(blob.size());
            msecs.push_back(secs);
            mnanos.push_back((uint32_t)nanos);
            if (slen != klen || std::memcmp(sp, key, slen) != 0) n_key_differs++;
        } else if (variant == 1) {
            const uint64_t ev = r.varint();                     // Error: 0 NotVideo, 1 VidProc(String), 2 NotEnoughFrames
            if (!r.ok || ev > 2) return VDF_E_INVAL;
            if (ev == 1) { const uint8_t *m; uint64_t ml; if (!r.str(&m, &ml)) return VDF_E_INVAL; }
            n_err++;
        } else {
            return VDF_E_INVAL;
        }
    }
    if (r.p != r.end) return VDF_E_INVAL;  // trailing bytes
    out->n_entries = n;
    out->n_ok = durs.size();
    out->n_err = n_err;
    out->n_key_differs = n_key_differs;
    out->hashes = dup_array(hashes);
    out->durations = dup_array(durs);
    out->path_offsets = dup_array(offs);
    out->paths = dup_array(blob);
    out->mtime_secs = dup_array(msecs);
    out->mtime_nanos = dup_array(mnanos);
    if (!out->hashes || !out->durations || !out->path_offsets || !out->paths || !out->mtime_secs || !out->mtime_nanos) {
        vdf_cache_free(out);
        return VDF_E_OOM;
    }
    return VDF_OK;
}

int vdf_cache_encode(uint64_t n, const uint64_t *hashes, const uint32_t *durations, const uint64_t *path_offsets,
                     const char *paths, const uint64_t *mtime_secs, const uint32_t *mtime_nanos, uint8_t **out_data,
                     size_t *out_len)
{
    if (!out_data || !out_len || (n && (!hashes || !durations || !path_offsets || !paths))) return VDF_E_INVAL;
    std::vector<uint8_t> o;
    o.reserve((size_t)n * 200 + 16);
    put_varint(o, n);
    for (uint64_t e = 0; e < n; e++) {
        const char *s = paths + path_offsets[e];
        const uint64_t slen = path_offsets[e + 1] - path_offsets[e];
        put_str(o, s, slen);                                        // key
        put_varint(o, mtime_secs ? mtime_secs[e] : 0);
        put_varint(o, mtime_nanos ? mtime_nanos[e] : 0);
        put_varint(o, 0);                                           // Ok
        for (int i = 0; i < VDF_HASH_WORDS; i++) put_varint(o, hashes[e * VDF_HASH_WORDS + i]);
        put_str(o, s, slen);                                        // VideoHash.src_path
        put_varint(o, durations[e]);
    }
    uint8_t *buf = (uint8_t *)std::malloc(std::max<size_t>(o.size(), 1));
    if (!buf) return VDF_E_OOM;
    std::memcpy(buf, o.data(), o.size());
    *out_data = buf;
    *out_len = o.size();
    return VDF_OK;
}

void vdf_buffer_free(void *p) { std::free(p); }

}  // extern "C"
