// Rust's `PathBuf: Ord` for a whole cache at once (SURVEY.md section 8f, row N1 -> a7).  Host only, no GPU call.
//
// Search::sort keys on (duration, src_path.to_owned())  (vid_dup_finder_lib/src/video_hashing/search_algorithm.rs:55-61) and
// PathBuf orders by COMPONENTS, not bytes: std::path::Path::cmp = Iterator::cmp over components(), where on Unix
//   components() = [RootDir if the path starts with '/'] or [CurDir if it is "." or starts with "./"], then one
//   Normal(bytes) per non-empty piece between '/' - except "." pieces, which are skipped, and "..", which is ParentDir;
//   Component derives Ord in declaration order: Prefix < RootDir < CurDir < ParentDir < Normal(bytes by memcmp, shorter first);
//   a path that is a strict component prefix of another is the smaller one.
// So "a/b" < "a.b" ('/' ends the component "a"; as bytes 0x2F > 0x2E would say the opposite), "a//b" == "a/b" == "a/./b".
// vdf_path_ranks turns the path blob of a decoded cache (vdf_cache_soa: blob + offsets) into one u32 per entry - its
// position in that order, equal paths sharing a rank - which is what vdf_sort_order_device takes as the second half of its key:
// no per-entry objects, no per-entry allocation, all host threads (sample sort over an index array: the comparisons of the
// partition pass run against ~4 k splitter paths that stay in cache, the buckets are sorted independently).
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "../../include/vdf.h"

namespace {

// One path's components, produced on the fly.
struct CompIter {
    const unsigned char *p, *end;
    int state;  // 0: before the leading RootDir / CurDir, 1: body
    CompIter(const char *s, size_t n) : p(reinterpret_cast<const unsigned char *>(s)), end(p + n), state(0) {}
    // kind: 0 = end of path, 1 = RootDir, 2 = CurDir, 3 = ParentDir, 4 = Normal [*b, *e)
    int next(const unsigned char **b, const unsigned char **e)
    {
        if (state == 0) {
            state = 1;
            if (p < end && *p == '/') return 1;
            if (p < end && *p == '.' && (p + 1 == end || p[1] == '/')) { p += 1; return 2; }
        }
        for (;;) {
            while (p < end && *p == '/') p++;
            if (p == end) return 0;
            const unsigned char *s = p;
            while (p < end && *p != '/') p++;
            const size_t len = (size_t)(p - s);
            if (len == 1 && s[0] == '.') continue;  // an inner "." is not a component
            if (len == 2 && s[0] == '.' && s[1] == '.') return 3;
            *b = s; *e = p;
            return 4;
        }
    }
};

int path_cmp(const char *a, size_t la, const char *b, size_t lb)
{
    // The paths of one cache share long prefixes ("/mnt/videos/..."): identical bytes decide nothing, so skip to the first
    // difference, eight bytes at a time, and back up to the start of the component it falls in (std's compare_components does
    // the same).  From there both iterators are in the body state; a difference in the very first component, or one path being a
    // byte prefix of the other, takes the walk from the start.
    size_t m = 0;
    const size_t lmin = std::min(la, lb);
    while (m + 8 <= lmin) {
        uint64_t x, y;
        std::memcpy(&x, a + m, 8);
        std::memcpy(&y, b + m, 8);
        if (x != y) { m += (size_t)(__builtin_ctzll(x ^ y) >> 3); break; }
        m += 8;
    }
    while (m < lmin && a[m] == b[m]) m++;
    CompIter ia(a, la), ib(b, lb);
    if (m < lmin) {
        size_t s = m;
        while (s > 0 && a[s - 1] != '/') s--;
        if (s > 0) {
            ia.p += s; ib.p += s;
            ia.state = ib.state = 1;
        }
    }
    for (;;) {
        const unsigned char *ab = nullptr, *ae = nullptr, *bb = nullptr, *be = nullptr;
        const int ka = ia.next(&ab, &ae), kb = ib.next(&bb, &be);
        if (ka != kb) return ka < kb ? -1 : 1;  // end of path (0) sorts first: the shorter component list is the smaller
        if (ka == 0) return 0;
        if (ka == 4) {
            const size_t na = (size_t)(ae - ab), nb = (size_t)(be - bb);
            const int c = std::memcmp(ab, bb, std::min(na, nb));
            if (c) return c < 0 ? -1 : 1;
            if (na != nb) return na < nb ? -1 : 1;
        }
    }
}

// A PLAIN path is one whose bytes spell its components and nothing else: no empty component ("//", a trailing '/' - the root "/"
// itself excepted), no "." or ".." component, no NUL byte.  That is what a directory walk produces, i.e. every path of a real cache.
// For two plain paths Path::cmp is the byte order in which '/' sorts below every other byte (a component that ends is smaller than
// one that goes on; RootDir - a leading '/' - is smaller than any Normal component; a path that ends is smaller than one that goes
// on): one pass to the first differing byte decides, no component iteration, no backing up.
bool is_plain(const char *s, size_t n)
{
    if (n == 0 || (n == 1 && s[0] == '/')) return true;
    if (s[n - 1] == '/') return false;
    size_t len = 0, dots = 0;  // of the component being read
    bool bad = false;
    for (size_t i = s[0] == '/' ? 1 : 0; i < n; i++) {
        const char c = s[i];
        if (c == '/') {
            bad = bad || len == 0 || (dots == len && len <= 2);
            len = dots = 0;
        } else {
            bad = bad || c == 0;
            len++;
            dots += c == '.';
        }
    }
    return !(bad || (dots == len && len <= 2));  // (the last component is not empty: the path does not end in '/')
}

// both plain; the first `skip` bytes are known to be equal
inline int plain_cmp(const char *a, size_t la, const char *b, size_t lb, size_t skip)
{
    size_t m = skip;
    const size_t lmin = std::min(la, lb);
    while (m + 8 <= lmin) {
        uint64_t x, y;
        std::memcpy(&x, a + m, 8);
        std::memcpy(&y, b + m, 8);
        if (x != y) { m += (size_t)(__builtin_ctzll(x ^ y) >> 3); break; }
        m += 8;
    }
    while (m < lmin && a[m] == b[m]) m++;
    if (m >= lmin) return la < lb ? -1 : la > lb ? 1 : 0;
    const unsigned x = (unsigned char)a[m] == '/' ? 0u : (unsigned char)a[m], y = (unsigned char)b[m] == '/' ? 0u : (unsigned char)b[m];
    return x < y ? -1 : 1;
}

inline size_t common_prefix(const char *a, size_t la, const char *b, size_t lb)
{
    size_t m = 0;
    const size_t lmin = std::min(la, lb);
    while (m < lmin && a[m] == b[m]) m++;
    return m;
}

struct Blob {
    const char *paths;
    const uint64_t *off;
    bool plain = false;  // every path of the blob is plain: plain_cmp decides
    size_t shared = 0;   // plain: leading bytes all paths of the blob have in common
    int cmp(uint32_t i, uint32_t j, size_t skip = 0) const
    {
        const char *a = paths + off[i], *b = paths + off[j];
        const size_t la = (size_t)(off[i + 1] - off[i]), lb = (size_t)(off[j + 1] - off[j]);
        return plain ? plain_cmp(a, la, b, lb, std::max(skip, shared)) : path_cmp(a, la, b, lb);
    }
    // PLAIN paths: the 8 bytes from `skip` on as one big-endian word in the order's alphabet ('/' -> 0; beyond the end -> 0).  Monotone:
    // key8(i) < key8(j) implies path i < path j; equal words decide nothing (a path that ends and one that goes on with '/' tie).
    uint64_t key8(uint32_t i, size_t skip) const
    {
        const unsigned char *a = reinterpret_cast<const unsigned char *>(paths + off[i]);
        const size_t la = (size_t)(off[i + 1] - off[i]);
        uint64_t k = 0;
        for (size_t q = 0; q < 8; q++) {
            const unsigned c = skip + q < la ? a[skip + q] : 0u;
            k = (k << 8) | (c == '/' ? 0u : c);
        }
        return k;
    }
    size_t lcp(uint32_t i, uint32_t j) const  // bytes two PLAIN paths share (everything ordered between them shares them too)
    {
        return common_prefix(paths + off[i], (size_t)(off[i + 1] - off[i]), paths + off[j], (size_t)(off[j + 1] - off[j]));
    }
};

// Sorting a group of PLAIN paths that share their first `depth` bytes without comparing paths: find what the whole group shares
// (one pass), let the next 8 bytes ride along with each index as one integer (Blob::key8), sort the integers, and descend into the
// runs of equal words 8 bytes further on.  Directory-walk paths differ late ("/lib/show_0412/season_03/clip_000012" + 2 digits), so a
// comparison sort spends its time re-reading what all its operands share; here every byte of a path is looked at about once.
struct KeyIdx { uint64_t key; uint32_t idx; };

void keyed_sort(const Blob &B, KeyIdx *v, size_t n, size_t depth)
{
    auto by_path = [&](const KeyIdx &x, const KeyIdx &y) { const int c = B.cmp(x.idx, y.idx, depth); return c ? c < 0 : x.idx < y.idx; };
    while (n > 1) {
        if (n <= 8) { std::sort(v, v + n, by_path); return; }
        // what the group shares beyond `depth`, measured against its first member
        const char *p0 = B.paths + B.off[v[0].idx];
        const size_t l0 = (size_t)(B.off[v[0].idx + 1] - B.off[v[0].idx]);
        size_t g = l0;
        for (size_t i = 1; i < n && g > depth; i++) {
            const char *q = B.paths + B.off[v[i].idx];
            const size_t lq = std::min<size_t>((size_t)(B.off[v[i].idx + 1] - B.off[v[i].idx]), g);
            size_t m = depth;
            while (m < lq && p0[m] == q[m]) m++;
            g = m;
        }
        depth = std::max(depth, std::min(g, l0));
        bool ends = false;  // a member that ends inside the window: its word ties with one that goes on with '/', the comparator decides
        for (size_t i = 0; i < n; i++) {
            v[i].key = B.key8(v[i].idx, depth);
            ends = ends || (size_t)(B.off[v[i].idx + 1] - B.off[v[i].idx]) < depth + 8;
        }
        if (ends) {
            std::sort(v, v + n, [&](const KeyIdx &x, const KeyIdx &y) { return x.key != y.key ? x.key < y.key : by_path(x, y); });
            return;
        }
        std::sort(v, v + n, [](const KeyIdx &x, const KeyIdx &y) { return x.key < y.key; });
        // runs of equal words share depth + 8 bytes: all but the last are sorted by recursion, the last by this loop
        size_t last_b = 0, last_n = 0;
        for (size_t b = 0; b < n;) {
            size_t e = b + 1;
            while (e < n && v[e].key == v[b].key) e++;
            if (e - b > 1) {
                if (last_n > 1) keyed_sort(B, v + last_b, last_n, depth + 8);
                last_b = b; last_n = e - b;
            }
            b = e;
        }
        if (last_n <= 1) return;
        v += last_b; n = last_n; depth += 8;
    }
}

// Worker threads that live for one vdf_path_ranks call and run its phases one after the other (threads started per phase spend
// the short phases being placed by the scheduler: 8 threads gave the partition pass no speed-up at all on the build container).
class Pool {
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_job_, cv_done_;
    const std::function<void(size_t)> *f_ = nullptr;
    size_t n_items_ = 0, generation_ = 0, busy_ = 0;
    std::atomic<size_t> next_{0};
    bool quit_ = false;

    void work()
    {
        size_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_job_.wait(lk, [&] { return quit_ || generation_ != seen; });
                if (quit_) return;
                seen = generation_;
            }
            for (size_t i; (i = next_.fetch_add(1)) < n_items_;) (*f_)(i);
            std::lock_guard<std::mutex> lk(m_);
            if (--busy_ == 0) cv_done_.notify_all();
        }
    }

public:
    explicit Pool(unsigned n_threads)
    {
        try {
            for (unsigned t = 1; t < n_threads; t++) th_.emplace_back([this] { work(); });  // the caller is worker 0
        } catch (...) {  // a pids cgroup / RLIMIT_NPROC: the threads that did start (possibly none) and the caller do the work
        }
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_job_.notify_all();
        for (auto &t : th_) t.join();
    }
    void run(size_t n_items, const std::function<void(size_t)> &f)
    {  // f(item) for item in [0, n_items), items handed out one at a time; returns when all are done
        if (th_.empty() || n_items <= 1) { for (size_t i = 0; i < n_items; i++) f(i); return; }
        {
            std::lock_guard<std::mutex> lk(m_);
            f_ = &f; n_items_ = n_items; next_ = 0; busy_ = th_.size(); generation_++;
        }
        cv_job_.notify_all();
        for (size_t i; (i = next_.fetch_add(1)) < n_items;) f(i);
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] { return busy_ == 0; });
    }
};

}  // namespace

extern "C" {

int vdf_path_compare(const char *a, size_t len_a, const char *b, size_t len_b)
{
    return path_cmp(a ? a : "", a ? len_a : 0, b ? b : "", b ? len_b : 0);
}

static int path_ranks_impl(const char *paths, const uint64_t *path_offsets, size_t n, uint32_t *out_rank, int n_threads);

int vdf_path_ranks(const char *paths, const uint64_t *path_offsets, size_t n, uint32_t *out_rank, int n_threads)
{
    try {
        return path_ranks_impl(paths, path_offsets, n, out_rank, n_threads);
    } catch (const std::bad_alloc &) {  // nothing may be thrown through the C ABI
        return VDF_E_OOM;
    } catch (...) {
        return VDF_E_INVAL;
    }
}

static int path_ranks_impl(const char *paths, const uint64_t *path_offsets, size_t n, uint32_t *out_rank, int n_threads)
{
    if (n == 0) return VDF_OK;
    if (!paths || !path_offsets || !out_rank || n >= 0xFFFFFFFFull) return VDF_E_INVAL;
    for (size_t i = 0; i < n; i++)
        if (path_offsets[i + 1] < path_offsets[i]) return VDF_E_INVAL;
    // automatic thread count: at most 32 - measured on the 256-thread host of the GPU box at 10 M paths (profiles/r05_cache_ingest.txt):
    // 8 threads 463 ms, 16 281, 32 209, 64 277, 256 396: beyond 32 the serial steps between the phases (sample sort, the scan over
    // buckets x slices) and the memory system take back what the parallel phases gain
    unsigned nt = n_threads > 0 ? (unsigned)n_threads : std::min(32u, std::thread::hardware_concurrency());
    nt = std::max(1u, std::min(nt, 256u));
    if (n < 4096) nt = 1;
    Blob B{paths, path_offsets};
    Pool pool(nt);
    std::vector<uint32_t> idx(n);
    {   // are all paths plain (path_order.cpp: is_plain)?  One linear pass; a single exception sends the whole call down the general road
        // (the same pass finds how many leading bytes ALL paths share - "/mnt/videos/" - which no comparison needs to look at again)
        const size_t n_sl = (size_t)nt * 4;
        std::atomic<bool> all_plain{true};
        std::vector<size_t> sl_lcp(n_sl, SIZE_MAX);
        const char *p0 = paths + path_offsets[0];
        const size_t l0 = (size_t)(path_offsets[1] - path_offsets[0]);
        pool.run(n_sl, [&](size_t sl) {
            const size_t lo = sl * n / n_sl, hi = (sl + 1) * n / n_sl;
            size_t g = l0;
            for (size_t i = lo; i < hi && all_plain.load(std::memory_order_relaxed); i++) {
                const char *q = paths + path_offsets[i];
                const size_t lq = (size_t)(path_offsets[i + 1] - path_offsets[i]);
                if (!is_plain(q, lq)) all_plain = false;
                if (g) g = common_prefix(p0, std::min(g, l0), q, lq);
            }
            sl_lcp[sl] = g;
        });
        B.plain = all_plain.load();
        if (B.plain) B.shared = *std::min_element(sl_lcp.begin(), sl_lcp.end());
    }
    std::vector<uint8_t> is_new(n);     // position k of the sorted order starts a new (distinct) path
    std::vector<uint8_t> new_known(n, 0);  // ... already decided while its bucket was sorted
    // ---- sample sort: splitters -> bucket of every entry -> counting scatter -> per-bucket sort
    // (also on one thread: a bucket's paths fit the cache, a std::sort over the whole blob misses on every comparison - 2 M paths 6.5 s against 2.4)
    const size_t n_buckets = n < 4096 ? 1 : std::min<size_t>(std::max<size_t>(n / 2048, 2), 8192);
    if (n_buckets == 1) {
        for (size_t i = 0; i < n; i++) idx[i] = (uint32_t)i;
        std::sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { const int c = B.cmp(x, y); return c ? c < 0 : x < y; });
    } else {
        // oversampled splitters (8 per bucket), taken at a fixed stride: cache files hold their entries in HashMap order (arbitrary)
        const size_t n_samples = std::min(n, n_buckets * 8);
        std::vector<uint32_t> samples(n_samples);
        for (size_t s = 0; s < n_samples; s++) samples[s] = (uint32_t)((s * n) / n_samples);
        if (B.plain) {  // (64 k samples: a comparison sort of them on one thread was 30 ms of the call)
            std::vector<KeyIdx> ks(n_samples);
            for (size_t q = 0; q < n_samples; q++) ks[q] = KeyIdx{0, samples[q]};
            keyed_sort(B, ks.data(), ks.size(), B.shared);
            for (size_t q = 0; q < n_samples; q++) samples[q] = ks[q].idx;
        } else {
            std::sort(samples.begin(), samples.end(), [&](uint32_t x, uint32_t y) { const int c = B.cmp(x, y); return c ? c < 0 : x < y; });
        }
        std::vector<uint32_t> split(n_buckets - 1);
        for (size_t k = 1; k < n_buckets; k++) split[k - 1] = samples[k * n_samples / n_buckets];
        // bucket(i) = number of splitters <= path i in (path, index) order: equal paths may straddle a splitter only by index,
        // and the per-bucket sort uses the same (path, index) order, so the concatenation is sorted
        // plain paths: what the smallest and the largest SAMPLE share is not shared by everything (entries outside the samples' range), so
        // the partition compares from byte 0; inside bucket k everything lies between two splitters and shares what they share
        std::vector<uint32_t> bucket_skip(n_buckets, 0);
        if (B.plain)
            for (size_t k = 1; k + 1 < n_buckets; k++) bucket_skip[k] = (uint32_t)B.lcp(split[k - 1], split[k]);
        // plain paths: the splitters' key words (ascending, like the splitters) narrow the binary search with integer comparisons
        // before the first path comparison: only splitters whose word equals the entry's remain to be compared byte by byte
        std::vector<uint64_t> split_key(B.plain ? split.size() : 0);
        for (size_t k = 0; k < split_key.size(); k++) split_key[k] = B.key8(split[k], B.shared);
        std::vector<uint16_t> bucket(n);
        const size_t n_slices = (size_t)nt * 4;
        std::vector<std::vector<uint32_t>> counts(n_slices, std::vector<uint32_t>(n_buckets, 0));
        // and the splitters' bytes (beyond what everything shares) sit back to back in one small array: the binary search of an entry then
        // walks ~0.4 MB of cache-resident bytes instead of 8 k offsets and 8 k paths scattered over the whole blob
        std::vector<char> split_bytes;
        std::vector<uint32_t> split_at(B.plain ? split.size() + 1 : 0, 0);
        if (B.plain) {
            for (size_t k = 0; k < split.size(); k++) {
                const size_t len = (size_t)(path_offsets[split[k] + 1] - path_offsets[split[k]]);
                split_at[k + 1] = split_at[k] + (uint32_t)(len - B.shared);
            }
            split_bytes.resize(split_at.back() + 8);
            for (size_t k = 0; k < split.size(); k++)
                std::memcpy(split_bytes.data() + split_at[k], paths + path_offsets[split[k]] + B.shared, split_at[k + 1] - split_at[k]);
        }
        pool.run(n_slices, [&](size_t sl) {
            const size_t lo = sl * n / n_slices, hi = (sl + 1) * n / n_slices;
            std::vector<uint32_t> &cnt = counts[sl];
            for (size_t i = lo; i < hi; i++) {
                size_t a = 0, b = split.size();  // first splitter that is greater than entry i
                if (B.plain) {
                    const uint64_t ki = B.key8((uint32_t)i, B.shared);
                    a = (size_t)(std::lower_bound(split_key.begin(), split_key.end(), ki) - split_key.begin());
                    b = (size_t)(std::upper_bound(split_key.begin() + (ptrdiff_t)a, split_key.end(), ki) - split_key.begin());
                    const char *q = paths + path_offsets[i] + B.shared;
                    const size_t lq = (size_t)(path_offsets[i + 1] - path_offsets[i]) - B.shared;
                    while (a < b) {
                        const size_t m = (a + b) / 2;
                        const int c = plain_cmp(split_bytes.data() + split_at[m], split_at[m + 1] - split_at[m], q, lq, 0);
                        if (c < 0 || (c == 0 && split[m] <= (uint32_t)i)) a = m + 1; else b = m;
                    }
                }
                while (a < b) {
                    const size_t m = (a + b) / 2;
                    const int c = B.cmp(split[m], (uint32_t)i);
                    if (c < 0 || (c == 0 && split[m] <= (uint32_t)i)) a = m + 1; else b = m;
                }
                bucket[i] = (uint16_t)a;
                cnt[a]++;
            }
        });
        // exclusive scan over (bucket, slice)
        std::vector<size_t> bucket_begin(n_buckets + 1, 0);
        {
            size_t run = 0;
            for (size_t k = 0; k < n_buckets; k++) {
                bucket_begin[k] = run;
                for (size_t sl = 0; sl < n_slices; sl++) { const uint32_t c = counts[sl][k]; counts[sl][k] = (uint32_t)run; run += c; }
            }
            bucket_begin[n_buckets] = run;
        }
        pool.run(n_slices, [&](size_t sl) {
            const size_t lo = sl * n / n_slices, hi = (sl + 1) * n / n_slices;
            std::vector<uint32_t> &at = counts[sl];
            for (size_t i = lo; i < hi; i++) idx[at[bucket[i]]++] = (uint32_t)i;
        });
        pool.run(n_buckets, [&](size_t k) {
            const size_t skip = std::max<size_t>(bucket_skip[k], B.shared);
            const size_t b0 = bucket_begin[k], b1 = bucket_begin[k + 1];
            if (!B.plain) {
                std::sort(idx.begin() + (ptrdiff_t)b0, idx.begin() + (ptrdiff_t)b1,
                          [&](uint32_t x, uint32_t y) { const int c = B.cmp(x, y); return c ? c < 0 : x < y; });
                return;
            }
            // plain: every path of the bucket shares `skip` bytes (keyed_sort takes it from there)
            std::vector<KeyIdx> tmp(b1 - b0);
            for (size_t q = b0; q < b1; q++) tmp[q - b0] = KeyIdx{0, idx[q]};
            keyed_sort(B, tmp.data(), tmp.size(), skip);
            for (size_t q = b0; q < b1; q++) idx[q] = tmp[q - b0].idx;
            // "does a new path start here" for all but the bucket's first position, while the bucket's paths are still in the cache
            for (size_t q = b0 + 1; q < b1; q++) { is_new[q] = B.cmp(idx[q - 1], idx[q], skip) != 0; new_known[q] = 1; }
        });
    }
    // ---- dense ranks: equal paths share one
    const size_t n_slices = nt == 1 ? 1 : (size_t)nt * 4;
    std::vector<uint32_t> first(n_slices + 1, 0);  // distinct-run starts inside each slice (position 0 of the array counts as one)
    pool.run(n_slices, [&](size_t sl) {
        const size_t lo = sl * n / n_slices, hi = (sl + 1) * n / n_slices;
        uint32_t c = 0;
        for (size_t k = lo; k < hi; k++) {
            const bool nw = new_known[k] ? is_new[k] != 0 : (k == 0 || B.cmp(idx[k - 1], idx[k]) != 0);
            is_new[k] = nw;
            c += nw;
        }
        first[sl + 1] = c;
    });
    for (size_t sl = 0; sl < n_slices; sl++) first[sl + 1] += first[sl];
    pool.run(n_slices, [&](size_t sl) {
        const size_t lo = sl * n / n_slices, hi = (sl + 1) * n / n_slices;
        uint32_t r = first[sl];  // runs started before this slice
        for (size_t k = lo; k < hi; k++) {
            r += is_new[k];
            out_rank[idx[k]] = r - 1;
        }
    });
    return VDF_OK;
}

}  // extern "C"
