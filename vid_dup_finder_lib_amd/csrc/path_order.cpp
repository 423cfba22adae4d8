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

struct Blob {
    const char *paths;
    const uint64_t *off;
    int cmp(uint32_t i, uint32_t j) const { return path_cmp(paths + off[i], (size_t)(off[i + 1] - off[i]), paths + off[j], (size_t)(off[j + 1] - off[j])); }
};

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
        for (unsigned t = 1; t < n_threads; t++) th_.emplace_back([this] { work(); });  // the caller is worker 0
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

int vdf_path_ranks(const char *paths, const uint64_t *path_offsets, size_t n, uint32_t *out_rank, int n_threads)
{
    if (n == 0) return VDF_OK;
    if (!paths || !path_offsets || !out_rank || n >= 0xFFFFFFFFull) return VDF_E_INVAL;
    for (size_t i = 0; i < n; i++)
        if (path_offsets[i + 1] < path_offsets[i]) return VDF_E_INVAL;
    unsigned nt = n_threads > 0 ? (unsigned)n_threads : std::thread::hardware_concurrency();
    nt = std::max(1u, std::min(nt, 256u));
    if (n < 4096) nt = 1;
    const Blob B{paths, path_offsets};
    Pool pool(nt);
    std::vector<uint32_t> idx(n);
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
        std::sort(samples.begin(), samples.end(), [&](uint32_t x, uint32_t y) { const int c = B.cmp(x, y); return c ? c < 0 : x < y; });
        std::vector<uint32_t> split(n_buckets - 1);
        for (size_t k = 1; k < n_buckets; k++) split[k - 1] = samples[k * n_samples / n_buckets];
        // bucket(i) = number of splitters <= path i in (path, index) order: equal paths may straddle a splitter only by index,
        // and the per-bucket sort uses the same (path, index) order, so the concatenation is sorted
        std::vector<uint16_t> bucket(n);
        const size_t n_slices = (size_t)nt * 4;
        std::vector<std::vector<uint32_t>> counts(n_slices, std::vector<uint32_t>(n_buckets, 0));
        pool.run(n_slices, [&](size_t sl) {
            const size_t lo = sl * n / n_slices, hi = (sl + 1) * n / n_slices;
            std::vector<uint32_t> &cnt = counts[sl];
            for (size_t i = lo; i < hi; i++) {
                size_t a = 0, b = split.size();  // first splitter that is greater than entry i
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
            std::sort(idx.begin() + (ptrdiff_t)bucket_begin[k], idx.begin() + (ptrdiff_t)bucket_begin[k + 1],
                      [&](uint32_t x, uint32_t y) { const int c = B.cmp(x, y); return c ? c < 0 : x < y; });
        });
    }
    // ---- dense ranks: equal paths share one
    const size_t n_slices = nt == 1 ? 1 : (size_t)nt * 4;
    std::vector<uint32_t> first(n_slices + 1, 0);  // distinct-run starts inside each slice (position 0 of the array counts as one)
    std::vector<uint8_t> is_new(n);
    pool.run(n_slices, [&](size_t sl) {
        const size_t lo = sl * n / n_slices, hi = (sl + 1) * n / n_slices;
        uint32_t c = 0;
        for (size_t k = lo; k < hi; k++) {
            const bool nw = k == 0 || B.cmp(idx[k - 1], idx[k]) != 0;
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
