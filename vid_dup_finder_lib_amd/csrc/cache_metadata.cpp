// The metadata sidecar of the reference app's hash cache.  Host only, no GPU call.
//
// Next to <dir>/<stem>.<ext> the app keeps <dir>/<stem>.metadata.txt
//   vid_dup_finder_app/src/video_hash_filesystem_cache/video_hash_filesystem_cache.rs:76-139 (validate_or_create_metadata_file:
//     cache present + sidecar absent => the app exits with status 1; sidecar present => try_parse + validate, errors refuse the cache)
//   .../cache_metadata.rs:45-51 (VdfCacheMetadata), :54-78 (new: Unix | Windows by target, FfmpegBackend unless the gstreamer
//     feature, cache_version 1), :80-89 (to_disk_fmt = "{:?},{:?},{:?},{},{}"), :91-125 (try_parse: split on ',', exactly five
//     fields; the first two trim + lowercase, crop through enum_utils::FromStr = the exact variant name, then str::parse::<f64>
//     and str::parse::<u64>, neither of which trims), :127-168 (validate: field by field against new(exp_crop, exp_skip), first
//     difference reported).
// A cache written by vdf_cache_encode alone is therefore not something the app loads; vdf_cache_metadata_format gives the
// sidecar's bytes, vdf_cache_metadata_path its name, and a reader of an app-written cache checks with parse + validate that the
// hashes it is about to mix were made with the same crop detection (a Cropdetect::None cache holds other hashes for letterboxed
// files than a Letterbox one: video_hash_builder.rs:188-212).
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/vdf.h"

namespace {

void set_err(char *err, size_t cap, const std::string &msg)
{
    if (!err || cap == 0) return;
    const size_t n = std::min(msg.size(), cap - 1);
    std::memcpy(err, msg.data(), n);
    err[n] = 0;
}

// length of the Unicode White_Space character (what str::trim removes) that STARTS at s[i] / ENDS at s[i - 1], 0 if none:
// U+0009..000D, U+0020, U+0085, U+00A0, U+1680, U+2000..200A, U+2028, U+2029, U+202F, U+205F, U+3000 in UTF-8
size_t ws_at(const std::string &s, size_t i)
{
    const auto u = [&](size_t k) { return k < s.size() ? (unsigned char)s[k] : 0u; };
    const unsigned c = u(i);
    if (c == ' ' || (c >= 9 && c <= 13)) return 1;
    if (c == 0xC2 && (u(i + 1) == 0x85 || u(i + 1) == 0xA0)) return 2;
    if (c == 0xE1 && u(i + 1) == 0x9A && u(i + 2) == 0x80) return 3;
    if (c == 0xE2 && u(i + 1) == 0x80 && ((u(i + 2) >= 0x80 && u(i + 2) <= 0x8A) || u(i + 2) == 0xA8 || u(i + 2) == 0xA9 || u(i + 2) == 0xAF)) return 3;
    if (c == 0xE2 && u(i + 1) == 0x81 && u(i + 2) == 0x9F) return 3;
    if (c == 0xE3 && u(i + 1) == 0x80 && u(i + 2) == 0x80) return 3;
    return 0;
}

std::string trim_lower(const std::string &s)
{  // str::trim + to_lowercase (the keywords are ASCII, so ASCII lower-casing decides every comparison the same way)
    size_t a = 0, b = s.size();
    for (size_t w; a < b && (w = ws_at(s, a)) != 0;) a += w;
    for (;;) {
        size_t w = 0;
        for (size_t len = 1; len <= 3 && len <= b - a; len++)
            if (ws_at(s, b - len) == len) { w = len; break; }
        if (!w) break;
        b -= w;
    }
    std::string r = s.substr(a, b - a);
    for (char &c : r)
        if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
    return r;
}

// Rust's `impl Display for f64` ({}): the shortest digits that round-trip, always positional (never an exponent), "NaN", "inf".
std::string rust_f64_display(double v)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char buf[400];
    const auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
    return std::string(buf, r.ptr);
}

// Rust's `impl Debug for f64` ({:?}, used by validate's messages): as Display, but integral values keep a ".0" and very large /
// small magnitudes switch to the exponent form (>= 1e16 or < 1e-4).
std::string rust_f64_debug(double v)
{
    if (std::isnan(v) || std::isinf(v)) return rust_f64_display(v);
    const double a = std::fabs(v);
    if (a != 0.0 && (a >= 1e16 || a < 1e-4)) {
        char buf[64];
        const auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
        std::string s(buf, r.ptr);  // d.ddde+XX -> Rust prints d.ddde16 / d.ddde-7 (no '+', no leading zeros)
        const size_t e = s.find('e');
        std::string mant = s.substr(0, e), ex = s.substr(e + 1);
        bool neg = false;
        if (!ex.empty() && (ex[0] == '+' || ex[0] == '-')) { neg = ex[0] == '-'; ex.erase(0, 1); }
        while (ex.size() > 1 && ex[0] == '0') ex.erase(0, 1);
        return mant + "e" + (neg ? "-" : "") + ex;
    }
    std::string s = rust_f64_display(v);
    if (s.find('.') == std::string::npos) s += ".0";
    return s;
}

// str::parse::<f64> (core::num::dec2flt): [+-] ( "inf" | "infinity" | "nan" (any case) | digits [. digits] [(e|E) [+-] digits] ) with at
// least one mantissa digit; nothing else - no white space, no hex, no "nan(..)".
bool rust_parse_f64(const std::string &s, double *out)
{
    size_t i = 0;
    bool neg = false;
    if (i < s.size() && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; i++; }
    const std::string rest = s.substr(i);
    std::string low = rest;
    for (char &c : low)
        if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
    if (low == "inf" || low == "infinity") { *out = neg ? -INFINITY : INFINITY; return true; }
    if (low == "nan") { *out = NAN; return true; }
    size_t j = 0, mant_digits = 0;
    while (j < rest.size() && rest[j] >= '0' && rest[j] <= '9') { j++; mant_digits++; }
    if (j < rest.size() && rest[j] == '.') {
        j++;
        while (j < rest.size() && rest[j] >= '0' && rest[j] <= '9') { j++; mant_digits++; }
    }
    if (mant_digits == 0) return false;
    if (j < rest.size() && (rest[j] == 'e' || rest[j] == 'E')) {
        j++;
        if (j < rest.size() && (rest[j] == '+' || rest[j] == '-')) j++;
        size_t ed = 0;
        while (j < rest.size() && rest[j] >= '0' && rest[j] <= '9') { j++; ed++; }
        if (ed == 0) return false;
    }
    if (j != rest.size()) return false;
    // the grammar is a subset of what from_chars takes (it wants no '+' and no leading '.': handled by parsing `rest` with a 0 in front)
    const std::string num = "0" + rest;
    double v = 0.0;
    const auto r = std::from_chars(num.data(), num.data() + num.size(), v);
    if (r.ec == std::errc::result_out_of_range) {  // Rust rounds to inf / 0, it does not fail
        // the exponent's sign decides (a mantissa of zeros gives 0 either way)
        const size_t e = low.find('e');
        const bool neg_exp = e != std::string::npos && e + 1 < low.size() && low[e + 1] == '-';
        bool mant_zero = true;
        for (size_t k = 0; k < (e == std::string::npos ? low.size() : e); k++)
            if (low[k] >= '1' && low[k] <= '9') mant_zero = false;
        v = (mant_zero || neg_exp) ? 0.0 : INFINITY;
    } else if (r.ec != std::errc() || r.ptr != num.data() + num.size()) {
        return false;
    }
    *out = neg ? -v : v;
    return true;
}

// str::parse::<u64>: an optional '+', then decimal digits only; overflow is an error.
bool rust_parse_u64(const std::string &s, uint64_t *out)
{
    size_t i = 0;
    if (i < s.size() && s[i] == '+') i++;
    if (i == s.size()) return false;
    uint64_t v = 0;
    for (; i < s.size(); i++) {
        if (s[i] < '0' || s[i] > '9') return false;
        const uint64_t d = (uint64_t)(s[i] - '0');
        if (v > (0xFFFFFFFFFFFFFFFFull - d) / 10) return false;
        v = v * 10 + d;
    }
    *out = v;
    return true;
}

const char *os_name(int32_t v) { return v == VDF_CACHE_OS_WINDOWS ? "Windows" : "Unix"; }
const char *backend_name(int32_t v) { return v == VDF_CACHE_BACKEND_GSTREAMER ? "GstreamerBackend" : "FfmpegBackend"; }
const char *crop_name(int32_t v) { return v == VDF_CROPDETECT_NONE ? "None" : v == VDF_CROPDETECT_LETTERBOX ? "Letterbox" : "Motion"; }

bool fields_valid(const vdf_cache_metadata *m)
{
    return m && (m->operating_system == VDF_CACHE_OS_WINDOWS || m->operating_system == VDF_CACHE_OS_UNIX) &&
           (m->decode_backend == VDF_CACHE_BACKEND_FFMPEG || m->decode_backend == VDF_CACHE_BACKEND_GSTREAMER) &&
           (m->crop == VDF_CROPDETECT_NONE || m->crop == VDF_CROPDETECT_LETTERBOX || m->crop == VDF_CROPDETECT_MOTION);
}

}  // namespace

extern "C" {

int vdf_cache_metadata_new(int32_t crop, double skip_forward_amount, vdf_cache_metadata *out)
{
    if (!out || !(crop == VDF_CROPDETECT_NONE || crop == VDF_CROPDETECT_LETTERBOX || crop == VDF_CROPDETECT_MOTION)) return VDF_E_INVAL;
    out->operating_system = VDF_CACHE_OS_UNIX;        // target_family = "unix": this library is Linux / ROCm only
    out->decode_backend = VDF_CACHE_BACKEND_FFMPEG;   // the app's default features (no gstreamer_backend)
    out->crop = crop;
    out->reserved = 0;
    out->skip_forward_amount = skip_forward_amount;
    out->cache_version = 1;
    return VDF_OK;
}

int vdf_cache_metadata_format(const vdf_cache_metadata *m, char *buf, size_t cap, size_t *out_len)
{
    if (!fields_valid(m) || !out_len) return VDF_E_INVAL;
    const std::string s = std::string(os_name(m->operating_system)) + "," + backend_name(m->decode_backend) + "," + crop_name(m->crop) + "," +
                          rust_f64_display(m->skip_forward_amount) + "," + std::to_string((unsigned long long)m->cache_version);
    *out_len = s.size();
    if (!buf || cap < s.size()) return VDF_E_OVERFLOW;
    std::memcpy(buf, s.data(), s.size());
    if (cap > s.size()) buf[s.size()] = 0;
    return VDF_OK;
}

int vdf_cache_metadata_parse(const char *text, size_t len, vdf_cache_metadata *out, char *err, size_t err_cap)
{
    if (!out || (len && !text)) return VDF_E_INVAL;
    const std::string val(text ? text : "", len);
    std::string f[5];
    size_t n_fields = 1, at = 0;
    for (size_t i = 0; i < val.size(); i++)
        if (val[i] == ',') n_fields++;
    if (n_fields != 5) { set_err(err, err_cap, "Could not parse cache metadata. Got " + val); return VDF_E_INVAL; }
    for (int k = 0; k < 5; k++) {
        const size_t c = k < 4 ? val.find(',', at) : val.size();
        f[k] = val.substr(at, c - at);
        at = c + 1;
    }
    const std::string os = trim_lower(f[0]), be = trim_lower(f[1]);
    if (os == "windows") out->operating_system = VDF_CACHE_OS_WINDOWS;
    else if (os == "unix") out->operating_system = VDF_CACHE_OS_UNIX;
    else { set_err(err, err_cap, "Could not parse operating_system. Got " + f[0]); return VDF_E_INVAL; }
    if (be == "ffmpegbackend") out->decode_backend = VDF_CACHE_BACKEND_FFMPEG;
    else if (be == "gstreamerbackend") out->decode_backend = VDF_CACHE_BACKEND_GSTREAMER;
    else { set_err(err, err_cap, "Could not parse decode_backend. Got " + f[1]); return VDF_E_INVAL; }
    if (f[2] == "None") out->crop = VDF_CROPDETECT_NONE;
    else if (f[2] == "Letterbox") out->crop = VDF_CROPDETECT_LETTERBOX;
    else if (f[2] == "Motion") out->crop = VDF_CROPDETECT_MOTION;
    else { set_err(err, err_cap, "Could not parse crop. Got " + f[2]); return VDF_E_INVAL; }
    out->reserved = 0;
    if (!rust_parse_f64(f[3], &out->skip_forward_amount)) { set_err(err, err_cap, "Could not parse skip_forward amount. Got " + f[3]); return VDF_E_INVAL; }
    if (!rust_parse_u64(f[4], &out->cache_version)) { set_err(err, err_cap, "Could not parse cache_version. Got " + f[4]); return VDF_E_INVAL; }
    return VDF_OK;
}

int vdf_cache_metadata_validate(const vdf_cache_metadata *act, int32_t exp_crop, double exp_skip_forward_amount, char *err, size_t err_cap)
{
    vdf_cache_metadata exp;
    if (!fields_valid(act) || vdf_cache_metadata_new(exp_crop, exp_skip_forward_amount, &exp) != VDF_OK) return VDF_E_INVAL;
    std::string msg;
    if (act->operating_system != exp.operating_system)
        msg = std::string("operating_system mismatch: Act: ") + os_name(act->operating_system) + ", Exp: " + os_name(exp.operating_system);
    else if (act->decode_backend != exp.decode_backend)
        msg = std::string("decode_backend mismatch: Act: ") + backend_name(act->decode_backend) + ", Exp: " + backend_name(exp.decode_backend);
    else if (act->crop != exp.crop)
        msg = std::string("crop mismatch: Act: ") + crop_name(act->crop) + ", Exp: " + crop_name(exp.crop);
    else if (act->skip_forward_amount != exp.skip_forward_amount)  // f64 `!=`: NaN never validates, as in the app
        msg = "skip_forward_amount mismatch: Act: " + rust_f64_debug(act->skip_forward_amount) + ", Exp: " + rust_f64_debug(exp.skip_forward_amount);
    else if (act->cache_version != exp.cache_version)
        msg = "cache_version mismatch: Act: " + std::to_string((unsigned long long)act->cache_version) + ", Exp: " +
              std::to_string((unsigned long long)exp.cache_version);
    if (msg.empty()) return VDF_OK;
    set_err(err, err_cap, msg);
    return VDF_E_INVAL;
}

int vdf_cache_metadata_path(const char *cache_path, size_t len, char *buf, size_t cap, size_t *out_len)
{
    // Path::file_stem + with_file_name("{stem}.metadata.txt") (video_hash_filesystem_cache.rs:93-104).  file_name() is the last
    // Normal component (trailing separators and "." pieces do not count; a path ending in ".." or "/" alone has none: the app
    // reports EINVAL); the stem drops the part from the last '.' on, unless the name starts with its only '.'.
    if (!cache_path || !out_len) return VDF_E_INVAL;
    size_t end = len;
    for (;;) {  // strip trailing "/" and "/." (components() normalises them away)
        while (end > 0 && cache_path[end - 1] == '/') end--;
        if (end >= 2 && cache_path[end - 1] == '.' && cache_path[end - 2] == '/') { end -= 1; continue; }
        break;
    }
    size_t start = end;
    while (start > 0 && cache_path[start - 1] != '/') start--;
    const std::string name(cache_path + start, end - start);
    if (name.empty() || name == ".." || (name == "." )) return VDF_E_INVAL;
    std::string stem = name;
    const size_t dot = name.rfind('.');
    if (dot != std::string::npos && dot != 0) stem = name.substr(0, dot);
    // the directory part as Path::parent() yields it (with_file_name pops to the parent, then pushes): trailing separators and "."
    // pieces go, a lone root stays "/", a leading "." stays
    size_t dend = start;
    for (;;) {
        while (dend > 1 && cache_path[dend - 1] == '/') dend--;
        if (dend >= 3 && cache_path[dend - 1] == '.' && cache_path[dend - 2] == '/') { dend -= 1; continue; }
        break;
    }
    std::string dir(cache_path, dend);
    if (dir == "/" && start > 0) dir = "/";
    else if (!dir.empty() && dir.back() == '/' && dir.size() > 1) dir.pop_back();
    if (!dir.empty() && dir.back() != '/') dir += '/';
    const std::string res = dir + stem + ".metadata.txt";
    *out_len = res.size();
    if (!buf || cap < res.size()) return VDF_E_OVERFLOW;
    std::memcpy(buf, res.data(), res.size());
    if (cap > res.size()) buf[res.size()] = 0;
    return VDF_OK;
}

}  // extern "C"
