// Raster wire formats on both sides of the hot path: the covariate
// GeoTIFFs machisplin reads with terra::rast() -- the bundled rasters are INT2S, LZW/deflate,
// NoData -32768, georeferenced by tags or by .tfw sidecars (inst/extdata/*.tif.ovr, *.tfw) -- are
// decoded on host threads band by band and streamed into device planes while the next band is
// being decoded; the result planes are written as FLT4S GeoTIFF, terra::writeRaster's default
// for doubles (machisplin.write.geotiff, V73:1011,1020).
//
// Reader: classic TIFF and BigTIFF, either byte order, strips or tiles, compression none / LZW /
// deflate, predictor 1 / 2 / 3, one sample per pixel, 8/16/32/64-bit integer or float samples.
// Writer: little-endian classic TIFF (BigTIFF above 4 GB), float32 strips, deflate or none, with
// ModelPixelScale / ModelTiepoint / GeoKeyDirectory (EPSG:4326) / GDAL_NODATA tags.
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include "common.h"

namespace mhs {

struct Ifd {
    int64_t width = 0, height = 0;
    int bits = 0, fmt = 1, comp = 1, pred = 1, spp = 1, planar = 1;
    int64_t tile_w = 0, tile_h = 0, rows_per_strip = 0;
    std::vector<uint64_t> offsets, counts;
    double nodata = NAN;
    bool has_scale = false, has_tie = false;
    double scale[3] = {0, 0, 0}, tie[6] = {0, 0, 0, 0, 0, 0};
};

struct TiffFile {
    FILE *f = nullptr;
    bool big = false, swap = false;
    uint64_t size = 0;   // bytes in the file: every offset / count a header field names is checked against it
    std::vector<Ifd> ifds;
    ~TiffFile() { if (f) fclose(f); }
};

static bool host_is_le() { const uint16_t x = 1; return *(const uint8_t *)&x == 1; }

template <typename T>
static T bswap(T v) {
    uint8_t *p = (uint8_t *)&v;
    std::reverse(p, p + sizeof(T));
    return v;
}

static bool rd(FILE *f, uint64_t off, void *dst, size_t n) {
    if (fseeko(f, (off_t)off, SEEK_SET) != 0) return false;
    return fread(dst, 1, n, f) == n;
}

template <typename T>
static bool rdv(TiffFile &t, uint64_t off, T *v) {
    if (!rd(t.f, off, v, sizeof(T))) return false;
    if (t.swap) *v = bswap(*v);
    return true;
}

static const int TYPE_SIZE[] = {0, 1, 1, 2, 4, 8, 1, 1, 2, 4, 8, 4, 8, 4, 0, 0, 8, 8, 8};

// numeric values of one directory entry as doubles (and raw bytes for ASCII)
static bool entry_values(TiffFile &t, int type, uint64_t count, uint64_t valoff_pos, std::vector<double> *out,
                         std::string *ascii) {
    if (type <= 0 || type > 18 || TYPE_SIZE[type] == 0) return false;
    if (count > t.size) return false;   // a value array cannot be longer than the file (bounds the allocations below)
    const uint64_t bytes = (uint64_t)TYPE_SIZE[type] * count;
    const uint64_t inline_cap = t.big ? 8 : 4;
    uint64_t pos = valoff_pos;
    if (bytes > inline_cap) {
        if (t.big) { uint64_t o; if (!rdv(t, valoff_pos, &o)) return false; pos = o; }
        else { uint32_t o; if (!rdv(t, valoff_pos, &o)) return false; pos = o; }
    }
    if (pos > t.size || bytes > t.size - pos) return false;
    std::vector<uint8_t> raw((size_t)bytes);
    if (bytes && !rd(t.f, pos, raw.data(), (size_t)bytes)) return false;
    if (type == 2) { if (ascii) ascii->assign((const char *)raw.data(), (size_t)bytes); return true; }
    if (!out) return true;
    out->resize((size_t)count);
    for (uint64_t i = 0; i < count; ++i) {
        const uint8_t *p = raw.data() + i * TYPE_SIZE[type];
        double v = 0;
#define RDT(T) { T x; memcpy(&x, p, sizeof(T)); if (t.swap) x = bswap(x); v = (double)x; }
        switch (type) {
            case 1: case 7: v = p[0]; break;
            case 6: v = (int8_t)p[0]; break;
            case 3: RDT(uint16_t) break;
            case 8: RDT(int16_t) break;
            case 4: case 13: RDT(uint32_t) break;
            case 9: RDT(int32_t) break;
            case 16: case 18: RDT(uint64_t) break;
            case 17: RDT(int64_t) break;
            case 11: RDT(float) break;
            case 12: RDT(double) break;
            case 5: { uint32_t a, b; memcpy(&a, p, 4); memcpy(&b, p + 4, 4); if (t.swap) { a = bswap(a); b = bswap(b); } v = b ? (double)a / b : 0; } break;
            case 10: { int32_t a, b; memcpy(&a, p, 4); memcpy(&b, p + 4, 4); if (t.swap) { a = bswap(a); b = bswap(b); } v = b ? (double)a / b : 0; } break;
            default: return false;
        }
#undef RDT
        (*out)[(size_t)i] = v;
    }
    return true;
}

static int open_tiff(const char *path, TiffFile *t) {
    t->f = fopen(path, "rb");
    if (!t->f) { set_error("cannot open %s", path); return MHS_ERR_INVALID; }
    if (fseeko(t->f, 0, SEEK_END) != 0) { set_error("%s: cannot seek", path); return MHS_ERR_INVALID; }
    { const off_t end = ftello(t->f); t->size = end > 0 ? (uint64_t)end : 0; }
    uint8_t hdr[16];
    if (!rd(t->f, 0, hdr, 8)) { set_error("%s: not a TIFF (short file)", path); return MHS_ERR_INVALID; }
    const bool le = hdr[0] == 'I' && hdr[1] == 'I', be = hdr[0] == 'M' && hdr[1] == 'M';
    if (!le && !be) { set_error("%s: not a TIFF (bad byte-order mark)", path); return MHS_ERR_INVALID; }
    t->swap = le != host_is_le();
    uint16_t magic;
    memcpy(&magic, hdr + 2, 2);
    if (t->swap) magic = bswap(magic);
    uint64_t next = 0;
    if (magic == 42) { uint32_t o; memcpy(&o, hdr + 4, 4); if (t->swap) o = bswap(o); next = o; }
    else if (magic == 43) { t->big = true; if (!rdv(*t, 8, &next)) return MHS_ERR_INVALID; }
    else { set_error("%s: not a TIFF (magic %d)", path, (int)magic); return MHS_ERR_INVALID; }
    for (int guard = 0; next != 0 && guard < 64; ++guard) {
        uint64_t n_entries = 0;
        if (t->big) { if (!rdv(*t, next, &n_entries)) break; }
        else { uint16_t n16; if (!rdv(*t, next, &n16)) break; n_entries = n16; }
        const uint64_t esz = t->big ? 20 : 12, base = next + (t->big ? 8 : 2);
        if (n_entries > 4096 || base > t->size || n_entries * esz > t->size - base) { set_error("%s: corrupt image directory", path); return MHS_ERR_INVALID; }
        Ifd d;
        for (uint64_t e = 0; e < n_entries; ++e) {
            const uint64_t p = base + e * esz;
            uint16_t tag, type;
            if (!rdv(*t, p, &tag) || !rdv(*t, p + 2, &type)) return MHS_ERR_INVALID;
            uint64_t count;
            if (t->big) { if (!rdv(*t, p + 4, &count)) return MHS_ERR_INVALID; }
            else { uint32_t c32; if (!rdv(*t, p + 4, &c32)) return MHS_ERR_INVALID; count = c32; }
            const uint64_t vpos = p + (t->big ? 12 : 8);
            std::vector<double> v;
            std::string a;
            switch (tag) {
                case 256: case 257: case 258: case 259: case 277: case 278: case 284: case 317: case 322: case 323: case 339:
                case 273: case 279: case 324: case 325: case 33550: case 33922:
                    if (!entry_values(*t, type, count, vpos, &v, nullptr) || v.empty()) { set_error("%s: unreadable tag %d", path, (int)tag); return MHS_ERR_INVALID; }
                    break;
                case 42113:
                    if (!entry_values(*t, type, count, vpos, nullptr, &a)) return MHS_ERR_INVALID;
                    break;
                default: continue;
            }
            switch (tag) {
                case 256: d.width = (int64_t)v[0]; break;
                case 257: d.height = (int64_t)v[0]; break;
                case 258: d.bits = (int)v[0]; break;
                case 259: d.comp = (int)v[0]; break;
                case 277: d.spp = (int)v[0]; break;
                case 278: d.rows_per_strip = (int64_t)v[0]; break;
                case 284: d.planar = (int)v[0]; break;
                case 317: d.pred = (int)v[0]; break;
                case 322: d.tile_w = (int64_t)v[0]; break;
                case 323: d.tile_h = (int64_t)v[0]; break;
                case 339: d.fmt = (int)v[0]; break;
                case 273: case 324: d.offsets.assign(v.size(), 0); for (size_t i = 0; i < v.size(); ++i) d.offsets[i] = (uint64_t)v[i]; break;
                case 279: case 325: d.counts.assign(v.size(), 0); for (size_t i = 0; i < v.size(); ++i) d.counts[i] = (uint64_t)v[i]; break;
                case 33550: if (v.size() >= 2) { d.has_scale = true; d.scale[0] = v[0]; d.scale[1] = v[1]; } break;
                case 33922: if (v.size() >= 6) { d.has_tie = true; for (int i = 0; i < 6; ++i) d.tie[i] = v[i]; } break;
                case 42113: d.nodata = atof(a.c_str()); break;
            }
        }
        if (d.rows_per_strip <= 0 || d.rows_per_strip > d.height) d.rows_per_strip = d.height;
        t->ifds.push_back(d);
        const uint64_t np = base + n_entries * esz;
        if (t->big) { if (!rdv(*t, np, &next)) break; }
        else { uint32_t o; if (!rdv(*t, np, &o)) break; next = o; }
    }
    if (t->ifds.empty()) { set_error("%s: no image directory", path); return MHS_ERR_INVALID; }
    return MHS_OK;
}

static int check_ifd(const char *path, const Ifd &d) {
    if (d.width <= 0 || d.height <= 0) { set_error("%s: bad dimensions", path); return MHS_ERR_INVALID; }
    if (d.spp != 1) { set_error("%s: %d samples per pixel (single-band rasters only)", path, d.spp); return MHS_ERR_INVALID; }
    if (!(d.bits == 8 || d.bits == 16 || d.bits == 32 || d.bits == 64)) { set_error("%s: %d bits per sample unsupported", path, d.bits); return MHS_ERR_INVALID; }
    if (!(d.comp == 1 || d.comp == 5 || d.comp == 8 || d.comp == 32946)) { set_error("%s: compression %d unsupported (none, LZW, deflate)", path, d.comp); return MHS_ERR_INVALID; }
    if (!(d.pred == 1 || d.pred == 2 || d.pred == 3)) { set_error("%s: predictor %d unsupported", path, d.pred); return MHS_ERR_INVALID; }
    if (d.offsets.empty() || d.offsets.size() != d.counts.size()) { set_error("%s: missing strip/tile offsets", path); return MHS_ERR_INVALID; }
    return MHS_OK;
}

// Everything the decoder derives from header fields, checked BEFORE any buffer is sized from them: a crafted
// width / height / tile size must not overflow the byte counts, every chunk the directory names must lie inside
// the file, and the directory must name as many chunks as the geometry needs.
constexpr int64_t TIFF_MAX_DIM = (int64_t)1 << 30;          // samples per row / rows
constexpr int64_t TIFF_MAX_CHUNK_BYTES = (int64_t)1 << 31;  // one decoded strip / tile
static int check_layout(const char *path, const TiffFile &t, const Ifd &d) {
    if (int rc = check_ifd(path, d)) return rc;
    const int64_t bps = d.bits / 8;
    if (d.width > TIFF_MAX_DIM || d.height > TIFF_MAX_DIM || d.width > ((int64_t)1 << 46) / d.height / bps) {
        set_error("%s: image dimensions %lld x %lld out of range", path, (long long)d.width, (long long)d.height);
        return MHS_ERR_INVALID;
    }
    if ((d.tile_w > 0) != (d.tile_h > 0) || d.tile_w < 0 || d.tile_h < 0 || d.tile_w > TIFF_MAX_DIM || d.tile_h > TIFF_MAX_DIM) {
        set_error("%s: bad tile dimensions", path);
        return MHS_ERR_INVALID;
    }
    const bool tiled = d.tile_w > 0;
    const int64_t cw = tiled ? d.tile_w : d.width, chh = tiled ? d.tile_h : d.rows_per_strip;
    if (chh <= 0 || cw > TIFF_MAX_CHUNK_BYTES / bps / chh) { set_error("%s: strip/tile of %lld x %lld samples is too large", path, (long long)cw, (long long)chh); return MHS_ERR_INVALID; }
    const int64_t across = tiled ? (d.width + cw - 1) / cw : 1, down = (d.height + chh - 1) / chh;
    if ((uint64_t)across * (uint64_t)down > d.offsets.size()) { set_error("%s: directory names %zu strips/tiles, the geometry needs %lld", path, d.offsets.size(), (long long)(across * down)); return MHS_ERR_INVALID; }
    for (size_t k = 0; k < d.offsets.size(); ++k)
        if (d.offsets[k] > t.size || d.counts[k] > t.size - d.offsets[k]) { set_error("%s: strip/tile %zu lies outside the file", path, k); return MHS_ERR_INVALID; }
    return MHS_OK;
}

// TIFF LZW (MSB-first codes, 9..12 bits, early change)
static bool lzw_decode(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_n) {
    struct Entry { int prev; uint8_t ch; uint16_t len; };
    static thread_local std::vector<Entry> tab;
    tab.resize(4096);
    for (int i = 0; i < 256; ++i) tab[i] = Entry{-1, (uint8_t)i, 1};
    int next = 258, bits = 9, prev = -1;
    uint32_t acc = 0;
    int nacc = 0;
    size_t ip = 0, op = 0;
    for (;;) {
        while (nacc < bits && ip < n) { acc = (acc << 8) | src[ip++]; nacc += 8; }
        if (nacc < bits) break;
        const int code = (int)((acc >> (nacc - bits)) & ((1u << bits) - 1));
        nacc -= bits;
        if (code == 257) break;
        if (code == 256) { next = 258; bits = 9; prev = -1; continue; }
        int cur = code;
        uint8_t first;
        if (prev < 0) {
            if (code > 255) return false;
            if (op >= cap) break;
            dst[op++] = (uint8_t)code;
            prev = code;
            continue;
        }
        if (code < next) {
            // emit string(code)
        } else if (code == next) {
            cur = prev;  // string(prev) + first(prev)
        } else return false;
        const int len = tab[cur].len + (code == next ? 1 : 0);
        if (op + (size_t)len > cap) {  // clip at the buffer end (padding in the last strip)
            std::vector<uint8_t> tmp((size_t)len);
            int c2 = cur, pos = tab[cur].len - 1;
            while (c2 >= 0) { tmp[(size_t)pos--] = tab[c2].ch; c2 = tab[c2].prev; }
            if (code == next) tmp[(size_t)len - 1] = tmp[0];
            const size_t room = cap - op;
            memcpy(dst + op, tmp.data(), room);
            op += room;
            break;
        }
        int c2 = cur, pos = tab[cur].len - 1;
        while (c2 >= 0) { dst[op + (size_t)pos--] = tab[c2].ch; c2 = tab[c2].prev; }
        first = dst[op];
        if (code == next) dst[op + (size_t)len - 1] = first;
        op += (size_t)len;
        if (next < 4096) {
            tab[next] = Entry{prev, first, (uint16_t)(tab[prev].len + 1)};
            ++next;
            if (next + 1 >= (1 << bits) && bits < 12) ++bits;  // early change
        }
        prev = code;
    }
    *out_n = op;
    return true;
}

// decode chunk k (strip or tile) into `buf` (native sample layout, host byte order, predictor undone);
// chunk rows x chunk cols samples
static bool decode_chunk(TiffFile &t, const Ifd &d, size_t k, std::vector<uint8_t> &raw, std::vector<uint8_t> &buf,
                         int64_t cw, int64_t ch) {
    const size_t bps = (size_t)d.bits / 8, need = (size_t)(cw * ch) * bps;
    buf.resize(need);
    raw.resize((size_t)d.counts[k]);
    {
        static std::mutex io;  // one FILE*, many decoder threads
        std::lock_guard<std::mutex> lk(io);
        if (d.counts[k] && !rd(t.f, d.offsets[k], raw.data(), raw.size())) return false;
    }
    if (d.comp == 1) {
        memcpy(buf.data(), raw.data(), std::min(need, raw.size()));
        if (raw.size() < need) memset(buf.data() + raw.size(), 0, need - raw.size());
    } else if (d.comp == 5) {
        size_t got = 0;
        if (!lzw_decode(raw.data(), raw.size(), buf.data(), need, &got)) return false;
        if (got < need) memset(buf.data() + got, 0, need - got);
    } else {
        uLongf got = (uLongf)need;
        const int rc = uncompress(buf.data(), &got, raw.data(), (uLong)raw.size());
        if (rc != Z_OK && rc != Z_BUF_ERROR) return false;
        if ((size_t)got < need) memset(buf.data() + got, 0, need - (size_t)got);
    }
    const bool file_le = host_is_le() != t.swap;
    if (d.pred == 3) {  // floating-point predictor: byte-plane differencing, planes in big-endian order
        std::vector<uint8_t> row((size_t)cw * bps);
        for (int64_t r = 0; r < ch; ++r) {
            uint8_t *p = buf.data() + (size_t)(r * cw) * bps;
            for (size_t i = 1; i < (size_t)cw * bps; ++i) p[i] = (uint8_t)(p[i] + p[i - 1]);
            memcpy(row.data(), p, row.size());
            for (int64_t c = 0; c < cw; ++c)
                for (size_t b = 0; b < bps; ++b) {
                    const uint8_t v = row[b * (size_t)cw + (size_t)c];  // plane b = byte b of the big-endian value
                    p[(size_t)c * bps + (host_is_le() ? bps - 1 - b : b)] = v;
                }
        }
        return true;
    }
    if (t.swap && bps > 1) {
        for (size_t i = 0; i + bps <= need; i += bps) std::reverse(buf.data() + i, buf.data() + i + bps);
    }
    (void)file_le;
    if (d.pred == 2) {
        for (int64_t r = 0; r < ch; ++r) {
            uint8_t *p = buf.data() + (size_t)(r * cw) * bps;
            if (bps == 1) for (int64_t c = 1; c < cw; ++c) p[c] = (uint8_t)(p[c] + p[c - 1]);
            else if (bps == 2) { uint16_t *q = (uint16_t *)p; for (int64_t c = 1; c < cw; ++c) q[c] = (uint16_t)(q[c] + q[c - 1]); }
            else if (bps == 4) { uint32_t *q = (uint32_t *)p; for (int64_t c = 1; c < cw; ++c) q[c] += q[c - 1]; }
            else { uint64_t *q = (uint64_t *)p; for (int64_t c = 1; c < cw; ++c) q[c] += q[c - 1]; }
        }
    }
    return true;
}

// decode image rows [r0, r1) into dst (row-major, `width` samples per row); r0 must sit on a chunk-row boundary
static int decode_rows(TiffFile &t, const Ifd &d, int64_t r0, int64_t r1, uint8_t *dst, int nthreads) {
    const size_t bps = (size_t)d.bits / 8;
    const bool tiled = d.tile_w > 0 && d.tile_h > 0;
    const int64_t cw = tiled ? d.tile_w : d.width, chh = tiled ? d.tile_h : d.rows_per_strip;
    const int64_t across = tiled ? (d.width + cw - 1) / cw : 1;
    const int64_t k0 = r0 / chh, k1 = (r1 + chh - 1) / chh;
    const int64_t nchunks = (k1 - k0) * across;
    std::atomic<int64_t> next(0);
    std::atomic<int> bad(0);
    auto work = [&]() {
      try {   // an exception leaving a std::thread calls std::terminate -- and would take the R process with it
        std::vector<uint8_t> raw, buf;
        for (int64_t j = next++; j < nchunks; j = next++) {
            const int64_t kr = k0 + j / across, kc = j % across;
            const size_t k = (size_t)(kr * across + kc);
            if (k >= d.offsets.size()) { bad = 1; return; }
            const int64_t rows_here = tiled ? chh : std::min(chh, d.height - kr * chh);
            if (!decode_chunk(t, d, k, raw, buf, cw, rows_here)) { bad = 1; return; }
            for (int64_t rr = 0; rr < rows_here; ++rr) {
                const int64_t row = kr * chh + rr;
                if (row < r0 || row >= r1 || row >= d.height) continue;
                const int64_t c0 = kc * cw, ncopy = std::min(cw, d.width - c0);
                memcpy(dst + ((size_t)(row - r0) * (size_t)d.width + (size_t)c0) * bps, buf.data() + (size_t)(rr * cw) * bps,
                       (size_t)ncopy * bps);
            }
        }
      } catch (...) { bad = 1; }
    };
    nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, nchunks));
    std::vector<std::thread> pool;
    for (int i = 1; i < nthreads; ++i) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
    if (bad) { set_error("corrupt or truncated strip/tile data"); return MHS_ERR_INVALID; }
    return MHS_OK;
}

static int decode_threads() {
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(16u, hc ? hc : 1u));
}

static int dtype_of(const Ifd &d) {  // MHS_* dtype a device plane of this file would have, or -1
    if (d.fmt == 3 && d.bits == 64) return MHS_F64;
    if (d.fmt == 3 && d.bits == 32) return MHS_F32;
    if (d.fmt == 2 && d.bits == 16) return MHS_I16;
    return -1;
}

__global__ __launch_bounds__(256) void f64_to_f32_kernel(const double *__restrict__ src, int64_t ld, int64_t nrow,
                                                         int64_t ncol, float nodata, int use_nodata,
                                                         float *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrow * ncol) return;
    const int64_t r = i / ncol, c = i - r * ncol;
    const double v = src[r * ld + c];
    dst[i] = (use_nodata && isnan(v)) ? nodata : (float)v;
}

// ---- writer helpers -------------------------------------------------------------------------
struct TagW { uint16_t tag, type; uint64_t count; std::vector<uint8_t> data; };

template <typename T>
static void put(std::vector<uint8_t> &b, T v) { const uint8_t *p = (const uint8_t *)&v; b.insert(b.end(), p, p + sizeof(T)); }

static int write_f32_tiff(const char *path, const mhs_grid &g, const float *data, double nodata, int compression) {
    const int64_t W = g.ncol, H = g.nrow;
    const int64_t rps = std::max<int64_t>(1, std::min<int64_t>(H, (1 << 20) / std::max<int64_t>(1, W * 4)));  // ~1 MB strips
    const int64_t nstrips = (H + rps - 1) / rps;
    std::vector<std::vector<uint8_t>> strips((size_t)nstrips);
    std::atomic<int64_t> next(0);
    std::atomic<int> bad(0);
    auto work = [&]() {
      try {
        for (int64_t s = next++; s < nstrips; s = next++) {
            const int64_t r0 = s * rps, nr = std::min(rps, H - r0);
            const uint8_t *src = (const uint8_t *)(data + r0 * W);
            const size_t n = (size_t)(nr * W) * 4;
            if (compression == 1) strips[(size_t)s].assign(src, src + n);
            else {
                uLongf cap = compressBound((uLong)n);
                strips[(size_t)s].resize(cap);
                if (compress2(strips[(size_t)s].data(), &cap, src, (uLong)n, 6) != Z_OK) { bad = 1; return; }
                strips[(size_t)s].resize(cap);
            }
        }
      } catch (...) { bad = 1; }
    };
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(decode_threads(), nstrips));
    std::vector<std::thread> pool;
    for (int i = 1; i < nt; ++i) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
    if (bad) { set_error("deflate failed"); return MHS_ERR_INVALID; }
    uint64_t total = 0;
    for (auto &s : strips) total += s.size();
    const bool big = total + (uint64_t)nstrips * 16 + 4096 > 0xFFFF0000ull;
    FILE *f = fopen(path, "wb");
    if (!f) { set_error("cannot create %s", path); return MHS_ERR_INVALID; }
    std::vector<uint8_t> hdr;
    hdr.push_back('I'); hdr.push_back('I');
    if (big) { put<uint16_t>(hdr, 43); put<uint16_t>(hdr, 8); put<uint16_t>(hdr, 0); put<uint64_t>(hdr, 0); }
    else { put<uint16_t>(hdr, 42); put<uint32_t>(hdr, 0); }
    fwrite(hdr.data(), 1, hdr.size(), f);
    std::vector<uint64_t> offs((size_t)nstrips), cnts((size_t)nstrips);
    uint64_t pos = hdr.size();
    for (int64_t s = 0; s < nstrips; ++s) {
        offs[(size_t)s] = pos; cnts[(size_t)s] = strips[(size_t)s].size();
        fwrite(strips[(size_t)s].data(), 1, strips[(size_t)s].size(), f);
        pos += strips[(size_t)s].size();
    }
    if (pos & 1) { fputc(0, f); ++pos; }
    // directory entries (ascending tag order)
    std::vector<TagW> tags;
    auto add_long = [&](uint16_t tag, std::vector<uint64_t> v) {
        TagW t{tag, (uint16_t)(big ? 16 : 4), v.size(), {}};
        for (uint64_t x : v) { if (big) put<uint64_t>(t.data, x); else put<uint32_t>(t.data, (uint32_t)x); }
        tags.push_back(t);
    };
    auto add_short = [&](uint16_t tag, std::vector<uint16_t> v) {
        TagW t{tag, 3, v.size(), {}};
        for (uint16_t x : v) put<uint16_t>(t.data, x);
        tags.push_back(t);
    };
    auto add_double = [&](uint16_t tag, std::vector<double> v) {
        TagW t{tag, 12, v.size(), {}};
        for (double x : v) put<double>(t.data, x);
        tags.push_back(t);
    };
    add_long(256, {(uint64_t)W}); add_long(257, {(uint64_t)H});
    add_short(258, {32}); add_short(259, {(uint16_t)(compression == 1 ? 1 : 8)}); add_short(262, {1});
    add_long(273, offs); add_short(277, {1}); add_long(278, {(uint64_t)rps}); add_long(279, cnts);
    add_short(284, {1}); add_short(339, {3});
    add_double(33550, {g.xres, g.yres, 0.0});
    add_double(33922, {0.0, 0.0, 0.0, g.xmin, g.ymax, 0.0});
    // GeoKeyDirectory: version 1.1.0, 3 keys: GTModelType = 2 (geographic), GTRasterType = 1 (PixelIsArea),
    // GeographicType = 4326 (WGS 84; the reference hard-codes +proj=longlat +datum=WGS84, V73:164,775)
    add_short(34735, {1, 1, 0, 3, 1024, 0, 1, 2, 1025, 0, 1, 1, 2048, 0, 1, 4326});
    if (!std::isnan(nodata)) {
        char buf[64];
        snprintf(buf, sizeof(buf), "%.17g", nodata);
        TagW t{42113, 2, strlen(buf) + 1, {}};
        t.data.assign(buf, buf + strlen(buf) + 1);
        tags.push_back(t);
    }
    const uint64_t ifd_pos = pos;
    const uint64_t esz = big ? 20 : 12, inl = big ? 8 : 4;
    const uint64_t dir_bytes = (big ? 8 : 2) + tags.size() * esz + (big ? 8 : 4);
    uint64_t extra = ifd_pos + dir_bytes;
    std::vector<uint8_t> dir, tail;
    if (big) put<uint64_t>(dir, tags.size()); else put<uint16_t>(dir, (uint16_t)tags.size());
    for (auto &t : tags) {
        put<uint16_t>(dir, t.tag); put<uint16_t>(dir, t.type);
        if (big) put<uint64_t>(dir, t.count); else put<uint32_t>(dir, (uint32_t)t.count);
        if (t.data.size() <= inl) {
            std::vector<uint8_t> v = t.data;
            v.resize(inl, 0);
            dir.insert(dir.end(), v.begin(), v.end());
        } else {
            if (big) put<uint64_t>(dir, extra + tail.size()); else put<uint32_t>(dir, (uint32_t)(extra + tail.size()));
            tail.insert(tail.end(), t.data.begin(), t.data.end());
            if (tail.size() & 1) tail.push_back(0);
        }
    }
    if (big) put<uint64_t>(dir, 0); else put<uint32_t>(dir, 0);
    fwrite(dir.data(), 1, dir.size(), f);
    fwrite(tail.data(), 1, tail.size(), f);
    // patch the first-IFD offset in the header
    if (big) { fseeko(f, 8, SEEK_SET); fwrite(&ifd_pos, 8, 1, f); }
    else { const uint32_t o = (uint32_t)ifd_pos; fseeko(f, 4, SEEK_SET); fwrite(&o, 4, 1, f); }
    const bool ok = fclose(f) == 0;
    if (!ok) { set_error("write to %s failed", path); return MHS_ERR_INVALID; }
    return MHS_OK;
}

}  // namespace mhs

using namespace mhs;

// No exception crosses the C ABI (the R shim would longjmp over live C++ frames): the bodies below run inside
// guard(), which maps std::bad_alloc to MHS_ERR_ALLOC and anything else to MHS_ERR_INVALID.
template <typename F>
static int guard(const char *what, F &&body) {
    try { return body(); }
    catch (const std::bad_alloc &) { set_error("%s: out of memory", what); return MHS_ERR_ALLOC; }
    catch (const std::exception &e) { set_error("%s: %s", what, e.what()); return MHS_ERR_INVALID; }
    catch (...) { set_error("%s: unexpected exception", what); return MHS_ERR_INVALID; }
}

static int mhs_tiff_info_read_impl(const char *path, int ifd, mhs_tiff_info *out) {
    MHS_REQUIRE(path && out && ifd >= 0, "bad arguments");
    TiffFile t;
    if (int rc = open_tiff(path, &t)) return rc;
    if ((size_t)ifd >= t.ifds.size()) { set_error("%s has %zu image directories", path, t.ifds.size()); return MHS_ERR_INVALID; }
    const Ifd &d = t.ifds[(size_t)ifd];
    if (int rc = check_layout(path, t, d)) return rc;
    out->width = d.width; out->height = d.height; out->bits = d.bits; out->sample_format = d.fmt;
    out->compression = d.comp; out->n_ifd = (int32_t)t.ifds.size(); out->nodata = d.nodata;
    out->dtype = dtype_of(d);
    out->has_geo = d.has_scale && d.has_tie;
    out->xres = d.has_scale ? d.scale[0] : NAN; out->yres = d.has_scale ? d.scale[1] : NAN;
    // tiepoint (i, j) -> (x, y): the north-west corner of the extent
    out->xmin = out->has_geo ? d.tie[3] - d.tie[0] * d.scale[0] : NAN;
    out->ymax = out->has_geo ? d.tie[4] + d.tie[1] * d.scale[1] : NAN;
    return MHS_OK;
}

static int mhs_tiff_read_host_impl(const char *path, int ifd, void *out, int64_t out_bytes) {
    MHS_REQUIRE(path && out && ifd >= 0, "bad arguments");
    TiffFile t;
    if (int rc = open_tiff(path, &t)) return rc;
    if ((size_t)ifd >= t.ifds.size()) { set_error("%s has %zu image directories", path, t.ifds.size()); return MHS_ERR_INVALID; }
    const Ifd &d = t.ifds[(size_t)ifd];
    if (int rc = check_layout(path, t, d)) return rc;
    MHS_REQUIRE(out_bytes >= d.width * d.height * (d.bits / 8), "output buffer too small");
    return decode_rows(t, d, 0, d.height, (uint8_t *)out, decode_threads());
}

static int mhs_tiff_read_dev_impl(const char *path, int ifd, void *out_dev, int64_t ld_elems, void *stream) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(path && out_dev && ifd >= 0, "bad arguments");
    TiffFile t;
    if (int rc = open_tiff(path, &t)) return rc;
    if ((size_t)ifd >= t.ifds.size()) { set_error("%s has %zu image directories", path, t.ifds.size()); return MHS_ERR_INVALID; }
    const Ifd &d = t.ifds[(size_t)ifd];
    if (int rc = check_layout(path, t, d)) return rc;
    MHS_REQUIRE(dtype_of(d) >= 0, "sample type has no device plane type (need int16, float32 or float64)");
    MHS_REQUIRE(ld_elems >= d.width, "ld smaller than the image width");
    hipStream_t s = pick_stream(stream);
    const size_t bps = (size_t)d.bits / 8;
    const bool tiled = d.tile_w > 0 && d.tile_h > 0;
    const int64_t chh = tiled ? d.tile_h : d.rows_per_strip;
    // bands of ~32 MB, a whole number of chunk rows; two pinned buffers: decode band k+1 while band k uploads
    int64_t band = std::max<int64_t>(chh, ((32 << 20) / std::max<int64_t>(1, d.width * (int64_t)bps)) / chh * chh);
    band = std::min(band, (d.height + chh - 1) / chh * chh);
    uint8_t *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2];
    for (int i = 0; i < 2; ++i) {
        MHS_HIP(hipHostMalloc((void **)&pin[i], (size_t)band * (size_t)d.width * bps, hipHostMallocDefault));
        MHS_HIP(hipEventCreate(&ev[i]));
    }
    int rc = MHS_OK, b = 0;
    for (int64_t r0 = 0; r0 < d.height && !rc; r0 += band, b ^= 1) {
        const int64_t r1 = std::min(d.height, r0 + band);
        if (r0 >= 2 * band && hipEventSynchronize(ev[b]) != hipSuccess) { rc = MHS_ERR_HIP; break; }
        rc = decode_rows(t, d, r0, r1, pin[b], decode_threads());
        if (rc) break;
        if (hipMemcpy2DAsync((uint8_t *)out_dev + (size_t)r0 * (size_t)ld_elems * bps, (size_t)ld_elems * bps, pin[b],
                             (size_t)d.width * bps, (size_t)d.width * bps, (size_t)(r1 - r0), hipMemcpyHostToDevice, s) != hipSuccess ||
            hipEventRecord(ev[b], s) != hipSuccess) rc = MHS_ERR_HIP;
    }
    (void)hipStreamSynchronize(s);
    for (int i = 0; i < 2; ++i) { (void)hipHostFree(pin[i]); (void)hipEventDestroy(ev[i]); }
    if (rc == MHS_ERR_HIP) set_error("HIP error while uploading %s", path);
    return rc;
}

static int mhs_tiff_write_f32_host_impl(const char *path, const mhs_grid *g, const float *data, double nodata, int compression) {
    MHS_REQUIRE(path && g && data && g->nrow > 0 && g->ncol > 0, "bad arguments");
    MHS_REQUIRE(compression == 1 || compression == 8, "compression must be 1 (none) or 8 (deflate)");
    return write_f32_tiff(path, *g, data, nodata, compression);
}

static int mhs_tiff_write_f32_dev_impl(const char *path, const mhs_grid *g, const double *plane_dev, int64_t ld, double nodata,
                           int compression, void *stream) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(path && g && plane_dev && g->nrow > 0 && g->ncol > 0 && ld >= g->ncol, "bad arguments");
    MHS_REQUIRE(compression == 1 || compression == 8, "compression must be 1 (none) or 8 (deflate)");
    hipStream_t s = pick_stream(stream);
    const int64_t n = g->nrow * g->ncol;
    DevBuf<float> tmp;
    MHS_HIP(tmp.alloc((size_t)n));
    hipLaunchKernelGGL(f64_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, plane_dev, ld, g->nrow, g->ncol,
                       (float)nodata, !std::isnan(nodata), tmp.p);
    MHS_HIP(hipGetLastError());
    float *host = nullptr;
    MHS_HIP(hipHostMalloc((void **)&host, sizeof(float) * (size_t)n, hipHostMallocDefault));
    int rc = MHS_OK;
    if (hipMemcpyAsync(host, tmp.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) { set_error("HIP error while downloading the plane"); rc = MHS_ERR_HIP; }
    if (!rc) rc = write_f32_tiff(path, *g, host, nodata, compression);
    (void)hipHostFree(host);
    return rc;
}

static int mhs_tfw_read_impl(const char *path, double *six) {
    MHS_REQUIRE(path && six, "bad arguments");
    FILE *f = fopen(path, "r");
    if (!f) { set_error("cannot open %s", path); return MHS_ERR_INVALID; }
    int n = 0;
    while (n < 6 && fscanf(f, "%lf", &six[n]) == 1) ++n;
    fclose(f);
    if (n != 6) { set_error("%s: a world file needs six numbers", path); return MHS_ERR_INVALID; }
    return MHS_OK;
}

extern "C" {

int mhs_tiff_info_read(const char *path, int ifd, mhs_tiff_info *out) {
    return guard("mhs_tiff_info_read", [&] { return mhs_tiff_info_read_impl(path, ifd, out); });
}
int mhs_tiff_read_host(const char *path, int ifd, void *out, int64_t out_bytes) {
    return guard("mhs_tiff_read_host", [&] { return mhs_tiff_read_host_impl(path, ifd, out, out_bytes); });
}
int mhs_tiff_read_dev(const char *path, int ifd, void *out_dev, int64_t ld_elems, void *stream) {
    return guard("mhs_tiff_read_dev", [&] { return mhs_tiff_read_dev_impl(path, ifd, out_dev, ld_elems, stream); });
}
int mhs_tiff_write_f32_host(const char *path, const mhs_grid *g, const float *data, double nodata, int compression) {
    return guard("mhs_tiff_write_f32_host", [&] { return mhs_tiff_write_f32_host_impl(path, g, data, nodata, compression); });
}
int mhs_tiff_write_f32_dev(const char *path, const mhs_grid *g, const double *plane_dev, int64_t ld, double nodata,
                           int compression, void *stream) {
    return guard("mhs_tiff_write_f32_dev", [&] { return mhs_tiff_write_f32_dev_impl(path, g, plane_dev, ld, nodata, compression, stream); });
}
int mhs_tfw_read(const char *path, double *six) {
    return guard("mhs_tfw_read", [&] { return mhs_tfw_read_impl(path, six); });
}

}  // extern "C"
