// Tile bookkeeping of machisplin.mltps Step 3/4/5 and machisplin.tiles.create/merge.
//
// Host part (integer windows, bit-exact): terra's extent snapping restated
// (SpatRaster::origin / align(e,"near") / crop -> colFromX,rowFromY), the Step-3 tile
// grid (V73:656-681), the tiles.create boxes (V73:1170-1197) and the seam strips
// (ext(as.points(A+B)) -> crop, V73:772-781 / 829-836 / 1408-1416 / 1469-1476).
// Device part (HBM-bound, one pass per tile / strip over its own window only):
// NA-aware mean mosaic, linear cross-fade of the seam strips, "first non-NA" overlay
// (V73:740-746, 783-806, 853-895, 1419-1546) and the Step-5 station gather (V73:910).
#include <algorithm>
#include <cmath>
#include <mutex>
#include <vector>
#include "common.h"

namespace mhs {

// ------------------------------------------------------------ terra snapping --
static inline double c_round(double x) { return round(x); }  // half away from zero, as terra (C++)

struct Ext { double xmin, xmax, ymin, ymax; };
struct Win { int64_t r0, r1, c0, c1; };

static inline double g_xmax(const mhs_grid &g) { return g.xmin + (double)g.ncol * g.xres; }
static inline double g_ymin(const mhs_grid &g) { return g.ymax - (double)g.nrow * g.yres; }

static void grid_origin(const mhs_grid &g, double *ox, double *oy) {
    double x = g.xmin - g.xres * c_round(g.xmin / g.xres);
    double y = g.ymax - g.yres * c_round(g.ymax / g.yres);
    if (fabs((g.xres + x) - fabs(x)) <= 1e-12 * g.xres) x = fabs(x);
    if (fabs((g.yres + y) - fabs(y)) <= 1e-12 * g.yres) y = fabs(y);
    *ox = x; *oy = y;
}

static int64_t col_from_x(const mhs_grid &g, double x) {
    const double xmax = g_xmax(g);
    if (x == xmax) return g.ncol - 1;
    if (x < g.xmin || x > xmax) return -1;
    return (int64_t)floor((x - g.xmin) / g.xres);
}

static int64_t row_from_y(const mhs_grid &g, double y) {
    const double ymin = g_ymin(g);
    if (y == ymin) return g.nrow - 1;
    if (y < ymin || y > g.ymax) return -1;
    return (int64_t)floor((g.ymax - y) / g.yres);
}

static Ext align_near(const mhs_grid &g, const Ext &e) {
    double ox, oy;
    grid_origin(g, &ox, &oy);
    Ext a;
    a.xmin = c_round((e.xmin - ox) / g.xres) * g.xres + ox;
    a.xmax = c_round((e.xmax - ox) / g.xres) * g.xres + ox;
    a.ymin = c_round((e.ymin - oy) / g.yres) * g.yres + oy;
    a.ymax = c_round((e.ymax - oy) / g.yres) * g.yres + oy;
    if (a.xmin == a.xmax) { if (a.xmin <= e.xmin) a.xmax = a.xmax + g.xres; else a.xmin = a.xmin - g.xres; }
    if (a.ymin == a.ymax) { if (a.ymin <= e.ymin) a.ymax = a.ymax + g.yres; else a.ymin = a.ymin - g.yres; }
    return a;
}

// terra::crop(x, e): false if the extents do not overlap
static bool crop_window(const mhs_grid &g, const Ext &e, Win *w) {
    const Ext a = align_near(g, e);
    const double xmn = std::max(a.xmin, g.xmin), xmx = std::min(a.xmax, g_xmax(g));
    const double ymn = std::max(a.ymin, g_ymin(g)), ymx = std::min(a.ymax, g.ymax);
    if (!(xmn < xmx && ymn < ymx)) return false;
    w->c0 = col_from_x(g, xmn + 0.5 * g.xres);
    w->c1 = col_from_x(g, xmx - 0.5 * g.xres) + 1;
    w->r0 = row_from_y(g, ymx - 0.5 * g.yres);
    w->r1 = row_from_y(g, ymn + 0.5 * g.yres) + 1;
    return true;
}

static mhs_grid window_geom(const mhs_grid &g, const Win &w) {
    mhs_grid s;
    s.xmin = g.xmin + (double)w.c0 * g.xres;
    s.ymax = g.ymax - (double)w.r0 * g.yres;
    s.xres = g.xres; s.yres = g.yres;
    s.nrow = w.r1 - w.r0; s.ncol = w.c1 - w.c0;
    return s;
}

static int check_grid(const mhs_grid *g) {
    MHS_REQUIRE(g && g->nrow > 0 && g->ncol > 0 && g->xres > 0 && g->yres > 0, "bad grid geometry");
    return MHS_OK;
}

// ------------------------------------------------------------------- kernels --
// bounding box (rows/cols, grid indices) of the cells of window w where A + B is not NA -- every seam in one launch
// (blockIdx.y = seam; seams whose tiles do not overlap have an empty window)
struct TileDesc { Win w; const double *p; int64_t ld; };
struct SeamDesc { int a, b; Win w; };
__global__ __launch_bounds__(256) void bbox_all_kernel(const TileDesc *__restrict__ T, const SeamDesc *__restrict__ S,
                                                       int *__restrict__ box /* per seam: rmin,rmax,cmin,cmax */) {
    const SeamDesc sd = S[blockIdx.y];
    const Win w = sd.w;
    const int64_t nc = w.c1 - w.c0, total = (w.r1 - w.r0) * nc;
    if (total <= 0) return;
    const TileDesc ta = T[sd.a], tb = T[sd.b];
    int rmin = INT32_MAX, rmax = -1, cmin = INT32_MAX, cmax = -1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = w.r0 + i / nc, c = w.c0 + i % nc;
        const double a = ta.p[(r - ta.w.r0) * ta.ld + (c - ta.w.c0)];
        const double b = tb.p[(r - tb.w.r0) * tb.ld + (c - tb.w.c0)];
        if (!isnan(a + b)) { rmin = min(rmin, (int)r); rmax = max(rmax, (int)r); cmin = min(cmin, (int)c); cmax = max(cmax, (int)c); }
    }
    if (rmax >= 0) {
        int *bx = box + 4 * (int64_t)blockIdx.y;
        atomicMin(&bx[0], rmin); atomicMax(&bx[1], rmax);
        atomicMin(&bx[2], cmin); atomicMax(&bx[3], cmax);
    }
}

// Step 4 in ONE pass over the output (round 6).  Rounds 1-5 scattered: two sum planes and two count planes zeroed,
// every tile and every strip added into them by a launch of its own (49 + 84 launches and 2 x 18 bytes per cell of
// scratch traffic at cfg3), a last pass composing the result -- 8.8 ms for 1e8 cells, the largest term of the tiled
// Step 3 + 4 once the tiles' fits went into one launch.  Here a block owns a 32 x 128 patch of the output, finds the
// few tiles and strips that touch it (bitmaps in LDS), and every cell gathers: the tiles that cover it in the
// reference's order (the sprc is built by prepending: last tile first), then the strips in collection order, the
// same additions in the same order as the scattered form -- bit-identical planes, 16 bytes per cell of traffic.
struct StripDesc { int a, b, axis_y, pad; Win w; double origin, res, cmin, delta; };
constexpr int MOSAIC_PR = 32, MOSAIC_PC = 128;
constexpr int MOSAIC_MAXWORDS = 1024;      // up to 32 768 tiles / strips per call
// Rows [row_lo, row_hi) of the output only (a device's row band); out points at row row_lo.
__global__ __launch_bounds__(256) void mosaic_fused_kernel(const TileDesc *__restrict__ T, int n_tiles,
                                                           const StripDesc *__restrict__ S, int n_strips, int64_t row_lo,
                                                           int64_t row_hi, int64_t ncol, double *__restrict__ out, int64_t ld) {
    __shared__ unsigned tmap[MOSAIC_MAXWORDS], smap[MOSAIC_MAXWORDS];
    const int tw = (n_tiles + 31) >> 5, sw = (n_strips + 31) >> 5;
    for (int i = threadIdx.x; i < tw; i += 256) tmap[i] = 0u;
    for (int i = threadIdx.x; i < sw; i += 256) smap[i] = 0u;
    __syncthreads();
    const int64_t pr0 = row_lo + (int64_t)blockIdx.y * MOSAIC_PR, pc0 = (int64_t)blockIdx.x * MOSAIC_PC;
    const int64_t pr1 = min(pr0 + MOSAIC_PR, row_hi), pc1 = min(pc0 + MOSAIC_PC, ncol);
    for (int h = threadIdx.x; h < n_tiles; h += 256) {
        const Win w = T[h].w;
        if (w.r0 < pr1 && w.r1 > pr0 && w.c0 < pc1 && w.c1 > pc0) atomicOr(&tmap[h >> 5], 1u << (h & 31));
    }
    for (int q = threadIdx.x; q < n_strips; q += 256) {
        const Win w = S[q].w;
        if (w.r0 < pr1 && w.r1 > pr0 && w.c0 < pc1 && w.c1 > pc0) atomicOr(&smap[q >> 5], 1u << (q & 31));
    }
    __syncthreads();
    const int64_t c = pc0 + (threadIdx.x & (MOSAIC_PC - 1));
    const int64_t rbase = pr0 + (threadIdx.x / MOSAIC_PC) * (MOSAIC_PR / (256 / MOSAIC_PC));
    constexpr int RPT = MOSAIC_PR / (256 / MOSAIC_PC);      // rows per thread
    double bsum[RPT], ssum[RPT];
    int bcnt[RPT], scnt[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) { bsum[k] = 0.0; ssum[k] = 0.0; bcnt[k] = 0; scnt[k] = 0; }
    // mean mosaic of the tiles: last tile first
    for (int wi = tw - 1; wi >= 0; --wi) {
        unsigned bits = tmap[wi];
        while (bits) {
            const int bit = 31 - __clz(bits);
            bits &= ~(1u << bit);
            const TileDesc t = T[wi * 32 + bit];
            if (c >= t.w.c0 && c < t.w.c1) {
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    const int64_t r = rbase + k;
                    if (r >= t.w.r0 && r < t.w.r1) {
                        const double v = t.p[(r - t.w.r0) * t.ld + (c - t.w.c0)];
                        if (!isnan(v)) { bsum[k] = bsum[k] + v; ++bcnt[k]; }
                    }
                }
            }
        }
    }
    // seam strips in collection order: feath = B * t + A * (1 - t), t = (coord - cmin) / delta   (V73:787-798)
    for (int wi = 0; wi < sw; ++wi) {
        unsigned bits = smap[wi];
        while (bits) {
            const int bit = __ffs(bits) - 1;
            bits &= ~(1u << bit);
            const StripDesc sd = S[wi * 32 + bit];
            if (c >= sd.w.c0 && c < sd.w.c1) {
                const TileDesc ta = T[sd.a], tb = T[sd.b];
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    const int64_t r = rbase + k;
                    if (r >= sd.w.r0 && r < sd.w.r1) {
                        const int64_t lr = r - sd.w.r0, lc = c - sd.w.c0;
                        const bool ina = r >= ta.w.r0 && r < ta.w.r1 && c >= ta.w.c0 && c < ta.w.c1;
                        const bool inb = r >= tb.w.r0 && r < tb.w.r1 && c >= tb.w.c0 && c < tb.w.c1;
                        const double a = ina ? ta.p[(r - ta.w.r0) * ta.ld + (c - ta.w.c0)] : NAN;   // extend(): NA outside
                        const double b = inb ? tb.p[(r - tb.w.r0) * tb.ld + (c - tb.w.c0)] : NAN;
                        // strip raster's own cell centres: xmin_s + (col + 0.5) xres  /  ymax_s - (row + 0.5) yres
                        const double coord = sd.axis_y ? sd.origin - ((double)lr + 0.5) * sd.res : sd.origin + ((double)lc + 0.5) * sd.res;
                        const double stD2 = (coord - sd.cmin) / sd.delta;
                        const double stD1 = 1.0 - (coord - sd.cmin) / sd.delta;
                        const double v = b * stD2 + a * stD1;
                        if (!isnan(v)) { ssum[k] = ssum[k] + v; ++scnt[k]; }
                    }
                }
            }
        }
    }
    // final.TPS = first non-NA of (mean of strips, mean of tiles)   (V73:887-889, 1528-1540)
    if (c < ncol) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int64_t r = rbase + k;
            if (r < row_hi) {
                double v;
                if (scnt[k] > 0) v = ssum[k] / (double)scnt[k];
                else if (bcnt[k] > 0) v = bsum[k] / (double)bcnt[k];
                else v = NAN;
                out[(r - row_lo) * ld + c] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void gather_kernel(const double *__restrict__ plane, int64_t ld,
                                                     const int64_t *__restrict__ rows,
                                                     const int64_t *__restrict__ cols, int64_t n,
                                                     double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (rows[i] < 0 || cols[i] < 0) ? NAN : plane[rows[i] * ld + cols[i]];
}

struct Seam { int64_t a, b; int axis_y; };

// seams in creation order: vertical (V73:764-806) then horizontal (V73:815-877)
static std::vector<Seam> seam_list(int64_t nRx, int64_t nCx) {
    std::vector<Seam> s;
    for (int64_t j = 1; j <= nRx; ++j)
        for (int64_t h = 1; h <= nCx; ++h) {
            const int64_t v = h + (j * nCx) - nCx;
            if (h < nCx) s.push_back(Seam{v - 1, v, 0});
        }
    int64_t f_clock = 0;
    for (int64_t j = 1; j <= nRx; ++j)
        for (int64_t h = 1; h <= nCx; ++h) {
            ++f_clock;
            const int64_t f_timer = (nRx * nCx) - nCx + 1;
            const int64_t v = h + (j * nCx) - nCx;
            if (f_clock < f_timer) s.push_back(Seam{v - 1, v + nCx - 1, 1});
        }
    return s;
}

static bool intersect(const Win &a, const Win &b, Win *o) {
    o->r0 = std::max(a.r0, b.r0); o->r1 = std::min(a.r1, b.r1);
    o->c0 = std::max(a.c0, b.c0); o->c1 = std::min(a.c1, b.c1);
    return o->r0 < o->r1 && o->c0 < o->c1;
}

static unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace mhs

using namespace mhs;

extern "C" {

int mhs_crop_window(const mhs_grid *g, const double *ext4, int64_t *win4) {
    if (int rc = check_grid(g)) return rc;
    MHS_REQUIRE(ext4 && win4, "NULL argument");
    Win w;
    if (!crop_window(*g, Ext{ext4[0], ext4[1], ext4[2], ext4[3]}, &w)) {
        set_error("mhs_crop_window: extents do not overlap");
        return MHS_ERR_INVALID;
    }
    win4[0] = w.r0; win4[1] = w.r1; win4[2] = w.c0; win4[3] = w.c1;
    return MHS_OK;
}

int mhs_step3_tile_windows(const mhs_grid *g, int64_t tile_edge, double fit_overlap, double keep_overlap,
                           int64_t *nRx_out, int64_t *nCx_out, int64_t *fit_win, int64_t *keep_win,
                           int64_t capacity) {
    if (int rc = check_grid(g)) return rc;
    MHS_REQUIRE(tile_edge > 0 && nRx_out && nCx_out, "bad arguments");
    const int64_t nRx = (int64_t)ceil((double)g->nrow / (double)tile_edge);
    const int64_t nCx = (int64_t)ceil((double)g->ncol / (double)tile_edge);
    *nRx_out = nRx; *nCx_out = nCx;
    if (!fit_win && !keep_win) return MHS_OK;  // size query
    MHS_REQUIRE(fit_win && keep_win && capacity >= nRx * nCx, "window arrays too small");
    const double xmin = g->xmin, xmax = g_xmax(*g), ymin = g_ymin(*g), ymax = g->ymax;
    const double longDist = (xmax - xmin) / (double)nCx;
    const double latDist = (ymax - ymin) / (double)nRx;
    int64_t m = 0;
    for (int64_t j = 1; j <= nRx; ++j)
        for (int64_t h = 1; h <= nCx; ++h, ++m) {
            const double hm1 = (double)(h - 1), hh = (double)h, jm1 = (double)(j - 1), jj = (double)j;
            const Ext b{xmin + ((longDist * hm1) - (longDist * fit_overlap)), xmin + ((longDist * hh) + (longDist * fit_overlap)),
                        (ymin + ((latDist * jm1))) - (latDist * fit_overlap), (ymin + ((latDist * jj))) + (latDist * fit_overlap)};
            const Ext d{xmin + ((longDist * hm1) - (longDist * keep_overlap)), xmin + ((longDist * hh) + (longDist * keep_overlap)),
                        (ymin + ((latDist * jm1))) - (latDist * keep_overlap), (ymin + ((latDist * jj))) + (latDist * keep_overlap)};
            Win wf, wk;
            if (!crop_window(*g, b, &wf)) { set_error("step3: empty fit window"); return MHS_ERR_NUMERIC; }
            const mhs_grid gf = window_geom(*g, wf);
            if (!crop_window(gf, d, &wk)) { set_error("step3: empty keep window"); return MHS_ERR_NUMERIC; }
            fit_win[4 * m + 0] = wf.r0; fit_win[4 * m + 1] = wf.r1; fit_win[4 * m + 2] = wf.c0; fit_win[4 * m + 3] = wf.c1;
            keep_win[4 * m + 0] = wf.r0 + wk.r0; keep_win[4 * m + 1] = wf.r0 + wk.r1;
            keep_win[4 * m + 2] = wf.c0 + wk.c0; keep_win[4 * m + 3] = wf.c0 + wk.c1;
        }
    return MHS_OK;
}

int mhs_tiles_create_windows(const mhs_grid *g, int64_t out_ncol, int64_t out_nrow, double feather_d,
                             double *boxes, int64_t *win) {
    if (int rc = check_grid(g)) return rc;
    MHS_REQUIRE(out_ncol >= 1 && out_nrow >= 1 && boxes && win, "bad arguments");
    feather_d = feather_d / 2;
    const double xmin = g->xmin, xmax = g_xmax(*g), ymin = g_ymin(*g), ymax = g->ymax;
    const double long_pix = (xmax - xmin) / (double)g->ncol, lat_pix = (ymax - ymin) / (double)g->nrow;
    const double longDist = (xmax - xmin) / (double)out_ncol, latDist = (ymax - ymin) / (double)out_nrow;
    int64_t m = 0;
    for (int64_t j = 1; j <= out_nrow; ++j)
        for (int64_t h = 1; h <= out_ncol; ++h, ++m) {
            const double hm1 = (double)(h - 1), hh = (double)h, jm1 = (double)(j - 1), jj = (double)j;
            const Ext b{xmin + ((longDist * hm1) - (long_pix * feather_d)), xmin + ((longDist * hh) + (long_pix * feather_d)),
                        (ymin + ((latDist * jm1))) - (lat_pix * feather_d), (ymin + ((latDist * jj))) + (lat_pix * feather_d)};
            boxes[4 * m + 0] = b.xmin; boxes[4 * m + 1] = b.xmax; boxes[4 * m + 2] = b.ymin; boxes[4 * m + 3] = b.ymax;
            Win w;
            if (!crop_window(*g, b, &w)) { set_error("tiles.create: empty tile"); return MHS_ERR_NUMERIC; }
            win[4 * m + 0] = w.r0; win[4 * m + 1] = w.r1; win[4 * m + 2] = w.c0; win[4 * m + 3] = w.c1;
        }
    return MHS_OK;
}

}  // extern "C"

// finite_tiles: the caller vouches that no tile holds an NA (thin-plate-spline planes: mhs_tps_surface) -- every seam's
// A + B is then non-NA on the whole overlap and the bounding-box pass (a kernel, a copy back and a host wait) is skipped
// [row_lo, row_hi): the rows to produce (out_dev holds row row_lo first); tiles that do not reach them may be NULL
int mhs::mosaic_feather_impl(const mhs_grid *g, int64_t nRx, int64_t nCx, const int64_t *tile_win,
                             const double *const *tile_dev, int merge_mode, double *out_dev, int64_t ld,
                             int64_t *seam_win_out, void *stream, bool finite_tiles, int64_t row_lo, int64_t row_hi) {
    if (int rc = require_ready()) return rc;
    if (int rc = check_grid(g)) return rc;
    MHS_REQUIRE(nRx >= 1 && nCx >= 1 && tile_win && tile_dev && out_dev && ld >= g->ncol, "bad arguments");
    if (row_hi < 0) row_hi = g->nrow;
    MHS_REQUIRE(0 <= row_lo && row_lo <= row_hi && row_hi <= g->nrow, "bad row band");
    MHS_REQUIRE(finite_tiles || (row_lo == 0 && row_hi == g->nrow), "a row band needs tiles without NA");
    if (row_lo == row_hi) return MHS_OK;
    hipStream_t s = pick_stream(stream);
    const int64_t n = nRx * nCx;
    MHS_REQUIRE(n <= 32 * MOSAIC_MAXWORDS, "too many tiles");
    std::vector<Win> tw((size_t)n);
    std::vector<TileDesc> td((size_t)n);
    for (int64_t h = 0; h < n; ++h) {
        tw[h] = Win{tile_win[4 * h], tile_win[4 * h + 1], tile_win[4 * h + 2], tile_win[4 * h + 3]};
        MHS_REQUIRE(0 <= tw[h].r0 && tw[h].r0 < tw[h].r1 && tw[h].r1 <= g->nrow && 0 <= tw[h].c0 &&
                    tw[h].c0 < tw[h].c1 && tw[h].c1 <= g->ncol, "bad tile window");
        MHS_REQUIRE(tile_dev[h] || tw[h].r1 <= row_lo || tw[h].r0 >= row_hi, "a tile that reaches the requested rows is NULL");
        td[h] = TileDesc{tw[h], tile_dev[h], tw[h].c1 - tw[h].c0};
    }
    std::vector<Seam> seams = n > 1 ? seam_list(nRx, nCx) : std::vector<Seam>();
    const size_t ns = seams.size();
    MHS_REQUIRE((int64_t)ns <= 32 * MOSAIC_MAXWORDS, "too many seams");
    // descriptors and the seams' boxes live in a small grow-only arena of the slot (one call at a time uses it; the
    // call ends with a stream synchronisation)
    std::lock_guard<std::mutex> arena_lock(mosaic_mutex());
    TileDesc *td_dev; SeamDesc *sd_dev; StripDesc *st_dev; int *box_dev;
    {
        Context &c = ctx();
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t o_td = 0, o_sd = up(o_td + sizeof(TileDesc) * (size_t)n), o_st = up(o_sd + sizeof(SeamDesc) * (ns + 1)),
                     o_box = up(o_st + sizeof(StripDesc) * (ns + 1)), need = up(o_box + 16 * (ns + 1));
        const int lane = mosaic_lane();
        if (need > c.mosaic_arena_cap[lane]) {
            if (c.mosaic_arena[lane]) { (void)hipStreamSynchronize(s); (void)hipFree(c.mosaic_arena[lane]); c.mosaic_arena[lane] = nullptr; c.mosaic_arena_cap[lane] = 0; }
            MHS_HIP(hipMalloc((void **)&c.mosaic_arena[lane], need * 2));
            c.mosaic_arena_cap[lane] = need * 2;
        }
        char *a = c.mosaic_arena[lane];
        td_dev = (TileDesc *)(a + o_td); sd_dev = (SeamDesc *)(a + o_sd); st_dev = (StripDesc *)(a + o_st); box_dev = (int *)(a + o_box);
    }
    MHS_HIP(hipMemcpyAsync(td_dev, td.data(), sizeof(TileDesc) * (size_t)n, hipMemcpyHostToDevice, s));
    std::vector<StripDesc> strips;
    if (ns > 0) {
        std::vector<int> hbox(4 * ns);
        std::vector<Win> inter(ns);
        std::vector<char> has(ns, 0);
        std::vector<SeamDesc> sd(ns);
        int64_t max_tot = 0;
        for (size_t k = 0; k < ns; ++k) {
            const Win &wa = tw[seams[k].a], &wb = tw[seams[k].b];
            has[k] = intersect(wa, wb, &inter[k]) ? 1 : 0;
            sd[k] = SeamDesc{(int)seams[k].a, (int)seams[k].b, has[k] ? inter[k] : Win{0, 0, 0, 0}};
            if (has[k]) max_tot = std::max(max_tot, (inter[k].r1 - inter[k].r0) * (inter[k].c1 - inter[k].c0));
        }
        if (finite_tiles) {
            for (size_t k = 0; k < ns; ++k) {
                if (has[k]) { hbox[4 * k] = (int)inter[k].r0; hbox[4 * k + 1] = (int)inter[k].r1 - 1; hbox[4 * k + 2] = (int)inter[k].c0; hbox[4 * k + 3] = (int)inter[k].c1 - 1; }
                else { hbox[4 * k] = INT32_MAX; hbox[4 * k + 1] = -1; hbox[4 * k + 2] = INT32_MAX; hbox[4 * k + 3] = -1; }
            }
        } else {
            for (size_t k = 0; k < ns; ++k) { hbox[4 * k] = INT32_MAX; hbox[4 * k + 1] = -1; hbox[4 * k + 2] = INT32_MAX; hbox[4 * k + 3] = -1; }
            MHS_HIP(hipMemcpyAsync(box_dev, hbox.data(), sizeof(int) * 4 * ns, hipMemcpyHostToDevice, s));
            MHS_HIP(hipMemcpyAsync(sd_dev, sd.data(), sizeof(SeamDesc) * ns, hipMemcpyHostToDevice, s));
            if (max_tot > 0) {
                const unsigned gx = (unsigned)std::min<int64_t>((max_tot + 255) / 256, 1024);
                hipLaunchKernelGGL(bbox_all_kernel, dim3(gx, (unsigned)ns), dim3(256), 0, s, td_dev, sd_dev, box_dev);
                MHS_HIP(hipGetLastError());
            }
            MHS_HIP(hipMemcpyAsync(hbox.data(), box_dev, sizeof(int) * 4 * ns, hipMemcpyDeviceToHost, s));
            MHS_HIP(hipStreamSynchronize(s));
        }
        // strips enter the mean in collection order: creation order for Step 4 (the stack is built by
        // prepending and the sprc prepends again), reversed for tiles.merge (c() appends, sprc prepends).
        // With exactly two tiles there is one seam and terra::merge takes it as is.
        for (size_t q = 0; q < ns; ++q) {
            const size_t k = merge_mode ? ns - 1 - q : q;
            Win w{-1, -1, -1, -1};
            if (has[k] && hbox[4 * k + 1] >= 0) {
                // blend.ext = bbox of the CENTRES of the non-NA cells; crop() snaps it back to cells
                const Ext e{g->xmin + ((double)hbox[4 * k + 2] + 0.5) * g->xres, g->xmin + ((double)hbox[4 * k + 3] + 0.5) * g->xres,
                            g->ymax - ((double)hbox[4 * k + 1] + 0.5) * g->yres, g->ymax - ((double)hbox[4 * k] + 0.5) * g->yres};
                if (!crop_window(*g, e, &w)) w = Win{-1, -1, -1, -1};
            }
            if (seam_win_out) { seam_win_out[4 * k] = w.r0; seam_win_out[4 * k + 1] = w.r1; seam_win_out[4 * k + 2] = w.c0; seam_win_out[4 * k + 3] = w.c1; }
            if (w.r0 < 0) continue;
            const mhs_grid gs = window_geom(*g, w);
            double origin, res, cfirst, clast;
            if (seams[k].axis_y) {
                origin = gs.ymax; res = gs.yres;
                cfirst = gs.ymax - (0.0 + 0.5) * gs.yres; clast = gs.ymax - ((double)(gs.nrow - 1) + 0.5) * gs.yres;
            } else {
                origin = gs.xmin; res = gs.xres;
                cfirst = gs.xmin + (0.0 + 0.5) * gs.xres; clast = gs.xmin + ((double)(gs.ncol - 1) + 0.5) * gs.xres;
            }
            const double cmin = std::min(cfirst, clast), cmax = std::max(cfirst, clast);
            const double delta = cmax - cmin;  // 0 for a one-cell-wide strip: 0/0 = NA, as in R
            strips.push_back(StripDesc{(int)seams[k].a, (int)seams[k].b, seams[k].axis_y, 0, w, origin, res, cmin, delta});
        }
        if (!strips.empty()) MHS_HIP(hipMemcpyAsync(st_dev, strips.data(), sizeof(StripDesc) * strips.size(), hipMemcpyHostToDevice, s));
    }
    dim3 grid((unsigned)((g->ncol + MOSAIC_PC - 1) / MOSAIC_PC), (unsigned)((row_hi - row_lo + MOSAIC_PR - 1) / MOSAIC_PR));
    MHS_REQUIRE(grid.y <= 65535u, "too many rows for one launch");
    hipLaunchKernelGGL(mosaic_fused_kernel, grid, dim3(256), 0, s, td_dev, (int)n, st_dev, (int)strips.size(), row_lo, row_hi, g->ncol, out_dev, ld);
    MHS_HIP(hipGetLastError());
    MHS_HIP(hipStreamSynchronize(s));  // the arena (and the host vectors the copies read) are free for the next call on return
    return MHS_OK;
}

extern "C" int mhs_mosaic_feather_dev(const mhs_grid *g, int64_t nRx, int64_t nCx, const int64_t *tile_win,
                                      const double *const *tile_dev, int merge_mode, double *out_dev, int64_t ld,
                                      int64_t *seam_win_out, void *stream) {
    return mosaic_feather_impl(g, nRx, nCx, tile_win, tile_dev, merge_mode, out_dev, ld, seam_win_out, stream, false, 0, -1);
}

extern "C" {
int mhs_mosaic_feather(const mhs_grid *g, int64_t nRx, int64_t nCx, const int64_t *tile_win,
                       const double *const *tile_host, int merge_mode, double *out_host) {
    if (int rc = require_ready()) return rc;
    if (int rc = check_grid(g)) return rc;
    MHS_REQUIRE(nRx >= 1 && nCx >= 1 && tile_win && tile_host && out_host, "bad arguments");
    const int64_t n = nRx * nCx;
    // tiles and the merged plane live in the library's persistent arena (no hipMalloc / hipFree per call)
    std::vector<size_t> off((size_t)n + 1, 0);
    for (int64_t h = 0; h < n; ++h) {
        const int64_t cells = (tile_win[4 * h + 1] - tile_win[4 * h]) * (tile_win[4 * h + 3] - tile_win[4 * h + 2]);
        MHS_REQUIRE(cells > 0 && tile_host[h], "bad tile");
        off[(size_t)h + 1] = off[(size_t)h] + (((size_t)cells * sizeof(double) + 255) & ~(size_t)255);
    }
    const size_t out_bytes = sizeof(double) * (size_t)(g->nrow * g->ncol);
    std::lock_guard<std::mutex> lk(pipe_mutex());
    if (int rc = host_pipe(off[(size_t)n] + out_bytes)) return rc;
    hipStream_t s = ctx().pipe_comp;
    char *base = ctx().pipe_arena;
    std::vector<const double *> ptrs((size_t)n);
    for (int64_t h = 0; h < n; ++h) {
        const int64_t cells = (tile_win[4 * h + 1] - tile_win[4 * h]) * (tile_win[4 * h + 3] - tile_win[4 * h + 2]);
        MHS_HIP(hipMemcpyAsync(base + off[(size_t)h], tile_host[h], sizeof(double) * (size_t)cells, hipMemcpyHostToDevice, s));
        ptrs[(size_t)h] = (const double *)(base + off[(size_t)h]);
    }
    double *out = (double *)(base + off[(size_t)n]);
    if (int rc = mhs_mosaic_feather_dev(g, nRx, nCx, tile_win, ptrs.data(), merge_mode, out, g->ncol, nullptr, s)) return rc;
    MHS_HIP(hipMemcpyAsync(out_host, out, out_bytes, hipMemcpyDeviceToHost, s));
    MHS_HIP(hipStreamSynchronize(s));
    return MHS_OK;
}

int mhs_seam_count(int64_t nRx, int64_t nCx, int64_t *n_seams) {
    MHS_REQUIRE(nRx >= 1 && nCx >= 1 && n_seams, "bad arguments");
    *n_seams = (int64_t)seam_list(nRx, nCx).size();
    return MHS_OK;
}

int mhs_gather_cells_dev(const double *plane_dev, int64_t ld, const int64_t *rows, const int64_t *cols,
                         int64_t n, double *out_host, void *stream) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(plane_dev && rows && cols && out_host && n >= 0, "bad arguments");
    if (n == 0) return MHS_OK;
    hipStream_t s = pick_stream(stream);
    DevBuf<int64_t> dr, dc;
    DevBuf<double> dout;
    MHS_HIP(dr.alloc((size_t)n)); MHS_HIP(dc.alloc((size_t)n)); MHS_HIP(dout.alloc((size_t)n));
    MHS_HIP(hipMemcpyAsync(dr.p, rows, sizeof(int64_t) * n, hipMemcpyHostToDevice, s));
    MHS_HIP(hipMemcpyAsync(dc.p, cols, sizeof(int64_t) * n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(gather_kernel, dim3(nblk(n)), dim3(256), 0, s, plane_dev, ld, dr.p, dc.p, n, dout.p);
    MHS_HIP(hipGetLastError());
    MHS_HIP(hipMemcpyAsync(out_host, dout.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    MHS_HIP(hipStreamSynchronize(s));
    return MHS_OK;
}

/* terra::cellFromXY for the stations: rows/cols (-1 outside), host only */
int mhs_cells_from_xy(const mhs_grid *g, const double *xy, int64_t n, int64_t *rows, int64_t *cols) {
    if (int rc = check_grid(g)) return rc;
    MHS_REQUIRE((xy || n == 0) && rows && cols && n >= 0, "bad arguments");
    for (int64_t i = 0; i < n; ++i) {
        const double x = xy[i], y = xy[n + i];
        const int64_t c = std::isnan(x) ? -1 : col_from_x(*g, x);
        const int64_t r = std::isnan(y) ? -1 : row_from_y(*g, y);
        rows[i] = (c < 0 || r < 0) ? -1 : r;
        cols[i] = (c < 0 || r < 0) ? -1 : c;
    }
    return MHS_OK;
}

}  // extern "C"
