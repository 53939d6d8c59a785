// Thin-plate-spline evaluation on gfx950: predict.Krig over every cell centre of a
// raster window (terra::interpolate, V73:726,753) and over arbitrary points.
//
//   f(x,y) = d0 + d1 u + d2 v + sum_j c_j (1/8pi) 0.5 log(r2) r2 ,  r2 = |(u,v)-(u_j,v_j)|^2
//
// Regime: FP64 vector-ALU bound (8 N flop per cell against 8 B written), not HBM
// bound.  Layout of the work:
//   * a wave owns 64 consecutive columns x ROWS consecutive rows; lane = column, so
//     dx and dx^2 are shared by the lane's ROWS cells and the stores are coalesced;
//   * knots are wave-uniform, so they arrive through the scalar cache (s_load) and
//     cost no VALU or LDS bandwidth;
//   * log(r2) is computed in-kernel from a 1024-entry {1/c, log c} table staged in
//     LDS: r2 = 2^e m, log r2 = e ln2 + log c_i + log1p(m/c_i - 1), |m/c_i - 1| <= 2^-11,
//     cubic log1p => absolute error < 2e-14 (ocml's log costs ~70 FP64 instructions).
//     Neighbouring lanes see neighbouring r2, so table reads mostly broadcast.
//
// Large windows take the FAR-FIELD-INTERPOLATED path (tps_ff_*): the window is cut into
// tiles that are square in the spline's scaled coordinates; for a tile, knots outside the
// 3 x 3 block of tiles around it are at least 1.5 tile widths from its centre, where
// sum_j c_j phi(r_j) is analytic and a 16 x 16 tensor Chebyshev interpolant reproduces it to
// FP64 rounding (measured 1.5e-15 of sum_j |c_j phi_j| with every far knot on the block's
// boundary).  So the far knots are summed at the tile's 256 nodes only (one wave per tile, the
// same inner loop), each cell gets  Ly F Lx'  (32 FMAs) plus the direct sum over the few knots of
// the 3 x 3 block.  Same result as the direct sum to rounding, ~100x fewer kernel evaluations at
// 5 000 knots on a 10 000 x 10 000 grid.  mhs_tps_eval_mode() forces either path.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "common.h"
#include "devmath.h"
#include "tps_batch.h"

namespace mhs {

constexpr int EVAL_ROWS = 4;       // rows per lane
constexpr int EVAL_WAVES = 4;      // waves per block, stacked along rows
constexpr int EVAL_TILE_ROWS = EVAL_ROWS * EVAL_WAVES;

struct EvalGeom {
    double xmin, ymax, xres, yres;  // grid affine
    double cx, cy, sx, sy;          // fields transform (x.center, x.scale)
    double d0, d1, d2;
    int64_t r0, c0;                 // window origin in the grid
    int nr, nc;                     // window size
    int64_t ld;                     // output leading dimension
};

// acc[k] += sum_{j in [j0, j1)} cw_j phi(|(u, v_k) - knot_j|^2): the inner loop of every grid kernel
__device__ __forceinline__ void tps_accumulate(const Knot *__restrict__ knots, const int j0, const int j1,
                                               const double u, const double (&v)[EVAL_ROWS],
                                               double (&acc)[EVAL_ROWS], const double2 *tab) {
#pragma unroll 2
    for (int j = j0; j < j1; ++j) {
        const Knot kn = knots[j];
        const double dx = u - kn.u;
        const double dx2 = dx * dx;
#pragma unroll
        for (int k = 0; k < EVAL_ROWS; ++k) {
            const double dy = v[k] - kn.v;
            const double dd = fma(dy, dy, dx2);
            acc[k] = fma(kn.cw, r2logr2(dd, tab), acc[k]);
        }
    }
}

// one block of the direct sum: 64 columns x 16 rows of the window, block (bx, by); tab is the staged table
__device__ __forceinline__ void eval_direct_block(const Knot *__restrict__ knots, int n, const double2 *tab, const EvalGeom &g,
                                                  double *__restrict__ out, int bx, int by) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int col = bx * 64 + lane;
    const int row0 = by * EVAL_TILE_ROWS + wave * EVAL_ROWS;
    if (row0 >= g.nr) return;

    // cell centre -> scaled coordinates, same operation order as the host/oracle
    const double x = g.xmin + ((double)(g.c0 + col) + 0.5) * g.xres;
    const double u = (x - g.cx) / g.sx;
    double v[EVAL_ROWS], acc[EVAL_ROWS];
#pragma unroll
    for (int k = 0; k < EVAL_ROWS; ++k) {
        const double y = g.ymax - ((double)(g.r0 + row0 + k) + 0.5) * g.yres;
        v[k] = (y - g.cy) / g.sy;
        acc[k] = 0.0;
    }

    tps_accumulate(knots, 0, n, u, v, acc, tab);

    if (col < g.nc) {
#pragma unroll
        for (int k = 0; k < EVAL_ROWS; ++k)
            if (row0 + k < g.nr)
                out[(int64_t)(row0 + k) * g.ld + col] = g.d0 + g.d1 * u + g.d2 * v[k] + acc[k];
    }
}

__global__ __launch_bounds__(64 * EVAL_WAVES) void tps_eval_grid_kernel(
    const Knot *__restrict__ knots, int n, const double2 *__restrict__ gtab, EvalGeom g,
    double *__restrict__ out) {
    __shared__ double2 tab[LOG_TAB_N];
    stage_log_table(tab, gtab);
    eval_direct_block(knots, n, tab, g, out, blockIdx.x, blockIdx.y);
}

__global__ __launch_bounds__(256) void tps_eval_points_kernel(
    const Knot *__restrict__ knots, int n, const double2 *__restrict__ gtab, EvalGeom g,
    const double *__restrict__ px, const double *__restrict__ py, int64_t npts,
    double *__restrict__ out) {
    __shared__ double2 tab[LOG_TAB_N];
    stage_log_table(tab, gtab);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t ii = i < npts ? i : npts - 1;
    const double u = (px[ii] - g.cx) / g.sx;
    const double v = (py[ii] - g.cy) / g.sy;
    double acc = 0.0;
    for (int j = 0; j < n; ++j) {
        const Knot kn = knots[j];
        const double dx = u - kn.u, dy = v - kn.v;
        const double dd = fma(dy, dy, dx * dx);
        acc = fma(kn.cw, r2logr2(dd, tab), acc);
    }
    if (i < npts) out[i] = g.d0 + g.d1 * u + g.d2 * v + acc;
}

// ------------------------------------------------ far-field-interpolated evaluation --
constexpr int FF_N = 16;                 // Chebyshev nodes per dimension
constexpr int FF_NODES = FF_N * FF_N;    // per tile
constexpr int FF_PAD = 2;                // bins beyond the window on every side (never in a 3 x 3 block)

struct FarGeom {
    int tx, ty, ntx, nty;                // tile size (cells), tiles per direction
    double t[FF_N];                      // Chebyshev points of the first kind on [-1, 1]
    // rows of the window one launch writes, [row_lo, row_hi), and the tile rows that cover them, [ty_lo, ty_hi): the
    // whole window by default; a row band of it when several devices share the window's plan (tps_predict_rows_dev) --
    // tiles, nodes and every cell's arithmetic are then those of the one-piece evaluation, bit for bit
    int row_lo, row_hi, ty_lo, ty_hi;
};

// far-field sum (plus the affine part) at the 16 x 16 nodes of every tile; one wave per tile,
// lane = (node column a, group of 4 node rows); blk = index of the block of four tiles
__device__ __forceinline__ void ff_nodes_block(const Knot *__restrict__ knots, int n, const int *__restrict__ bin_start,
                                               const double2 *tab, const EvalGeom &g, const FarGeom &f,
                                               double *__restrict__ nodes, int blk) {
    const int lane = threadIdx.x & 63;
    const int tile = f.ty_lo * f.ntx + blk * 4 + (threadIdx.x >> 6);
    if (tile >= f.ty_hi * f.ntx) return;
    const int tyi = tile / f.ntx, txi = tile - tyi * f.ntx;
    const int a = lane & 15, bg = lane >> 4;
    const double hx = 0.5 * (double)f.tx * g.xres, hy = 0.5 * (double)f.ty * g.yres;
    const double xc = g.xmin + (double)(g.c0 + (int64_t)txi * f.tx) * g.xres + hx;
    const double yc = g.ymax - (double)(g.r0 + (int64_t)tyi * f.ty) * g.yres - hy;
    const double u = (xc + hx * f.t[a] - g.cx) / g.sx;
    double v[EVAL_ROWS], acc[EVAL_ROWS];
#pragma unroll
    for (int k = 0; k < EVAL_ROWS; ++k) {
        v[k] = (yc + hy * f.t[bg * EVAL_ROWS + k] - g.cy) / g.sy;
        acc[k] = 0.0;
    }
    const int nbx = f.ntx + 2 * FF_PAD;
    int j = 0;
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {        // skip the three bin-row ranges of the 3 x 3 block
        const int *row = bin_start + (int64_t)(tyi + FF_PAD - 1 + r) * nbx + (txi + FF_PAD - 1);
        const int s0 = row[0], e0 = row[3];
        tps_accumulate(knots, j, s0, u, v, acc, tab);
        j = e0;
    }
    tps_accumulate(knots, j, n, u, v, acc, tab);
#pragma unroll
    for (int k = 0; k < EVAL_ROWS; ++k)
        nodes[(int64_t)tile * FF_NODES + (bg * EVAL_ROWS + k) * FF_N + a] = g.d0 + g.d1 * u + g.d2 * v[k] + acc[k];
}

__global__ __launch_bounds__(256) void tps_ff_nodes_kernel(const Knot *__restrict__ knots, int n,
                                                           const int *__restrict__ bin_start,
                                                           const double2 *__restrict__ gtab, EvalGeom g,
                                                           FarGeom f, double *__restrict__ nodes) {
    __shared__ double2 tab[LOG_TAB_N];
    stage_log_table(tab, gtab);
    ff_nodes_block(knots, n, bin_start, tab, g, f, nodes, blockIdx.x);
}

// cells: interpolated far field + direct near field.  A block covers 64 columns x 16 rows of one
// tile (tile widths are multiples of 64, heights of 16); lane = column, 4 rows per lane.  Block (bx, by) of the window.
struct FfCellsShared {
    double2 tab[LOG_TAB_N];
    double sF[FF_NODES];
    double sG[FF_N][64];
};
__device__ __forceinline__ void ff_cells_block(const Knot *__restrict__ knots, const int *__restrict__ bin_start,
                                               const double2 *__restrict__ gtab, const EvalGeom &g, const FarGeom &f,
                                               const double *__restrict__ nodes, const double *__restrict__ lx,
                                               const double *__restrict__ ly, double *__restrict__ out, FfCellsShared &sh,
                                               int bx, int by_in) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int cchunks = f.tx / 64, rchunks = f.ty / EVAL_TILE_ROWS;
    const int txi = bx / cchunks, cc = bx - txi * cchunks;
    const int by = by_in + f.ty_lo * rchunks;
    const int tyi = by / rchunks, rc = by - tyi * rchunks;
    const int tile = tyi * f.ntx + txi;
    const int lcol = cc * 64 + lane;                                  // column within the tile
    const int lrow0 = rc * EVAL_TILE_ROWS + wave * EVAL_ROWS;         // first row within the tile
    const int col = txi * f.tx + lcol;
    const int row0 = tyi * f.ty + lrow0;
    if (tyi * f.ty + rc * EVAL_TILE_ROWS >= f.row_hi || tyi * f.ty + (rc + 1) * EVAL_TILE_ROWS <= f.row_lo ||
        txi * f.tx + cc * 64 >= g.nc) return;   // whole block outside
    for (int i = threadIdx.x; i < LOG_TAB_N; i += 64 * EVAL_WAVES) sh.tab[i] = gtab[i];
    sh.sF[threadIdx.x] = nodes[(int64_t)tile * FF_NODES + threadIdx.x];
    __syncthreads();
    {   // G[b][i] = sum_a F[b][a] Lx[i][a] for the block's 64 columns; this wave: b = 4 wave .. 4 wave + 3
        double lxr[FF_N];
#pragma unroll
        for (int a = 0; a < FF_N; ++a) lxr[a] = lx[(int64_t)lcol * FF_N + a];
#pragma unroll
        for (int k = 0; k < EVAL_ROWS; ++k) {
            const int b = wave * EVAL_ROWS + k;
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < FF_N; ++a) s = fma(sh.sF[b * FF_N + a], lxr[a], s);
            sh.sG[b][lane] = s;
        }
    }
    __syncthreads();
    const double x = g.xmin + ((double)(g.c0 + col) + 0.5) * g.xres;
    const double u = (x - g.cx) / g.sx;
    double v[EVAL_ROWS], acc[EVAL_ROWS];
#pragma unroll
    for (int k = 0; k < EVAL_ROWS; ++k) {
        const double y = g.ymax - ((double)(g.r0 + row0 + k) + 0.5) * g.yres;
        v[k] = (y - g.cy) / g.sy;
        const double *lyr = ly + (int64_t)(lrow0 + k) * FF_N;           // wave-uniform row of Ly
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < FF_N; ++b) s = fma(lyr[b], sh.sG[b][lane], s);
        acc[k] = s;
    }
    const int nbx = f.ntx + 2 * FF_PAD;
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {
        const int *brow = bin_start + (int64_t)(tyi + FF_PAD - 1 + r) * nbx + (txi + FF_PAD - 1);
        tps_accumulate(knots, brow[0], brow[3], u, v, acc, sh.tab);
    }
    if (col < g.nc) {
#pragma unroll
        for (int k = 0; k < EVAL_ROWS; ++k)
            if (row0 + k >= f.row_lo && row0 + k < f.row_hi) out[(int64_t)(row0 + k - f.row_lo) * g.ld + col] = acc[k];
    }
}

__global__ __launch_bounds__(64 * EVAL_WAVES) void tps_ff_cells_kernel(
    const Knot *__restrict__ knots, const int *__restrict__ bin_start, const double2 *__restrict__ gtab,
    EvalGeom g, FarGeom f, const double *__restrict__ nodes, const double *__restrict__ lx,
    const double *__restrict__ ly, double *__restrict__ out) {
    __shared__ FfCellsShared sh;
    ff_cells_block(knots, bin_start, gtab, g, f, nodes, lx, ly, out, sh, blockIdx.x, blockIdx.y);
}

// ---- the same three kernels over MANY windows in one launch each (a tiled Step 3: one window per tile spline, the
// splines' coefficients written on the device by tps_batch.hip).  blockIdx.y = window, blockIdx.x = block within it
// (the grid is as wide as the largest window needs; the others' surplus blocks leave at once).
struct EvalWindow {
    EvalGeom g;                 // d0..d2 are read from d3 on the device
    FarGeom f;
    const Knot *knots;          // far: in bin order; direct: any order
    const int *bin_start;
    double *nodes;
    const double *lx, *ly;
    double *out;
    const double *d3;
    int n, far;
    int node_blocks;            // far: blocks of four tiles
    int cgx, cgy;               // blocks of 64 columns x 16 rows
};
__global__ __launch_bounds__(256) void tps_batch_nodes_kernel(const EvalWindow *__restrict__ W, const double2 *__restrict__ gtab) {
    __shared__ double2 tab[LOG_TAB_N];
    const EvalWindow &D = W[blockIdx.y];
    if (!D.far || (int)blockIdx.x >= D.node_blocks) return;
    stage_log_table(tab, gtab);
    EvalGeom g = D.g;
    g.d0 = D.d3[0]; g.d1 = D.d3[1]; g.d2 = D.d3[2];
    ff_nodes_block(D.knots, D.n, D.bin_start, tab, g, D.f, D.nodes, blockIdx.x);
}
__global__ __launch_bounds__(64 * EVAL_WAVES) void tps_batch_cells_kernel(const EvalWindow *__restrict__ W, const double2 *__restrict__ gtab) {
    __shared__ FfCellsShared sh;
    const EvalWindow &D = W[blockIdx.y];
    if ((int)blockIdx.x >= D.cgx * D.cgy) return;
    const int by = blockIdx.x / D.cgx, bx = blockIdx.x - by * D.cgx;
    if (D.far) {
        ff_cells_block(D.knots, D.bin_start, gtab, D.g, D.f, D.nodes, D.lx, D.ly, D.out, sh, bx, by);
    } else {
        stage_log_table(sh.tab, gtab);
        EvalGeom g = D.g;
        g.d0 = D.d3[0]; g.d1 = D.d3[1]; g.d2 = D.d3[2];
        eval_direct_block(D.knots, D.n, sh.tab, g, D.out, bx, by);
    }
}

int upload_knots(mhs_tps *t) {
    std::vector<Knot> h((size_t)t->n);
    const double k = 0.5 / (8.0 * M_PI);
    for (int64_t j = 0; j < t->n; ++j) {
        h[j].u = t->knots_uv[j];
        h[j].v = t->knots_uv[t->n + j];
        h[j].cw = t->c[j] * k;
        h[j].pad = 0.0;
    }
    t->far.r0 = -1;   // coefficients changed: the far-field plan's sorted knots are stale
    if (t->knots_dev) { (void)hipDeviceSynchronize(); pool_release(t->knots_dev); t->knots_dev = nullptr; }   // re-fitted handle: rare
    t->knots_dev = (Knot *)pool_alloc(sizeof(Knot) * (size_t)(t->n ? t->n : 1));
    if (!t->knots_dev) return MHS_ERR_ALLOC;
    return h2d_sync(t->knots_dev, h.data(), sizeof(Knot) * (size_t)t->n);
}

static EvalGeom make_geom(const mhs_tps *t, const mhs_grid *g, int64_t r0, int64_t r1, int64_t c0,
                          int64_t c1, int64_t ld) {
    EvalGeom e;
    if (g) { e.xmin = g->xmin; e.ymax = g->ymax; e.xres = g->xres; e.yres = g->yres; }
    else { e.xmin = e.ymax = 0; e.xres = e.yres = 1; }
    e.cx = t->center[0]; e.cy = t->center[1]; e.sx = t->scale[0]; e.sy = t->scale[1];
    e.d0 = t->d[0]; e.d1 = t->d[1]; e.d2 = t->d[2];
    e.r0 = r0; e.c0 = c0; e.nr = (int)(r1 - r0); e.nc = (int)(c1 - c0); e.ld = ld;
    return e;
}

static int g_eval_mode = 0;   // 0 auto, 1 direct, 2 far-field-interpolated

template <typename T>
static int grow(T **p, size_t *cap, size_t need) {
    if (*cap >= need && *p) return MHS_OK;
    if (*p) { (void)hipDeviceSynchronize(); pool_release(*p); *p = nullptr; *cap = 0; }      // a plan for another window: rare
    *p = (T *)pool_alloc((need ? need : 1) * sizeof(T));
    if (!*p) return MHS_ERR_ALLOC;
    *cap = need;
    return MHS_OK;
}

// rows of the barycentric Chebyshev interpolation matrix L[i][a] = l_a(tau_i), tau_i the centre of cell
// i of a tile of `cells` cells mapped to (-1, 1); flip: rows run against the coordinate (raster rows)
static void cheb_matrix(int cells, bool flip, const double (&t)[FF_N], std::vector<double> &L) {
    double w[FF_N];
    for (int a = 0; a < FF_N; ++a) w[a] = ((a & 1) ? -1.0 : 1.0) * sin((2 * a + 1) * M_PI / (2.0 * FF_N));
    L.assign((size_t)cells * FF_N, 0.0);
    for (int i = 0; i < cells; ++i) {
        double tau = (2.0 * i + 1.0) / (double)cells - 1.0;
        if (flip) tau = -tau;
        int hit = -1;
        double q[FF_N], sum = 0.0;
        for (int a = 0; a < FF_N; ++a) {
            const double d = tau - t[a];
            if (d == 0.0) { hit = a; break; }
            q[a] = w[a] / d;
            sum += q[a];
        }
        for (int a = 0; a < FF_N; ++a) L[(size_t)i * FF_N + a] = hit >= 0 ? (a == hit ? 1.0 : 0.0) : q[a] / sum;
    }
}

// Host side of the far-field plan, from the knots' coordinates alone (coefficients play no part): tile sizes, and the
// counting sort of the knots by bin.  Shared by the per-handle plan below and the batched evaluation of a tiled Step 3.
// far_plan_choose: *use = false when the direct sum is expected to be at least as cheap (few knots, small windows).
static void far_plan_choose(const double *knots_uv, int64_t N, const EvalGeom &e, FarGeom *f, bool *use,
                            std::vector<double> &kc, std::vector<double> &kr) {
    *use = false;
    if (g_eval_mode == 1 || e.nc < 64 || e.nr < 16 || N < 32) return;
    // knots in window cell coordinates
    kc.resize((size_t)N); kr.resize((size_t)N);
    int64_t inside = 0;
    for (int64_t j = 0; j < N; ++j) {
        const double x = knots_uv[j] * e.sx + e.cx, y = knots_uv[N + j] * e.sy + e.cy;
        kc[j] = (x - e.xmin) / e.xres - (double)e.c0;
        kr[j] = (e.ymax - y) / e.yres - (double)e.r0;
        if (kc[j] >= 0 && kc[j] < e.nc && kr[j] >= 0 && kr[j] < e.nr) ++inside;
    }
    const double cells = (double)e.nr * (double)e.nc;
    const double rho = (double)(inside > 0 ? inside : 1) / cells;          // knots per cell near the window
    const double aspect = (e.xres / e.sx) / (e.yres / e.sy);               // scaled width / height of a cell
    double best = 1e300;
    int btx = 0, bty = 0;
    for (int tx = 64; tx <= 1024; tx += 64) {
        int ty = (int)llround((double)tx * aspect / 16.0) * 16;
        if (ty < 16) ty = 16;
        if (ty > 4096) continue;
        const double ntx = ceil((double)e.nc / tx), nty = ceil((double)e.nr / ty);
        const double cost = ntx * nty * FF_NODES * (double)N / cells + 9.0 * rho * tx * ty + 4.0;
        if (cost < best) { best = cost; btx = tx; bty = ty; }
    }
    if (btx == 0) return;
    if (g_eval_mode != 2 && best > 0.5 * (double)N) return;
    const double tile_aspect = ((double)btx * e.xres / e.sx) / ((double)bty * e.yres / e.sy);
    if (tile_aspect > 1.25 || tile_aspect < 0.8) return;                   // cannot make square tiles: direct
    f->tx = btx; f->ty = bty;
    f->ntx = (int)((e.nc + btx - 1) / btx); f->nty = (int)((e.nr + bty - 1) / bty);
    for (int a = 0; a < FF_N; ++a) f->t[a] = cos((2 * a + 1) * M_PI / (2.0 * FF_N));
    f->row_lo = 0; f->row_hi = e.nr; f->ty_lo = 0; f->ty_hi = f->nty;
    *use = true;
}
// counting sort of the knots by bin; bins beyond FF_PAD tiles outside the window collapse onto the rim.
// start: (nty + 4) * (ntx + 4) + 1 offsets; order[p] = index of the knot at sorted position p.
static void far_plan_sort(int64_t N, const EvalGeom &e, const FarGeom &f, const std::vector<double> &kc,
                          const std::vector<double> &kr, std::vector<int> &start, std::vector<int> &order,
                          int64_t *node_pairs, int64_t *cell_pairs) {
    const int btx = f.tx, bty = f.ty;
    const int nbx = f.ntx + 2 * FF_PAD, nby = f.nty + 2 * FF_PAD;
    std::vector<int> bin((size_t)N);
    start.assign((size_t)nbx * nby + 1, 0);
    for (int64_t j = 0; j < N; ++j) {
        double bx = floor(kc[j] / btx), by = floor(kr[j] / bty);
        bx = std::min(std::max(bx, (double)-FF_PAD), (double)(f.ntx + FF_PAD - 1));
        by = std::min(std::max(by, (double)-FF_PAD), (double)(f.nty + FF_PAD - 1));
        bin[j] = ((int)by + FF_PAD) * nbx + ((int)bx + FF_PAD);
        ++start[(size_t)bin[j] + 1];
    }
    for (size_t b = 0; b + 1 < start.size(); ++b) start[b + 1] += start[b];
    std::vector<int> fill(start.begin(), start.end() - 1);
    order.resize((size_t)N);
    for (int64_t j = 0; j < N; ++j) order[(size_t)fill[(size_t)bin[j]]++] = (int)j;
    int64_t np = 0, cp = 0;
    for (int tyi = 0; tyi < f.nty; ++tyi)
        for (int txi = 0; txi < f.ntx; ++txi) {
            int64_t near = 0;
            for (int r = 0; r < 3; ++r) {
                const size_t b = (size_t)(tyi + FF_PAD - 1 + r) * nbx + (size_t)(txi + FF_PAD - 1);
                near += start[b + 3] - start[b];
            }
            const int64_t tc = std::min<int64_t>(btx, e.nc - (int64_t)txi * btx) * std::min<int64_t>(bty, e.nr - (int64_t)tyi * bty);
            np += (int64_t)FF_NODES * (N - near);
            cp += tc * near;
        }
    if (node_pairs) *node_pairs = np;
    if (cell_pairs) *cell_pairs = cp;
}

// Choose tile sizes for the far-field-interpolated path and (re)build the handle's plan.
static int plan_far(mhs_tps *t, const mhs_grid *g, const EvalGeom &e, int64_t r1, int64_t c1, FarGeom *f, bool *use) {
    const int64_t N = t->n;
    std::vector<double> kc, kr;
    far_plan_choose(t->knots_uv.data(), N, e, f, use, kc, kr);
    if (!*use) return MHS_OK;
    const int btx = f->tx, bty = f->ty;
    mhs_tps::FarPlan &P = t->far;
    const bool same = P.sorted_dev && P.xmin == e.xmin && P.ymax == e.ymax && P.xres == e.xres && P.yres == e.yres &&
                      P.r0 == e.r0 && P.r1 == r1 && P.c0 == e.c0 && P.c1 == c1 && P.tx == btx && P.ty == bty;
    if (same) return MHS_OK;
    // The plan's device buffers are about to be rewritten (blocking copies on the null stream): kernels of the previous
    // evaluation, enqueued on a caller's non-blocking stream, may still be reading them.
    if (P.in_flight) { MHS_HIP(hipStreamSynchronize(P.last_stream)); P.in_flight = false; }
    std::vector<int> start, order;
    far_plan_sort(N, e, *f, kc, kr, start, order, &P.node_pairs, &P.cell_pairs);
    std::vector<Knot> sorted((size_t)N);
    const double kk = 0.5 / (8.0 * M_PI);
    for (int64_t p = 0; p < N; ++p) {
        const int j = order[(size_t)p];
        Knot &k = sorted[(size_t)p];
        k.u = t->knots_uv[j]; k.v = t->knots_uv[N + j]; k.cw = t->c[j] * kk; k.pad = 0.0;
    }
    std::vector<double> lx, ly;
    cheb_matrix(btx, false, f->t, lx);
    cheb_matrix(bty, true, f->t, ly);
    if (!P.sorted_dev) { P.sorted_dev = (Knot *)pool_alloc(sizeof(Knot) * (size_t)N); if (!P.sorted_dev) return MHS_ERR_ALLOC; }
    if (int rc = grow(&P.bin_start_dev, &P.bins_cap, start.size())) return rc;
    if (int rc = grow(&P.nodes_dev, &P.nodes_cap, (size_t)f->ntx * f->nty * FF_NODES)) return rc;
    if (int rc = grow(&P.lx_dev, &P.lx_cap, lx.size())) return rc;
    if (int rc = grow(&P.ly_dev, &P.ly_cap, ly.size())) return rc;
    if (int rc = h2d_sync(P.sorted_dev, sorted.data(), sizeof(Knot) * (size_t)N)) return rc;
    if (int rc = h2d_sync(P.bin_start_dev, start.data(), sizeof(int) * start.size())) return rc;
    if (int rc = h2d_sync(P.lx_dev, lx.data(), sizeof(double) * lx.size())) return rc;
    if (int rc = h2d_sync(P.ly_dev, ly.data(), sizeof(double) * ly.size())) return rc;
    P.xmin = e.xmin; P.ymax = e.ymax; P.xres = e.xres; P.yres = e.yres;
    P.r0 = e.r0; P.r1 = r1; P.c0 = e.c0; P.c1 = c1; P.tx = btx; P.ty = bty; P.ntx = f->ntx; P.nty = f->nty;
    return MHS_OK;
}

// ---- host side of the batched evaluation (declared in tps_batch.h)
struct EvalBatch {
    struct Item {
        EvalWindow w;               // pointers filled at launch
        int64_t knot_off = 0;       // the spline's knot records in the fit batch's output
        int res_index = 0;          // its SmallResult
        size_t bins_off = 0, nodes_off = 0;     // in ints / doubles
        int lxy = -1;               // index into mats
    };
    struct Mats { int tx, ty; std::vector<double> lx, ly; size_t lx_off = 0, ly_off = 0; };
    std::vector<Item> items;
    std::vector<EvalWindow> wh;     // the descriptors as uploaded (kept until the batch is destroyed: the copy is asynchronous)
    std::vector<int> bins;          // all windows' bin offsets
    std::vector<Mats> mats;         // interpolation matrices, one pair per distinct tile size
    size_t nodes_total = 0;
    unsigned max_node_blocks = 0, max_cell_blocks = 0;
};
EvalBatch *eval_batch_create() { return new EvalBatch(); }
void eval_batch_destroy(EvalBatch *B) { delete B; }

// the planning half of eval_batch_add: pure (no shared state), so the tiles of a surface are planned by several host threads
struct EvalPlan { EvalBatch::Item it; std::vector<int> start; bool far = false; };
static int eval_plan_one(const double *knots_uv, int n, const double *center, const double *scale, const mhs_grid *grid,
                         int64_t r0, int64_t r1, int64_t c0, int64_t c1, double *out_dev, int64_t ld, EvalPlan &P, std::vector<int> &perm) {
    MHS_REQUIRE(r1 - r0 < (1LL << 30) && c1 - c0 < (1LL << 30) && r1 > r0 && c1 > c0, "window too large or empty");
    EvalBatch::Item &it = P.it;
    memset(&it.w, 0, sizeof(it.w));
    EvalGeom &e = it.w.g;
    e.xmin = grid->xmin; e.ymax = grid->ymax; e.xres = grid->xres; e.yres = grid->yres;
    e.cx = center[0]; e.cy = center[1]; e.sx = scale[0]; e.sy = scale[1];
    e.d0 = e.d1 = e.d2 = 0.0;
    e.r0 = r0; e.c0 = c0; e.nr = (int)(r1 - r0); e.nc = (int)(c1 - c0); e.ld = ld;
    it.w.out = out_dev; it.w.n = n;
    std::vector<double> kc, kr;
    far_plan_choose(knots_uv, n, e, &it.w.f, &P.far, kc, kr);
    perm.resize((size_t)n);
    if (P.far) {
        std::vector<int> order;
        far_plan_sort(n, e, it.w.f, kc, kr, P.start, order, nullptr, nullptr);
        for (int p = 0; p < n; ++p) perm[(size_t)order[(size_t)p]] = p;
        it.w.far = 1;
        it.w.node_blocks = (it.w.f.ntx * it.w.f.nty + 3) / 4;
        it.w.cgx = it.w.f.ntx * (it.w.f.tx / 64);
        it.w.cgy = it.w.f.nty * (it.w.f.ty / EVAL_TILE_ROWS);
    } else {
        for (int p = 0; p < n; ++p) perm[(size_t)p] = p;
        it.w.far = 0; it.w.node_blocks = 0;
        it.w.cgx = (e.nc + 63) / 64;
        it.w.cgy = (e.nr + EVAL_TILE_ROWS - 1) / EVAL_TILE_ROWS;
    }
    return MHS_OK;
}
static void eval_plan_commit(EvalBatch *B, EvalPlan &P, int res_index, int64_t knot_off) {
    EvalBatch::Item &it = P.it;
    it.knot_off = knot_off; it.res_index = res_index;
    if (P.far) {
        it.bins_off = B->bins.size();
        B->bins.insert(B->bins.end(), P.start.begin(), P.start.end());
        it.nodes_off = B->nodes_total;
        B->nodes_total += (size_t)it.w.f.ntx * it.w.f.nty * FF_NODES;
        it.lxy = -1;
        for (size_t q = 0; q < B->mats.size(); ++q)
            if (B->mats[q].tx == it.w.f.tx && B->mats[q].ty == it.w.f.ty) it.lxy = (int)q;
        if (it.lxy < 0) {
            EvalBatch::Mats m;
            m.tx = it.w.f.tx; m.ty = it.w.f.ty;
            cheb_matrix(m.tx, false, it.w.f.t, m.lx);
            cheb_matrix(m.ty, true, it.w.f.t, m.ly);
            it.lxy = (int)B->mats.size();
            B->mats.push_back(std::move(m));
        }
    }
    B->max_node_blocks = std::max(B->max_node_blocks, (unsigned)it.w.node_blocks);
    B->max_cell_blocks = std::max(B->max_cell_blocks, (unsigned)(it.w.cgx * it.w.cgy));
    B->items.push_back(it);
}
int eval_batch_add(EvalBatch *B, const double *knots_uv, int n, const double *center, const double *scale, const mhs_grid *grid,
                   int64_t r0, int64_t r1, int64_t c0, int64_t c1, double *out_dev, int64_t ld, int res_index, int64_t knot_off,
                   std::vector<int> &perm) {
    EvalPlan P;
    if (int rc = eval_plan_one(knots_uv, n, center, scale, grid, r0, r1, c0, c1, out_dev, ld, P, perm)) return rc;
    eval_plan_commit(B, P, res_index, knot_off);
    return MHS_OK;
}
// two halves for callers that plan many windows side by side: plan (any thread), then commit (one thread, in job order)
EvalPlanHandle *eval_batch_plan(const double *knots_uv, int n, const double *center, const double *scale, const mhs_grid *grid,
                                int64_t r0, int64_t r1, int64_t c0, int64_t c1, double *out_dev, int64_t ld, std::vector<int> &perm, int *rc_out) {
    EvalPlan *P = new EvalPlan();
    const int rc = eval_plan_one(knots_uv, n, center, scale, grid, r0, r1, c0, c1, out_dev, ld, *P, perm);
    if (rc_out) *rc_out = rc;
    if (rc) { delete P; return nullptr; }
    return reinterpret_cast<EvalPlanHandle *>(P);
}
void eval_batch_commit(EvalBatch *B, EvalPlanHandle *h, int res_index, int64_t knot_off) {
    EvalPlan *P = reinterpret_cast<EvalPlan *>(h);
    eval_plan_commit(B, *P, res_index, knot_off);
    delete P;
}
void eval_plan_drop(EvalPlanHandle *h) { delete reinterpret_cast<EvalPlan *>(h); }

static size_t eb_up(size_t x) { return (x + 255) & ~(size_t)255; }
size_t eval_batch_device_bytes(const EvalBatch *B) {
    size_t off = 0;
    off = eb_up(off + sizeof(EvalWindow) * B->items.size());
    off = eb_up(off + sizeof(int) * B->bins.size());
    for (const EvalBatch::Mats &m : B->mats) off = eb_up(off + sizeof(double) * (m.lx.size() + m.ly.size()));
    off = eb_up(off + sizeof(double) * B->nodes_total);
    return off;
}

int eval_batch_launch(EvalBatch *B, char *dev, const Knot *knots_base, const SmallResult *res_base, hipStream_t s) {
    if (B->items.empty()) return MHS_OK;
    size_t off = 0;
    EvalWindow *wdev = (EvalWindow *)(dev + off); off = eb_up(off + sizeof(EvalWindow) * B->items.size());
    int *bins_dev = (int *)(dev + off); off = eb_up(off + sizeof(int) * B->bins.size());
    std::vector<double *> lxd, lyd;
    for (EvalBatch::Mats &m : B->mats) {
        double *p = (double *)(dev + off);
        off = eb_up(off + sizeof(double) * (m.lx.size() + m.ly.size()));
        lxd.push_back(p); lyd.push_back(p + m.lx.size());
        MHS_HIP(hipMemcpyAsync(p, m.lx.data(), sizeof(double) * m.lx.size(), hipMemcpyHostToDevice, s));
        MHS_HIP(hipMemcpyAsync(p + m.lx.size(), m.ly.data(), sizeof(double) * m.ly.size(), hipMemcpyHostToDevice, s));
    }
    double *nodes_dev = (double *)(dev + off);
    std::vector<EvalWindow> &w = B->wh;
    w.resize(B->items.size());
    for (size_t k = 0; k < B->items.size(); ++k) {
        const EvalBatch::Item &it = B->items[k];
        w[k] = it.w;
        w[k].knots = knots_base + it.knot_off;
        w[k].d3 = res_base[it.res_index].d;
        if (it.w.far) {
            w[k].bin_start = bins_dev + it.bins_off;
            w[k].nodes = nodes_dev + it.nodes_off;
            w[k].lx = lxd[(size_t)it.lxy]; w[k].ly = lyd[(size_t)it.lxy];
        }
    }
    MHS_HIP(hipMemcpyAsync(wdev, w.data(), sizeof(EvalWindow) * w.size(), hipMemcpyHostToDevice, s));
    if (!B->bins.empty()) MHS_HIP(hipMemcpyAsync(bins_dev, B->bins.data(), sizeof(int) * B->bins.size(), hipMemcpyHostToDevice, s));
    const unsigned nt = (unsigned)B->items.size();
    MHS_REQUIRE(nt <= 65535u, "too many windows for one launch");
    if (B->max_node_blocks > 0)
        hipLaunchKernelGGL(tps_batch_nodes_kernel, dim3(B->max_node_blocks, nt), dim3(256), 0, s, wdev, ctx().log_tab);
    hipLaunchKernelGGL(tps_batch_cells_kernel, dim3(B->max_cell_blocks, nt), dim3(64 * EVAL_WAVES), 0, s, wdev, ctx().log_tab);
    MHS_HIP(hipGetLastError());
    return MHS_OK;
}

}  // namespace mhs

using namespace mhs;

extern "C" {

int mhs_tps_from_coef(const double *knots_uv, const double *c, const double *d3, int64_t n,
                      double lambda, const double *center2, const double *scale2, mhs_tps **out) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(knots_uv && c && d3 && center2 && scale2 && out, "NULL argument");
    MHS_REQUIRE(n >= 1 && n < (1LL << 31), "n out of range");
    MHS_REQUIRE(scale2[0] > 0 && scale2[1] > 0, "scale must be positive");
    mhs_tps *t = new mhs_tps();
    t->n = n;
    t->lambda = lambda;
    t->eff_df = t->gcv = NAN;
    t->c.assign(c, c + n);
    t->knots_uv.assign(knots_uv, knots_uv + 2 * n);
    memcpy(t->d, d3, sizeof(t->d));
    memcpy(t->center, center2, sizeof(t->center));
    memcpy(t->scale, scale2, sizeof(t->scale));
    if (int rc = upload_knots(t)) { mhs_tps_free(t); return rc; }
    *out = t;
    return MHS_OK;
}

int mhs_tps_size(const mhs_tps *t, int64_t *n) {
    MHS_REQUIRE(t && n, "NULL argument");
    *n = t->n;
    return MHS_OK;
}

int mhs_tps_get(const mhs_tps *t, double *c, double *d3, double *knots_uv, double *lambda,
                double *center2, double *scale2, double *eff_df, double *gcv) {
    MHS_REQUIRE(t != nullptr, "NULL handle");
    if (c) memcpy(c, t->c.data(), sizeof(double) * (size_t)t->n);
    if (d3) memcpy(d3, t->d, sizeof(t->d));
    if (knots_uv) memcpy(knots_uv, t->knots_uv.data(), sizeof(double) * 2 * (size_t)t->n);
    if (lambda) *lambda = t->lambda;
    if (center2) memcpy(center2, t->center, sizeof(t->center));
    if (scale2) memcpy(scale2, t->scale, sizeof(t->scale));
    if (eff_df) *eff_df = t->eff_df;
    if (gcv) *gcv = t->gcv;
    return MHS_OK;
}

// drop a handle whose device buffers nothing in flight reads any more (the caller has synchronised): no device-wide wait
}  // extern "C"
int mhs::tps_free_quiet(mhs_tps *t) {
    if (!t) return MHS_OK;
    for (void *q : {(void *)t->knots_dev, (void *)t->far.sorted_dev, (void *)t->far.bin_start_dev, (void *)t->far.nodes_dev,
                    (void *)t->far.lx_dev, (void *)t->far.ly_dev}) pool_release(q);
    delete t;
    return MHS_OK;
}
extern "C" {

int mhs_tps_free(mhs_tps *t) {
    if (!t) return MHS_OK;
    // evaluations are asynchronous on the caller's streams: ONE device-wide wait (hipFree made six) before the blocks go back
    if (t->knots_dev && ctx().ready) (void)hipDeviceSynchronize();
    return tps_free_quiet(t);
}

int mhs_tps_eval_mode(int mode) {
    MHS_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0 (auto), 1 (direct) or 2 (far-field-interpolated)");
    g_eval_mode = mode;
    return MHS_OK;
}

int mhs_tps_eval_plan(const mhs_tps *t, int *tile_cols, int *tile_rows, int64_t *node_pairs, int64_t *cell_pairs) {
    MHS_REQUIRE(t != nullptr, "NULL handle");
    const mhs_tps::FarPlan &P = t->far;
    if (tile_cols) *tile_cols = P.last_used ? P.tx : 0;
    if (tile_rows) *tile_rows = P.last_used ? P.ty : 0;
    if (node_pairs) *node_pairs = P.last_used ? P.node_pairs : 0;
    if (cell_pairs) *cell_pairs = P.last_used ? P.cell_pairs : 0;
    return MHS_OK;
}

int mhs_tps_predict_grid_dev(const mhs_tps *t, const mhs_grid *g, int64_t r0, int64_t r1,
                             int64_t c0, int64_t c1, double *out_dev, int64_t ld, void *stream) {
    return tps_predict_rows_dev(t, g, r0, r1, c0, c1, r0, r1, out_dev, ld, stream);
}

int mhs_tps_predict_rows_dev(const mhs_tps *t, const mhs_grid *g, int64_t r0, int64_t r1, int64_t c0, int64_t c1,
                             int64_t b0, int64_t b1, double *out_dev, int64_t ld, void *stream) {
    return tps_predict_rows_dev(t, g, r0, r1, c0, c1, b0, b1, out_dev, ld, stream);
}

}  // extern "C"

// Rows [b0, b1) of the window [r0, r1) x [c0, c1), evaluated with the WINDOW's plan (far-field tile size, tile origin,
// path decision): out_dev holds row b0 at offset 0.  The multi-device drivers cut a window into row bands, one per device;
// every band's cells then get exactly the arithmetic of the one-piece evaluation.
int mhs::tps_predict_rows_dev(const mhs_tps *t, const mhs_grid *g, int64_t r0, int64_t r1, int64_t c0, int64_t c1,
                              int64_t b0, int64_t b1, double *out_dev, int64_t ld, void *stream) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(t && g && out_dev, "NULL argument");
    MHS_REQUIRE(g->nrow > 0 && g->ncol > 0 && g->xres > 0 && g->yres > 0, "bad grid geometry");
    MHS_REQUIRE(0 <= r0 && r0 <= r1 && r1 <= g->nrow && 0 <= c0 && c0 <= c1 && c1 <= g->ncol,
                "window outside the grid");
    MHS_REQUIRE(r0 <= b0 && b0 <= b1 && b1 <= r1, "row band outside the window");
    MHS_REQUIRE(ld >= c1 - c0, "ld smaller than the window width");
    MHS_REQUIRE(r1 - r0 < (1LL << 30) && c1 - c0 < (1LL << 30), "window too large");
    if (b1 == b0 || c1 == c0) return MHS_OK;
    const EvalGeom e = make_geom(t, g, r0, r1, c0, c1, ld);
    FarGeom f;
    bool far = false;
    mhs_tps *tm = const_cast<mhs_tps *>(t);
    std::lock_guard<std::mutex> lk(tm->mu);
    if (int rc = plan_far(tm, g, e, r1, c1, &f, &far)) return rc;
    tm->far.last_used = far;
    if (far) {
        mhs_tps::FarPlan &Pm = tm->far;
        // the node values are scratch of ONE evaluation: another stream must not start on them before the last one is done
        if (Pm.in_flight && Pm.last_stream != pick_stream(stream)) MHS_HIP(hipStreamSynchronize(Pm.last_stream));
        Pm.last_stream = pick_stream(stream);
        Pm.in_flight = true;
        const mhs_tps::FarPlan &P = t->far;
        f.row_lo = (int)(b0 - r0); f.row_hi = (int)(b1 - r0);
        f.ty_lo = f.row_lo / f.ty; f.ty_hi = (f.row_hi + f.ty - 1) / f.ty;
        const int ntiles = f.ntx * (f.ty_hi - f.ty_lo);
        dim3 cgrid((unsigned)(f.ntx * (f.tx / 64)), (unsigned)((f.ty_hi - f.ty_lo) * (f.ty / EVAL_TILE_ROWS)));
        MHS_REQUIRE(cgrid.y <= 65535u, "too many rows for one launch");
        hipLaunchKernelGGL(tps_ff_nodes_kernel, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, pick_stream(stream),
                           P.sorted_dev, (int)t->n, P.bin_start_dev, ctx().log_tab, e, f, P.nodes_dev);
        hipLaunchKernelGGL(tps_ff_cells_kernel, cgrid, dim3(64 * EVAL_WAVES), 0, pick_stream(stream), P.sorted_dev,
                           P.bin_start_dev, ctx().log_tab, e, f, P.nodes_dev, P.lx_dev, P.ly_dev, out_dev);
        MHS_HIP(hipGetLastError());
        return MHS_OK;
    }
    // the direct sum does not depend on the window: the band is evaluated as a window of its own
    const EvalGeom eb = make_geom(t, g, b0, b1, c0, c1, ld);
    dim3 grid((unsigned)((eb.nc + 63) / 64), (unsigned)((eb.nr + EVAL_TILE_ROWS - 1) / EVAL_TILE_ROWS));
    // gridDim.y is limited to 65535: 16 rows per block covers > 1e6 rows
    MHS_REQUIRE(grid.y <= 65535u, "too many rows for one launch");
    hipLaunchKernelGGL(tps_eval_grid_kernel, grid, dim3(64 * EVAL_WAVES), 0, pick_stream(stream),
                       t->knots_dev, (int)t->n, ctx().log_tab, eb, out_dev);
    MHS_HIP(hipGetLastError());
    return MHS_OK;
}

extern "C" {

int mhs_tps_predict_grid(const mhs_tps *t, const mhs_grid *g, int64_t r0, int64_t r1, int64_t c0,
                         int64_t c1, double *out_host) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(t && g && out_host, "NULL argument");
    MHS_REQUIRE(0 <= r0 && r0 <= r1 && 0 <= c0 && c0 <= c1, "bad window");
    const int64_t nr = r1 - r0, nc = c1 - c0;
    if (nr == 0 || nc == 0) return MHS_OK;
    // the plane comes from the library's persistent arena (no hipMalloc / hipFree -- a device-wide synchronisation each --
    // per call) and is evaluated on the library's own stream, the one the copy back is ordered on
    std::lock_guard<std::mutex> lk(pipe_mutex());
    if (int rc = host_pipe(sizeof(double) * (size_t)(nr * nc))) return rc;
    double *buf = (double *)ctx().pipe_arena;
    hipStream_t s = ctx().pipe_comp;
    if (int rc = mhs_tps_predict_grid_dev(t, g, r0, r1, c0, c1, buf, nc, s)) return rc;
    MHS_HIP(hipMemcpyAsync(out_host, buf, sizeof(double) * (size_t)(nr * nc), hipMemcpyDeviceToHost, s));
    MHS_HIP(hipStreamSynchronize(s));
    return MHS_OK;
}

int mhs_tps_predict_points(const mhs_tps *t, const double *xy, int64_t n, double *out_host) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(t && out_host && (xy || n == 0), "NULL argument");
    MHS_REQUIRE(n >= 0, "negative n");
    if (n == 0) return MHS_OK;
    DevBuf<double> dxy, dout;
    MHS_HIP(dxy.alloc((size_t)(2 * n)));
    MHS_HIP(dout.alloc((size_t)n));
    hipStream_t s = ctx().stream;
    MHS_HIP(hipMemcpyAsync(dxy.p, xy, sizeof(double) * 2 * (size_t)n, hipMemcpyHostToDevice, s));
    const EvalGeom e = make_geom(t, nullptr, 0, 0, 0, 0, 0);
    hipLaunchKernelGGL(tps_eval_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       t->knots_dev, (int)t->n, ctx().log_tab, e, dxy.p, dxy.p + n, n, dout.p);
    MHS_HIP(hipGetLastError());
    MHS_HIP(hipMemcpyAsync(out_host, dout.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
    MHS_HIP(hipStreamSynchronize(s));
    return MHS_OK;
}

}  // extern "C"
