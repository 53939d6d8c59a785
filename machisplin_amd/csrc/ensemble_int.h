// Internal to the ensemble translation units (ensemble.hip: lm, nnet, earth, ksvm, gbm, the member sequence and the C entry
// points; forest.hip: randomForest): the model handle, the window / stack descriptions every member kernel takes, the
// rank-key lookup gbm and randomForest share, and the host helpers that build their geometry-dependent tables.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <functional>
#include <mutex>
#include <vector>
#include "common.h"

enum { K_LM = 0, K_NNET = 1, K_EARTH = 2, K_SVR = 3, K_GBM = 4, K_RF = 5 };

namespace mhs {

constexpr int PMAX = 12;  // predictors supported by the register-resident kernels

struct PredGeom {
    double xmin, ymax, xres, yres;
    int64_t r0, c0;  // window origin in the grid
    int nr, nc;      // window size
    int64_t ld_out;
};

struct StackDev {
    const void *data;
    int C;            // planes
    int dtype;
    int64_t plane_stride, ld;
    double nodata;
    int has_nodata;
    int all_from_planes;  // points mode: every predictor (LONG, LAT too) comes from a plane
};

// 16-byte node record shared by gbm and randomForest walks
struct __attribute__((aligned(16))) Node {
    double val;              // split value, or the prediction at a terminal
    short var;               // 0-based predictor, -1 = terminal
    unsigned short left, right, missing;  // tree-local child indices
};

struct TreeChunk { int first_tree, n_trees, node_begin, node_count; };

}  // namespace mhs

struct mhs_model {
    int kind = -1;
    int p = 0;
    // lm / nnet / earth / svr parameters (device)
    double *dpar = nullptr;
    int *ipar = nullptr;
    int n0 = 0, n1 = 0, n2 = 0;   // nnet: size ; earth: nterms, nfactors ; svr: SVs kept, row stride, SVs with alpha > 0
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;  // nnet: y_scale,y_shift ; svr: b, sigma, y_center, y_scale, max|alpha|
    // trees
    mhs::Node *nodes = nullptr;
    int *tree_off = nullptr;       // n_trees + 1 node offsets
    mhs::TreeChunk *chunks = nullptr;
    int n_trees = 0, n_chunks = 0, max_chunk_nodes = 0;
    int64_t n_nodes = 0;
    double init_f = 0;
    bool lds_ok = true;
    double *split_scratch = nullptr;     // device, partial tree sums of the few-cells path (launch_trees)
    // gbm predicate-LUT fast path (trees with <= 6 splits): see gbm_lut_kernel
    int lut_S = 0;                       // splits per tree after padding (0 = path unavailable)
    double *lut = nullptr;               // device, n_trees_padded << lut_S leaf values
    int *lut_meta = nullptr;             // device, 12 dwords per tree: c[6] (float bits), key offset[6]
    void *lut_sorted = nullptr;          // device, sorted distinct key-space thresholds, predictor after predictor
                                         // (float keys for float32 / int16 planes, double keys for float64 planes)
    int *lut_sorted_off = nullptr;       // device, p + 1 offsets into lut_sorted
    int *axis_rank = nullptr;            // device, the LONG rank of every grid column, then the LAT rank of every grid row (publish_axis_ranks)
    int axis_ncol = 0;
    double *lut_rt = nullptr;            // device, the same leaf values with every tree's levels ordered uniform-first
    int *lut_rt_meta = nullptr;          // device, LUT_RT_DW dwords per tree (gbm_lutreg_rt_kernel)
    unsigned *lut_cls = nullptr;         // device, 5 class words per tree (rank threshold << 3 | predictor; gbm_coherent_kernel)
    int *gbm_probe = nullptr;            // device, GBC_PROBE_SLOTS x 8 ints: per launch, what the probe blocks summed
    std::atomic<unsigned> gbm_probe_next{0};
    // NA cells of a window, compacted for the MissingNode walk (round 4): 4 buffers in turn, {count, overflow, cell indices ...}
    unsigned *na_list[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t na_cap[4] = {0, 0, 0, 0};
    hipEvent_t na_done[4] = {nullptr, nullptr, nullptr, nullptr};   // recorded behind the last kernel that reads the buffer
    hipStream_t na_stream[4] = {nullptr, nullptr, nullptr, nullptr}; // ... on this stream
    std::atomic<unsigned> na_next{0};
    std::vector<double> lut_host;        // host copy of lut (the row-tile tables are permutations of it)
    std::vector<int> lut_var;            // host, n_trees x lut_S (-1 = padding)
    std::vector<double> lut_thr;         // host, n_trees x lut_S split values
    int n_trees_padded = 0;
    mhs_grid meta_grid = {0, 0, 0, 0, 0, 0};  // geometry lut_meta / rf_nodes were built for
    int meta_C = -1;
    int meta_key64 = -1;                 // key type lut_meta / rf_nodes were built for (1 = double keys)
    // The geometry-dependent tables above are IMMUTABLE once built: a rebuild (another grid, another plane type)
    // allocates fresh buffers and retires the old ones until mhs_model_free, so kernels already enqueued on any
    // stream keep reading what they were launched with; `mu` serialises rebuilds from several host threads.
    std::vector<void *> retired;
    std::mutex mu;
    // randomForest level-synchronous walk (rf_walk_kernel): available when every split node has
    // rightDaughter == leftDaughter + 1 (how randomForest numbers its nodes)
    bool rf_fast = false;
    unsigned long long *rf_nodes = nullptr;  // device, 8-byte records {(rank << 8) | key offset; left | right << 16}
    double *rf_lval = nullptr;               // device, node prediction by node id
    int *rf_depth = nullptr;                 // device, levels to descend per tree
    int *rf_dmin = nullptr;                  // device, depth of every tree's shallowest terminal node
    std::vector<double> rf_thr;              // host, split value per node
    std::vector<unsigned short> rf_left, rf_right, rf_var;  // host, per node (var 0xFFFF = terminal; a terminal's children are itself)
    int rf_max_nodes = 0;
    int rf_max_depth = 0;                    // deepest tree (rf_walk_ld_kernel packs a tree's level counts in 6 bits each)
    int rf_log2r = -1;                       // walks per lane rf_nodes were built for
    int rf_form = 0;                         // ... and in which form: RF_SMALL (16-bit byte addresses, predictions in LDS),
                                             // RF_BIG (node indices, predictions in global memory), RF_COMPACT (split nodes only)
    std::vector<int> rf_off;                 // host, n_trees + 1 node offsets
    int *rf_coff = nullptr;                  // device, n_trees + 1 record offsets of the COMPACT form
    int rf_cmax = 0;                         // COMPACT: most records in a tree (its split nodes + 1)
    int *rf_csub = nullptr;                  // device, per COMPACT record: {split nodes in the subtree below it (itself included),
                                             // first terminal of that subtree (terminals numbered in node order within the tree)}
    double *rf_clval = nullptr;              // device, COMPACT: the terminals' predictions, tree after tree in that numbering
    int rf_compact_ok = 0;                   // every tree's leaf codes fit 16 bits (8 * splits + nodes <= 65535)
    // Several device slots (mhs_init_devices): the buffers above live on ONE device.  The handle remembers the loader
    // call that built it (with copies of its flat arrays) and the multi-device drivers build a replica per slot on
    // first use (model_on_slot); replicas are owned by the handle and freed with it.
    int slot = 0, device = -1;               // where the buffers above live
    std::function<int(mhs_model **)> reload;
    mhs_model *replica[mhs::MAX_SLOTS] = {};
};

namespace mhs {

__device__ __forceinline__ double load_plane(const StackDev &s, int k, int64_t row, int64_t col) {
    const int64_t idx = (int64_t)k * s.plane_stride + row * s.ld + col;
    double v;
    if (s.dtype == MHS_F64) v = ((const double *)s.data)[idx];
    else if (s.dtype == MHS_F32) v = (double)((const float *)s.data)[idx];
    else v = (double)((const short *)s.data)[idx];
    if (s.has_nodata && v == s.nodata) v = NAN;
    return v;
}

// predictor k of the cell at window position (row, col): rast_stack layer order
__device__ __forceinline__ double predictor(const StackDev &s, const PredGeom &g, int k, int row, int col) {
    const int64_t ar = g.r0 + row, ac = g.c0 + col;
    if (k < s.C || s.all_from_planes) return load_plane(s, k, ar, ac);
    if (k == s.C) return g.xmin + ((double)ac + 0.5) * g.xres;  // LONG (V73:131-133)
    return g.ymax - ((double)ar + 0.5) * g.yres;                // LAT  (V73:128-130)
}

__device__ __forceinline__ void emit(double *out, int64_t idx, double pred, double weight, int accumulate) {
    double v = pred * weight;
    if (accumulate) v = out[idx] + v;
    out[idx] = v;
}


// ---- the rank keys of gbm_lut_kernel / the forest walks (see the comment above gbm_lut_kernel in ensemble.hip)
constexpr int LUT_R = 4;            // cells per lane
constexpr int LUT_CHUNK = 64;       // trees per LDS chunk
constexpr int LUT_META_DW = 12;     // dwords of meta per tree: c[6] (float), key offset[6]
constexpr int LUT_COARSE = 4096;    // floats of the coarse rank table (aliases the LUT chunk buffer)

typedef float float2v __attribute__((ext_vector_type(2)));

// rank[c] = #{sorted distinct tkeys of predictor j that are <= key[c]} for the lane's LUT_R cells: a
// binary search of a coarse table (every stride-th tkey, staged in LDS by the whole block) and a
// short fine search in global memory.  Must be called by every thread of the block.
// KT = float: planes whose values are exactly float-representable (float32 / int16); KT = double: float64 planes
// (what terra holds in RAM and the R shim hands over, V73:468-606) -- the search is 1 % of a tree kernel, so doing
// it in double costs nothing and the ranks that come out are the same small integers either way.
template <int LUT_R, int NT, typename KT>
__device__ __forceinline__ void lut_ranks_t(const int j, const KT *__restrict__ sorted,
                                            const int *__restrict__ sorted_off, KT *coarse,
                                            const StackDev &s, const PredGeom &g, const int (&row)[LUT_R],
                                            const int (&col)[LUT_R], bool (&na)[LUT_R], float (&rank)[LUT_R]) {
    constexpr int COARSE_N = LUT_COARSE * (int)sizeof(float) / (int)sizeof(KT);   // the scratch is LUT_COARSE floats
    const int o = sorted_off[j], n = sorted_off[j + 1] - o;
    const KT *T = sorted + o;
    const int stride = (n + COARSE_N - 1) / COARSE_N;
    const int nc = stride ? (n + stride - 1) / stride : 0;
    __syncthreads();
    for (int e = threadIdx.x; e < nc; e += (NT ? NT : (int)blockDim.x)) coarse[e] = T[(int64_t)e * stride];
    __syncthreads();
    KT k[LUT_R];
    int lo[LUT_R], cnt[LUT_R];
#pragma unroll
    for (int c = 0; c < LUT_R; ++c) {
        if (j < s.C) { const double xv = load_plane(s, j, g.r0 + row[c], g.c0 + col[c]); na[c] |= isnan(xv); k[c] = (KT)xv; }
        else if (j == s.C) k[c] = (KT)(g.c0 + col[c]);
        else k[c] = -(KT)(g.r0 + row[c]);
        lo[c] = 0; cnt[c] = 0;
    }
    int top = 1;
    while (top < nc) top <<= 1;
    for (int st = top; st > 0; st >>= 1) {      // lo = #{coarse <= k}
#pragma unroll
        for (int c = 0; c < LUT_R; ++c) {
            const int mid = lo[c] + st;
            if (mid <= nc && coarse[mid - 1] <= k[c]) lo[c] = mid;
        }
    }
    int ftop = 1;
    while (ftop < stride) ftop <<= 1;
    for (int st = ftop >> 1; st > 0; st >>= 1) {  // cnt = #{T in (base, base + stride) <= k}, base = (lo-1) stride
#pragma unroll
        for (int c = 0; c < LUT_R; ++c) {
            const int base = (lo[c] - 1) * stride;
            const int mid = cnt[c] + st;
            if (lo[c] > 0 && mid < stride && base + mid < n && T[base + mid] <= k[c]) cnt[c] = mid;
        }
    }
#pragma unroll
    for (int c = 0; c < LUT_R; ++c) rank[c] = (float)(lo[c] > 0 ? (lo[c] - 1) * stride + 1 + cnt[c] : 0);
}
template <int LUT_R, int NT>
__device__ __forceinline__ void lut_ranks(const int j, const void *__restrict__ sorted, const int key64,
                                          const int *__restrict__ sorted_off, float *coarse,
                                          const StackDev &s, const PredGeom &g, const int (&row)[LUT_R],
                                          const int (&col)[LUT_R], bool (&na)[LUT_R], float (&rank)[LUT_R]) {
    if (key64) lut_ranks_t<LUT_R, NT, double>(j, (const double *)sorted, sorted_off, (double *)coarse, s, g, row, col, na, rank);
    else lut_ranks_t<LUT_R, NT, float>(j, (const float *)sorted, sorted_off, coarse, s, g, row, col, na, rank);
}

// ---- host side: device copies, key-space thresholds, the geometry-dependent tables
template <typename T>
static int to_device(const T *h, size_t n, T **d) {
    MHS_HIP(hipMalloc((void **)d, sizeof(T) * (n ? n : 1)));
    if (n) MHS_HIP(hipMemcpy(*d, h, sizeof(T) * n, hipMemcpyHostToDevice));
    return MHS_OK;
}

constexpr int TREE_R = 2;
constexpr size_t LDS_LIMIT = 150 * 1024;     // of the 160 KiB per CU
constexpr size_t LDS_MAX = 160 * 1024;       // all of it (one block per CU)
constexpr int GBM_CHUNK_NODES = 1024;        // 16 KiB of node records per chunk

static float ceil_to_float(double thr) {  // smallest float >= thr
    float f = (float)thr;
    if ((double)f < thr) f = nextafterf(f, INFINITY);
    return f;
}
static float floor_to_float(double thr) {  // largest float <= thr
    float f = (float)thr;
    if ((double)f > thr) f = nextafterf(f, -INFINITY);
    return f;
}

// Key-space threshold of a split for this grid:  x < thr (gbm, LE = false)  or  x <= thr (randomForest, LE = true)
// <=>  key < tkey  EXACTLY, with key = the plane value as KT (float for float32 / int16 planes, which hold nothing
// but float-representable values; double for float64 planes), the column index for LONG, minus the row index for
// LAT (thresholds converted with the same double formula the kernels use for the cell centres).
template <typename KT, bool LE>
static KT split_tkey(int v, int C, double thr, const mhs_grid &grid) {
    KT tk;
    if (v < C) {
        if constexpr (sizeof(KT) == 4) tk = LE ? nextafterf(floor_to_float(thr), INFINITY) : ceil_to_float(thr);
        else tk = LE ? nextafter(thr, (double)INFINITY) : thr;
    } else if (v == C) {  // LONG: columns whose centre is < (<=) thr form a prefix [0, c*)
        int64_t lo = 0, hi = grid.ncol;
        while (lo < hi) {
            const int64_t mid = (lo + hi) / 2;
            const double x = grid.xmin + ((double)mid + 0.5) * grid.xres;
            if (LE ? x <= thr : x < thr) lo = mid + 1; else hi = mid;
        }
        tk = (KT)lo;
    } else {              // LAT: rows whose centre is < (<=) thr form a suffix [r*, nrow)
        int64_t lo = 0, hi = grid.nrow;
        while (lo < hi) {
            const int64_t mid = (lo + hi) / 2;
            const double y = grid.ymax - ((double)mid + 0.5) * grid.yres;
            if (LE ? y <= thr : y < thr) hi = mid; else lo = mid + 1;
        }
        tk = (KT)0.5 - (KT)lo;
    }
    if (tk != tk) tk = (KT)INFINITY;   // a NaN split value never sends a cell left or right by "<"
    return tk;
}

static bool same_meta(const mhs_model *m, const mhs_grid &grid, int C, int key64) {
    const mhs_grid &o = m->meta_grid;
    return m->meta_C == C && m->meta_key64 == key64 && o.xmin == grid.xmin && o.ymax == grid.ymax && o.xres == grid.xres &&
           o.yres == grid.yres && o.nrow == grid.nrow && o.ncol == grid.ncol;
}

enum { RF_SMALL = 0, RF_BIG = 1, RF_COMPACT = 2 };   // forms of the randomForest node records (build_rf_nodes_t)

// what a tree kernel launch reads of the geometry-dependent tables (a snapshot taken under the model's mutex)
struct TreeTables { const void *sorted; const int *sorted_off; const int *lut_meta; const unsigned long long *rf_nodes; const int *rf_coff;
                    const double *lut_rt; const int *lut_rt_meta; const unsigned *lut_cls; const int *axis_rank = nullptr; int axis_ncol = 0;
                    const int *rf_csub = nullptr; const double *rf_clval = nullptr; };

// fresh device copy of a host table; the buffer it replaces is retired, not freed (kernels in flight may read it)
template <typename T>
static int publish(mhs_model *m, const std::vector<T> &h, T **slot) {
    T *d = nullptr;
    if (int rc = to_device(h.data(), h.size(), &d)) return rc;
    if (*slot) m->retired.push_back((void *)*slot);
    *slot = d;
    return MHS_OK;
}

// sorted distinct key-space thresholds per predictor, flattened; returns the per-predictor lists for the rank lookup
template <typename KT>
static void sort_unique(std::vector<std::vector<KT>> &sorted, std::vector<int> &off, std::vector<KT> &flat) {
    off.assign(sorted.size() + 1, 0);
    for (size_t v = 0; v < sorted.size(); ++v) {
        std::vector<KT> &sv = sorted[v];
        std::sort(sv.begin(), sv.end());
        sv.erase(std::unique(sv.begin(), sv.end()), sv.end());
        off[v + 1] = off[v] + (int)sv.size();
        flat.insert(flat.end(), sv.begin(), sv.end());
    }
    if (flat.empty()) flat.push_back((KT)0);
}

// Round 4: the ranks of the two coordinate predictors by table.  LONG's key is the grid column and LAT's minus the grid row
// (lut_ranks_t), so rank = #{thresholds <= key} is a function of the column / of the row alone: one table entry per grid
// column, then one per grid row, instead of a coarse-table staging, two barriers and ~18 search steps per cell and predictor
// (two of cfg3's five predictors; the rank keys are 9 % of the forest kernel and 20 % of the coherent gbm kernel).
template <typename KT>
static int publish_axis_ranks(mhs_model *m, const std::vector<std::vector<KT>> &sorted, int C, const mhs_grid &grid) {
    if (m->p != C + 2 || grid.ncol <= 0 || grid.nrow <= 0 || (int64_t)grid.ncol + grid.nrow > (1 << 26)) {
        if (m->axis_rank) m->retired.push_back((void *)m->axis_rank);
        m->axis_rank = nullptr; m->axis_ncol = 0;
        return MHS_OK;
    }
    std::vector<int> ar((size_t)grid.ncol + (size_t)grid.nrow);
    const std::vector<KT> &sl = sorted[(size_t)C], &st = sorted[(size_t)C + 1];
    for (int64_t c = 0; c < grid.ncol; ++c) ar[(size_t)c] = (int)(std::upper_bound(sl.begin(), sl.end(), (KT)c) - sl.begin());
    for (int64_t r = 0; r < grid.nrow; ++r) ar[(size_t)grid.ncol + (size_t)r] = (int)(std::upper_bound(st.begin(), st.end(), -(KT)r) - st.begin());
    m->axis_ncol = (int)grid.ncol;
    return publish(m, ar, &m->axis_rank);
}


// ensemble.hip
int finish_trees(mhs_model *m, const std::vector<Node> &nodes, const std::vector<int> &off);
int check_common(int p, mhs_model **out);
// forest.hip: randomForest on a grid window with the level-synchronous walk kernels; *launched = false: none of them applies
// (points mode, trees too large, MHS_TREES_GENERIC) and the caller falls back to the generic node walk
int launch_forest(const mhs_model *m, const StackDev &s, const PredGeom &g, const mhs_grid *grid, double weight, int accumulate,
                  double *out, hipStream_t st, int64_t total, bool *launched);

}  // namespace mhs
