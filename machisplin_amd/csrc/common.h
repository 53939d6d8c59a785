// Internal declarations shared by the HIP translation units of libmachisplin_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <string>
#include <vector>
#include "machisplin_hip.h"

namespace mhs {

void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define MHS_HIP(call)                                                        \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) return mhs::hip_fail(e_, #call, __FILE__, __LINE__); \
    } while (0)

#define MHS_REQUIRE(cond, msg)                                               \
    do {                                                                     \
        if (!(cond)) { mhs::set_error("%s: %s", __func__, msg); return MHS_ERR_INVALID; } \
    } while (0)

// log(m) table for m in [1,2): 2^LOG_TAB_BITS intervals, entry = {1/c_i, -log(1/c_i)}
constexpr int LOG_TAB_BITS = 10;
constexpr int LOG_TAB_N = 1 << LOG_TAB_BITS;

// Everything one spline fit needs to run independently of another one: two streams (the panel
// factorisations run ahead of the trailing update, tps_fit.hip), a pool of timing-disabled events and a
// grow-only device arena its work buffers are carved from (no hipMalloc / hipFree -- a device-wide
// synchronisation -- per fit).  mhs_tps_fit uses lane 0; mhs_tps_surface fits its tiles on several lanes
// from host threads.
constexpr int FIT_PANEL_CUS = 32;      // compute units the 32-column route's trailing update stays off (a multiple of 8: the same ones in every XCD)
struct FitLane {
    hipStream_t s = nullptr, s2 = nullptr;
    hipStream_t ms = nullptr, ms2 = nullptr;   // the same pair confined to the compute units mhs_fit_reserve_cus keeps free
    hipStream_t s2r = nullptr;                 // s2's work of the 32-column route, kept OFF four compute units of every XCD (runtime.hip)
    std::vector<hipEvent_t> pool;
    char *arena = nullptr;
    size_t arena_cap = 0;
    double *pinned = nullptr;                  // page-locked host scratch of the band-32 GCV search (lambdas in, terms out)
};

struct Context {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;         // = lanes[0]->s: the stream of the blocking host entry points
    std::vector<FitLane *> lanes;         // lanes[0] is created by mhs_init, the others on demand (fit_lane)
    FitLane *batch = nullptr;             // the batched small fits' own lane: ONE plain stream + arena, no CU-masked streams (batch_lane)
    double2 *log_tab = nullptr;  // device, LOG_TAB_N entries
    double *exp_tab = nullptr;   // device, 4096 entries 2^(j/4096) (svr_kernel)
    double *points_arena = nullptr;       // grow-only device scratch of mhs_residual_points
    size_t points_arena_cap = 0;          // in doubles
    // grow-only device scratch of mhs_mosaic_feather_dev (sums, counts, seam boxes), one per mosaic lane: lane 0 = the calling
    // thread's own mosaics (Step 3 + 4 inside a tile), lane 1 = the merging helper threads of multi.hip, which run beside them
    char *mosaic_arena[2] = {nullptr, nullptr};
    size_t mosaic_arena_cap[2] = {0, 0};  // in bytes
    double *surface_arena = nullptr;      // grow-only device scratch of mhs_tps_surface (the tiles' keep windows)
    size_t surface_arena_cap = 0;         // in doubles
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // Pipelined host-pointer entry points (what the R shim calls: rasters live in host RAM, V73:468-606): persistent,
    // grow-only device arena + copy streams + events, so that a call makes no hipMalloc / hipFree (a device-wide
    // synchronisation each) and row band k + 1 travels host -> device, band k - 1 device -> host, under band k's kernels
    char *pipe_arena = nullptr;
    size_t pipe_arena_cap = 0;
    hipStream_t pipe_h2d = nullptr, pipe_d2h = nullptr, pipe_comp = nullptr;
    hipEvent_t pipe_in[8] = {}, pipe_done[8] = {}, pipe_out[8] = {};
    hipStream_t upload = nullptr;         // non-blocking stream of the small blocking host -> device copies (h2d_sync)
    int n_cu = 0;
    // mhs_fit_reserve_cus: a stream whose CU mask leaves compute units out, and the two events that order a kernel
    // launched on it between its neighbours on the caller's stream
    int reserved_cus = 0;                 // active setting (0 = off)
    int masked_cus = 0;                   // what masked_stream was created for
    hipStream_t masked_stream = nullptr;
    std::vector<uint32_t> comp_mask;      // the reserved units' mask (for lanes created later)
    hipEvent_t mask_ev0 = nullptr, mask_ev1 = nullptr;
};
// Device slots (round 5): the library drives up to MAX_SLOTS devices from ONE host process (what a single-threaded R
// session needs, SURVEY.md 8b: "library may use internal host threads + HIP streams (one per GPU)").  A slot = one
// Context (streams, arenas, pools, caches) bound to one physical device; several slots may name the same physical
// device (that is how the multi-device drivers are tested on a one-GPU box).  Every host thread has a CURRENT slot
// (thread-local, 0 by default -- the only slot mhs_init() creates): ctx(), pipe_mutex(), mask_mutex(), the block pool
// and the reduction cache all resolve through it.  A thread the library starts inherits its parent's slot explicitly
// (SlotBind), because both the slot and HIP's current device are per-thread state.
constexpr int MAX_SLOTS = 16;
int current_slot();
int slot_count();                     // slots initialised by mhs_init / mhs_init_devices
int bind_slot(int slot);              // make `slot` this thread's current slot and its device HIP's current device
struct SlotBind {                     // RAII: bind for a scope, restore the previous slot (and device) on exit
    int prev;
    explicit SlotBind(int slot) : prev(current_slot()) { (void)bind_slot(slot); }
    ~SlotBind() { (void)bind_slot(prev); }
    SlotBind(const SlotBind &) = delete;
    SlotBind &operator=(const SlotBind &) = delete;
};
std::mutex &pipe_mutex();             // one pipelined host-pointer call at a time (per slot)
int host_pipe(size_t arena_bytes);    // streams / events on first use; grows the arena (grow-only) to at least arena_bytes
std::mutex &mask_mutex();             // guards the fields above and the event pair's record / wait sequences (per slot)
std::mutex &mosaic_mutex();           // one user of the slot's mosaic arena (of the calling thread's lane) at a time
int cpu_budget();                     // CPUs this process may keep busy (hardware threads capped by the cgroup quota; tps_gcv_host.hip)
int mosaic_lane();                    // the calling thread's mosaic lane (0 unless set)
void set_mosaic_lane(int lane);
Context &ctx();                       // the current slot's context
Context &ctx_slot(int slot);
int fit_lane(int i, FitLane **out);   // lane i, created on first use (call from one thread at a time)
// The batch's lane (tps_batch.hip, the tiled Step 3): a single non-blocking stream and a grow-only arena.  NOT one of `lanes`:
// those carry up to three CU-masked streams each, and a process that holds a few dozen masked streams sees every later kernel on
// its other streams run ~1.8 x slower (measured in round 6: the hardware queues are shared and a queue keeps a mask) -- the
// tiled Step 3 must not create nine lanes to use one stream.
int batch_lane(FitLane **out);
std::mutex &batch_mutex();            // one user of the slot's batch lane (its stream and arena) at a time: mhs_tps_fit_many, the tiled Step 3
// mhs_tps_fit on a given lane; gcv_threads = host threads of the GCV search (0 = auto)
int tps_fit_lane(FitLane &L, const double *xy, const double *y, int64_t N, double lambda, int gcv_mode,
                 int gcv_threads, mhs_tps **out);
// Small device blocks (a spline's knots, its far-field plan) from a pool instead of hipMalloc / hipFree: the reference-tiled
// Step 3 builds and drops a spline per tile -- 16 per (tile, layer) unit at cfg4, ~6 buffers each -- and every hipFree is a
// device-wide synchronisation.  Power-of-two size classes, blocks never return to the driver before mhs_shutdown.
// pool_release does NOT synchronise: the caller guarantees that nothing in flight still reads the block.
void *pool_alloc(size_t bytes);
void pool_release(void *p);
void pool_clear();
int require_ready();
// tiles.hip: mhs_mosaic_feather_dev; finite_tiles = no tile holds an NA (spline planes): the seams' bounding-box pass is skipped;
// [row_lo, row_hi) (row_hi < 0: the whole grid): the rows to produce, out_dev holding row row_lo first -- a device's row band;
// tiles that do not reach those rows may be NULL
int mosaic_feather_impl(const mhs_grid *g, int64_t nRx, int64_t nCx, const int64_t *tile_win, const double *const *tile_dev,
                        int merge_mode, double *out_dev, int64_t ld, int64_t *seam_win_out, void *stream, bool finite_tiles,
                        int64_t row_lo = 0, int64_t row_hi = -1);
void reduction_cache_clear();           // tps_fit.hip: drop every cached reduction of the current slot (mhs_shutdown)
void multi_reset();                     // multi.hip: drop the multi-device drivers' per-slot state (mhs_shutdown, mhs_init_devices)
// Blocking host -> device copy that does NOT go through the NULL stream: a plain hipMemcpy synchronises with every
// blocking stream -- the CU-masked streams of mhs_fit_reserve_cus are blocking ones, so it would wait for the forest.
int h2d_sync(void *dst, const void *src, size_t bytes);
// _dev entry points launch on exactly the stream they are given; NULL is HIP's default
// (null) stream, which is also torch's default stream.
inline hipStream_t pick_stream(void *s) { return (hipStream_t)s; }

// device buffer with RAII for temporaries inside one ABI call
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) {
        if (p) { (void)hipFree(p); p = nullptr; }
        n = count;
        return hipMalloc((void **)&p, (count ? count : 1) * sizeof(T));
    }
    T *release() { T *q = p; p = nullptr; return q; }
};

// one knot as the evaluation kernels read it (scalar loads, 32 B per knot):
// scaled coordinates and the coefficient with fields' constants folded in,
// cw = c_j * 0.5/(8 pi)  so that  sum_j cw_j * d2 * log(d2) = sum_j c_j phi(d2).
struct Knot { double u, v, cw, pad; };

}  // namespace mhs

struct mhs_tps;
namespace mhs {
int upload_knots(mhs_tps *t);  // (re)build t->knots_dev from t->c / t->knots_uv
int tps_free_quiet(mhs_tps *t);   // mhs_tps_free without its device-wide wait (the caller has synchronised)
// rows [b0, b1) of the window [r0, r1) x [c0, c1) with the window's own plan (tps_eval.hip); out_dev holds row b0 first
int tps_predict_rows_dev(const mhs_tps *t, const mhs_grid *g, int64_t r0, int64_t r1, int64_t c0, int64_t c1,
                         int64_t b0, int64_t b1, double *out_dev, int64_t ld, void *stream);
}

struct mhs_model;
namespace mhs {
// ensemble.hip: the handle's twin on device slot `slot` (built on first use; owned by the handle)
int model_on_slot(const mhs_model *m, int slot, const mhs_model **out);
// ensemble.hip: pred.elev on grid rows [b0, b1) from a device buffer holding only those rows of every plane
int members_rows_dev(const mhs_model *const *models, const double *weights, int n_models, int accumulate, int scale, double wt_total,
                     const mhs_grid *g, const void *buf, int64_t buf_r0, int64_t buf_r1, int n_layers, int dtype, int64_t ld,
                     double nodata, int64_t b0, int64_t b1, double *out_dev, int64_t ld_out, hipStream_t st);
int ensemble_band_dev(const mhs_model *const *models, const double *weights, int n_models, double wt_total, const mhs_grid *g,
                      const void *band_data, int n_layers, int dtype, int64_t ld, double nodata, int64_t b0, int64_t b1,
                      double *out_dev, int64_t ld_out, hipStream_t st);
}

// fitted spline handle (opaque to callers)
struct mhs_tps {
    int64_t n = 0;
    double lambda = 0, eff_df = 0, gcv = 0;
    double center[2] = {0, 0}, scale[2] = {1, 1};
    double d[3] = {0, 0, 0};
    std::vector<double> c;        // n
    std::vector<double> knots_uv; // n x 2 column-major, scaled
    mhs::Knot *knots_dev = nullptr;
    // plan of the far-field-interpolated grid evaluation (tps_eval.hip), cached per window geometry
    struct FarPlan {
        double xmin = 0, ymax = 0, xres = 0, yres = 0;
        int64_t r0 = -1, r1 = -1, c0 = -1, c1 = -1;
        int tx = 0, ty = 0, ntx = 0, nty = 0;   // tile size in cells, tiles per direction
        mhs::Knot *sorted_dev = nullptr;        // knots ordered by bin (row-major)
        int *bin_start_dev = nullptr;           // (nty + 4) * (ntx + 4) + 1 offsets into sorted_dev
        double *nodes_dev = nullptr;            // ntx * nty * 256 far-field values at the tile nodes
        double *lx_dev = nullptr, *ly_dev = nullptr;  // interpolation matrices, tx x 16 and ty x 16
        size_t nodes_cap = 0, bins_cap = 0, lx_cap = 0, ly_cap = 0;
        bool last_used = false;                 // the last grid evaluation took this path
        hipStream_t last_stream = nullptr;      // stream of the last far-field evaluation (its kernels read the buffers above)
        bool in_flight = false;                 // ... which may still be running
        int64_t node_pairs = 0, cell_pairs = 0; // (node, far knot) and (cell, near knot) kernel evaluations
    } far;
    std::mutex mu;   // grid evaluations of one handle from several host threads: the plan is rebuilt under it
};
