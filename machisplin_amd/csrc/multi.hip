// Several devices behind the C ABI (round 5): ONE host process -- the single-threaded R session of the reference
// (V73:117 forces n.cores = 1) -- drives 1..16 MI355X through one call.
//
// The reference side of this is README.md:157-215 (machisplin.tiles.create -> machisplin.mltps per tile ->
// machisplin.tiles.merge; "embarrassingly parallel" tiles) and machisplin.mltps Steps 2-5 themselves (V73:442-930), whose
// cells are independent given the fitted members and the spline coefficients.  Two drivers:
//
//   * ROW BANDS (mhs_mltps_grid_multi*, BASELINE configs 3 and 5): the grid is cut into contiguous row bands, one per
//     device slot (cuts at multiples of 16 rows, the tile height of gbm's coherent kernel).  A host thread per slot
//     uploads its band of the covariates, predicts the ensemble on it, slot 0 also computes the station residuals and fits
//     the spline (its band is made shorter by slot0_share), every slot evaluates the spline on ITS band with the whole
//     grid's evaluation plan and adds.  The coefficient "broadcast" is a host-memory hand-over (one process).  The output
//     goes either straight down every device's own PCIe link into the caller's host plane, or -- `gather` -- is stitched
//     on every device by ONE RCCL all-gather over xGMI (chunks of equal height, slot 0's rows parked at the end of its
//     chunk, so the gathered chunks ARE the grid in place).  With the reference-tiled Step 3 (tile_edge > 0) the tiles are
//     dealt over the slots by cost and travel with the bands.
//   * (TILE, LAYER) UNITS (mhs_tiles_units_multi, BASELINE config 4): machisplin.tiles.create's user tiles x response
//     layers as independent mltps runs, unit u = layer * n_tiles + tile on slot u mod N, no exchange while units run; a
//     layer's tile planes are then brought to the layer's owner (l mod N) -- peer copies over xGMI, the same data movement
//     an all-gather restricted to the owner makes -- merged (machisplin.tiles.merge) and written to the caller's plane.
//
// Slots may alias one physical device (mhs_init_devices with repeated ids): every code path above then runs on a one-GPU
// box -- peer copies become device copies, and the RCCL step, which refuses two ranks on one device, is replaced by the
// same copies.  tests/test_multi_gpu.py compares 2 / 4 slots on GPU 0 with the one-device planes bit for bit.
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "common.h"

using namespace mhs;

namespace {

constexpr int64_t BAND_ROWS_ALIGN = 16;      // = ensemble.hip's BAND_ALIGN: bands of whole coherent-kernel tiles

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------- per-slot resources --
struct MultiSlot {
    hipStream_t s = nullptr;                 // the band work of this slot (non-blocking)
    hipEvent_t e0 = nullptr, e1 = nullptr;   // timing of the band kernels
    hipStream_t u = nullptr;                 // host -> device copies of a band that is still travelling (mhs_mltps_grid_multi)
    hipEvent_t up[5] = {}, dn[5] = {};       // ... one per sub-band up, one per finished sub-band down
    hipStream_t h = nullptr;                 // mhs_tiles_units_multi: the tiles' covariate crops, uploaded ahead of their first unit
};
MultiSlot g_ms[MAX_SLOTS];
std::mutex g_ms_mu;

int multi_slot(int slot, MultiSlot **out) {   // call with the thread bound to `slot`
    std::lock_guard<std::mutex> lk(g_ms_mu);
    MultiSlot &m = g_ms[slot];
    if (!m.s) {
        MHS_HIP(hipStreamCreateWithFlags(&m.s, hipStreamNonBlocking));
        MHS_HIP(hipEventCreate(&m.e0));
        MHS_HIP(hipEventCreate(&m.e1));
        // the copy streams at the LOWEST priority: streams of one priority share a few hardware queues, and a 32 MB copy chunk at the
        // head of a queue holds up every small kernel behind it -- measured on cfg4: the merged layers' copies down (on a stream of
        // the default priority) cost the units beside them as much time as the copies took
        int prio_lo = 0, prio_hi = 0;
        MHS_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        MHS_HIP(hipStreamCreateWithPriority(&m.u, hipStreamNonBlocking, prio_lo));
        MHS_HIP(hipStreamCreateWithPriority(&m.h, hipStreamNonBlocking, prio_lo));
        for (hipEvent_t &e : m.up) MHS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t &e : m.dn) MHS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    *out = &m;
    return MHS_OK;
}

// ---------------------------------------------------------------------------------------------- RCCL --
// Bound at run time (dlopen): the library has no link-time dependency on librccl, and a Python host has torch's copy of
// the same soname mapped already.  Single-process, one communicator per slot (ncclCommInitAll), one thread per slot.
typedef void *nccl_comm;
struct Rccl {
    void *h = nullptr;
    int (*CommInitAll)(nccl_comm *, int, const int *) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    nccl_comm comm[MAX_SLOTS] = {};
    int n = 0;                                // communicators live for this many slots
    bool tried = false, usable = false;
    std::string why;
};
Rccl g_rccl;
std::mutex g_rccl_mu;
constexpr int NCCL_FLOAT64 = 8;               // ncclFloat64 / ncclDouble (rccl.h: ncclDataType_t)

// true when every slot sits on a device of its own (RCCL refuses two ranks on one device)
bool slots_distinct() {
    const int n = slot_count();
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b)
            if (ctx_slot(a).device == ctx_slot(b).device) return false;
    return true;
}

// communicators for the current slots, or `usable = false` with the reason (aliased slots, library missing)
void rccl_prepare() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    const int n = slot_count();
    if (g_rccl.tried && g_rccl.n == n) return;
    g_rccl.tried = true; g_rccl.usable = false; g_rccl.n = n;
    // (one slot: the all-gather of one rank is a copy onto itself -- still taken through RCCL, so that the binding, the
    // communicator and the call are exercised on a one-GPU box too)
    if (!slots_distinct()) { g_rccl.why = "slots share a physical device (RCCL needs one device per rank): peer copies instead"; return; }
    if (getenv("MHS_MULTI_NO_RCCL")) { g_rccl.why = "MHS_MULTI_NO_RCCL is set: peer copies instead"; return; }
    if (!g_rccl.h) {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            g_rccl.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.h) break;
        }
        if (!g_rccl.h) { g_rccl.why = std::string("librccl not found: ") + dlerror(); return; }
        g_rccl.CommInitAll = (int (*)(nccl_comm *, int, const int *))dlsym(g_rccl.h, "ncclCommInitAll");
        g_rccl.CommDestroy = (int (*)(nccl_comm))dlsym(g_rccl.h, "ncclCommDestroy");
        g_rccl.AllGather = (int (*)(const void *, void *, size_t, int, nccl_comm, hipStream_t))dlsym(g_rccl.h, "ncclAllGather");
        g_rccl.GetErrorString = (const char *(*)(int))dlsym(g_rccl.h, "ncclGetErrorString");
        if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.AllGather) { g_rccl.why = "librccl lacks ncclCommInitAll / ncclAllGather"; return; }
    }
    int devs[MAX_SLOTS];
    for (int k = 0; k < n; ++k) devs[k] = ctx_slot(k).device;
    const int rc = g_rccl.CommInitAll(g_rccl.comm, n, devs);
    if (rc != 0) {
        g_rccl.why = std::string("ncclCommInitAll failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
        return;
    }
    g_rccl.usable = true;
    g_rccl.why = "rccl";
}

void rccl_reset() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.usable && g_rccl.CommDestroy)
        for (int k = 0; k < g_rccl.n; ++k) if (g_rccl.comm[k]) { (void)g_rccl.CommDestroy(g_rccl.comm[k]); g_rccl.comm[k] = nullptr; }
    g_rccl.tried = g_rccl.usable = false;
    g_rccl.n = 0;
}

// --------------------------------------------------------------------------------------- thread team --
struct Barrier {
    std::mutex mu;
    std::condition_variable cv;
    int n, waiting = 0;
    uint64_t phase = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t ph = phase;
        if (++waiting == n) { waiting = 0; ++phase; cv.notify_all(); }
        else cv.wait(lk, [&] { return phase != ph; });
    }
};

// One host thread per slot runs `body(slot)`; the first failure's status and message are carried to the caller.  A body
// must pass every barrier it shares with the others even after a failure (Team::failed() says when to skip the work).
struct Team {
    int n;
    Barrier bar;
    std::atomic<int> rc{MHS_OK};
    std::mutex err_mu;
    std::string err;
    explicit Team(int n_) : n(n_), bar(n_) {}
    bool failed() const { return rc.load() != MHS_OK; }
    void fail(int code) {
        int expected = MHS_OK;
        if (rc.compare_exchange_strong(expected, code)) {
            std::lock_guard<std::mutex> lk(err_mu);
            err = mhs_last_error();              // thread-local in the worker
        }
    }
    template <typename F>
    int run(F body) {
        const int home = current_slot();
        std::vector<std::thread> th;
        auto wrap = [&](int slot) {
            SlotBind bind(slot);
            body(slot);
        };
        for (int k = 1; k < n; ++k) th.emplace_back(wrap, k);
        wrap(0);
        for (std::thread &t : th) t.join();
        (void)bind_slot(home);
        if (failed()) { set_error("%s", err.c_str()); return rc.load(); }
        return MHS_OK;
    }
};

// `step` only if nobody has failed yet; a failure is recorded
#define TEAM_DO(team, expr)                                      \
    do {                                                         \
        if (!(team).failed()) { const int rc_ = (expr); if (rc_ != MHS_OK) (team).fail(rc_); } \
    } while (0)

// ------------------------------------------------------------------------------------------ row bands --
// Chunks of `band` rows, one per slot; slot 0 may hold fewer (n0) -- its rows sit at the END of its chunk, `lead` rows in
// -- so that chunk k starts at grid row k * band - lead and equal-sized chunks tile the grid (sharded.row_bands).
struct BandPlan {
    int64_t band = 0, lead = 0;
    std::vector<int64_t> r0, r1;
};

BandPlan plan_bands(int64_t nrow, int n, double slot0_share) {
    BandPlan p;
    p.r0.assign((size_t)n, 0); p.r1.assign((size_t)n, 0);
    if (n == 1) { p.band = nrow; p.r1[0] = nrow; return p; }
    const int64_t even = (nrow + n - 1) / n;
    const int64_t align = even >= BAND_ROWS_ALIGN ? BAND_ROWS_ALIGN : 1;
    auto up = [&](int64_t v) { return (v + align - 1) / align * align; };
    auto equal = [&] {
        p.band = up(even); p.lead = 0;
        for (int k = 0; k < n; ++k) { p.r0[(size_t)k] = std::min<int64_t>(k * p.band, nrow); p.r1[(size_t)k] = std::min<int64_t>((k + 1) * p.band, nrow); }
    };
    if (std::isnan(slot0_share)) { equal(); return p; }
    const double sh = std::min(std::max(slot0_share, 0.0), 1.0);
    int64_t n0 = std::min<int64_t>(nrow, (int64_t)llround(sh * (double)nrow / (double)align) * align);
    const int64_t rest = nrow - n0, others = n - 1;
    const int64_t h = up((rest + others - 1) / others);
    if (n0 > h) { equal(); return p; }          // slot 0 must not be the tallest band
    p.band = std::max(h, n0); p.lead = p.band - n0;
    p.r0[0] = 0; p.r1[0] = n0;
    for (int k = 1; k < n; ++k) {
        p.r0[(size_t)k] = std::min<int64_t>(n0 + (k - 1) * h, nrow);
        p.r1[(size_t)k] = std::min<int64_t>(n0 + k * h, nrow);
    }
    // chunk k (k >= 1) starts at grid row n0 + (k - 1) h = k band - lead only when h == band: true unless n0 > h (handled)
    return p;
}

}  // namespace

// one slot's share of a multi-device raster stack
struct MultiBand {
    int64_t r0 = 0, r1 = 0;
    char *cov = nullptr;          // C planes of rows [r0, r1), plane k at k * (r1 - r0) * ld elements
    double *ens = nullptr;        // pred.elev on the band (rows x ncol)
    double *tot = nullptr;        // pred.elev + final.TPS on the band
    double *full = nullptr;       // gather target: n * band rows x ncol (the grid starts `lead` rows in)
    double *tiles = nullptr;      // ... and every tile's keep window
    size_t tiles_cap = 0;
};

struct mhs_multi_stack {
    mhs_grid g{};
    int C = 0, dtype = 0;
    int64_t ld = 0;               // elements per stored row (= ncol)
    double nodata = NAN;
    int n = 0;
    int dev[MAX_SLOTS] = {};      // the physical device every band's buffers live on (slot k's device when the stack was built)
    BandPlan plan;
    MultiBand b[MAX_SLOTS];
    int used_tps = 0;             // which plane holds the last step's final: 1 = tot, 0 = ens
    bool gathered = false;
    bool have_result = false;
    // mhs_mltps_grid_multi: the caller's planes have NOT been uploaded yet -- the next step brings each band in sub-bands
    // under its own first member kernels
    const mhs_stack *pending = nullptr;
    double *pending_out = nullptr;   // ... and, for the global Step 3, sends every finished sub-band down to this plane
    bool downloaded = false;
    double upload_ms = 0;         // how long the slowest slot's helper thread spent in the copies up
    double download_ms = 0;       // what was left of the copies down when the slowest slot's last sub-band was final
};

namespace {

size_t elem_size(int dtype) { return dtype == MHS_F64 ? 8 : dtype == MHS_F32 ? 4 : 2; }

// what the last steps measured, for the automatic slot-0 share of the next stack of the same shape
struct Balance { int n = 0; int64_t nrow = 0, ncol = 0, stations = 0; double share = NAN; };
Balance g_balance;
std::mutex g_balance_mu;

// mhs_mltps_grid_multi's device buffers, kept between calls of the same shape (no hipMalloc / hipFree of gigabytes per layer)
mhs_multi_stack *g_host_ms = nullptr;
std::mutex g_host_mu;
// mhs_tiles_units_multi's device memory, one grow-only arena per slot kept between calls: hipMalloc / hipFree cost ~35 ms per
// GB here, and a cfg4 call needs 14 GB (crops, scratch, 48 unit planes, merge buffers) -- 0.4 s of a 0.84 s call went there
struct UnitsArena { char *base = nullptr; size_t cap = 0; };
UnitsArena g_units_arena[MAX_SLOTS];

// Device -> pageable host memory through the library's own pinned ring (per slot, kept between calls): the runtime's path for a
// pageable destination is ONE thread copying out of its staging buffer (16-20 GB/s here) and, measured on cfg4, slows the
// kernels beside it by a quarter.  Here a chunk travels by DMA into a pinned buffer, and SINK_THREADS host threads copy finished
// chunks to their place, several at a time.
constexpr int RING_BUFS = 8, SINK_THREADS = 3;      // (fewer threads where the host has few cores per slot)
constexpr size_t RING_CHUNK = (size_t)8 << 20;
struct PinnedRing { char *buf[RING_BUFS] = {}; hipEvent_t ev[RING_BUFS] = {}; bool ok = false; };
PinnedRing g_ring[MAX_SLOTS];

int ring_prepare(int slot) {                     // call with the thread bound to `slot`
    PinnedRing &R = g_ring[slot];
    if (R.ok) return MHS_OK;
    for (int b = 0; b < RING_BUFS; ++b) {
        if (!R.buf[b]) MHS_HIP(hipHostMalloc((void **)&R.buf[b], RING_CHUNK, hipHostMallocDefault));
        if (!R.ev[b]) MHS_HIP(hipEventCreateWithFlags(&R.ev[b], hipEventDisableTiming));
    }
    R.ok = true;
    return MHS_OK;
}

class HostSink {
  public:
    HostSink(int slot, hipStream_t st) : slot_(slot), st_(st), R_(g_ring[slot]) {
        for (int b = 0; b < RING_BUFS; ++b) free_[b] = true;
        const int nth = std::min(SINK_THREADS, std::max(1, cpu_budget() / (4 * std::max(1, slot_count()))));
        for (int w = 0; w < nth; ++w) th_.emplace_back([this] { work(); });
    }
    ~HostSink() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    // bytes from device memory (src, on this slot's device) to dst; returns when every byte is in place
    int download(void *dst, const void *src, size_t bytes) {
        size_t off = 0;
        for (int c = 0; off < bytes; ++c) {
            const size_t sz = std::min(RING_CHUNK, bytes - off);
            int b = -1;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { for (int q = 0; q < RING_BUFS; ++q) if (free_[q]) { b = q; return true; } return false; });
                free_[b] = false;
            }
            hipError_t e = hipMemcpyAsync(R_.buf[b], (const char *)src + off, sz, hipMemcpyDeviceToHost, st_);
            if (e == hipSuccess) e = hipEventRecord(R_.ev[b], st_);
            if (e != hipSuccess) {
                set_error("HostSink: %s", hipGetErrorString(e));
                std::lock_guard<std::mutex> lk(mu_);
                free_[b] = true; rc_ = MHS_ERR_HIP;
                break;
            }
            {
                std::lock_guard<std::mutex> lk(mu_);
                jobs_.push_back(Job{b, (char *)dst + off, sz});
                ++pending_;
            }
            cv_.notify_all();
            off += sz;
        }
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return pending_ == 0; });
        return rc_;
    }

  private:
    struct Job { int b; char *dst; size_t sz; };
    void work() {
        SlotBind bind(slot_);
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !jobs_.empty(); });
                if (jobs_.empty()) return;
                j = jobs_.front(); jobs_.pop_front();
            }
            const hipError_t e = hipEventSynchronize(R_.ev[j.b]);
            if (e == hipSuccess) memcpy(j.dst, R_.buf[j.b], j.sz);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (e != hipSuccess) rc_ = MHS_ERR_HIP;
                free_[j.b] = true; --pending_;
            }
            cv_.notify_all();
        }
    }
    int slot_;
    hipStream_t st_;
    PinnedRing &R_;
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Job> jobs_;
    bool free_[RING_BUFS];
    int pending_ = 0, rc_ = MHS_OK;
    bool stop_ = false;
};

// the stack's bands still sit on the devices its slots are bound to (round-5 advisor finding: mhs_init_devices with the same
// count but other ids would send slot k's kernels after another device's pointers)
bool stack_current(const mhs_multi_stack *ms) {
    if (ms->n != slot_count()) return false;
    for (int k = 0; k < ms->n; ++k) if (!ctx_slot(k).ready || ctx_slot(k).device != ms->dev[k]) return false;
    return true;
}

int free_stack(mhs_multi_stack *ms) {
    if (!ms) return MHS_OK;
    if (!stack_current(ms)) { delete ms; return MHS_OK; }      // its devices were re-initialised: their memory went with them
    const int home = current_slot();
    for (int k = 0; k < ms->n; ++k) {
        (void)bind_slot(k);
        MultiBand &b = ms->b[k];
        if (g_ms[k].s) (void)hipStreamSynchronize(g_ms[k].s);
        for (void *q : {(void *)b.cov, (void *)b.ens, (void *)b.tot, (void *)b.full, (void *)b.tiles})
            if (q) (void)hipFree(q);
    }
    (void)bind_slot(home);
    delete ms;
    return MHS_OK;
}

}  // namespace

// what the host-plane calls keep between calls: the band buffers of mhs_mltps_grid_multi, the unit arenas and pinned rings of
// mhs_tiles_units_multi
static void trim_caches() {
    std::lock_guard<std::mutex> lk(g_host_mu);
    if (g_host_ms) { bool alive = true; for (int k = 0; k < g_host_ms->n; ++k) alive = alive && ctx_slot(k).ready; if (alive) free_stack(g_host_ms); else delete g_host_ms; }
    g_host_ms = nullptr;
    const int home = current_slot();
    for (int k = 0; k < MAX_SLOTS; ++k) {
        if (ctx_slot(k).ready) {
            (void)bind_slot(k);
            if (g_units_arena[k].base) { (void)hipDeviceSynchronize(); (void)hipFree(g_units_arena[k].base); }
            for (int b = 0; b < RING_BUFS; ++b) {
                if (g_ring[k].buf[b]) (void)hipHostFree(g_ring[k].buf[b]);
                if (g_ring[k].ev[b]) (void)hipEventDestroy(g_ring[k].ev[b]);
            }
        }
        g_units_arena[k] = UnitsArena();
        g_ring[k] = PinnedRing();
    }
    (void)bind_slot(home);
}

extern "C" int mhs_multi_trim(void) {
    if (int rc = require_ready()) return rc;
    trim_caches();
    return MHS_OK;
}

void mhs::multi_reset() {
    trim_caches();
    rccl_reset();
    std::lock_guard<std::mutex> lk(g_ms_mu);
    for (int k = 0; k < MAX_SLOTS; ++k) {
        MultiSlot &m = g_ms[k];
        if (!m.s) continue;
        if (ctx_slot(k).ready) {
            SlotBind bind(k);
            (void)hipStreamSynchronize(m.s); (void)hipStreamDestroy(m.s);
            (void)hipEventDestroy(m.e0); (void)hipEventDestroy(m.e1);
            if (m.u) { (void)hipStreamSynchronize(m.u); (void)hipStreamDestroy(m.u); }
            if (m.h) { (void)hipStreamSynchronize(m.h); (void)hipStreamDestroy(m.h); }
            for (hipEvent_t e : m.up) if (e) (void)hipEventDestroy(e);
            for (hipEvent_t e : m.dn) if (e) (void)hipEventDestroy(e);
        }
        m = MultiSlot();
    }
}

extern "C" {

static int stack_build(const mhs_grid *g, const mhs_stack *covars_host, double slot0_share, bool upload, mhs_multi_stack **out) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(g && covars_host && covars_host->data && out, "NULL argument");
    MHS_REQUIRE(g->nrow > 0 && g->ncol > 0 && g->xres > 0 && g->yres > 0, "bad grid geometry");
    MHS_REQUIRE(covars_host->n_layers >= 1 && covars_host->n_layers <= 64, "bad number of layers");
    MHS_REQUIRE(covars_host->dtype == MHS_F64 || covars_host->dtype == MHS_F32 || covars_host->dtype == MHS_I16, "bad stack dtype");
    MHS_REQUIRE(covars_host->ld >= g->ncol && covars_host->plane_stride >= covars_host->ld * g->nrow, "stack strides smaller than the grid");
    const int n = slot_count();
    mhs_multi_stack *ms = new mhs_multi_stack();
    ms->g = *g; ms->C = covars_host->n_layers; ms->dtype = covars_host->dtype; ms->ld = g->ncol; ms->nodata = covars_host->nodata;
    ms->n = n;
    for (int k = 0; k < n; ++k) ms->dev[k] = ctx_slot(k).device;
    ms->plan = plan_bands(g->nrow, n, slot0_share);
    const size_t esz = elem_size(ms->dtype);
    Team team(n);
    const int rc = team.run([&](int slot) {
        MultiBand &b = ms->b[slot];
        b.r0 = ms->plan.r0[(size_t)slot]; b.r1 = ms->plan.r1[(size_t)slot];
        const int64_t nb = b.r1 - b.r0;
        auto work = [&]() -> int {
            MultiSlot *S = nullptr;
            if (int rc2 = multi_slot(slot, &S)) return rc2;
            if (nb == 0) return MHS_OK;
            const size_t plane = (size_t)nb * (size_t)ms->ld * esz;
            MHS_HIP(hipMalloc((void **)&b.cov, plane * (size_t)ms->C));
            MHS_HIP(hipMalloc((void **)&b.ens, sizeof(double) * (size_t)nb * (size_t)g->ncol));
            MHS_HIP(hipMalloc((void **)&b.tot, sizeof(double) * (size_t)nb * (size_t)g->ncol));
            if (!upload) return MHS_OK;
            for (int k = 0; k < ms->C; ++k) {
                const char *src = (const char *)covars_host->data + ((size_t)k * covars_host->plane_stride + (size_t)b.r0 * covars_host->ld) * esz;
                if (covars_host->ld == g->ncol)
                    MHS_HIP(hipMemcpyAsync(b.cov + plane * k, src, plane, hipMemcpyHostToDevice, S->s));
                else
                    MHS_HIP(hipMemcpy2DAsync(b.cov + plane * k, (size_t)ms->ld * esz, src, (size_t)covars_host->ld * esz, (size_t)g->ncol * esz,
                                             (size_t)nb, hipMemcpyHostToDevice, S->s));
            }
            MHS_HIP(hipStreamSynchronize(S->s));
            return MHS_OK;
        };
        TEAM_DO(team, work());
    });
    if (rc) { free_stack(ms); return rc; }
    *out = ms;
    return MHS_OK;
}

int mhs_multi_stack_create(const mhs_grid *g, const mhs_stack *covars_host, double slot0_share, mhs_multi_stack **out) {
    return stack_build(g, covars_host, slot0_share, true, out);
}

int mhs_multi_stack_free(mhs_multi_stack *ms) { return free_stack(ms); }

// host only (no GPU needed): the row bands mhs_multi_stack_create cuts for n slots
int mhs_plan_row_bands(int64_t nrow, int n_slots, double slot0_share, int64_t *r0, int64_t *r1, int64_t *band, int64_t *lead) {
    MHS_REQUIRE(nrow > 0 && n_slots >= 1 && n_slots <= MAX_SLOTS && r0 && r1, "bad arguments");
    const BandPlan p = plan_bands(nrow, n_slots, slot0_share);
    for (int k = 0; k < n_slots; ++k) { r0[k] = p.r0[(size_t)k]; r1[k] = p.r1[(size_t)k]; }
    if (band) *band = p.band;
    if (lead) *lead = p.lead;
    return MHS_OK;
}

int mhs_multi_stack_bands(const mhs_multi_stack *ms, int *n_slots, int64_t *r0, int64_t *r1) {
    MHS_REQUIRE(ms && n_slots, "NULL argument");
    *n_slots = ms->n;
    for (int k = 0; k < ms->n; ++k) {
        if (r0) r0[k] = ms->b[k].r0;
        if (r1) r1[k] = ms->b[k].r1;
    }
    return MHS_OK;
}

}  // extern "C"

namespace {

// rows [r0, r1) in sub-bands, cut at whole 16-row tiles of the grid; returns how many.  Copies up: the short ones first -- 4,
// 16, 40, 40 %, or for float64 planes (twice the bytes per row: the copies are then only ~2 x faster than the kernels that
// wait for them, see host_window_pipeline in ensemble.hip) 3, 6, 13, 28, 50 %.  Copies down: the short ones last, 48, 30, 14, 6, 2 %.
enum { BANDS_UP = 0, BANDS_UP_F64 = 1, BANDS_DOWN = 2 };
constexpr int MAX_SUB = 5;
int sub_bands(int64_t r0, int64_t r1, int mode, int64_t cut[MAX_SUB + 1]) {
    static const int pct[3][MAX_SUB] = {{4, 16, 40, 40, 0}, {3, 6, 13, 28, 50}, {48, 30, 14, 6, 2}};
    const int nq = mode == BANDS_UP ? 4 : 5;
    const int64_t nb = r1 - r0;
    cut[0] = r0;
    for (int q = 0, a = 0; q < nq; ++q) {
        a += pct[mode][q];
        cut[q + 1] = q == nq - 1 ? r1 : std::min(r1, std::max(cut[q], (r0 + nb * a / 100 + BAND_ROWS_ALIGN / 2) / BAND_ROWS_ALIGN * BAND_ROWS_ALIGN));
    }
    return nq;
}

// the helper thread that sits in a slot's copies from pageable memory while the slot's own thread launches kernels
struct Uploader {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    int issued = 0, rc = MHS_OK;
    std::string err;
    template <typename F>
    void start(int slot, int nq, F issue) {
        th = std::thread([this, slot, nq, issue] {
            SlotBind bind(slot);
            for (int q = 0; q < nq; ++q) {
                const int r = issue(q);
                std::lock_guard<std::mutex> lk(mu);
                if (r) { rc = r; err = mhs_last_error(); issued = nq; } else issued = q + 1;
                cv.notify_all();
                if (r) break;
            }
        });
    }
    int wait(int q) {                          // sub-band q's copies are in the stream and its event is recorded
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return issued > q; });
        if (rc) set_error("%s", err.c_str());
        return rc;
    }
    void join() { if (th.joinable()) th.join(); }
    ~Uploader() { join(); }
};

struct StepShared {
    // Step 2 at the stations / Step 3's fit, produced by slot 0
    std::vector<double> resid;
    double rsq_model = NAN, tss = NAN;
    // global fit: the coefficients every slot rebuilds its spline handle from
    int64_t nk = 0;
    std::vector<double> c, knots_uv;
    double d[3] = {0, 0, 0}, center[2] = {0, 0}, scale[2] = {1, 1}, lambda = NAN;
    // reference-tiled Step 3
    int64_t nRx = 1, nCx = 1;
    std::vector<int64_t> fit_win, keep_win;
    std::vector<int> owner;
    std::vector<size_t> tile_cells;           // cells of tile h's keep window, rounded up to 32
    // what every slot holds of the tiles: the ones it owns (fits + evaluates) and the ones that reach its rows (pulled from
    // their owners) -- offsets into its own tile area, -1 for the others; bytes pulled from peers
    std::vector<int64_t> slot_off[MAX_SLOTS];
    size_t slot_need[MAX_SLOTS] = {};
    int64_t pulled_bytes[MAX_SLOTS] = {};
    // Step 5
    std::vector<double> f_actual;
    int used_tps = 0;
    double rsq_final = NAN;
    double fit_ms = 0, band_ms[MAX_SLOTS] = {}, tiles_ms[MAX_SLOTS] = {}, upload_ms[MAX_SLOTS] = {}, download_ms[MAX_SLOTS] = {};
};

}  // namespace

extern "C" {

// Steps 2-5 of machisplin.mltps for one response layer on a resident multi-device stack (V73:442-930).
int mhs_mltps_grid_multi_dev(const mhs_model *const *models, const double *weights, int n_models, double wt_total,
                             mhs_multi_stack *ms, const double *X, const double *resp, int64_t n, int64_t tile_edge,
                             double lambda, int gcv_mode, int gather, mhs_mltps_info *info) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(models && weights && n_models >= 1 && n_models <= 8 && ms && X && resp && n > 3, "bad arguments");
    MHS_REQUIRE(stack_current(ms), "the stack was built for another set of device slots");
    MHS_REQUIRE(wt_total != 0.0 && !std::isnan(wt_total), "wt_total must be non-zero");
    const int N = ms->n;
    const mhs_grid g = ms->g;
    const int p = ms->C + 2;
    int kind0 = 0, p0 = 0;
    for (int k = 0; k < n_models; ++k) {
        MHS_REQUIRE(models[k] != nullptr, "NULL model");
        if (int rc = mhs_model_info(models[k], &kind0, &p0, nullptr)) return rc;
        MHS_REQUIRE(p0 == p, "a model's predictor count does not match the stack (layers + 2)");
    }
    const double t_step0 = now_ms();
    // knots = the LONG / LAT columns of dat_tps (V73:688,751): the last two columns of X; their cells for Step 5
    const double *knots = X + (size_t)(p - 2) * (size_t)n;
    std::vector<int64_t> rows((size_t)n), cols((size_t)n);
    if (int rc = mhs_cells_from_xy(&g, knots, n, rows.data(), cols.data())) return rc;
    StepShared S;
    S.f_actual.assign((size_t)n, NAN);
    int64_t nt = 1;
    if (tile_edge > 0) {
        if (int rc = mhs_step3_tile_windows(&g, tile_edge, 0.2, 0.025, &S.nRx, &S.nCx, nullptr, nullptr, 0)) return rc;
        nt = S.nRx * S.nCx;
    }
    const bool tiled = nt > 1;
    if (tiled) {
        S.fit_win.resize((size_t)nt * 4); S.keep_win.resize((size_t)nt * 4);
        if (int rc = mhs_step3_tile_windows(&g, tile_edge, 0.2, 0.025, &S.nRx, &S.nCx, S.fit_win.data(), S.keep_win.data(), nt)) return rc;
        // Ownership follows the rows (round 6): a tile is fitted and evaluated by the slot whose band holds most of its keep
        // window -- that slot needs the plane anyway -- and a slot pulls from its peers only the tiles that reach ITS rows,
        // then mosaics and feathers only those rows, straight into its band.  (Rounds 1-5 dealt the tiles by cost, pulled
        // EVERY tile to EVERY slot and mosaicked the whole grid N times: 0.8 GB per slot over the fabric at cfg3 and N x
        // redundant Step-4 work.)  Bands with no rows own nothing.
        S.tile_cells.resize((size_t)nt);
        S.owner.assign((size_t)nt, 0);
        for (int k = 0; k < N; ++k) S.slot_off[k].assign((size_t)nt, -1);
        for (int64_t h = 0; h < nt; ++h) {
            const int64_t *kw = &S.keep_win[(size_t)h * 4];
            S.tile_cells[(size_t)h] = ((size_t)((kw[1] - kw[0]) * (kw[3] - kw[2])) + 31) & ~(size_t)31;
            int64_t best_rows = -1;
            for (int k = 0; k < N; ++k) {
                const int64_t ov = std::min(kw[1], ms->b[k].r1) - std::max(kw[0], ms->b[k].r0);
                if (ov > best_rows) { best_rows = ov; S.owner[(size_t)h] = k; }
            }
            for (int k = 0; k < N; ++k) {
                const bool reaches = ms->b[k].r1 > ms->b[k].r0 && kw[0] < ms->b[k].r1 && kw[1] > ms->b[k].r0;
                if (reaches || S.owner[(size_t)h] == k) { S.slot_off[k][(size_t)h] = (int64_t)S.slot_need[k]; S.slot_need[k] += S.tile_cells[(size_t)h]; }
            }
        }
    }
    if (gather) rccl_prepare();
    const bool use_rccl = gather && g_rccl.usable;
    std::vector<const mhs_model *> mods((size_t)N * (size_t)n_models);
    std::vector<mhs_tps *> tps((size_t)N, nullptr);
    Team team(N);
    const int rc = team.run([&](int slot) {
        MultiBand &b = ms->b[slot];
        const int64_t nb = b.r1 - b.r0;
        MultiSlot *M = nullptr;
        TEAM_DO(team, multi_slot(slot, &M));
        // this slot's twins of the fitted members
        for (int k = 0; k < n_models; ++k) TEAM_DO(team, model_on_slot(models[k], slot, &mods[(size_t)slot * n_models + k]));
        const mhs_model *const *my = &mods[(size_t)slot * n_models];
        // ---- slot 0: res.FINAL at the stations (V73:477-620) and, for the global Step 3, the fit (V73:751)
        auto fit = [&]() -> int {
            S.resid.resize((size_t)n);
            if (int rc2 = mhs_residual_points(my, weights, n_models, wt_total, X, resp, n, S.resid.data())) return rc2;
            // R's sum() and mean() accumulate in long double (V73:912-917 run in R): so do these
            long double msum = 0.0L, ss = 0.0L, rs = 0.0L;
            for (int64_t i = 0; i < n; ++i) msum += resp[i];
            const double mean = (double)(msum / (long double)n);
            for (int64_t i = 0; i < n; ++i) { ss += (long double)((resp[i] - mean) * (resp[i] - mean)); rs += (long double)(S.resid[(size_t)i] * S.resid[(size_t)i]); }
            S.tss = (double)ss; S.rsq_model = 1.0 - (double)rs / (double)ss;
            if (tiled) return MHS_OK;
            const double t0 = now_ms();
            mhs_tps *t = nullptr;
            if (int rc2 = mhs_tps_fit(knots, S.resid.data(), n, lambda, gcv_mode, &t)) return rc2;
            S.fit_ms = now_ms() - t0;
            tps[0] = t;
            if (int rc2 = mhs_tps_size(t, &S.nk)) return rc2;
            S.c.resize((size_t)S.nk); S.knots_uv.resize((size_t)S.nk * 2);
            return mhs_tps_get(t, S.c.data(), S.d, S.knots_uv.data(), &S.lambda, S.center, S.scale, nullptr, nullptr);
        };
        // ---- Step 2 on the band (V73:447-619): enqueued, not waited for
        // Host planes (ms->pending): the band still lies in the caller's memory.  It travels in sub-bands of 4, 16, 40, 40 % (float64 planes: 3, 6, 13, 28, 50 %) of
        // its rows (cut at whole 16-row tiles), issued by a helper thread -- a copy from pageable memory blocks its caller until
        // staged --; every member but the last runs on a sub-band as soon as it has arrived: the exposed upload is the 4 %.
        // Slot 0 fits the spline beside its first members (a thread of its own); with the coefficients there before the last member
        // starts, that member runs on sub-bands of 48, 30, 14, 6, 2 % and each finished sub-band --
        // scaled, final.TPS added -- goes down to the caller's plane under the next one (`piped`): the exposed download is the
        // 2 %.  The plane that travels is pred.elev + final.TPS; should Step 5 keep pred.elev alone (V73:917-930) it is sent
        // afterwards.  Same cells, same members in the same order, same sums: same bits as the one-piece evaluation.
        const bool piped = ms->pending && ms->pending_out;
        const int banded = n_models > 1 ? n_models - 1 : 1;
        Uploader up;
        if (ms->pending && nb > 0 && !team.failed()) {
            const mhs_stack *h = ms->pending;
            const size_t esz = elem_size(ms->dtype), plane = (size_t)nb * (size_t)ms->ld * esz;
            int64_t cut[MAX_SUB + 1];
            const int nq = sub_bands(b.r0, b.r1, ms->dtype == MHS_F64 ? BANDS_UP_F64 : BANDS_UP, cut);
            const double t0 = now_ms();
            up.start(slot, nq, [=, &S](int q) -> int {
                const int64_t rows_q = cut[q + 1] - cut[q];
                for (int k = 0; k < ms->C && rows_q > 0; ++k) {
                    const char *src = (const char *)h->data + ((size_t)k * h->plane_stride + (size_t)cut[q] * h->ld) * esz;
                    char *dst = b.cov + plane * k + (size_t)(cut[q] - b.r0) * (size_t)ms->ld * esz;
                    if (h->ld == g.ncol)
                        MHS_HIP(hipMemcpyAsync(dst, src, (size_t)rows_q * (size_t)ms->ld * esz, hipMemcpyHostToDevice, M->u));
                    else
                        MHS_HIP(hipMemcpy2DAsync(dst, (size_t)ms->ld * esz, src, (size_t)h->ld * esz, (size_t)g.ncol * esz, (size_t)rows_q,
                                                 hipMemcpyHostToDevice, M->u));
                }
                MHS_HIP(hipEventRecord(M->up[q], M->u));
                if (q == nq - 1) S.upload_ms[slot] = now_ms() - t0;
                return MHS_OK;
            });
            // slot 0's fit runs on a thread of its own beside this one (which goes on launching sub-bands as they arrive): the fit
            // call blocks its caller for its whole length, and with it inline the device ran out of queued kernels behind the
            // short sub-bands (float64 planes: +15 ms on cfg3)
            std::thread fitter;
            int fit_rc = MHS_OK;
            std::string fit_err;
            if (slot == 0)
                fitter = std::thread([&] {
                    SlotBind bind(slot);
                    fit_rc = fit();
                    if (fit_rc) fit_err = mhs_last_error();
                });
            auto launch = [&]() -> int {
                MHS_HIP(hipEventRecord(M->e0, M->s));
                for (int q = 0; q < nq; ++q) {
                    if (int rc2 = up.wait(q)) return rc2;
                    MHS_HIP(hipStreamWaitEvent(M->s, M->up[q], 0));
                    if (int rc2 = members_rows_dev(my, weights, banded, 0, 0, wt_total, &g, b.cov, b.r0, b.r1, ms->C, ms->dtype, ms->ld, ms->nodata,
                                                   cut[q], cut[q + 1], b.ens + (size_t)(cut[q] - b.r0) * (size_t)g.ncol, g.ncol, M->s)) return rc2;
                }
                if (!piped) {
                    if (int rc2 = members_rows_dev(my + banded, weights + banded, n_models - banded, 1, 1, wt_total, &g, b.cov, b.r0, b.r1,
                                                   ms->C, ms->dtype, ms->ld, ms->nodata, b.r0, b.r1, b.ens, g.ncol, M->s)) return rc2;
                    MHS_HIP(hipEventRecord(M->e1, M->s));
                }
                return MHS_OK;
            };
            TEAM_DO(team, launch());
            if (fitter.joinable()) {
                fitter.join();
                if (fit_rc) { set_error("%s", fit_err.c_str()); team.fail(fit_rc); }
            }
            up.join();
            // the caller's planes may be released once the entry point returns: no copy may still be reading them
            if (team.failed()) (void)hipStreamSynchronize(M->u);
        } else {
            if (nb > 0 && !team.failed()) {
                auto launch = [&]() -> int {
                    MHS_HIP(hipEventRecord(M->e0, M->s));
                    if (int rc2 = ensemble_band_dev(my, weights, n_models, wt_total, &g, b.cov, ms->C, ms->dtype, ms->ld, ms->nodata, b.r0, b.r1,
                                                    b.ens, g.ncol, M->s)) return rc2;
                    MHS_HIP(hipEventRecord(M->e1, M->s));
                    return MHS_OK;
                };
                TEAM_DO(team, launch());
            }
            if (slot == 0) TEAM_DO(team, fit());
        }
        team.bar.wait();                                               // residuals (and the global fit) are there
        // the last member sub-band by sub-band, each scaled, final.TPS (b.tot) added and sent down under the next (host planes)
        auto down = [&]() -> int {
            int64_t cut[MAX_SUB + 1];
            const int nd = sub_bands(b.r0, b.r1, BANDS_DOWN, cut);
            for (int q = 0; q < nd; ++q) {
                const size_t off = (size_t)(cut[q] - b.r0) * (size_t)g.ncol;
                if (cut[q + 1] > cut[q]) {
                    if (int rc2 = members_rows_dev(my + banded, weights + banded, n_models - banded, 1, 1, wt_total, &g, b.cov, b.r0, b.r1, ms->C,
                                                   ms->dtype, ms->ld, ms->nodata, cut[q], cut[q + 1], b.ens + off, g.ncol, M->s)) return rc2;
                    if (int rc2 = mhs_scale_add_dev(b.ens + off, 1.0, b.tot + off, b.tot + off, (cut[q + 1] - cut[q]) * g.ncol, M->s)) return rc2;
                }
                MHS_HIP(hipEventRecord(M->dn[q], M->s));
            }
            MHS_HIP(hipEventRecord(M->e1, M->s));
            const double t0 = now_ms();
            for (int q = 0; q < nd; ++q) {
                if (cut[q + 1] == cut[q]) continue;
                const size_t off = (size_t)(cut[q] - b.r0) * (size_t)g.ncol;
                MHS_HIP(hipStreamWaitEvent(M->u, M->dn[q], 0));
                if (q == nd - 1) S.download_ms[slot] = -now_ms();      // what is left once the last sub-band is final
                MHS_HIP(hipMemcpyAsync(ms->pending_out + (size_t)b.r0 * (size_t)g.ncol + off, b.tot + off,
                                       sizeof(double) * (size_t)(cut[q + 1] - cut[q]) * (size_t)g.ncol, hipMemcpyDeviceToHost, M->u));
            }
            MHS_HIP(hipStreamSynchronize(M->u));
            S.download_ms[slot] = S.download_ms[slot] < 0 ? S.download_ms[slot] + now_ms() : now_ms() - t0;
            return MHS_OK;
        };
        if (!tiled) {
            // ---- Step 3, global: every slot evaluates ITS rows with the whole grid's plan (V73:753)
            if (slot != 0 && nb > 0)
                TEAM_DO(team, mhs_tps_from_coef(S.knots_uv.data(), S.c.data(), S.d, S.nk, S.lambda, S.center, S.scale, &tps[(size_t)slot]));
            if (nb > 0)
                TEAM_DO(team, tps_predict_rows_dev(tps[(size_t)slot], &g, 0, g.nrow, 0, g.ncol, b.r0, b.r1, b.tot, g.ncol, M->s));
            if (piped && nb > 0) {
                const int rcd = team.failed() ? MHS_OK : down();
                if (rcd) { (void)hipStreamSynchronize(M->u); team.fail(rcd); }
            }
        } else {
            // ---- Step 3 the way the reference computes it above 1500 px (V73:636-747): this slot's tiles, then every tile
            // from its owner (peer copies), mosaic + feather on the whole grid, this slot's rows of it
            const double t0 = now_ms();
            auto my_tiles = [&]() -> int {
                if (b.tiles_cap < S.slot_need[slot]) {
                    if (b.tiles) { MHS_HIP(hipStreamSynchronize(M->s)); MHS_HIP(hipFree(b.tiles)); b.tiles = nullptr; b.tiles_cap = 0; }
                    MHS_HIP(hipMalloc((void **)&b.tiles, sizeof(double) * S.slot_need[slot]));
                    b.tiles_cap = S.slot_need[slot];
                }
                std::vector<int64_t> ids;
                std::vector<double *> outs;
                for (int64_t h = 0; h < nt; ++h)
                    if (S.owner[(size_t)h] == slot) { ids.push_back(h); outs.push_back(b.tiles + S.slot_off[slot][(size_t)h]); }
                if (ids.empty()) return MHS_OK;
                return mhs_tps_tiles_dev(&g, knots, S.resid.data(), n, X /* covariate 1 at the stations */, tile_edge, lambda, gcv_mode,
                                         ids.data(), (int64_t)ids.size(), outs.data());
            };
            TEAM_DO(team, my_tiles());
            S.tiles_ms[slot] = now_ms() - t0;
            team.bar.wait();                                           // every tile is final on its owner
            auto bring = [&]() -> int {
                if (nb == 0) return MHS_OK;
                std::vector<const double *> ptrs((size_t)nt, nullptr);
                for (int64_t h = 0; h < nt; ++h) {
                    const int64_t *kw = &S.keep_win[(size_t)h * 4];
                    if (!(kw[0] < b.r1 && kw[1] > b.r0)) continue;     // does not reach this slot's rows
                    const int o = S.owner[(size_t)h];
                    ptrs[(size_t)h] = b.tiles + S.slot_off[slot][(size_t)h];
                    if (o == slot) continue;
                    const size_t bytes = sizeof(double) * S.tile_cells[(size_t)h];
                    MHS_HIP(hipMemcpyPeerAsync(b.tiles + S.slot_off[slot][(size_t)h], ctx_slot(slot).device,
                                               ms->b[o].tiles + S.slot_off[o][(size_t)h], ctx_slot(o).device, bytes, M->s));
                    S.pulled_bytes[slot] += (int64_t)bytes;
                }
                // Step 4 on this slot's rows only, into its band (the tiles are spline planes: no NA)
                return mosaic_feather_impl(&g, S.nRx, S.nCx, S.keep_win.data(), ptrs.data(), 0, b.tot, g.ncol, nullptr, M->s, true, b.r0, b.r1);
            };
            TEAM_DO(team, bring());
            if (piped && nb > 0) {
                const int rcd = team.failed() ? MHS_OK : down();
                if (rcd) { (void)hipStreamSynchronize(M->u); team.fail(rcd); }
            }
        }
        // ---- Step 5 on the band (V73:906-917): sum, the stations' cells
        if (nb > 0 && !team.failed()) {
            auto step5 = [&]() -> int {
                if (!piped) if (int rc2 = mhs_scale_add_dev(b.ens, 1.0, b.tot, b.tot, nb * g.ncol, M->s)) return rc2;
                std::vector<int64_t> rr, cc, idx;
                for (int64_t i = 0; i < n; ++i)
                    if (rows[(size_t)i] >= b.r0 && rows[(size_t)i] < b.r1 && cols[(size_t)i] >= 0) {
                        rr.push_back(rows[(size_t)i] - b.r0); cc.push_back(cols[(size_t)i]); idx.push_back(i);
                    }
                std::vector<double> f(rr.size());
                if (int rc2 = mhs_gather_cells_dev(b.tot, g.ncol, rr.data(), cc.data(), (int64_t)rr.size(), f.data(), M->s)) return rc2;
                for (size_t q = 0; q < idx.size(); ++q) S.f_actual[(size_t)idx[q]] = f[q];
                return MHS_OK;
            };
            TEAM_DO(team, step5());
        }
        team.bar.wait();                                               // every station's cell has been read
        if (slot == 0 && !team.failed()) {                             // V73:917-930: keep the sum iff it improves R^2
            long double rs = 0.0L;
            for (int64_t i = 0; i < n; ++i) { const double e = resp[i] - S.f_actual[(size_t)i]; rs += (long double)(e * e); }
            S.rsq_final = 1.0 - (double)rs / S.tss;
            S.used_tps = S.rsq_final > S.rsq_model ? 1 : 0;
        }
        team.bar.wait();
        // ---- read here, right behind the barrier and before anything that can fail: every thread takes the same branch at
        // the collective below (a collective must not be entered by some ranks only; round-5 advisor finding: the re-send
        // in between can call team.fail())
        const bool go = !team.failed();
        if (piped && nb > 0 && go && !S.used_tps) {
            auto resend = [&]() -> int {
                MHS_HIP(hipMemcpyAsync(ms->pending_out + (size_t)b.r0 * (size_t)g.ncol, b.ens, sizeof(double) * (size_t)nb * (size_t)g.ncol,
                                       hipMemcpyDeviceToHost, M->s));
                return MHS_OK;
            };
            TEAM_DO(team, resend());
        }
        // ---- the one collective: stitch the final plane on every device
        if (gather && go) {
            auto stitch = [&]() -> int {
                const int64_t band = ms->plan.band;
                if (!b.full) MHS_HIP(hipMalloc((void **)&b.full, sizeof(double) * (size_t)band * (size_t)N * (size_t)g.ncol));
                const double *src = S.used_tps ? b.tot : b.ens;
                // this slot's rows at their place in the gather target (slot 0's at the end of its chunk)
                double *mine = b.full + ((size_t)ms->plan.lead + (size_t)b.r0) * (size_t)g.ncol;
                if (nb > 0) MHS_HIP(hipMemcpyAsync(mine, src, sizeof(double) * (size_t)nb * (size_t)g.ncol, hipMemcpyDeviceToDevice, M->s));
                return MHS_OK;
            };
            int rcs = stitch();
            if (rcs) team.fail(rcs);
            team.bar.wait();                                           // every slot's target exists
            const bool go2 = !team.failed();
            if (go2 && use_rccl) {
                const size_t count = (size_t)ms->plan.band * (size_t)g.ncol;
                const int rcn = g_rccl.AllGather(b.full + (size_t)slot * count, b.full, count, NCCL_FLOAT64, g_rccl.comm[slot], M->s);
                if (rcn != 0) { set_error("ncclAllGather failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rcn) : "?"); team.fail(MHS_ERR_HIP); }
            } else if (go2 && nb > 0) {
                // aliased slots (or no librccl): every slot PUSHES its rows into the other slots' targets -- the mesh pattern of
                // SURVEY.md section 8e
                auto push = [&]() -> int {
                    const double *mine = b.full + ((size_t)ms->plan.lead + (size_t)b.r0) * (size_t)g.ncol;
                    for (int o = 0; o < N; ++o) {
                        if (o == slot) continue;
                        double *dst = ms->b[o].full + ((size_t)ms->plan.lead + (size_t)b.r0) * (size_t)g.ncol;
                        MHS_HIP(hipMemcpyPeerAsync(dst, ctx_slot(o).device, mine, ctx_slot(slot).device, sizeof(double) * (size_t)nb * (size_t)g.ncol, M->s));
                    }
                    return MHS_OK;
                };
                rcs = push();
                if (rcs) team.fail(rcs);
            }
        }
        if (M) (void)hipStreamSynchronize(M->s);
        if (nb > 0 && M && !team.failed()) {
            float ms_f = 0.f;
            if (hipEventElapsedTime(&ms_f, M->e0, M->e1) == hipSuccess) S.band_ms[slot] = (double)ms_f;
        }
        team.bar.wait();                                               // pushes have landed everywhere
        if (tps[(size_t)slot]) { (void)mhs_tps_free(tps[(size_t)slot]); tps[(size_t)slot] = nullptr; }
    });
    if (ms->pending) {
        ms->downloaded = !rc && ms->pending_out;
        ms->pending = nullptr; ms->pending_out = nullptr;
        ms->upload_ms = ms->download_ms = 0;
        for (int k = 0; k < N; ++k) { ms->upload_ms = std::max(ms->upload_ms, S.upload_ms[k]); ms->download_ms = std::max(ms->download_ms, S.download_ms[k]); }
    }
    if (rc) return rc;
    ms->used_tps = S.used_tps;
    ms->gathered = gather != 0;
    ms->have_result = true;
    const double step_ms = now_ms() - t_step0;
    // the share of the rows that would have balanced this step: slot 0's band + fit against the other slots' bands
    double suggested = NAN;
    if (N > 1 && !tiled) {
        double ms_per_row = 0.0;
        int64_t rows_timed = 0;
        for (int k = 0; k < N; ++k) if (ms->b[k].r1 > ms->b[k].r0) { ms_per_row += S.band_ms[k]; rows_timed += ms->b[k].r1 - ms->b[k].r0; }
        if (rows_timed > 0 && ms_per_row > 0) {
            const double cells_ms = ms_per_row / (double)rows_timed * (double)g.nrow;
            const double s0 = 1.0 / N - S.fit_ms * (N - 1) / ((double)N * cells_ms);
            suggested = std::min(std::max(s0, 0.0), 1.0 / N);
            std::lock_guard<std::mutex> lk(g_balance_mu);
            // hysteresis (round-5 advisor finding): the suggestion moves with the jitter of fit_ms / band_ms, and a plan that
            // moves slot 0's band by one 16-row unit makes mhs_mltps_grid_multi free and rebuild gigabytes of cached
            // device buffers -- the remembered share only follows a change of more than 5 % of a band
            const bool same_shape = g_balance.n == N && g_balance.nrow == g.nrow && g_balance.ncol == g.ncol && g_balance.stations == n;
            if (!same_shape || !(fabs(suggested - g_balance.share) <= 0.05 / (double)N))
                g_balance = Balance{N, g.nrow, g.ncol, n, suggested};
        }
    }
    if (info) {
        memset(info, 0, sizeof(*info));
        info->rsq_model = S.rsq_model; info->rsq_final = S.rsq_final; info->lambda = tiled ? NAN : S.lambda;
        info->n_knots = tiled ? 0 : S.nk; info->used_tps = S.used_tps; info->n_slots = N;
        info->tiles_rows = S.nRx; info->tiles_cols = S.nCx;
        info->collective = !gather ? 0 : use_rccl ? 1 : 2;
        info->fit_ms = S.fit_ms; info->step_ms = step_ms; info->suggested_slot0_share = suggested;
        for (int k = 0; k < N; ++k) {
            info->band_r0[k] = ms->b[k].r0; info->band_r1[k] = ms->b[k].r1; info->band_ms[k] = S.band_ms[k]; info->tiles_ms[k] = S.tiles_ms[k];
            info->tiles_pulled_bytes[k] = S.pulled_bytes[k];
            info->tiles_owned[k] = 0;
            for (size_t h = 0; h < S.owner.size(); ++h) if (S.owner[h] == k) ++info->tiles_owned[k];
        }
    }
    return MHS_OK;
}

// The last step's final plane, every slot's rows straight down its own PCIe link into the caller's plane (row-major, ld = ncol)
int mhs_multi_final_download(const mhs_multi_stack *ms, double *final_host) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(ms && final_host && ms->have_result, "no result to download");
    MHS_REQUIRE(stack_current(ms), "the stack was built for another set of device slots");
    Team team(ms->n);
    return team.run([&](int slot) {
        const MultiBand &b = ms->b[slot];
        const int64_t nb = b.r1 - b.r0;
        auto work = [&]() -> int {
            if (nb == 0) return MHS_OK;
            MultiSlot *M = nullptr;
            if (int rc2 = multi_slot(slot, &M)) return rc2;
            const double *src = ms->used_tps ? b.tot : b.ens;
            MHS_HIP(hipMemcpyAsync(final_host + (size_t)b.r0 * (size_t)ms->g.ncol, src, sizeof(double) * (size_t)nb * (size_t)ms->g.ncol,
                                   hipMemcpyDeviceToHost, M->s));
            MHS_HIP(hipStreamSynchronize(M->s));
            return MHS_OK;
        };
        TEAM_DO(team, work());
    });
}

// Device pointers of the last step's planes on one slot: the slot's own rows [r0, r1) (band_dev, ld = ncol) and, after a
// step with `gather`, the stitched grid (full_dev, nrow x ncol).  Any output pointer may be NULL.
int mhs_multi_final_dev(const mhs_multi_stack *ms, int slot, double **band_dev, int64_t *r0, int64_t *r1, double **full_dev) {
    MHS_REQUIRE(ms && slot >= 0 && slot < ms->n && ms->have_result, "bad arguments");
    const MultiBand &b = ms->b[slot];
    if (band_dev) *band_dev = ms->used_tps ? b.tot : b.ens;
    if (r0) *r0 = b.r0;
    if (r1) *r1 = b.r1;
    if (full_dev) *full_dev = (ms->gathered && b.full) ? b.full + (size_t)ms->plan.lead * (size_t)ms->g.ncol : nullptr;
    return MHS_OK;
}

// The same from host planes to a host plane in ONE call -- what the R shim binds (terra holds the rasters in RAM,
// V73:468-606): upload the bands, Steps 2-5, download.  slot0_share: NaN = automatic (equal bands the first time, then
// what the last step of the same shape measured).
int mhs_mltps_grid_multi(const mhs_model *const *models, const double *weights, int n_models, double wt_total, const mhs_grid *g,
                         const mhs_stack *covars_host, const double *X, const double *resp, int64_t n, int64_t tile_edge,
                         double lambda, int gcv_mode, double slot0_share, double *final_host, mhs_mltps_info *info) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(g && covars_host && final_host, "NULL argument");
    double share = slot0_share;
    if (std::isnan(share)) {
        std::lock_guard<std::mutex> lk(g_balance_mu);
        if (g_balance.n == slot_count() && g_balance.nrow == g->nrow && g_balance.ncol == g->ncol && g_balance.stations == n) share = g_balance.share;
    }
    // validated by stack_build below when the buffers are (re)built; a cached stack needs the same checks
    MHS_REQUIRE(covars_host->data && g->nrow > 0 && g->ncol > 0, "bad arguments");
    MHS_REQUIRE(covars_host->dtype == MHS_F64 || covars_host->dtype == MHS_F32 || covars_host->dtype == MHS_I16, "bad stack dtype");
    MHS_REQUIRE(covars_host->ld >= g->ncol && covars_host->plane_stride >= covars_host->ld * g->nrow, "stack strides smaller than the grid");
    std::lock_guard<std::mutex> lk(g_host_mu);
    const double t0 = now_ms();
    const BandPlan want = plan_bands(g->nrow, slot_count(), share);
    mhs_multi_stack *ms = g_host_ms;
    if (ms && !(stack_current(ms) && ms->g.nrow == g->nrow && ms->g.ncol == g->ncol && ms->C == covars_host->n_layers &&
                ms->dtype == covars_host->dtype && ms->plan.r0 == want.r0 && ms->plan.r1 == want.r1)) {
        free_stack(ms);
        ms = g_host_ms = nullptr;
    }
    if (!ms) {
        if (int rc = stack_build(g, covars_host, share, false, &ms)) return rc;
        g_host_ms = ms;
    }
    ms->g = *g; ms->nodata = covars_host->nodata; ms->have_result = false;
    ms->pending = covars_host;                       // the step brings each band up under its own first kernels ...
    ms->pending_out = final_host;                    // ... and sends it down under its last ones
    ms->downloaded = false;
    const double t1 = now_ms();
    int rc = mhs_mltps_grid_multi_dev(models, weights, n_models, wt_total, ms, X, resp, n, tile_edge, lambda, gcv_mode, 0, info);
    ms->pending = nullptr; ms->pending_out = nullptr;
    const double t2 = now_ms();
    if (!rc && !ms->downloaded) rc = mhs_multi_final_download(ms, final_host);
    // upload_ms: buffers (first call of a shape) + the time the slowest slot's helper spent in its copies, most of it under that
    // slot's kernels; download_ms: the copies down that were NOT under kernels
    if (info && !rc) { info->upload_ms = (t1 - t0) + ms->upload_ms; info->download_ms = ms->downloaded ? ms->download_ms : now_ms() - t2; }
    return rc;
}

}  // extern "C"

// ------------------------------------------------------------------------------- (tile, layer) units --
namespace {

struct UnitSlot {                 // what one slot uses while its units run (all of it carved out of the slot's UnitsArena)
    std::vector<char *> cov;      // tile t's covariate rows x cols, uploaded when the slot's first unit of the tile starts
    double *ens = nullptr, *tps = nullptr;      // scratch planes of the largest tile
    std::vector<double *> merge_in;             // a layer's tile planes on the merging slot
    double *merged = nullptr;
};

// which response layers have all their tiles final (the merging slots wait on it)
struct LayerBoard {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int64_t> done;
    bool stop = false;
    void finished(int l) { std::lock_guard<std::mutex> lk(mu); ++done[(size_t)l]; cv.notify_all(); }
    void abort() { std::lock_guard<std::mutex> lk(mu); stop = true; cv.notify_all(); }
    bool wait(int l, int64_t n_tiles) {       // false: the call failed somewhere
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || done[(size_t)l] >= n_tiles; });
        return !stop;
    }
};

}  // namespace

extern "C" int mhs_tiles_units_multi(const mhs_grid *g, const mhs_stack *covars_host, int64_t out_ncol, int64_t out_nrow,
                                     double feather_d, int n_layers, const mhs_unit *units, int tps, int64_t tile_edge,
                                     double lambda, int gcv_mode, double *const *merged_host, double *rsq,
                                     mhs_units_info *info) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(g && covars_host && covars_host->data && units && merged_host, "NULL argument");
    MHS_REQUIRE(g->nrow > 0 && g->ncol > 0 && g->xres > 0 && g->yres > 0, "bad grid geometry");
    MHS_REQUIRE(out_ncol >= 1 && out_nrow >= 1 && out_ncol * out_nrow <= 4096 && n_layers >= 1, "bad tile layout");
    MHS_REQUIRE(covars_host->dtype == MHS_F64 || covars_host->dtype == MHS_F32 || covars_host->dtype == MHS_I16, "bad stack dtype");
    MHS_REQUIRE(covars_host->ld >= g->ncol && covars_host->plane_stride >= covars_host->ld * g->nrow, "stack strides smaller than the grid");
    const int N = slot_count();
    const int64_t n_tiles = out_ncol * out_nrow, n_units = n_tiles * n_layers;
    const int C = covars_host->n_layers, p = C + 2;
    const size_t esz = elem_size(covars_host->dtype);
    std::vector<double> boxes((size_t)n_tiles * 4);
    std::vector<int64_t> win((size_t)n_tiles * 4);
    if (int rc = mhs_tiles_create_windows(g, out_ncol, out_nrow, feather_d, boxes.data(), win.data())) return rc;   // V73:1165-1208
    int64_t max_cells = 0;
    for (int64_t t = 0; t < n_tiles; ++t) max_cells = std::max(max_cells, (win[4 * t + 1] - win[4 * t]) * (win[4 * t + 3] - win[4 * t + 2]));
    for (int64_t u = 0; u < n_units; ++u) {
        const mhs_unit &U = units[u];
        MHS_REQUIRE(U.models && U.weights && U.n_models >= 1 && U.n_models <= 8 && U.X && U.resp && U.n > 3, "bad unit");
        MHS_REQUIRE(U.wt_total != 0.0 && !std::isnan(U.wt_total), "a unit's wt_total must be non-zero");
    }
    std::lock_guard<std::mutex> arena_lk(g_host_mu);                // the slots' arenas serve one call at a time
    const double t_start = now_ms();
    std::vector<double *> plane((size_t)n_units, nullptr);          // unit u's final plane, on its owner's device
    std::vector<double> rsq_model((size_t)n_units, NAN), rsq_final((size_t)n_units, NAN), unit_ms((size_t)n_units, 0.0);
    std::vector<UnitSlot> us((size_t)N);
    LayerBoard board;
    board.done.assign((size_t)n_layers, 0);
    Team team(N);
    const int rc = team.run([&](int slot) {
        UnitSlot &L = us[(size_t)slot];
        MultiSlot *M = nullptr;
        TEAM_DO(team, multi_slot(slot, &M));
        auto alloc = [&]() -> int {
            // everything this slot needs, carved out of its arena: the crops of its tiles, two scratch planes, its units' final
            // planes and -- if it merges a layer -- every tile's plane once more and the merged grid
            auto cells_of = [&](int64_t t) { return (size_t)((win[4 * t + 1] - win[4 * t]) * (win[4 * t + 3] - win[4 * t + 2])); };
            auto up = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
            std::vector<char> mine((size_t)n_tiles, 0);
            size_t need = 2 * up(sizeof(double) * (size_t)max_cells);
            for (int64_t u = slot; u < n_units; u += N) { mine[(size_t)(u % n_tiles)] = 1; need += up(sizeof(double) * cells_of(u % n_tiles)); }
            for (int64_t t = 0; t < n_tiles; ++t) if (mine[(size_t)t]) need += up(cells_of(t) * esz * (size_t)C);
            bool merges = false;
            for (int l = slot; l < n_layers; l += N) merges = merges || merged_host[l] != nullptr;
            if (merges) {
                for (int64_t t = 0; t < n_tiles; ++t) need += up(sizeof(double) * cells_of(t));
                need += up(sizeof(double) * (size_t)g->nrow * (size_t)g->ncol);
            }
            UnitsArena &A = g_units_arena[slot];
            if (need > A.cap) {
                if (A.base) { MHS_HIP(hipDeviceSynchronize()); MHS_HIP(hipFree(A.base)); A.base = nullptr; A.cap = 0; }
                MHS_HIP(hipMalloc((void **)&A.base, need));
                A.cap = need;
            }
            char *q = A.base;
            auto take = [&](size_t bytes) { char *r = q; q += up(bytes); return r; };
            L.ens = (double *)take(sizeof(double) * (size_t)max_cells);
            L.tps = (double *)take(sizeof(double) * (size_t)max_cells);
            L.cov.assign((size_t)n_tiles, nullptr);
            for (int64_t t = 0; t < n_tiles; ++t) if (mine[(size_t)t]) L.cov[(size_t)t] = take(cells_of(t) * esz * (size_t)C);
            for (int64_t u = slot; u < n_units; u += N)        // (read by other slots' mergers only after the unit is reported finished)
                plane[(size_t)u] = (double *)take(sizeof(double) * cells_of(u % n_tiles));
            if (merges) {
                L.merge_in.assign((size_t)n_tiles, nullptr);
                for (int64_t t = 0; t < n_tiles; ++t) L.merge_in[(size_t)t] = (double *)take(sizeof(double) * cells_of(t));
                L.merged = (double *)take(sizeof(double) * (size_t)g->nrow * (size_t)g->ncol);
            }
            return MHS_OK;
        };
        TEAM_DO(team, alloc());
        // a tile's response layers share its stations: the Step-3 tiles' reductions are built by the first layer this slot
        // runs on the tile and reused by the others (bit-identical fits); nothing outlives the call
        TEAM_DO(team, mhs_tps_reduction_cache(1));
        // ---- terra::crop of every tile this slot works on (V73:1207), uploaded by a helper thread in the order of first use, ahead
        // of the units: only the first tile's copy is waited for with nothing to run (288 GB of HBM: every crop stays resident)
        std::vector<int64_t> order;
        std::vector<int> pos((size_t)n_tiles, -1);
        for (int64_t u = slot; u < n_units; u += N)
            if (pos[(size_t)(u % n_tiles)] < 0) { pos[(size_t)(u % n_tiles)] = (int)order.size(); order.push_back(u % n_tiles); }
        std::vector<hipEvent_t> tev(order.size(), nullptr);
        Uploader pre;
        if (!order.empty() && !team.failed())
            pre.start(slot, (int)order.size(), [&](int q) -> int {
                const int64_t t = order[(size_t)q];
                const int64_t *w = &win[(size_t)t * 4];
                const int64_t nr = w[1] - w[0], nc = w[3] - w[2];
                for (int k = 0; k < C; ++k) {
                    const char *src = (const char *)covars_host->data + ((size_t)k * covars_host->plane_stride + (size_t)w[0] * covars_host->ld + (size_t)w[2]) * esz;
                    MHS_HIP(hipMemcpy2DAsync(L.cov[(size_t)t] + (size_t)k * (size_t)nr * (size_t)nc * esz, (size_t)nc * esz, src, (size_t)covars_host->ld * esz,
                                             (size_t)nc * esz, (size_t)nr, hipMemcpyHostToDevice, M->h));
                }
                MHS_HIP(hipEventCreateWithFlags(&tev[(size_t)q], hipEventDisableTiming));
                MHS_HIP(hipEventRecord(tev[(size_t)q], M->h));
                return MHS_OK;
            });
        std::vector<char> ready((size_t)n_tiles, 0);
        // ---- machisplin.tiles.merge (V73:1392-1548) of layer l on slot l mod N, by a helper thread of that slot: as soon as
        // the layer's last tile is final anywhere its tiles come over xGMI, are mosaicked and feathered on the slot's second
        // stream and go down to the caller's plane -- under the units of the following layers (the copy to pageable memory
        // blocks its thread, which is why it has its own)
        std::thread merger;
        {
            bool any = false;
            for (int l = slot; l < n_layers; l += N) any = any || merged_host[l] != nullptr;
            if (any && !team.failed()) merger = std::thread([&, slot] {
                SlotBind bind(slot);
                set_mosaic_lane(1);              // its own scratch: the units' Step-3 / Step-4 mosaics do not wait for a merge
                if (const int rcr = ring_prepare(slot)) { team.fail(rcr); board.abort(); return; }
                HostSink sink(slot, M->u);
                for (int l = slot; l < n_layers; l += N) {
                    if (!merged_host[l]) continue;
                    if (!board.wait(l, n_tiles)) return;
                    auto merge = [&]() -> int {
                        std::vector<const double *> ptrs((size_t)n_tiles);
                        for (int64_t t = 0; t < n_tiles; ++t) {
                            const int64_t u = (int64_t)l * n_tiles + t;
                            const int owner = (int)(u % N);
                            const size_t bytes = sizeof(double) * (size_t)((win[4 * t + 1] - win[4 * t]) * (win[4 * t + 3] - win[4 * t + 2]));
                            if (owner == slot) { ptrs[(size_t)t] = plane[(size_t)u]; continue; }
                            MHS_HIP(hipMemcpyPeerAsync(L.merge_in[(size_t)t], ctx_slot(slot).device, plane[(size_t)u], ctx_slot(owner).device, bytes, M->u));
                            ptrs[(size_t)t] = L.merge_in[(size_t)t];
                        }
                        if (int rc2 = mhs_mosaic_feather_dev(g, out_nrow, out_ncol, win.data(), ptrs.data(), 1, L.merged, g->ncol, nullptr, M->u)) return rc2;
                        return sink.download(merged_host[l], L.merged, sizeof(double) * (size_t)g->nrow * (size_t)g->ncol);
                    };
                    const int rcm = merge();
                    if (rcm) { team.fail(rcm); board.abort(); (void)hipStreamSynchronize(M->u); return; }
                }
            });
        }
        // ---- this slot's units, layer-major (u = layer * n_tiles + tile, slot u mod N: sharded.unit_owner)
        for (int64_t u = slot; u < n_units && !team.failed(); u += N) {
            const int64_t t = u % n_tiles;
            const mhs_unit &U = units[u];
            const int64_t *w = &win[(size_t)t * 4];
            const int64_t nr = w[1] - w[0], nc = w[3] - w[2];
            auto run = [&]() -> int {
                const double t0 = now_ms();
                // terra::crop(rast, e.ext[[t]]): the tile's geometry and its rows x cols of every plane (V73:1207)
                mhs_grid gt = {g->xmin + (double)w[2] * g->xres, g->ymax - (double)w[0] * g->yres, g->xres, g->yres, nr, nc};
                if (!ready[(size_t)t]) {       // the tile's crop is (being) uploaded by the helper: this stream waits for it once
                    if (int rc2 = pre.wait(pos[(size_t)t])) return rc2;
                    MHS_HIP(hipStreamWaitEvent(M->s, tev[(size_t)pos[(size_t)t]], 0));
                    ready[(size_t)t] = 1;
                }
                std::vector<const mhs_model *> my((size_t)U.n_models);
                int p0 = 0;
                for (int k = 0; k < U.n_models; ++k) {
                    MHS_REQUIRE(U.models[k] != nullptr, "NULL model in a unit");
                    if (int rc2 = mhs_model_info(U.models[k], nullptr, &p0, nullptr)) return rc2;
                    MHS_REQUIRE(p0 == p, "a unit's model does not match the stack (layers + 2 predictors)");
                    if (int rc2 = model_on_slot(U.models[k], slot, &my[(size_t)k])) return rc2;
                }
                // Step 2 (V73:447-620)
                if (int rc2 = ensemble_band_dev(my.data(), U.weights, U.n_models, U.wt_total, &gt, L.cov[(size_t)t], C, covars_host->dtype, nc,
                                                covars_host->nodata, 0, nr, L.ens, nc, M->s)) return rc2;
                std::vector<double> res((size_t)U.n);
                if (int rc2 = mhs_residual_points(my.data(), U.weights, U.n_models, U.wt_total, U.X, U.resp, U.n, res.data())) return rc2;
                long double msum = 0.0L, tss_l = 0.0L, rs = 0.0L;      // R's sum() / mean(): long double accumulators
                for (int64_t i = 0; i < U.n; ++i) msum += U.resp[i];
                const double mean = (double)(msum / (long double)U.n);
                for (int64_t i = 0; i < U.n; ++i) { tss_l += (long double)((U.resp[i] - mean) * (U.resp[i] - mean)); rs += (long double)(res[(size_t)i] * res[(size_t)i]); }
                const double tss = (double)tss_l;
                rsq_model[(size_t)u] = 1.0 - (double)rs / tss;
                const double *fin = L.ens;
                if (tps) {
                    // Step 3 + 4 (V73:636-897) on the tile, Step 5 (V73:902-930)
                    const double *knots = U.X + (size_t)(p - 2) * (size_t)U.n;
                    if (int rc2 = mhs_tps_surface_dev(&gt, knots, res.data(), U.n, U.X, tile_edge, lambda, gcv_mode, L.tps, nc, nullptr, M->s)) return rc2;
                    if (int rc2 = mhs_scale_add_dev(L.ens, 1.0, L.tps, L.tps, nr * nc, M->s)) return rc2;
                    std::vector<int64_t> rows((size_t)U.n), cols((size_t)U.n);
                    if (int rc2 = mhs_cells_from_xy(&gt, knots, U.n, rows.data(), cols.data())) return rc2;
                    std::vector<double> f((size_t)U.n);
                    if (int rc2 = mhs_gather_cells_dev(L.tps, nc, rows.data(), cols.data(), U.n, f.data(), M->s)) return rc2;
                    rs = 0.0L;
                    for (int64_t i = 0; i < U.n; ++i) { const double e = U.resp[i] - f[(size_t)i]; rs += (long double)(e * e); }
                    rsq_final[(size_t)u] = 1.0 - (double)rs / tss;
                    if (rsq_final[(size_t)u] > rsq_model[(size_t)u]) fin = L.tps;
                }
                MHS_HIP(hipMemcpyAsync(plane[(size_t)u], fin, sizeof(double) * (size_t)nr * (size_t)nc, hipMemcpyDeviceToDevice, M->s));
                MHS_HIP(hipStreamSynchronize(M->s));
                unit_ms[(size_t)u] = now_ms() - t0;
                return MHS_OK;
            };
            TEAM_DO(team, run());
            if (team.failed()) board.abort(); else board.finished((int)(u / n_tiles));
        }
        if (team.failed()) board.abort();
        if (ctx().ready) (void)mhs_tps_reduction_cache(0);
        pre.join();
        if (M && M->h) (void)hipStreamSynchronize(M->h);       // the caller's planes are not read after the call, whatever happened
        for (hipEvent_t e : tev) if (e) (void)hipEventDestroy(e);
        if (merger.joinable()) merger.join();
        team.bar.wait();                                               // nobody reads a unit plane any more
        if (M) (void)hipStreamSynchronize(M->s);
    });
    if (rc) return rc;
    if (rsq) for (int64_t u = 0; u < n_units; ++u) { rsq[2 * u] = rsq_model[(size_t)u]; rsq[2 * u + 1] = rsq_final[(size_t)u]; }
    if (info) {
        memset(info, 0, sizeof(*info));
        info->n_slots = N; info->n_units = n_units; info->step_ms = now_ms() - t_start;
        for (int64_t u = 0; u < n_units; ++u) {
            info->unit_ms_sum += unit_ms[(size_t)u];
            info->unit_ms_max = std::max(info->unit_ms_max, unit_ms[(size_t)u]);
            info->slot_ms[u % N] += unit_ms[(size_t)u];
        }
    }
    return MHS_OK;
}
