// randomForest on a grid window: the level-synchronous walk kernels (trees in LDS as key-space node records, a lane's
// cells walking them in lock step) and the loader that numbers a forest's nodes for them -- predict.randomForest inside
// terra::predict(rast_stack, mod.rf) of machisplin.mltps Step 2 (V73:447-619).  LDS-latency bound: see DESIGN.md section 4.
#include <type_traits>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ensemble_int.h"
#include "devmath.h"

namespace mhs {

// ------------------------------------------------- randomForest: level-synchronous walk --
// One tree at a time lives in LDS (from byte 0) as 8-byte node records plus the node predictions.
// Terminals point at themselves, so every lane descends a fixed, wave-uniform number of levels (the
// tree's depth) with no divergent control flow.  As in gbm_lut_kernel the cell's predictors are
// first replaced by their RANK among the forest's sorted distinct key-space thresholds of that
// predictor ("x <= split" <=> key < tkey_j <=> rank <= j), which makes a level three VALU
// instructions around its two dependent LDS reads:
//   node = {(j << 8) | (var * 4R),  left byte address | right byte address << 16}
//   key address  = lane base + BYTE_0(node.x)             (v_add_u32 with an SDWA byte select)
//   go right     = (rank << 8) > node.x                   (the var byte cannot flip the compare)
//   next address = right ? WORD_1(node.y) : WORD_0(node.y)  (v_cndmask_b32 with SDWA word selects)
// Keys are parked in LDS as [lane][var][cell slot] with an odd lane stride (conflict-free when the
// lanes of a wave read the same predictor).  R independent walks per lane keep the two dependent
// LDS reads of a level in flight.
constexpr int RF_COARSE_BYTES = LUT_COARSE * (int)sizeof(float);
constexpr unsigned RF_LEAF_WORD = 0xffffff00u;     // RF_SMALL: word 0 of a terminal node's record (rf_walk_loop5x.inc tests for it)
// LDS accesses by 32-bit byte address (the walk's node addresses come out of LDS data, so no pointer
// arithmetic may be attached to them); the kernel's dynamic LDS starts at address 0 (no static LDS)
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2v lds_u2(unsigned a) { return *(__attribute__((address_space(3))) const uint2v *)(uintptr_t)a; }
__device__ __forceinline__ unsigned lds_u32(unsigned a) { return *(__attribute__((address_space(3))) const unsigned *)(uintptr_t)a; }
__device__ __forceinline__ double lds_f64(unsigned a) { return *(__attribute__((address_space(3))) const double *)(uintptr_t)a; }

// walks per lane for the "log2r" code the forest tables are built with: 1 -> 2, 2 -> 4, 3 -> 5 (the double-buffered
// kernel only: five walks' keys still fit beside two tree buffers when the trees are small)
__host__ __device__ constexpr int rf_walks(int code) { return code >= 3 ? code + 2 : (1 << code); }   // 1 2 4 | 5 6 7 8


// A lane's R cells in the forest walk kernels.  STRIPS (grids): the same column of R adjacent rows -- the rows are cut
// into strips of R, a lane index runs along a strip and on into the next one -- so that a wave's 64 R cells are
// neighbours, end in neighbouring leaves, and the wave can leave a tree at its cells' deepest leaf instead of the tree's.
// Otherwise (few rows, e.g. the stations' point list): cell i0 + c * ceil(total / R).
template <int R>
__device__ __forceinline__ void rf_lane_cells(const PredGeom &g, int64_t i0, int strips, int (&row)[R], int (&col)[R], bool (&live)[R]) {
    const int64_t total = (int64_t)g.nr * g.nc;
    const int64_t part = (total + R - 1) / R;
#pragma unroll
    for (int c = 0; c < R; ++c) {
        if (strips) {
            const int64_t sr = i0 / g.nc;
            col[c] = (int)(i0 - sr * g.nc);
            const int64_t r = sr * R + c;
            live[c] = r < g.nr;
            row[c] = (int)(live[c] ? r : g.nr - 1);
        } else {
            int64_t i = i0 + c * part;
            live[c] = i0 < part && i < total;
            if (i >= total) i = total - 1;
            row[c] = (int)(i / g.nc); col[c] = (int)(i - (int64_t)row[c] * g.nc);
        }
    }
}
static int64_t rf_lane_count(const PredGeom &g, int R, int strips) {      // lanes a launch needs
    const int64_t total = (int64_t)g.nr * g.nc;
    return strips ? (((int64_t)g.nr + R - 1) / R) * g.nc : (total + R - 1) / R;
}
static int rf_strips(const PredGeom &g, int R) { return g.nr >= 4 * R; }

// WAVE-UNIFORM PREFIX of the forest walks (round 3; grids).  With LANE = TREE (64 trees at a time, node records read from
// global memory) every tree is descended for as long as the split threshold lies outside the wave's [min, max] rank of
// the split's predictor (the ranks are the keys already parked in LDS).  entry[b], lane l = tree 64 b + l: entry node in
// the low 16 bits, levels descended in bits 16..30, bit 31 = the entry node is terminal.
constexpr int RF_ENTRY_BATCHES = 16;                                // batches of 64 trees held in registers
constexpr int RF_PREFIX_MAX_P = 12;                                 // the wave's [min, max] ranks are held for this many predictors;
                                                                    // forests with more walk from the root (prefix off)
template <int R>
__device__ __forceinline__ void rf_prefix_entries(unsigned (&entry)[RF_ENTRY_BATCHES], const uint2 *__restrict__ gnodes,
                                                  const int *__restrict__ tree_off, int n_trees, int p, const char *smem,
                                                  unsigned lane_base, const bool (&na)[R]) {
    const int lane = threadIdx.x & 63;
    int mn[RF_PREFIX_MAX_P], mx[RF_PREFIX_MAX_P];                  // callers guarantee p <= RF_PREFIX_MAX_P
#pragma unroll
    for (int v = 0; v < 12; ++v) {
        mn[v] = 0x7fffffff; mx[v] = -1;
        if (v < p) {
            int a = 0x7fffffff, b = -1;
#pragma unroll
            for (int c = 0; c < R; ++c)
                if (!na[c]) {
                    const int rk = (int)(*(const unsigned *)(smem + lane_base + (unsigned)(v * R + c) * 4u) >> 8);
                    a = min(a, rk); b = max(b, rk);
                }
#pragma unroll
            for (int q = 32; q > 0; q >>= 1) { a = min(a, __shfl_xor(a, q)); b = max(b, __shfl_xor(b, q)); }
            mn[v] = __builtin_amdgcn_readfirstlane(a); mx[v] = __builtin_amdgcn_readfirstlane(b);
        }
    }
#pragma unroll
    for (int b = 0; b < RF_ENTRY_BATCHES; ++b) {
        if (b * 64 < n_trees) {
            const int t = b * 64 + lane;
            bool walking = t < n_trees;
            const int o = tree_off[min(t, n_trees - 1)];
            unsigned nd = 0u, plen = 0u, term = 0u;
            while (__builtin_amdgcn_ballot_w64(walking)) {
                if (walking) {
                    const uint2 rec = gnodes[o + (int)nd];
                    if (rec.x == RF_LEAF_WORD) { term = 1u; walking = false; }
                    else {
                        const int j = (int)(rec.x >> 8), v = (int)(rec.x & 0xFFu) / (4 * R);
                        int lo = mn[0], hi = mx[0];
#pragma unroll
                        for (int q = 1; q < 12; ++q) if (q < p && v == q) { lo = mn[q]; hi = mx[q]; }
                        if (lo > j) { nd = (rec.y >> 16) >> 3; ++plen; }            // every cell's rank > j: right
                        else if (hi <= j) { nd = (rec.y & 0xFFFFu) >> 3; ++plen; }   // every cell's rank <= j: left
                        else walking = false;
                    }
                }
            }
            entry[b] = nd | (plen << 16) | (term << 31);
        }
    }
}

// Double-buffered form for trees of up to 4095 nodes (two buffers stay within the 16-bit child addresses): the
// next tree travels global -> registers -> the other LDS buffer WHILE this one is walked, the node predictions are
// read from global memory one tree behind (issued after a walk, added after the next one, in tree order), and a
// tree costs one barrier.  In the single-buffer form a third of the kernel was staging: every wave idle while
// 48 KB are copied between two barriers, 500 times per block.
template <int LOG2R, bool K64>
__global__ __launch_bounds__(1024) void rf_walk_db_kernel(const uint2 *__restrict__ gnodes,
                                                          const double *__restrict__ glval,
                                                          const int *__restrict__ tree_off,
                                                          const int *__restrict__ depth,
                                                          const void *__restrict__ sorted, int key64,
                                                          const int *__restrict__ sorted_off, int n_trees,
                                                          int max_nodes, int p, StackDev s, PredGeom g,
                                                          double weight, int accumulate,
                                                          double *__restrict__ out, const int *__restrict__ dmin, int strips, int prefix) {
    constexpr int R = rf_walks(LOG2R);
    constexpr int PF = 4;                                          // node records per thread in flight (max_nodes <= 4095)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned buf_bytes = (unsigned)max_nodes * 8u;           // one tree's nodes; the two buffers sit at 0 and buf_bytes
    const unsigned tree_bytes = max(2u * buf_bytes, (unsigned)RF_COARSE_BYTES);
    float *coarse = (float *)smem;                                 // rank search scratch (before the first tree)
    const unsigned stride = (unsigned)(p * R) | 1u;                // dwords of keys per lane
    const unsigned lane_base = tree_bytes + threadIdx.x * stride * 4u;
    if ((unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap();
    const int64_t i0 = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    int row[R], col[R];
    bool na[R], live[R];
    double acc[R], pending[R];
    unsigned node[R];
    rf_lane_cells<R>(g, i0, strips, row, col, live);
#pragma unroll
    for (int c = 0; c < R; ++c) { na[c] = false; acc[c] = 0.0; pending[c] = 0.0; }
    for (int j = 0; j < p; ++j) {
        float r[R];
        if constexpr (K64) lut_ranks_t<R, 1024, double>(j, (const double *)sorted, sorted_off, (double *)coarse, s, g, row, col, na, r);
        else lut_ranks_t<R, 1024, float>(j, (const float *)sorted, sorted_off, coarse, s, g, row, col, na, r);
#pragma unroll
        for (int c = 0; c < R; ++c) *(unsigned *)(smem + lane_base + (unsigned)(j * R + c) * 4u) = (unsigned)r[c] << 8;
    }
    // The wave's 64 R cells are neighbours: near the root of a tree they all go the same way.  Where that stops -- at a
    // terminal node, then the whole wave shares the tree's prediction, or at the first split that separates the wave's cells
    // -- is where the cell walks of the tree loop below START (rf_prefix_entries).  The cells end at the same nodes as
    // from the root: identical planes.
    constexpr int EB = RF_ENTRY_BATCHES;
    unsigned entry[EB];
#pragma unroll
    for (int b = 0; b < EB; ++b) entry[b] = 0u;
    if (prefix && p <= RF_PREFIX_MAX_P && n_trees <= 64 * EB) rf_prefix_entries<R>(entry, gnodes, tree_off, n_trees, p, smem, lane_base, na);
    __syncthreads();                                               // coarse table no longer needed: buffer 0 may be written
    {
        const int o = tree_off[0], cnt = tree_off[1] - o;
        for (int e = threadIdx.x; e < cnt; e += 1024) ((uint2 *)smem)[e] = gnodes[o + e];
    }
    __syncthreads();
    // tree t's scalars (offsets, depth) are fetched one iteration ahead: an s_load at the top of every tree
    // would stall all 16 waves for its latency, 500 times
    int o = tree_off[0], o1 = tree_off[1], o2 = n_trees > 1 ? tree_off[2] : o1, o3 = n_trees > 2 ? tree_off[3] : o2;
    int levels = depth[0], levels1 = n_trees > 1 ? depth[1] : 0;
    int shallow = dmin ? dmin[0] : levels, shallow1 = n_trees > 1 ? (dmin ? dmin[1] : levels1) : 0;
    unsigned ecur = 0u;
    // The node records travel global -> registers -> LDS one tree AND one iteration ahead: tree t + 2 is requested at the top
    // of iteration t and parked at the top of iteration t + 1 (its buffer, tree t's, is free after the barrier that ends
    // iteration t), so the request has a whole iteration to arrive even when the walks are short (most of a forest's trees
    // end, for a wave of neighbouring cells, at or near the entry node).
    uint2 pn[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) {
        const int e = threadIdx.x + q * 1024;
        if (n_trees > 1 && e < o2 - o1) pn[q] = gnodes[o1 + e];
    }
    for (int t = 0; t < n_trees; ++t) {
        if ((t & 63) == 0) {
            ecur = 0u;
#pragma unroll
            for (int b = 0; b < EB; ++b) if ((t >> 6) == b) ecur = entry[b];
        }
        const unsigned ent = (unsigned)__builtin_amdgcn_readlane((int)ecur, t & 63);
        const unsigned boff = (t & 1) ? buf_bytes : 0u, noff = (t & 1) ? 0u : buf_bytes;
        const int cnt1 = t + 1 < n_trees ? o2 - o1 : 0, cnt2 = t + 2 < n_trees ? o3 - o2 : 0;
        const int o4 = t + 4 <= n_trees ? tree_off[t + 4] : o3;        // consumed two iterations from now
        const int levels2 = t + 2 < n_trees ? depth[t + 2] : 0;
        const int shallow2 = t + 2 < n_trees ? (dmin ? dmin[t + 2] : levels2) : 0;
        {   // park tree t + 1 in the other buffer, child addresses moved there; then request tree t + 2
            const unsigned reloc = noff * 0x10001u;
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int e = threadIdx.x + q * 1024;
                if (e < cnt1) { uint2 nd = pn[q]; nd.y += reloc; *(uint2 *)(smem + noff + (unsigned)e * 8u) = nd; }
            }
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int e = threadIdx.x + q * 1024;
                if (e < cnt2) pn[q] = gnodes[o2 + e];
            }
        }
        // the walks start at the wave's entry node of this tree and descend what is left of the tree's depth
        const int plen = (int)((ent >> 16) & 0x7FFFu);
        const int lev = (ent >> 31) ? 0 : levels - plen, shal = max(shallow - plen, 0);
#pragma unroll
        for (int c = 0; c < R; ++c) node[c] = boff + ((ent & 0xFFFFu) << 3);
        if constexpr (R == 5) {
            // five walks: the level loop by hand (tools/gen_rf_walk_asm.py) -- the walks rotated so that the two wait states an
            // SDWA select needs after v_cmp's write of VCC are the previous walk's next node read and the next walk's key wait
            // lev - 1 of them with a next level: the first min(shallowest leaf, lev - 1) untested, the others leave the
            // loop when every walk of the wave has reached a terminal node; then the last level
            int c0 = min(shal, lev - 1), cnt = lev - 1 - c0;
            if (lev > 0)
                asm volatile(
#include "rf_walk_loop5x.inc"
                    : [n0] "+v"(node[0]), [n1] "+v"(node[1]), [n2] "+v"(node[2]), [n3] "+v"(node[3]), [n4] "+v"(node[4]), [cnt] "+s"(cnt),
                      [c0] "+s"(c0)
                    : [lb] "v"(lane_base)
                    : "memory", "vcc", "scc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
                      "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120");
        } else if constexpr (R == 4) {
            int c0 = min(shal, lev - 1), cnt = lev - 1 - c0;
            if (lev > 0)
                asm volatile(
#include "rf_walk_loop4x.inc"
                    : [n0] "+v"(node[0]), [n1] "+v"(node[1]), [n2] "+v"(node[2]), [n3] "+v"(node[3]), [cnt] "+s"(cnt), [c0] "+s"(c0)
                    : [lb] "v"(lane_base)
                    : "memory", "vcc", "scc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113",
                      "v115", "v116", "v117", "v118", "v120");
        } else
        for (int l = 0; l < lev; ++l) {
#pragma unroll
            for (int c = 0; c < R; ++c) {
                const uint2v nd = lds_u2(node[c]);
                const unsigned k = lds_u32(lane_base + (nd.x & 0xFFu) + c * 4);
                asm("v_cmp_gt_u32_e32 vcc, %1, %2\n\t"
                    "s_nop 1\n\t"
                    "v_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"
                    : "=v"(node[c]) : "v"(k), "v"(nd.x), "v"(nd.y) : "vcc");
            }
        }
        // the previous tree's predictions have arrived by now; this tree's are requested
#pragma unroll
        for (int c = 0; c < R; ++c) {
            acc[c] = acc[c] + pending[c];
            pending[c] = glval[o + (int)((node[c] - boff) >> 3)];
        }
        __syncthreads();                                           // every wave has left tree t; tree t + 1 is parked
        o = o1; o1 = o2; o2 = o3; o3 = o4;
        levels = levels1; levels1 = levels2;
        shallow = shallow1; shallow1 = shallow2;
    }
#pragma unroll
    for (int c = 0; c < R; ++c) {
        acc[c] = acc[c] + pending[c];
        if (live[c])
            emit(out, (int64_t)row[c] * g.ld_out + col[c], na[c] ? NAN : acc[c] / (double)n_trees, weight, accumulate);
    }
}

// TRIPLE-buffered form (round 3): the double-buffered kernel above still meets at one s_barrier per tree, and the 16 waves
// of a block do not finish a tree together (their random node reads conflict differently): measured, a quarter of that
// kernel was waves waiting at the barrier for the slowest one while the LDS pipe -- the bound of the walk -- ran dry.
// Here there is NO barrier in the tree loop.  Three node buffers at a compile-time STRIDE (the walk's ds_read_b64 carries
// the buffer in its immediate offset, so the records hold buffer-relative child addresses and need no relocation; the
// tree loop is unrolled by three); a wave that has walked tree t (buffer t % 3) parks its share of tree t + 2 in buffer
// (t + 2) % 3 -- which held tree t - 1 -- and goes on to tree t + 1, which was parked during tree t - 1.  Two monotonic
// LDS counters per buffer order this: walked[b] (+1 per wave and tree walked in b) guards the overwrite, staged[b] (+1
// per wave and tree parked in b) guards the walk.  The LDS unit executes a wave's operations in order, so "parked,
// then ds_add" and "last node read, then ds_add" need no fence; the polls are ds_read + s_waitcnt, normally satisfied at
// the first read.  A wave may run up to a whole tree ahead of the slowest one.
__device__ __forceinline__ void lds_wait_ge(unsigned addr, unsigned target) {
    for (;;) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) >= target) break;
        __builtin_amdgcn_s_sleep(2);
    }
}
__device__ __forceinline__ void lds_signal(unsigned addr) {      // the wave's first lane adds 1
    if ((threadIdx.x & 63) == 0) asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(1u) : "memory");
    else asm volatile("" ::: "memory");
}

// LOADER-WAVE form (round 4).  Round 3's ablation of the kernel above: of 64.6 ms (8 000 x 8 000 cells, 500 trees) the walks are 23
// and the "bare tree loop" 25 -- every one of the 16 waves spends ~260 instructions per tree on its share of the staging (eight
// address computations, PF predicated global loads, PF predicated LDS stores, two counter polls).  Here NL waves of the block
// (the last ones) do nothing but stage, tree u being loader u % NL's; the other 16 - NL waves (x 64 lanes x R cells) only walk.
// A loader's REGISTERS are the fourth buffer: it requests its next tree (STRIDE bytes = PF x 16 bytes per lane, all in flight,
// straight-line code) as soon as it has written the previous one to LDS, and the records wait in registers until every walker
// has left the buffer's previous tree.  A walker's step: four readlanes (prefix entry, first record, depth, shallowest leaf --
// the scalars of 64 trees come in one vector load each), a poll that is skipped while the buffer's cached counter says the tree
// is parked (the three staged counters come in one ds_read_b128), the hand-scheduled level loop, one ds_add, R prediction
// loads that are consumed FIVE trees later.  Same buffers, counters, records and order of additions as rf_walk_tb_kernel:
// identical planes.  The records' array carries STRIDE bytes of padding behind the last tree for the loaders' over-read.
// Measured on the way (profiles/r04_forest_variants.txt): staging by LDS-DMA (global_load_lds_dwordx4, M0 reaches all 160 KB, data
// visible at the issuer's vmcnt(0): tools/micro/lds_dma_range.hip) runs at the DMA path's own cadence, 1.5 us per 24.8 KB tree
// whatever is in flight and even from L2 -- no faster than one register loader (1.35 us, bound by one load round trip per tree).
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {      // 64 lanes x 16 bytes global -> LDS at lds_dst + 16 lane
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_dst) : "memory");
}
// (round 4's timing experiments -- walks, staging or synchronisation switched off, two loaders, no LDS-DMA -- are in
// profiles/r04_forest_variants.txt and in the history at commit 55aadea; round 5 removed their switches)
enum { RF_LD_PREFIX = 1 };

template <int LOG2R, bool K64, int STRIDE>
__global__ __launch_bounds__(1024) void rf_walk_ld_kernel(const uint2 *__restrict__ gnodes,
                                                          const double *__restrict__ glval,
                                                          const int *__restrict__ tree_off,
                                                          const int *__restrict__ depth,
                                                          const void *__restrict__ sorted,
                                                          const int *__restrict__ sorted_off, int n_trees,
                                                          int p, StackDev s, PredGeom g,
                                                          double weight, int accumulate,
                                                          double *__restrict__ out, const int *__restrict__ dmin, int strips, int flags,
                                                          const int *__restrict__ axis_rank, int axis_ncol) {
    constexpr int R = rf_walks(LOG2R), NL = 1, WALKERS = 16 - NL;
    constexpr unsigned TREE_BYTES = 3u * STRIDE;
    static_assert(TREE_BYTES >= (unsigned)RF_COARSE_BYTES && 2 * STRIDE <= 65535, "buffer bases inside the 16-bit immediate offsets");
    constexpr unsigned CNT = TREE_BYTES;                           // staged[3] at CNT, walked[3] at CNT + 16
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *coarse = (float *)smem;                                 // rank search scratch (before the first tree)
    const unsigned stride = (unsigned)(p * R) | 1u;                // dwords of keys per lane
    const unsigned lane_base = TREE_BYTES + 32u + threadIdx.x * stride * 4u;
    if ((unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap();
    const bool loader = threadIdx.x >= 64 * WALKERS;               // wave-uniform
    // the loaders' lanes shadow the block's first waves (they take part in the cooperative rank search and emit nothing)
    const int64_t i0 = (int64_t)blockIdx.x * (64 * WALKERS) + (loader ? threadIdx.x - 64 * WALKERS : threadIdx.x);
    int row[R], col[R];
    bool na[R], live[R];
    if (strips == 2) {
        // COMPACT wave tiles: 16 columns x 4 R rows (lane = column + 16 x row group, a lane's R walks on adjacent rows) instead of
        // 64 columns x R rows -- on smooth rasters the wave's predictor ranges are narrower, its prefix longer and its deepest
        // leaf nearer (CPU study tools/r04_rf_slice_sim.py: 3.6 -> 1.8 levels walked per wave and tree on the 8d planes)
        const int wave = (int)(threadIdx.x >> 6), lane_ = (int)(threadIdx.x & 63u);
        const int64_t tile = (int64_t)blockIdx.x * WALKERS + (loader ? wave - WALKERS : wave);
        const int tiles_x = (g.nc + 15) / 16;
        const int64_t ty = tile / tiles_x;
        const int tx = (int)(tile - ty * tiles_x);
#pragma unroll
        for (int c = 0; c < R; ++c) {
            const int64_t r = ty * (4 * R) + (lane_ >> 4) * R + c;
            const int cc = tx * 16 + (lane_ & 15);
            live[c] = r < g.nr && cc < g.nc;
            row[c] = (int)min(r, (int64_t)g.nr - 1); col[c] = min(cc, g.nc - 1);
        }
    } else
    rf_lane_cells<R>(g, i0, strips, row, col, live);
#pragma unroll
    for (int c = 0; c < R; ++c) na[c] = false;
    for (int j = 0; j < p; ++j) {
        float r[R];
        if (axis_rank && j >= s.C && !s.all_from_planes) {          // LONG / LAT: the rank is a function of the column / the row (publish_axis_ranks)
#pragma unroll
            for (int c = 0; c < R; ++c)
                r[c] = (float)(j == s.C ? axis_rank[g.c0 + col[c]] : axis_rank[(int64_t)axis_ncol + g.r0 + row[c]]);
        } else if constexpr (K64) lut_ranks_t<R, 1024, double>(j, (const double *)sorted, sorted_off, (double *)coarse, s, g, row, col, na, r);
        else lut_ranks_t<R, 1024, float>(j, (const float *)sorted, sorted_off, coarse, s, g, row, col, na, r);
#pragma unroll
        for (int c = 0; c < R; ++c) *(unsigned *)(smem + lane_base + (unsigned)(j * R + c) * 4u) = (unsigned)r[c] << 8;
    }
    unsigned entry[RF_ENTRY_BATCHES];
#pragma unroll
    for (int b = 0; b < RF_ENTRY_BATCHES; ++b) entry[b] = 0u;
    // (measured and not kept, profiles/r04_forest_variants.txt: the chains of eight batches interleaved -- finished chains re-read,
    // 6.1 -> 8.5 ms of 48 on 8 000 x 8 000 cells --; the upper levels descended once per block with the block's ranges and each
    // wave continuing from there: 6.7 ms.  The prefix is bound by the gather rate of its lane = tree record loads.)
    if (!loader && (flags & RF_LD_PREFIX) && p <= RF_PREFIX_MAX_P && n_trees <= 64 * RF_ENTRY_BATCHES)
        rf_prefix_entries<R>(entry, gnodes, tree_off, n_trees, p, smem, lane_base, na);
    __syncthreads();                                               // coarse table no longer needed
    if (threadIdx.x < 8) *(unsigned *)(smem + CNT + threadIdx.x * 4u) = 0u;
    __syncthreads();
    if (loader) {
        constexpr int PF = (STRIDE + 1023) / 1024;                  // 16-byte pieces per lane and tree
        static_assert(PF <= 25, "one named register quad per piece below");
        const unsigned lane16 = (threadIdx.x & 63u) * 16u;
        // named quads, not an array: hipcc keeps a 25 x 16-byte array in scratch memory; always PF pieces, straight-line: under
        // `piece < pieces of this tree` it waits vmcnt(0) between the loads
        uint4 r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15, r16, r17, r18, r19, r20, r21, r22, r23, r24;
        const char *src = nullptr;
#define MHS_RF_ALL(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15) F(16) F(17) F(18) F(19) F(20) \
                      F(21) F(22) F(23) F(24)
#define MHS_RF_LOAD(Q) if constexpr (Q < PF) r##Q = *(const uint4 *)(src + (size_t)Q * 1024u);
#define MHS_RF_STORE(Q) if constexpr (Q < PF) *(uint4 *)(smem + slot * (unsigned)STRIDE + (unsigned)Q * 1024u + lane16) = r##Q;
#define MHS_RF_REQUEST(U) { \
            src = (const char *)(gnodes + tree_off[(U)]) + lane16; \
            MHS_RF_ALL(MHS_RF_LOAD) }
        {
            // ONE loader, TWO trees in flight: even trees through the registers (requested while their buffer is still being
            // walked), odd trees by LDS-DMA (no registers, issued once the buffer is free; the data is in LDS at vmcnt(0)).
            // Either path alone is one round trip per tree (1.2 - 1.5 us); alternating, a pair costs about one.
            MHS_RF_REQUEST(0)
            for (int u = 0; u < n_trees; u += 2) {
                unsigned slot = (unsigned)u % 3u;
                if (u >= 3) lds_wait_ge(CNT + 16u + 4u * slot, (unsigned)WALKERS * (unsigned)(u / 3));
                MHS_RF_ALL(MHS_RF_STORE)
                lds_signal(CNT + 4u * slot);
                if (u + 2 < n_trees) MHS_RF_REQUEST(u + 2)
                if (u + 1 < n_trees) {
                    slot = (unsigned)(u + 1) % 3u;
                    if (u + 1 >= 3) lds_wait_ge(CNT + 16u + 4u * slot, (unsigned)WALKERS * (unsigned)((u + 1) / 3));
                    {
                        const char *dsrc = (const char *)(gnodes + tree_off[u + 1]) + lane16;
#pragma unroll
                        for (int q = 0; q < PF; ++q)
                            glds16(dsrc + (size_t)q * 1024u, (unsigned)__builtin_amdgcn_readfirstlane((int)(slot * (unsigned)STRIDE + (unsigned)q * 1024u)));
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    lds_signal(CNT + 4u * slot);
                }
            }
            return;
        }
#undef MHS_RF_REQUEST
#undef MHS_RF_STORE
#undef MHS_RF_LOAD
#undef MHS_RF_ALL
        return;
    }
    constexpr int PD = 6;                                          // a tree's predictions are added PD - 1 trees after their request
    double acc[R], pend[PD][R];                                    // pend[t % PD]: predictions of tree t
#pragma unroll
    for (int c = 0; c < R; ++c) {
        acc[c] = 0.0;
#pragma unroll
        for (int k = 0; k < PD; ++k) pend[k][c] = 0.0;
    }
    unsigned staged[3] = {0u, 0u, 0u};                             // cached counters: buffer b has held staged[b] trees so far
    int ocur = 0;                                                  // lane l: first record of tree 64 b + l
    unsigned wcur = 0u;                                            // lane l: its walk -- entry node's byte address | c0 << 19 | cnt << 25 | walks << 31
    double tpcur = 0.0;                                            // lane l: the prediction of its tree's entry node where the wave does not walk the tree
    const int lane = threadIdx.x & 63;
    auto step = [&](auto slot_tag, auto pidx_tag, const int t, const unsigned k3) {      // k3 = t / 3
        constexpr int SLOT = decltype(slot_tag)::value, PIDX = decltype(pidx_tag)::value;
        if ((t & 63) == 0) {
            // The next 64 trees' scalars, lane = tree: one vector load each instead of three scalar loads (and their waits) per
            // tree, and the walk's loop counts (levels without / with the exit test) formed here on the vector unit, 64 trees per
            // instruction, instead of a dozen scalar instructions per tree on the CU's one scalar unit (the PMC pass counts
            // 0.8 scalar per vector instruction in this kernel)
            unsigned ecur = 0u;
#pragma unroll
            for (int b = 0; b < RF_ENTRY_BATCHES; ++b) if ((t >> 6) == b) ecur = entry[b];
            const int tl = min(t + lane, n_trees - 1);
            ocur = tree_off[tl];
            const int dcur = depth[tl], mcur = dmin ? dmin[tl] : dcur;
            const int plen = (int)((ecur >> 16) & 0x7FFFu);
            const int levels = (ecur >> 31) ? 0 : dcur - plen, shallow = max(mcur - plen, 0);
            const int c0 = min(shallow, levels - 1), cnt = levels - 1 - c0;              // levels <= 63: depth of a tree of <= 3 200 nodes
            wcur = ((ecur & 0xFFFFu) << 3) | (levels > 0 ? ((unsigned)c0 << 19) | ((unsigned)cnt << 25) | 0x80000000u : 0u);
            // a tree the wave does not walk (its cells share the entry node: two thirds of the trees on smooth rasters) has ONE
            // prediction for all of them: fetched here, lane = tree, instead of by four 64-lane loads of one address in its step
            tpcur = levels > 0 ? 0.0 : glval[ocur + (int)(ecur & 0xFFFFu)];
            // the loads are awaited HERE: left pending, hipcc puts s_waitcnt vmcnt(0) in front of every step's readlanes (it
            // cannot count the loads issued since around the loop) and every step then waits for the previous steps'
            // prediction loads as well
            asm volatile("" : "+v"(ocur), "+v"(wcur), "+v"(tpcur));
        }
        const unsigned wk = (unsigned)__builtin_amdgcn_readlane((int)wcur, t & 63);
        const int o = __builtin_amdgcn_readlane(ocur, t & 63);
        // tree t is parked?  (also asked by a wave that skips the tree: its ds_add below must not come before the loader has
        // counted every wave out of tree t - 3, which is what "tree t is parked" implies)
        if (staged[SLOT] <= k3)
            for (;;) {
                uint4v cv;                                                   // staged[0..2] and the zero word behind them
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(cv) : "v"(CNT) : "memory");
                staged[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)cv.x);
                staged[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)cv.y);
                staged[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)cv.z);
                if (staged[SLOT] > k3) break;
                __builtin_amdgcn_s_sleep(1);
            }
        unsigned node[R];
#pragma unroll
        for (int c = 0; c < R; ++c) node[c] = wk & 0x7FFF8u;
        int c0 = (int)((wk >> 19) & 63u), cnt = (int)((wk >> 25) & 63u);
        if constexpr (R == 4) {
            if (wk >> 31)
                asm volatile(
#include "rf_walk_loop4xo.inc"
                    : [n0] "+v"(node[0]), [n1] "+v"(node[1]), [n2] "+v"(node[2]), [n3] "+v"(node[3]), [cnt] "+s"(cnt), [c0] "+s"(c0)
                    : [lb] "v"(lane_base), [off] "n"(SLOT * STRIDE)
                    : "memory", "vcc", "scc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113",
                      "v115", "v116", "v117", "v118", "v120");
        } else {
            static_assert(R == 4 || R == 5, "hand loops exist for four and five walks");
            if (wk >> 31)
                asm volatile(
#include "rf_walk_loop5xo.inc"
                    : [n0] "+v"(node[0]), [n1] "+v"(node[1]), [n2] "+v"(node[2]), [n3] "+v"(node[3]), [n4] "+v"(node[R - 1]), [cnt] "+s"(cnt),
                      [c0] "+s"(c0)
                    : [lb] "v"(lane_base), [off] "n"(SLOT * STRIDE)
                    : "memory", "vcc", "scc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
                      "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120");
        }
        // this wave has left tree t: lane 0 adds 1 (every lane is active here: no branch around the ds_add)
        asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" :: "v"(CNT + 16u + 4u * SLOT), "v"(1u) : "memory");
        const char *lv = (const char *)(glval + o);                      // node[] are byte addresses of 8-byte records = of the doubles
#pragma unroll
        for (int c = 0; c < R; ++c) acc[c] = acc[c] + pend[(PIDX + 1) % PD][c];      // tree t - (PD - 1)'s; still in tree order
        if (wk >> 31) {
#pragma unroll
            for (int c = 0; c < R; ++c) pend[PIDX][c] = *(const double *)(lv + node[c]);
        } else {
            const double tv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tpcur), t & 63), __builtin_amdgcn_readlane(__double2loint(tpcur), t & 63));
#pragma unroll
            for (int c = 0; c < R; ++c) pend[PIDX][c] = tv;
        }
    };
    static_assert(PD == 6, "the tree loop is unrolled by the least common multiple of the 3 buffers and PD");
    unsigned k3 = 0u;
    for (int t = 0; t < n_trees; t += 6, k3 += 2u) {
        step(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, t, k3);
        if (t + 1 < n_trees) step(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, t + 1, k3);
        if (t + 2 < n_trees) step(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}, t + 2, k3);
        if (t + 3 < n_trees) step(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, t + 3, k3 + 1u);
        if (t + 4 < n_trees) step(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{}, t + 4, k3 + 1u);
        if (t + 5 < n_trees) step(std::integral_constant<int, 2>{}, std::integral_constant<int, 5>{}, t + 5, k3 + 1u);
    }
    // trees n - (PD - 1) .. n - 1 are still pending (slots never written hold 0.0)
    for (int k = n_trees - (PD - 1); k < n_trees; ++k) {
        const int q = ((k % PD) + PD) % PD;
#pragma unroll
        for (int c = 0; c < R; ++c) {
            double v = pend[0][c];
#pragma unroll
            for (int z = 1; z < PD; ++z) if (q == z) v = pend[z][c];
            acc[c] = acc[c] + v;
        }
    }
#pragma unroll
    for (int c = 0; c < R; ++c) {
        if (live[c])
            emit(out, (int64_t)row[c] * g.ld_out + col[c], na[c] ? NAN : acc[c] / (double)n_trees, weight, accumulate);
    }
}

// SUBTREE form (round 5): the loader stages, per block of cells, only the SUBTREE the block's cells can reach, several trees to a buffer.
// Round 4's loader-wave kernel copies every tree whole into one of three buffers -- 12 MB per block of 3 840 cells, 196 GB per
// 1e8 cells past the L2 (PMC) -- and is bound by one load round trip per tree and buffer (0.75 us per tree with two in flight:
// the loader alone takes 38 of the kernel's 44 ms on 8 000 x 8 000 cells).  But a block's cells are neighbours: with the block's
// [min, max] rank of every predictor (the waves' ranges, reduced through LDS) the loader descends every tree from its root while
// the split falls the same way for the whole block -- lane = tree, the walkers' own prefix one level up -- and needs only the
// subtree below that node.  mhs_rf_load numbers nodes in PRE-ORDER, so that subtree is the contiguous record range
// [entry, entry + size): on the 8d planes 3.7 KB per tree instead of 24 (tools/r04_rf_slice_sim.py with 80 x 48-cell blocks),
// and a fifth of the trees need nothing at all.  The range goes, in 1 KB chunks, to the SAME offsets of the tree's buffer
// (tree t: buffer t % 3) it has in a whole copy -- so every record's child addresses stay valid and the chunks travel by
// LDS-DMA (global_load_lds, no registers, dozens in flight) -- and trees whose ranges do not overlap share a buffer: the
// loader keeps, per buffer and chunk, the last tree that used it (lane = chunk) and waits, before it overwrites a chunk,
// for every walker to have LEFT that tree (the walkers' progress words, one per wave, plain stores); the walkers wait for
// the loader's "trees staged" word -- both normally satisfied by the cached value.  A round of the loader: the next trees'
// chunks up to ~40 in flight, one s_waitcnt vmcnt(0), one store of the new count.  With whole trees (no prefix, rough
// rasters) this degenerates into round 4's schedule.
// Same nodes visited, same additions in the same order: identical planes (test_forest_walk_kernels_equal_each_other_...).
template <int LOG2R, bool K64, int STRIDE>
__global__ __launch_bounds__(1024) void rf_walk_sub_kernel(const uint2 *__restrict__ gnodes,
                                                           const double *__restrict__ glval,
                                                           const int *__restrict__ tree_off,
                                                           const int *__restrict__ depth,
                                                           const void *__restrict__ sorted,
                                                           const int *__restrict__ sorted_off, int n_trees,
                                                           int p, StackDev s, PredGeom g,
                                                           double weight, int accumulate,
                                                           double *__restrict__ out, const int *__restrict__ dmin, int tiles, int flags,
                                                           const int *__restrict__ axis_rank, int axis_ncol) {
    constexpr int R = rf_walks(LOG2R), WALKERS = 15;
    constexpr unsigned TREE_BYTES = 3u * STRIDE;
    static_assert(TREE_BYTES >= (unsigned)RF_COARSE_BYTES + 2048u && 2 * STRIDE <= 65535 && STRIDE % 1024 == 0, "buffer bases inside the 16-bit immediate offsets");
    constexpr unsigned CNT = TREE_BYTES, STG = CNT, PROG = CNT + 4u;   // 16 words: trees staged; the walker waves' progress
    constexpr unsigned RANGES = (unsigned)RF_COARSE_BYTES;           // 15 waves x 12 predictors x (min, max), behind the rank search's coarse table
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *coarse = (float *)smem;
    const unsigned stride = (unsigned)(p * R) | 1u;                // dwords of keys per lane
    const unsigned lane_base = CNT + 64u + threadIdx.x * stride * 4u;
    if ((unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap();
    const bool loader = threadIdx.x >= 64 * WALKERS;               // wave-uniform
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    const int cw = loader ? wave - WALKERS : wave;                 // the loader's lanes shadow wave 0 (cooperative rank search; they emit nothing)
    int row[R], col[R];
    bool na[R], live[R];
    if (tiles) {
        // wave tiles of 16 columns x 4 R rows (lane = column + 16 x row group, a lane's R walks on adjacent rows); a block's 15
        // tiles are 5 x 3 neighbours -- 80 x 12 R cells, the compact footprint that keeps the BLOCK's rank ranges narrow
        const int tiles_x = (g.nc + 15) / 16, bx_n = (tiles_x + 4) / 5;
        const int by = (int)(blockIdx.x / (unsigned)bx_n), bx = (int)(blockIdx.x - (unsigned)by * (unsigned)bx_n);
        const int tx = bx * 5 + cw % 5;
        const int64_t ty = (int64_t)by * 3 + cw / 5;
#pragma unroll
        for (int c = 0; c < R; ++c) {
            const int64_t r = ty * (4 * R) + (lane >> 4) * R + c;
            const int cc = tx * 16 + (lane & 15);
            live[c] = r < g.nr && cc < g.nc;
            row[c] = (int)min(r, (int64_t)g.nr - 1); col[c] = min(cc, g.nc - 1);
        }
    } else {
        const int64_t i0 = (int64_t)blockIdx.x * (64 * WALKERS) + (loader ? threadIdx.x - 64 * WALKERS : threadIdx.x);
        rf_lane_cells<R>(g, i0, 0, row, col, live);
    }
#pragma unroll
    for (int c = 0; c < R; ++c) na[c] = false;
    for (int j = 0; j < p; ++j) {
        float r[R];
        if (axis_rank && j >= s.C && !s.all_from_planes) {          // LONG / LAT: the rank is a function of the column / the row (publish_axis_ranks)
#pragma unroll
            for (int c = 0; c < R; ++c)
                r[c] = (float)(j == s.C ? axis_rank[g.c0 + col[c]] : axis_rank[(int64_t)axis_ncol + g.r0 + row[c]]);
        } else if constexpr (K64) lut_ranks_t<R, 1024, double>(j, (const double *)sorted, sorted_off, (double *)coarse, s, g, row, col, na, r);
        else lut_ranks_t<R, 1024, float>(j, (const float *)sorted, sorted_off, coarse, s, g, row, col, na, r);
#pragma unroll
        for (int c = 0; c < R; ++c) *(unsigned *)(smem + lane_base + (unsigned)(j * R + c) * 4u) = (unsigned)r[c] << 8;
    }
    const bool prefix = (flags & RF_LD_PREFIX) && p <= RF_PREFIX_MAX_P && n_trees <= 64 * RF_ENTRY_BATCHES;
    // the wave's [min, max] rank of every predictor over its cells that are not NA (wave-uniform: scalar registers)
    int mn[RF_PREFIX_MAX_P], mx[RF_PREFIX_MAX_P];
#pragma unroll
    for (int v = 0; v < RF_PREFIX_MAX_P; ++v) {
        mn[v] = 0x7fffffff; mx[v] = -1;
        if (prefix && v < p) {
            int a = 0x7fffffff, b = -1;
#pragma unroll
            for (int c = 0; c < R; ++c)
                if (!na[c]) {
                    const int rk = (int)(*(const unsigned *)(smem + lane_base + (unsigned)(v * R + c) * 4u) >> 8);
                    a = min(a, rk); b = max(b, rk);
                }
#pragma unroll
            for (int q = 32; q > 0; q >>= 1) { a = min(a, __shfl_xor(a, q)); b = max(b, __shfl_xor(b, q)); }
            mn[v] = __builtin_amdgcn_readfirstlane(a); mx[v] = __builtin_amdgcn_readfirstlane(b);
            // an NA cell's walk is thrown away, but it must stay INSIDE the staged subtree: it takes the wave's smallest rank
#pragma unroll
            for (int c = 0; c < R; ++c)
                if (na[c] && mx[v] >= 0) *(unsigned *)(smem + lane_base + (unsigned)(v * R + c) * 4u) = (unsigned)mn[v] << 8;
        }
    }
    __syncthreads();                                               // the coarse table is no longer needed: the ranges go behind it
    if (!loader && lane < 2 * RF_PREFIX_MAX_P) {
        int val = 0;
#pragma unroll
        for (int v = 0; v < RF_PREFIX_MAX_P; ++v) { if (lane == 2 * v) val = mn[v]; if (lane == 2 * v + 1) val = mx[v]; }
        *(int *)(smem + RANGES + (unsigned)(wave * 2 * RF_PREFIX_MAX_P + lane) * 4u) = val;
    }
    if (threadIdx.x < 16) *(unsigned *)(smem + CNT + threadIdx.x * 4u) = 0u;
    __syncthreads();
    // lane = tree descent from node `nd` while the split falls the same way for every rank in [lo, hi] of its predictor; `end` follows
    // (pre-order: the left subtree of a node k is [k + 1, right), the right one [right, end))
    auto descend = [&](const int (&lo_)[RF_PREFIX_MAX_P], const int (&hi_)[RF_PREFIX_MAX_P], int t, unsigned &nd, unsigned &end, unsigned &plen,
                       unsigned &term) {
        bool walking = t < n_trees;
        const int o = tree_off[min(t, n_trees - 1)];
        while (__builtin_amdgcn_ballot_w64(walking)) {
            if (walking) {
                const uint2 rec = gnodes[o + (int)nd];
                if (rec.x == RF_LEAF_WORD) { term = 1u; walking = false; }
                else {
                    const int j = (int)(rec.x >> 8), v = (int)(rec.x & 0xFFu) / (4 * R);
                    int lo = lo_[0], hi = hi_[0];
#pragma unroll
                    for (int q = 1; q < RF_PREFIX_MAX_P; ++q) if (q < p && v == q) { lo = lo_[q]; hi = hi_[q]; }
                    if (lo > j) { nd = (rec.y >> 16) >> 3; ++plen; }                                   // every rank > j: right
                    else if (hi <= j) { end = (rec.y >> 16) >> 3; nd = (rec.y & 0xFFFFu) >> 3; ++plen; }  // every rank <= j: left
                    else walking = false;
                }
            }
        }
    };
    unsigned entry[RF_ENTRY_BATCHES];                              // walkers: the wave's entry nodes; loader: the block's chunk ranges
#pragma unroll
    for (int b = 0; b < RF_ENTRY_BATCHES; ++b) entry[b] = 0u;
    if (loader) {
        // ---- the block's ranges and subtrees: entry[b], lane l = tree 64 b + l: chunks | first chunk << 8 ---------------
        int bmn[RF_PREFIX_MAX_P], bmx[RF_PREFIX_MAX_P];
#pragma unroll
        for (int v = 0; v < RF_PREFIX_MAX_P; ++v) {
            int a = 0x7fffffff, b = -1;
            if (prefix && v < p && lane < WALKERS) {
                a = *(const int *)(smem + RANGES + (unsigned)(lane * 2 * RF_PREFIX_MAX_P + 2 * v) * 4u);
                b = *(const int *)(smem + RANGES + (unsigned)(lane * 2 * RF_PREFIX_MAX_P + 2 * v + 1) * 4u);
            }
#pragma unroll
            for (int q = 32; q > 0; q >>= 1) { a = min(a, __shfl_xor(a, q)); b = max(b, __shfl_xor(b, q)); }
            bmn[v] = __builtin_amdgcn_readfirstlane(a); bmx[v] = __builtin_amdgcn_readfirstlane(b);
        }
#pragma unroll
        for (int b = 0; b < RF_ENTRY_BATCHES; ++b) {
            if (b * 64 < n_trees) {
                const int t = b * 64 + lane;
                const int tl = min(t, n_trees - 1);
                unsigned nd = 0u, end = (unsigned)(tree_off[tl + 1] - tree_off[tl]), plen = 0u, term = 0u;
                if (prefix) descend(bmn, bmx, t, nd, end, plen, term);
                const unsigned q0 = (nd * 8u) >> 10;
                const unsigned nq = (term || t >= n_trees) ? 0u : ((end * 8u + 1023u) >> 10) - q0;
                entry[b] = nq | (q0 << 8);
            }
        }
    } else if (prefix) {
        // ---- the wave's own prefix: where its cells first part ways ------------------------------------------------------
#pragma unroll
        for (int b = 0; b < RF_ENTRY_BATCHES; ++b) {
            if (b * 64 < n_trees) {
                unsigned nd = 0u, end = 0u, plen = 0u, term = 0u;
                descend(mn, mx, b * 64 + lane, nd, end, plen, term);
                entry[b] = nd | (plen << 16) | (term << 31);
            }
        }
    }
    __syncthreads();                                               // the ranges have been read: the buffers are free
    if (loader) {
        // ---- staging ----------------------------------------------------------------------------------------------------
        const unsigned lane16 = (unsigned)lane * 16u;
        unsigned last0 = 0u, last1 = 0u, last2 = 0u;               // lane c: 1 + the last tree that used chunk c of buffer 0 / 1 / 2
        unsigned prog = 0u;                                        // trees every walker has left (cached)
        unsigned cur = 0u;                                         // the current 64 trees' chunk ranges, lane = tree
        int offv = 0;                                              // ... and first records
        int inflight = 0;
        auto poll = [&]() {                                        // the slowest walker's progress
            unsigned v = 0xffffffffu;
            if (lane < WALKERS) v = *(volatile const unsigned *)(smem + PROG + (unsigned)lane * 4u);
#pragma unroll
            for (int q = 32; q > 0; q >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, q));
            prog = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
        };
        auto publish = [&](int done) {                             // every chunk issued so far has landed: trees 0 .. done - 1 are whole
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) *(volatile unsigned *)(smem + STG) = (unsigned)done;
            inflight = 0;
        };
        auto stage = [&](auto slot_tag, const int u) {
            constexpr int SLOT = decltype(slot_tag)::value;
            if ((u & 63) == 0) {
#pragma unroll
                for (int b = 0; b < RF_ENTRY_BATCHES; ++b) if ((u >> 6) == b) cur = entry[b];
                offv = tree_off[min(u + lane, n_trees - 1)];
                asm volatile("" : "+v"(cur), "+v"(offv));
            }
            const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)cur, u & 63);
            const int nq = (int)(w & 0xFFu), q0 = (int)(w >> 8);
            if (nq == 0) return;
            unsigned &last = SLOT == 0 ? last0 : SLOT == 1 ? last1 : last2;
            const bool mine = lane >= q0 && lane < q0 + nq;
            unsigned need = mine ? last : 0u;
#pragma unroll
            for (int q = 32; q > 0; q >>= 1) need = max(need, (unsigned)__shfl_xor((int)need, q));
            need = (unsigned)__builtin_amdgcn_readfirstlane((int)need);
            if (prog < need) {
                poll();
                if (prog < need) {
                    if (inflight) publish(u);                      // the walkers may be waiting for what is in flight
                    while (prog < need) { __builtin_amdgcn_s_sleep(2); poll(); }
                }
            }
            const char *src = (const char *)(gnodes + __builtin_amdgcn_readlane(offv, u & 63)) + ((size_t)q0 << 10) + lane16;
            unsigned dst = (unsigned)(SLOT * STRIDE) + ((unsigned)q0 << 10);
            for (int i = 0; i < nq; ++i) { glds16(src, dst); src += 1024; dst += 1024u; }
            if (mine) last = (unsigned)u + 1u;
            inflight += nq;
            if (inflight >= 40) publish(u + 1);
        };
        for (int u = 0; u < n_trees; u += 3) {
            stage(std::integral_constant<int, 0>{}, u);
            if (u + 1 < n_trees) stage(std::integral_constant<int, 1>{}, u + 1);
            if (u + 2 < n_trees) stage(std::integral_constant<int, 2>{}, u + 2);
            // a short forest tail / trees without chunks: keep the count moving (cheap when nothing is in flight)
            if (inflight >= 16 || (u % 24) == 21) publish(min(u + 3, n_trees));
        }
        publish(n_trees);
        return;
    }
    // ---- the walkers ------------------------------------------------------------------------------------------------
    constexpr int PD = 6;                                          // a tree's predictions are added PD - 1 trees after their request
    double acc[R], pend[PD][R];                                    // pend[t % PD]: predictions of tree t
#pragma unroll
    for (int c = 0; c < R; ++c) {
        acc[c] = 0.0;
#pragma unroll
        for (int k = 0; k < PD; ++k) pend[k][c] = 0.0;
    }
    unsigned staged = 0u;                                          // cached: trees 0 .. staged - 1 are in their buffers
    int ocur = 0;                                                  // lane l: first record of tree 64 b + l
    unsigned wcur = 0u;                                            // lane l: its walk -- entry node's byte address | c0 << 19 | cnt << 25 | walks << 31
    double tpcur = 0.0;                                            // lane l: the prediction of its tree's entry node where the wave does not walk the tree
    auto step = [&](auto slot_tag, auto pidx_tag, const int t) {
        constexpr int SLOT = decltype(slot_tag)::value, PIDX = decltype(pidx_tag)::value;
        if ((t & 63) == 0) {
            // the next 64 trees' scalars, lane = tree (one vector load each), and the walk's loop counts formed on the vector unit
            unsigned ecur = 0u;
#pragma unroll
            for (int b = 0; b < RF_ENTRY_BATCHES; ++b) if ((t >> 6) == b) ecur = entry[b];
            const int tl = min(t + lane, n_trees - 1);
            ocur = tree_off[tl];
            const int dcur = depth[tl], mcur = dmin ? dmin[tl] : dcur;
            const int plen = (int)((ecur >> 16) & 0x7FFFu);
            const int levels = (ecur >> 31) ? 0 : dcur - plen, shallow = max(mcur - plen, 0);
            const int c0 = min(shallow, levels - 1), cnt = levels - 1 - c0;              // levels <= 63
            wcur = ((ecur & 0xFFFFu) << 3) | (levels > 0 ? ((unsigned)c0 << 19) | ((unsigned)cnt << 25) | 0x80000000u : 0u);
            // a tree the wave does not walk (its cells share the entry node) has ONE prediction for all of them: fetched here
            tpcur = levels > 0 ? 0.0 : glval[ocur + (int)(ecur & 0xFFFFu)];
            asm volatile("" : "+v"(ocur), "+v"(wcur), "+v"(tpcur));
        }
        const unsigned wk = (unsigned)__builtin_amdgcn_readlane((int)wcur, t & 63);
        const int o = __builtin_amdgcn_readlane(ocur, t & 63);
        unsigned node[R];
#pragma unroll
        for (int c = 0; c < R; ++c) node[c] = wk & 0x7FFF8u;
        if (wk >> 31) {
            if (staged <= (unsigned)t)                              // tree t is in its buffer?
                for (;;) {
                    unsigned v;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(STG) : "memory");
                    staged = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
                    if (staged > (unsigned)t) break;
                    __builtin_amdgcn_s_sleep(1);
                }
            int c0 = (int)((wk >> 19) & 63u), cnt = (int)((wk >> 25) & 63u);
            if constexpr (R == 4) {
                asm volatile(
#include "rf_walk_loop4xo.inc"
                    : [n0] "+v"(node[0]), [n1] "+v"(node[1]), [n2] "+v"(node[2]), [n3] "+v"(node[3]), [cnt] "+s"(cnt), [c0] "+s"(c0)
                    : [lb] "v"(lane_base), [off] "n"(SLOT * STRIDE)
                    : "memory", "vcc", "scc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113",
                      "v115", "v116", "v117", "v118", "v120");
            } else {
                static_assert(R == 4 || R == 5, "hand loops exist for four and five walks");
                asm volatile(
#include "rf_walk_loop5xo.inc"
                    : [n0] "+v"(node[0]), [n1] "+v"(node[1]), [n2] "+v"(node[2]), [n3] "+v"(node[3]), [n4] "+v"(node[R - 1]), [cnt] "+s"(cnt),
                      [c0] "+s"(c0)
                    : [lb] "v"(lane_base), [off] "n"(SLOT * STRIDE)
                    : "memory", "vcc", "scc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
                      "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120");
            }
        }
        // this wave has left tree t (or never entered it): its progress word, lane 0 (every lane is active here)
        asm volatile("s_mov_b64 exec, 1\n\tds_write_b32 %0, %1\n\ts_mov_b64 exec, -1" :: "v"(PROG + 4u * (unsigned)wave), "v"((unsigned)t + 1u) : "memory");
        const char *lv = (const char *)(glval + o);                      // node[] are byte addresses of 8-byte records = of the doubles
#pragma unroll
        for (int c = 0; c < R; ++c) acc[c] = acc[c] + pend[(PIDX + 1) % PD][c];      // tree t - (PD - 1)'s; still in tree order
        if (wk >> 31) {
#pragma unroll
            for (int c = 0; c < R; ++c) pend[PIDX][c] = *(const double *)(lv + node[c]);
        } else {
            const double tv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tpcur), t & 63), __builtin_amdgcn_readlane(__double2loint(tpcur), t & 63));
#pragma unroll
            for (int c = 0; c < R; ++c) pend[PIDX][c] = tv;
        }
    };
    static_assert(PD == 6, "the tree loop is unrolled by the least common multiple of the 3 buffers and PD");
    for (int t = 0; t < n_trees; t += 6) {
        step(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, t);
        if (t + 1 < n_trees) step(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < n_trees) step(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}, t + 2);
        if (t + 3 < n_trees) step(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, t + 3);
        if (t + 4 < n_trees) step(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{}, t + 4);
        if (t + 5 < n_trees) step(std::integral_constant<int, 2>{}, std::integral_constant<int, 5>{}, t + 5);
    }
    // trees n - (PD - 1) .. n - 1 are still pending (slots never written hold 0.0)
    for (int k = n_trees - (PD - 1); k < n_trees; ++k) {
        const int q = ((k % PD) + PD) % PD;
#pragma unroll
        for (int c = 0; c < R; ++c) {
            double v = pend[0][c];
#pragma unroll
            for (int z = 1; z < PD; ++z) if (q == z) v = pend[z][c];
            acc[c] = acc[c] + v;
        }
    }
#pragma unroll
    for (int c = 0; c < R; ++c) {
        if (live[c])
            emit(out, (int64_t)row[c] * g.ld_out + col[c], na[c] ? NAN : acc[c] / (double)n_trees, weight, accumulate);
    }
}

// rf_prefix_entries for the COMPACT records (split nodes only; a state is a split record's byte address or D + the number of a
// terminal node among the tree's terminals, D = 8 x the tree's splits): entry state in the low 16 bits, levels descended above them.
template <int R>
__device__ __forceinline__ void rf_prefix_entries_compact(unsigned (&entry)[RF_ENTRY_BATCHES], const uint2 *__restrict__ gnodes,
                                                          const int *__restrict__ coff, int n_trees, int p, const char *smem,
                                                          unsigned lane_base, const bool (&na)[R]) {
    const int lane = threadIdx.x & 63;
    int mn[12], mx[12];
#pragma unroll
    for (int v = 0; v < 12; ++v) {
        mn[v] = 0x7fffffff; mx[v] = -1;
        if (v < p) {
            int a = 0x7fffffff, b = -1;
#pragma unroll
            for (int c = 0; c < R; ++c)
                if (!na[c]) {
                    const int rk = (int)(*(const unsigned *)(smem + lane_base + (unsigned)(v * R + c) * 4u) >> 8);
                    a = min(a, rk); b = max(b, rk);
                }
#pragma unroll
            for (int q = 32; q > 0; q >>= 1) { a = min(a, __shfl_xor(a, q)); b = max(b, __shfl_xor(b, q)); }
            mn[v] = __builtin_amdgcn_readfirstlane(a); mx[v] = __builtin_amdgcn_readfirstlane(b);
        }
    }
#pragma unroll
    for (int b = 0; b < RF_ENTRY_BATCHES; ++b) {
        if (b * 64 < n_trees) {
            const int t = min(b * 64 + lane, n_trees - 1);
            const int cb = coff[t];
            const unsigned D = (unsigned)(coff[t + 1] - cb - 1) * 8u;
            unsigned state = 0u, plen = 0u;
            bool walking = b * 64 + lane < n_trees && state < D;
            while (__builtin_amdgcn_ballot_w64(walking)) {
                if (walking) {
                    const uint2 rec = gnodes[cb + (int)(state >> 3)];
                    const int j = (int)(rec.x >> 8), v = (int)(rec.x & 0xFFu) / (4 * R);
                    int lo = mn[0], hi = mx[0];
#pragma unroll
                    for (int q = 1; q < 12; ++q) if (q < p && v == q) { lo = mn[q]; hi = mx[q]; }
                    if (lo > j) { state = rec.y >> 16; ++plen; }
                    else if (hi <= j) { state = rec.y & 0xFFFFu; ++plen; }
                    else walking = false;
                    if (state >= D) walking = false;
                }
            }
            entry[b] = state | (plen << 16);
        }
    }
}

// COMPACT form for trees beyond the double-buffered kernel's 4 095 nodes (a 20 000-station forest: ~12 000 nodes per
// tree).  Half of a tree's nodes are terminals, which the walk never needs to READ -- it only has to remember which one
// it reached.  LDS holds the records of the split nodes alone plus one all-zero record at byte address D = 8 * splits
// (build_rf_nodes_t, RF_COMPACT): ~49 KB instead of 97 KB for such a tree, which leaves room for the keys of
// 4 cells x 960 lanes (15 waves; the BIG form fits 2 x 1024 or 4 x 512 -- half the walks in flight, and this loop
// is bound by the latency of its two dependent LDS reads).  A lane's state is the byte address of a split node's
// record or, once it has reached a terminal, D + that node's number among the tree's terminals:
//       rec = LDS[min(state, D)];  child = key > rec.rank ? rec.right : rec.left;  state = max(child, state)
// (children follow their parent in randomForest's numbering and every terminal code is >= D, so max() leaves a split
// node's state to its child and a terminal's state alone: the all-zero record's children are 0).  The next tree's
// records travel global -> registers during the walk (PF x 8 bytes per thread) and registers -> LDS between two
// barriers after it, so staging costs the block one LDS write pass per tree instead of a round trip to L2.
template <int PF, bool K64>
__global__ __launch_bounds__(1024) void rf_walk_compact_kernel(const uint2 *__restrict__ gnodes, const double *__restrict__ glval,
                                                               const int *__restrict__ tree_off, const int *__restrict__ coff,
                                                               const int *__restrict__ depth, const void *__restrict__ sorted,
                                                               const int *__restrict__ sorted_off, int n_trees, int cmax, int p,
                                                               StackDev s, PredGeom g, double weight, int accumulate,
                                                               double *__restrict__ out, const int *__restrict__ dmin, int strips, int prefix) {
    constexpr int R = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned nt = blockDim.x;
    const unsigned tree_bytes = max((unsigned)cmax * 8u, (unsigned)RF_COARSE_BYTES);
    float *coarse = (float *)smem;
    const unsigned stride = (unsigned)(p * R) | 1u;
    const unsigned lane_base = tree_bytes + threadIdx.x * stride * 4u;
    if ((unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap();
    const int64_t i0 = (int64_t)blockIdx.x * nt + threadIdx.x;
    int row[R], col[R];
    bool na[R], live[R];
    double acc[R], pending[R];
    unsigned node[R];
    rf_lane_cells<R>(g, i0, strips, row, col, live);
#pragma unroll
    for (int c = 0; c < R; ++c) { na[c] = false; acc[c] = 0.0; pending[c] = 0.0; }
    for (int j = 0; j < p; ++j) {
        float r[R];
        if constexpr (K64) lut_ranks_t<R, 0, double>(j, (const double *)sorted, sorted_off, (double *)coarse, s, g, row, col, na, r);
        else lut_ranks_t<R, 0, float>(j, (const float *)sorted, sorted_off, coarse, s, g, row, col, na, r);
#pragma unroll
        for (int c = 0; c < R; ++c) *(unsigned *)(smem + lane_base + (unsigned)(j * R + c) * 4u) = (unsigned)r[c] << 8;
    }
    unsigned entry[RF_ENTRY_BATCHES];                              // where each tree's walks start for this wave
#pragma unroll
    for (int b = 0; b < RF_ENTRY_BATCHES; ++b) entry[b] = 0u;
    if (prefix && p <= RF_PREFIX_MAX_P && n_trees <= 64 * RF_ENTRY_BATCHES) rf_prefix_entries_compact<R>(entry, gnodes, coff, n_trees, p, smem, lane_base, na);
    unsigned ecur = 0u;
    __syncthreads();                                               // coarse table no longer needed
    {
        const int o = coff[0], cnt = coff[1] - o;
        for (int e = threadIdx.x; e < cnt; e += (int)nt) ((uint2 *)smem)[e] = gnodes[o + e];
    }
    __syncthreads();
    // scalars of the trees ahead are fetched early, as in rf_walk_db_kernel
    // o: the tree's first terminal in glval (the terminals' predictions, tree after tree: tree_off[t] - (coff[t] - t))
    int c0 = coff[0], c1 = coff[1], c2 = n_trees > 1 ? coff[2] : c1;
    int o = tree_off[0] - c0, o1 = n_trees > 1 ? tree_off[1] - c1 + 1 : 0;
    int levels = depth[0], levels1 = n_trees > 1 ? depth[1] : 0;
    int shallow = dmin ? dmin[0] : levels, shallow1 = n_trees > 1 ? (dmin ? dmin[1] : levels1) : 0;
    for (int t = 0; t < n_trees; ++t) {
        const int cnt1 = t + 1 < n_trees ? c2 - c1 : 0;
        const int c3 = t + 3 <= n_trees ? coff[t + 3] : c2;
        const int o2 = t + 2 < n_trees ? tree_off[t + 2] - c2 + (t + 2) : 0;
        const int levels2 = t + 2 < n_trees ? depth[t + 2] : 0;
        const int shallow2 = t + 2 < n_trees ? (dmin ? dmin[t + 2] : levels2) : 0;
        const unsigned D = (unsigned)(c1 - c0 - 1) * 8u;
        uint2 pn[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int e = (int)threadIdx.x + q * (int)nt;
            if (e < cnt1) pn[q] = gnodes[c1 + e];
        }
        if ((t & 63) == 0) {
            ecur = 0u;
#pragma unroll
            for (int b = 0; b < RF_ENTRY_BATCHES; ++b) if ((t >> 6) == b) ecur = entry[b];
        }
        const unsigned ent = (unsigned)__builtin_amdgcn_readlane((int)ecur, t & 63);
        const int plen = (int)(ent >> 16);
        const int lev = (ent & 0xFFFFu) >= D ? 0 : levels - plen, shal = min(max(shallow - plen, 0), lev);
#pragma unroll
        for (int c = 0; c < R; ++c) node[c] = ent & 0xFFFFu;
        auto level = [&]() {
            // (one walk after the other, as the compiler lays it out: issuing the four walks' reads side by side -- as the batched
            // walk of rf_walk_cbs_kernel does -- measured 6 % SLOWER here, where 15 waves walk ~20 levels a tree and the LDS pipe is the bound)
#pragma unroll
            for (int c = 0; c < R; ++c) {
                const uint2v nd = lds_u2(min(node[c], D));
                const unsigned k = lds_u32(lane_base + (nd.x & 0xFFu) + c * 4);
                unsigned child;
                asm("v_cmp_gt_u32_e32 vcc, %1, %2\n\t"
                    "s_nop 1\n\t"
                    "v_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"
                    : "=v"(child) : "v"(k), "v"(nd.x), "v"(nd.y) : "vcc");
                node[c] = max(child, node[c]);
            }
        };
        // no state is terminal above the tree's shallowest leaf; from there on the wave leaves the tree as soon as every
        // state of every lane is a terminal code (>= D)
        for (int l = 0; l < shal; ++l) level();
        for (int l = shal; l < lev; ++l) {
            const unsigned lowest = min(min(node[0], node[1]), min(node[2], node[3]));
            if (!__builtin_amdgcn_ballot_w64(lowest < D)) break;
            level();
        }
#pragma unroll
        for (int c = 0; c < R; ++c) {
            acc[c] = acc[c] + pending[c];
            pending[c] = glval[o + (int)(node[c] - D)];
        }
        __syncthreads();                                           // every wave has left this tree
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int e = (int)threadIdx.x + q * (int)nt;
            if (e < cnt1) *(uint2 *)(smem + (unsigned)e * 8u) = pn[q];
        }
        __syncthreads();
        o = o1; o1 = o2;
        c0 = c1; c1 = c2; c2 = c3;
        levels = levels1; levels1 = levels2;
        shallow = shallow1; shallow1 = shallow2;
    }
#pragma unroll
    for (int c = 0; c < R; ++c) {
        acc[c] = acc[c] + pending[c];
        if (live[c])
            emit(out, (int64_t)row[c] * g.ld_out + col[c], na[c] ? NAN : acc[c] / (double)n_trees, weight, accumulate);
    }
}


// BLOCK-SUBTREE form of the compact walk (round 6): the trees of a 20 000-station forest (~6 000 split records, 48 KB each)
// no longer travel whole.  rf_walk_compact_kernel copies every tree into LDS for every block of 3 840 cells -- two barriers a
// tree, every wave waiting for the block's deepest walk, and a cost that is linear in the trees whether the forest fits the L2
// or not (tools/r06_forest_big.py: the CU's own global -> LDS path is the limit, so the bytes staged are what to cut) -- yet a
// block's cells are neighbours and reach a sliver of each tree.  Here a block is 80 x 48 cells (TW x TH wave tiles of 16 x 16, a
// lane's 4 walks on adjacent rows) and, after the rank keys are parked:
//   1. BLOCK PREFIX, lane = tree: waves 0..7 descend 64 trees each from the root (records from global memory, the ranges from a
//      table in LDS) for as long as the split falls the same way for the block's whole [min, max] rank of its predictor.  Where
//      that ends is the block's entry into the tree: a terminal (its prediction is all the block needs) or a split record a,
//      below which lie -- nodes are numbered in pre-order -- the n = csub[a] consecutive records [a, a + n) and the n + 1
//      consecutive terminals from csub's second word on (the compact records number terminals in node order; rf_clval holds
//      their predictions in that order).
//   2. WAVE PREFIX: every wave goes on from the block's entries with its own, narrower ranges -- its walks start there.
//   3. The trees are taken in order, as many per BATCH as their ranges fit the LDS region beside the keys (64-aligned groups,
//      lane = tree: one prefix sum places them, another deals the batch's 16-byte units out, 256 to a wave).  A batch is
//      requested (registers) before the batch ahead of it is walked and parked in LDS between two barriers after; every wave then
//      walks the batch's trees by itself, in tree order -- no barrier, no waiting for another wave's deepest leaf.  A (wave, tree)
//      pair entered at a terminal -- three in four on cfg5's planes -- costs two readlanes: the predictions of such pairs are
//      read lane = tree, once per batch.
// A staged subtree keeps its records as they are: child fields are byte addresses relative to the TREE, so a walk reads
//       rec = LDS[min(state, Dz) + delta],   Dz = a + 8 n (the zero record),  delta = 8 * slot - a
// and otherwise steps as the compact kernel does (state = max(child, state); terminal codes are >= D >= Dz); a terminal code's
// prediction is LDS[vbase + 8 code].  A subtree too large to sit in LDS with its predictions (a whole tree) leaves them in
// global memory; a block whose subtrees average more than rough_slots LDS slots (noisy rasters; MHS_RF_PLAIN) takes the
// compact kernel's whole-tree loop instead, inside this kernel.  Same leaf for every cell, predictions added in tree order: the
// planes are those of every other forest kernel, bit for bit (test_forest_walk_kernels_equal_each_other_and_the_node_walk,
// test_forest_in_the_compact_form_equals_the_node_walk).  cfg5's step: 1 137 -> 455 ms for 4e8 cells and 500 trees.
constexpr int RF_CBS_BATCHES = 8;                                   // x 64 trees, held lane = tree in registers
#ifdef RF_CBS_STATS
__device__ unsigned long long g_cbs_stats[16];
#define CBS_ADD(i, v) atomicAdd(&g_cbs_stats[i], (unsigned long long)(v))
#define CBS_CLOCK() __builtin_readcyclecounter()
#else
#define CBS_ADD(i, v) ((void)0)
#define CBS_CLOCK() 0ull
#endif
template <bool K64>
__global__ __launch_bounds__(1024) void rf_walk_cbs_kernel(const uint2 *__restrict__ gnodes, const double *__restrict__ glval,
                                                           const int *__restrict__ tree_off, const int *__restrict__ coff,
                                                           const int *__restrict__ csub, const int *__restrict__ depth,
                                                           const void *__restrict__ sorted, const int *__restrict__ sorted_off,
                                                           int n_trees, unsigned region_bytes, int p, StackDev s, PredGeom g,
                                                           double weight, int accumulate, double *__restrict__ out,
                                                           const int *__restrict__ dmin, const int *__restrict__ axis_rank,
                                                           int axis_ncol, int tw, int th, int prefix, unsigned rough_slots) {
    constexpr int R = 4, NB = RF_CBS_BATCHES, P = RF_PREFIX_MAX_P, PF = 8;
    constexpr unsigned RANGES = 0u, TAB = 2048u;                    // 16 waves' + the block's 12 x (min, max); then {entry, records, first terminal} per tree
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *coarse = (float *)smem;
    const int nw = (int)(blockDim.x >> 6), wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    const unsigned stride = (unsigned)(p * R) | 1u;
    const unsigned lane_base = region_bytes + threadIdx.x * stride * 4u;
    if ((unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap();
    int row[R], col[R];
    bool na[R], live[R];
    [[maybe_unused]] const unsigned long long st_t0 = CBS_CLOCK();
    [[maybe_unused]] unsigned long long st_stage = 0, st_walk = 0, st_levels = 0, st_term = 0;
    {
        const int tiles_x = (g.nc + 15) / 16, bx_n = (tiles_x + tw - 1) / tw;
        const int by = (int)(blockIdx.x / (unsigned)bx_n), bx = (int)(blockIdx.x - (unsigned)by * (unsigned)bx_n);
        const int tx = bx * tw + wave % tw;
        const int64_t ty = (int64_t)by * th + wave / tw;
#pragma unroll
        for (int c = 0; c < R; ++c) {
            const int64_t r = ty * (4 * R) + (lane >> 4) * R + c;
            const int cc = tx * 16 + (lane & 15);
            live[c] = r < g.nr && cc < g.nc;
            row[c] = (int)min(r, (int64_t)g.nr - 1); col[c] = min(cc, g.nc - 1);
            na[c] = false;
        }
    }
    for (int j = 0; j < p; ++j) {
        float r[R];
        if (axis_rank && j >= s.C && !s.all_from_planes) {          // LONG / LAT: a function of the column / the row (publish_axis_ranks)
#pragma unroll
            for (int c = 0; c < R; ++c)
                r[c] = (float)(j == s.C ? axis_rank[g.c0 + col[c]] : axis_rank[(int64_t)axis_ncol + g.r0 + row[c]]);
        } else if constexpr (K64) lut_ranks_t<R, 0, double>(j, (const double *)sorted, sorted_off, (double *)coarse, s, g, row, col, na, r);
        else lut_ranks_t<R, 0, float>(j, (const float *)sorted, sorted_off, coarse, s, g, row, col, na, r);
#pragma unroll
        for (int c = 0; c < R; ++c) *(unsigned *)(smem + lane_base + (unsigned)(j * R + c) * 4u) = (unsigned)r[c] << 8;
    }
    __syncthreads();                                               // the coarse table is no longer needed
    // the wave's [min, max] rank of every predictor over its cells that are not NA -> LDS (lane 2v: min, 2v + 1: max)
    {
        int mine = lane & 1 ? -1 : 0x7fffffff;
#pragma unroll
        for (int v = 0; v < P; ++v) {
            if (v < p) {
                int a = 0x7fffffff, b = -1;
#pragma unroll
                for (int c = 0; c < R; ++c)
                    if (!na[c]) {
                        const int rk = (int)(*(const unsigned *)(smem + lane_base + (unsigned)(v * R + c) * 4u) >> 8);
                        a = min(a, rk); b = max(b, rk);
                    }
#pragma unroll
                for (int q = 32; q > 0; q >>= 1) { a = min(a, __shfl_xor(a, q)); b = max(b, __shfl_xor(b, q)); }
                // an NA cell's walk is thrown away, but it must stay inside the staged subtree: it takes the wave's smallest rank
#pragma unroll
                for (int c = 0; c < R; ++c)
                    if (na[c] && b >= 0) *(unsigned *)(smem + lane_base + (unsigned)(v * R + c) * 4u) = (unsigned)a << 8;
                if (lane == 2 * v) mine = a;
                if (lane == 2 * v + 1) mine = b;
            }
        }
        if (lane < 2 * P) *(int *)(smem + RANGES + (unsigned)(wave * 2 * P + lane) * 4u) = mine;
    }
    __syncthreads();
    // the block's ranges: the waves' merged (behind theirs in the table)
    constexpr unsigned BRANGES = RANGES + 16u * 2u * P * 4u;
    if (threadIdx.x < 2 * P) {
        int val = lane & 1 ? -1 : 0x7fffffff;
        for (int w = 0; w < nw; ++w) {
            const int x = *(const int *)(smem + RANGES + (unsigned)(w * 2 * P + lane) * 4u);
            val = lane & 1 ? max(val, x) : min(val, x);
        }
        *(int *)(smem + BRANGES + (unsigned)lane * 4u) = val;
    }
    __syncthreads();
    // lane = tree: descend from `state` while the split falls the same way for every rank in [lo, hi] of its predictor -- the
    // ranges table at LDS byte address rbase (one ds_read_b64 a level: kept in registers, the compiler indexes them through scratch)
    auto descend = [&](unsigned rbase, bool walking, int cb, unsigned D, unsigned &state, unsigned &plen) {
        walking = walking && state < D;
        while (__builtin_amdgcn_ballot_w64(walking)) {
            if (walking) {
                const uint2 rec = gnodes[cb + (int)(state >> 3)];
                const int j = (int)(rec.x >> 8);
                const unsigned v = (rec.x & 0xFFu) / (4 * R);
                const int2 r = *(const int2 *)(smem + rbase + v * 8u);
                if (r.x > j) { state = rec.y >> 16; ++plen; }
                else if (r.y <= j) { state = rec.y & 0xFFFFu; ++plen; }
                else walking = false;
                if (state >= D) walking = false;
            }
        }
    };
    const int nbatch = (n_trees + 63) >> 6;
    [[maybe_unused]] const unsigned long long st_t1 = CBS_CLOCK();
    {   // 1. the block's entries
        for (int b = wave; b < nbatch; b += nw) {
            const int t = b * 64 + lane, tl = min(t, n_trees - 1);
            const int cb = coff[tl];
            const unsigned D = (unsigned)(coff[tl + 1] - cb - 1) * 8u;
            unsigned state = 0u, plen = 0u;
            if (prefix) descend(BRANGES, t < n_trees, cb, D, state, plen);
            // the subtree below the entry: n split records from `state` on and n + 1 terminals from lf on; a terminal entry: itself
            const int2 sub = (t < n_trees && state < D) ? *(const int2 *)(csub + 2 * (cb + (int)(state >> 3))) : make_int2(0, (int)(state - D));
            if (t < n_trees) *(uint4 *)(smem + TAB + (unsigned)t * 16u) = make_uint4(state | (plen << 16), (unsigned)sub.x, (unsigned)sub.y, 0u);
        }
    }
    __syncthreads();
    [[maybe_unused]] const unsigned long long st_t2 = CBS_CLOCK();
    unsigned entry[NB], bent[NB], bsz[NB];                          // bent: the block's entry | its first terminal << 16
    {   // 2. the wave's entries
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            entry[b] = 0u; bent[b] = 0u; bsz[b] = 0u;
            if (b * 64 < n_trees) {
                const int t = b * 64 + lane, tl = min(t, n_trees - 1);
                const uint4 e = *(const uint4 *)(smem + TAB + (unsigned)tl * 16u);
                const int cb = coff[tl];
                const unsigned D = (unsigned)(coff[tl + 1] - cb - 1) * 8u;
                unsigned state = e.x & 0xFFFFu, plen = e.x >> 16;
                bent[b] = state | (e.z << 16); bsz[b] = e.y;
                if (prefix) descend(RANGES + (unsigned)wave * 2u * P * 4u, t < n_trees, cb, D, state, plen);
                entry[b] = state | (plen << 16);
#ifdef RF_CBS_STATS
                {   // the subtree below the WAVE's entry
                    unsigned long long sl = t < n_trees ? (state < D ? 2ull * (unsigned)csub[2 * (cb + (int)(state >> 3))] + 2ull : 1ull) : 0ull;
                    unsigned long long big = sl > 400, s64 = sl > 64, sbig = big ? sl : 0;
                    for (int q = 32; q > 0; q >>= 1) { sl += __shfl_xor(sl, q); big += __shfl_xor(big, q); s64 += __shfl_xor(s64, q); sbig += __shfl_xor(sbig, q); }
                    if (lane == 0) { CBS_ADD(14, sl - sbig); CBS_ADD(15, big); }
                }
#endif
            }
        }
    }
    [[maybe_unused]] const unsigned long long st_t3 = CBS_CLOCK();
#ifdef RF_CBS_STATS
    if (wave == 0) {
        unsigned long long sl = 0, nz = 0, pb = 0, pw = 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) if (b * 64 + lane < n_trees) { sl += bsz[b] ? 2 * bsz[b] + 2 : 1; nz += bsz[b] == 0; pw += entry[b] >> 16; }
        for (int q = 32; q > 0; q >>= 1) { sl += __shfl_xor(sl, q); nz += __shfl_xor(nz, q); pw += __shfl_xor(pw, q); }
        if (lane == 0) { CBS_ADD(0, 1); CBS_ADD(1, sl); CBS_ADD(3, nz); CBS_ADD(7, pw); }
        (void)pb;
    }
#endif
    // slots (8-byte words) the region holds -- and no more than the waves fetch in one turn (256 16-byte units each, less the
    // ragged ends of 64 trees' ranges)
    const unsigned cap = min(region_bytes >> 3, 512u * (unsigned)nw - 256u);
    double acc[R], pending[R];
    unsigned node[R];
#pragma unroll
    for (int c = 0; c < R; ++c) { acc[c] = 0.0; pending[c] = 0.0; }
    // A ROUGH block -- its subtrees average more than rough_slots LDS slots a tree: noisy rasters, MHS_RF_PLAIN -- gains nothing
    // from batches of one or two trees and takes rf_walk_compact_kernel's loop instead: whole trees, the next one travelling
    // through registers while this one is walked (same region, same keys, the waves' entries as computed above).
    unsigned tot = 0u;
#pragma unroll
    for (int b = 0; b < NB; ++b) if (b * 64 + lane < n_trees) tot += bsz[b] ? 2u * bsz[b] + 4u : 2u;
#pragma unroll
    for (int q = 32; q > 0; q >>= 1) tot += (unsigned)__shfl_xor((int)tot, q);
    if (__builtin_amdgcn_readfirstlane((int)tot) > rough_slots * (unsigned)n_trees) {
        const unsigned nt = blockDim.x;
        __syncthreads();                                           // the prefixes' tables are no longer needed
        {
            const int o = coff[0], cnt = coff[1] - o;
            for (int e = threadIdx.x; e < cnt; e += (int)nt) ((uint2 *)smem)[e] = gnodes[o + e];
        }
        __syncthreads();
        int c0 = coff[0], c1 = coff[1], c2 = n_trees > 1 ? coff[2] : c1;
        int o = tree_off[0] - c0, o1 = n_trees > 1 ? tree_off[1] - c1 + 1 : 0;
        int levels = depth[0], levels1 = n_trees > 1 ? depth[1] : 0;
        int shallow = dmin ? dmin[0] : levels, shallow1 = n_trees > 1 ? (dmin ? dmin[1] : levels1) : 0;
        unsigned ecur = 0u;
        for (int t = 0; t < n_trees; ++t) {
            const int cnt1 = t + 1 < n_trees ? c2 - c1 : 0;
            const int c3 = t + 3 <= n_trees ? coff[t + 3] : c2;
            const int o2 = t + 2 < n_trees ? tree_off[t + 2] - c2 + (t + 2) : 0;
            const int levels2 = t + 2 < n_trees ? depth[t + 2] : 0;
            const int shallow2 = t + 2 < n_trees ? (dmin ? dmin[t + 2] : levels2) : 0;
            const unsigned D = (unsigned)(c1 - c0 - 1) * 8u;
            uint2 pn[PF];
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int e = (int)threadIdx.x + q * (int)nt;
                if (e < cnt1) pn[q] = gnodes[c1 + e];
            }
            if ((t & 63) == 0) {
                ecur = 0u;
#pragma unroll
                for (int b = 0; b < NB; ++b) if ((t >> 6) == b) ecur = entry[b];
            }
            const unsigned ent = (unsigned)__builtin_amdgcn_readlane((int)ecur, t & 63);
            const int plen = (int)(ent >> 16);
            const int lev = (ent & 0xFFFFu) >= D ? 0 : levels - plen, shal = min(max(shallow - plen, 0), lev);
#pragma unroll
            for (int c = 0; c < R; ++c) node[c] = ent & 0xFFFFu;
            auto level = [&]() {
                // (one walk after the other, as the compiler lays it out: issuing the four walks' reads side by side -- as the batched
                // walk of rf_walk_cbs_kernel does -- measured 6 % SLOWER here, where 15 waves walk ~20 levels a tree and the LDS pipe is the bound)
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    const uint2v nd = lds_u2(min(node[c], D));
                    const unsigned k = lds_u32(lane_base + (nd.x & 0xFFu) + c * 4);
                    unsigned child;
                    asm("v_cmp_gt_u32_e32 vcc, %1, %2\n\t"
                        "s_nop 1\n\t"
                        "v_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"
                        : "=v"(child) : "v"(k), "v"(nd.x), "v"(nd.y) : "vcc");
                    node[c] = max(child, node[c]);
                }
            };
            for (int l = 0; l < shal; ++l) level();
            for (int l = shal; l < lev; ++l) {
                const unsigned lowest = min(min(node[0], node[1]), min(node[2], node[3]));
                if (!__builtin_amdgcn_ballot_w64(lowest < D)) break;
                level();
            }
#pragma unroll
            for (int c = 0; c < R; ++c) {
                acc[c] = acc[c] + pending[c];
                pending[c] = glval[o + (int)(node[c] - D)];
            }
            __syncthreads();                                       // every wave has left this tree
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int e = (int)threadIdx.x + q * (int)nt;
                if (e < cnt1) *(uint2 *)(smem + (unsigned)e * 8u) = pn[q];
            }
            __syncthreads();
            o = o1; o1 = o2;
            c0 = c1; c1 = c2; c2 = c3;
            levels = levels1; levels1 = levels2;
            shallow = shallow1; shallow1 = shallow2;
        }
#pragma unroll
        for (int c = 0; c < R; ++c) {
            acc[c] = acc[c] + pending[c];
            if (live[c])
                emit(out, (int64_t)row[c] * g.ld_out + col[c], na[c] ? NAN : acc[c] / (double)n_trees, weight, accumulate);
        }
        return;
    }
    // Per tree of a batch, lane = tree: where its ranges go.  Records: global words [gr, gr + n) + one word that is stored as
    // zeros (the zero record) -> LDS slots [sr, sr + n]; predictions: global words [gv, gv + m) -> slots [sv, sv + m).  A range sits
    // at the parity of its global word index, so it travels in 16-byte units (ragged ends: the valid half alone).
    struct Place { unsigned gr, m, sr, sv, ur, uv; };
    auto place = [&](unsigned n, unsigned a, unsigned base, unsigned cb, unsigned gv) {
        Place P;
        P.gr = cb + (a >> 3);
        P.m = 2u * n + 4u <= cap ? (n ? n + 1u : 1u) : 0u;
        P.sr = base + ((base ^ P.gr) & 1u);
        P.sv = n ? P.sr + n + 1u : base;
        P.sv += (P.sv ^ gv) & 1u;
        P.ur = n ? ((P.gr + n + 2u) >> 1) - (P.gr >> 1) : 0u;
        P.uv = P.m ? ((gv + P.m + 1u) >> 1) - (gv >> 1) : 0u;
        return P;
    };
#pragma nounroll
    for (int b = 0; b < nbatch; ++b) {
        unsigned ecur = 0u, acur = 0u, ncur = 0u, fcur = 0u;
#pragma unroll
        for (int q = 0; q < NB; ++q) if (q == b) { ecur = entry[q]; acur = bent[q] & 0xFFFFu; ncur = bsz[q]; fcur = bent[q] >> 16; }
        const int tl = min(b * 64 + lane, n_trees - 1);
        const int cbv = coff[tl];
        const int ov = tree_off[tl] - cbv + tl + (int)fcur;        // the subtree's first terminal in glval
        const unsigned Dv = (unsigned)(coff[tl + 1] - cbv - 1) * 8u;
        const int levv = depth[tl], shv = dmin ? dmin[tl] : levv;
        // LDS slots of a tree: n records, the zero record, n + 1 predictions (+ a slot before each range for its parity); a
        // terminal entry: its prediction alone; a subtree too large for that -- a whole tree, MHS_RF_PLAIN -- leaves its
        // predictions in global memory
        const unsigned slots = !ncur ? 2u : 2u * ncur + 4u <= cap ? 2u * ncur + 4u : ncur + 2u;
        const int lend = min(64, n_trees - b * 64);
        // the batch from tree l0 on: the trees [l0, l1) whose ranges fit the region together, and where each goes
        auto plan = [&](int l0, unsigned &basev) {
            unsigned cum = lane >= l0 ? slots : 0u;
#pragma unroll
            for (int q = 1; q < 64; q <<= 1) { const unsigned up = (unsigned)__shfl_up((int)cum, q); if (lane >= q) cum += up; }
            const unsigned long long fit = __builtin_amdgcn_ballot_w64(lane >= l0 && lane < lend && cum <= cap);
            basev = cum - slots;
            return l0 + max(1, (int)__builtin_popcountll(fit));
        };
        // this wave's share of a batch's 16-byte units (the batch's ranges end to end, 256 units a wave, 4 a lane): requested
        // here, parked in LDS by park() -- a batch is fetched while the batch before it is walked.  Only the trees whose units
        // overlap the wave's 256 are looked at (lane = tree: a prefix sum of the units).
        uint4 w[4];
        unsigned waddr[4], wflag = 0u;                              // per unit: LDS byte address; bits 4q..4q+3: lo, hi valid, lo, hi zero
        auto fetch = [&](int l0, int l1, unsigned basev) {
            const unsigned x0 = 256u * (unsigned)wave;
            const Place V = place(ncur, acur, basev, (unsigned)cbv, (unsigned)ov);
            const unsigned unv = lane >= l0 && lane < l1 ? V.ur + V.uv : 0u;
            unsigned uend = unv;
#pragma unroll
            for (int q = 1; q < 64; q <<= 1) { const unsigned up = (unsigned)__shfl_up((int)uend, q); if (lane >= q) uend += up; }
            const unsigned long long mine = __builtin_amdgcn_ballot_w64(unv && uend > x0 && uend - unv < x0 + 256u);
            unsigned wsrc[4] = {0u, 0u, 0u, 0u};                   // the unit's index in its array; bit 31: the predictions
            wflag = 0u;
            for (unsigned long long left = mine; left; left &= left - 1ull) {
                const int l = __builtin_ctzll(left);
                const unsigned n = (unsigned)__builtin_amdgcn_readlane((int)ncur, l);
                const unsigned gr = (unsigned)__builtin_amdgcn_readlane((int)V.gr, l), gv = (unsigned)__builtin_amdgcn_readlane(ov, l);
                const unsigned m = (unsigned)__builtin_amdgcn_readlane((int)V.m, l);
                const unsigned sr = (unsigned)__builtin_amdgcn_readlane((int)V.sr, l), sv = (unsigned)__builtin_amdgcn_readlane((int)V.sv, l);
                const unsigned ur = (unsigned)__builtin_amdgcn_readlane((int)V.ur, l);
                const unsigned un = (unsigned)__builtin_amdgcn_readlane((int)unv, l);
                const unsigned ustart = (unsigned)__builtin_amdgcn_readlane((int)uend, l) - un;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned rel = x0 + (unsigned)(q * 64 + lane) - ustart;
                    if (rel < un) {
                        const bool isrec = rel < ur;
                        const unsigned g0 = isrec ? gr : gv, cnt = isrec ? n + 1u : m, s0 = isrec ? sr : sv;
                        const unsigned u = (g0 >> 1) + (isrec ? rel : rel - ur), w0 = 2u * u;
                        const unsigned lo = w0 >= g0 && w0 < g0 + cnt, hi = w0 + 1u >= g0 && w0 + 1u < g0 + cnt;
                        const unsigned zlo = isrec && w0 == gr + n, zhi = isrec && w0 + 1u == gr + n;
                        waddr[q] = (s0 + w0 - g0) * 8u;
                        wflag |= (lo | (hi << 1) | (zlo << 2) | (zhi << 3)) << (4 * q);
                        wsrc[q] = u | (isrec ? 0u : 0x80000000u);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)                             // all requests together, nothing waits for them here
                if ((wflag >> (4 * q)) & 3u) {
                    const uint4 *src = wsrc[q] >> 31 ? (const uint4 *)glval : (const uint4 *)gnodes;
                    w[q] = src[wsrc[q] & 0x7FFFFFFFu];
                }
        };
        auto park = [&]() {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned f = (wflag >> (4 * q)) & 15u;
                if (f & 3u) {
                    uint4 d = w[q];
                    if (f & 4u) { d.x = 0u; d.y = 0u; }
                    if (f & 8u) { d.z = 0u; d.w = 0u; }
                    if ((f & 3u) == 3u) *(uint4 *)(smem + waddr[q]) = d;
                    else if (f & 1u) *(uint2 *)(smem + waddr[q]) = make_uint2(d.x, d.y);
                    else *(uint2 *)(smem + waddr[q] + 8u) = make_uint2(d.z, d.w);
                }
            }
        };
        int l0 = 0;
        unsigned basev = 0u, basen = 0u;
        int l1 = plan(l0, basev);
        fetch(l0, l1, basev);                                       // the group's first batch: nothing to walk meanwhile
        while (l0 < lend) {
            [[maybe_unused]] const unsigned long long st_a = CBS_CLOCK();
            __syncthreads();                                       // every wave has left the region's previous trees
            park();
            __syncthreads();
            [[maybe_unused]] const unsigned long long st_b = CBS_CLOCK();
            st_stage += st_b - st_a;
            if (wave == 0 && lane == 0) CBS_ADD(2, 1);
            // the batch's trees, lane = tree: the walk's three scalars -- rec = LDS[min(state, Dz) + delta], a terminal code's
            // prediction at LDS[vbase + 8 code] -- and, where the wave enters the tree at a terminal (every cell of the wave gets
            // that prediction: three (wave, tree) pairs in four on cfg5's planes), the prediction itself
            const Place V = place(ncur, acur, basev, (unsigned)cbv, (unsigned)ov);
            const unsigned Dzv = ncur ? acur + 8u * ncur : 0u, deltav = 8u * V.sr - acur, vbasev = 8u * V.sv - 8u * (Dv + fcur);
            const bool inb = lane >= l0 && lane < l1;
            const bool quick = inb && V.m && (ecur & 0xFFFFu) >= Dzv;
            double tval = 0.0;
            if (quick) tval = lds_f64(vbasev + 8u * (ecur & 0xFFFFu));
            const unsigned long long quickm = __builtin_amdgcn_ballot_w64(quick);
            const unsigned long long bigm = __builtin_amdgcn_ballot_w64(inb && !V.m);
            int l2 = l1;
            if (l1 < lend) { l2 = plan(l1, basen); fetch(l1, l2, basen); }      // the next batch travels while this one is walked
            // (two copies of the walk: the one for batches whose predictions are all in LDS holds no global load, so nothing in it
            // waits for the batch that is on its way)
            auto walk = [&](int l, auto all_in_lds) {
                if ((quickm >> l) & 1ull) {
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(tval);
                    const unsigned vlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, l);
                    const unsigned vhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), l);
                    const double v = __longlong_as_double((long long)(((unsigned long long)vhi << 32) | vlo));
                    st_term += 1;
#pragma unroll
                    for (int c = 0; c < R; ++c) { acc[c] = acc[c] + pending[c]; pending[c] = v; }
                    return;
                }
                const unsigned ent = (unsigned)__builtin_amdgcn_readlane((int)ecur, l);
                const unsigned Dz = (unsigned)__builtin_amdgcn_readlane((int)Dzv, l), delta = (unsigned)__builtin_amdgcn_readlane((int)deltav, l);
                const unsigned vbase = (unsigned)__builtin_amdgcn_readlane((int)vbasev, l);
                const int levels = __builtin_amdgcn_readlane(levv, l), shallow = __builtin_amdgcn_readlane(shv, l);
                const int plen = (int)(ent >> 16);
                const int lev = (ent & 0xFFFFu) >= Dz ? 0 : levels - plen, shal = min(max(shallow - plen, 0), lev);
#pragma unroll
                for (int c = 0; c < R; ++c) node[c] = ent & 0xFFFFu;
                // the R walks' reads side by side (three passes: the compiler keeps the asm statements in order and would
                // otherwise wait for each walk's two reads before it issues the next walk's)
                auto level = [&]() {
                    uint2v nd[R];
                    unsigned k[R];
#pragma unroll
                    for (int c = 0; c < R; ++c) nd[c] = lds_u2(min(node[c], Dz) + delta);
#pragma unroll
                    for (int c = 0; c < R; ++c) k[c] = lds_u32(lane_base + (nd[c].x & 0xFFu) + c * 4);
#pragma unroll
                    for (int c = 0; c < R; ++c) {
                        unsigned child;
                        asm("v_cmp_gt_u32_e32 vcc, %1, %2\n\t"
                            "s_nop 1\n\t"
                            "v_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"
                            : "=v"(child) : "v"(k[c]), "v"(nd[c].x), "v"(nd[c].y) : "vcc");
                        node[c] = max(child, node[c]);
                    }
                };
                st_term += lev == 0;
                for (int q = 0; q < shal; ++q) { level(); ++st_levels; }
                for (int q = shal; q < lev; ++q) {
                    const unsigned lowest = min(min(node[0], node[1]), min(node[2], node[3]));
                    if (!__builtin_amdgcn_ballot_w64(lowest < Dz)) break;
                    level(); ++st_levels;
                }
                const bool in_lds = decltype(all_in_lds)::value || !((bigm >> l) & 1ull);
                // a terminal code c = D + the terminal's number: its prediction sits at LDS word sv + (number - lf)
                // (predictions left in global memory: glval[gv + (number - lf)], number = code - D)
                const unsigned gofs = in_lds ? 0u : (unsigned)__builtin_amdgcn_readlane(ov, l) - (unsigned)__builtin_amdgcn_readlane((int)Dv, l)
                                                    - (unsigned)__builtin_amdgcn_readlane((int)fcur, l);
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    acc[c] = acc[c] + pending[c];
                    if (in_lds) pending[c] = lds_f64(vbase + 8u * node[c]);
                    else pending[c] = glval[gofs + node[c]];
                }
            };
            if (!bigm)
                for (int l = l0; l < l1; ++l) walk(l, std::true_type());
            else
                for (int l = l0; l < l1; ++l) walk(l, std::false_type());
            st_walk += CBS_CLOCK() - st_b;
            l0 = l1; l1 = l2; basev = basen;
        }
    }
#ifdef RF_CBS_STATS
    if (lane == 0) {
        CBS_ADD(4, st_levels); CBS_ADD(5, st_term); CBS_ADD(6, 1);
        CBS_ADD(8, st_t1 - st_t0); CBS_ADD(9, st_t2 - st_t1); CBS_ADD(10, st_t3 - st_t2); CBS_ADD(11, st_stage); CBS_ADD(12, st_walk);
        CBS_ADD(13, CBS_CLOCK() - st_t0);
    }
#endif
#pragma unroll
    for (int c = 0; c < R; ++c) {
        acc[c] = acc[c] + pending[c];
        if (live[c])
            emit(out, (int64_t)row[c] * g.ld_out + col[c], na[c] ? NAN : acc[c] / (double)n_trees, weight, accumulate);
    }
}
#ifdef RF_CBS_STATS
extern "C" __attribute__((visibility("default"))) int mhs_debug_cbs_stats(unsigned long long *out16) {
    unsigned long long z[16] = {};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_cbs_stats), sizeof(z)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_cbs_stats), z, sizeof(z)) != hipSuccess;
}
#endif


// MHS_RF_KERNEL = ld | db | compact pins one of the three forest walk kernels where it applies (the equality tests and the
// benchmarks' comparisons); unset: the loader-wave kernel, else the double-buffered one, else the split-node one, else the
// generic node walk.  MHS_RF_PLAIN=1: no wave-uniform prefix and every tree to its full depth (the walk as round 2 had it).
enum { RF_PICK_AUTO = 0, RF_PICK_LD, RF_PICK_DB, RF_PICK_COMPACT, RF_PICK_SUB, RF_PICK_CBS };
static int rf_pick() {
    const char *e = getenv("MHS_RF_KERNEL");
    if (!e) return RF_PICK_AUTO;
    return !strcmp(e, "ld") ? RF_PICK_LD : !strcmp(e, "db") ? RF_PICK_DB : !strcmp(e, "compact") ? RF_PICK_COMPACT : !strcmp(e, "sub") ? RF_PICK_SUB
         : !strcmp(e, "cbs") ? RF_PICK_CBS : RF_PICK_AUTO;
}
static bool rf_plain() { return getenv("MHS_RF_PLAIN") != nullptr; }

// the double-buffered kernel: two node buffers within 16-bit byte addresses, predictions in global memory.  The predictor's
// key offset is one byte.
static size_t rf_walk_db_lds(const mhs_model *m, int log2r) {
    const size_t tree_bytes = std::max((size_t)m->rf_max_nodes * 16, (size_t)RF_COARSE_BYTES);
    return tree_bytes + (size_t)1024 * (((size_t)m->p * rf_walks(log2r)) | 1) * 4;
}
static int rf_walk_db_log2r(const mhs_model *m) {
    if (m->rf_max_nodes > 4095) return -1;
    for (int l2 = 3; l2 >= 1; --l2)
        if ((m->p * rf_walks(l2) * 4) <= 255 && rf_walk_db_lds(m, l2) <= LDS_LIMIT) return l2;
    return -1;
}

// loader-wave kernel (three node buffers): buffer stride (bytes, a template parameter) and walks per lane; false = does not apply
static bool rf_walk_ld_config(const mhs_model *m, int *log2r, int *stride) {
    if (m->rf_max_depth > 63) return false;                       // a tree's level counts travel in 6 bits each
    for (int st : {16384, 24576, 25600}) {
        if ((size_t)m->rf_max_nodes * 8 > (size_t)st) continue;
        for (int l2 = st != 25600 ? 3 : 2; l2 >= 2; --l2)      // the widest stride is instantiated for four walks only
            if ((m->p * rf_walks(l2) * 4) <= 255 &&
                (size_t)3 * st + 32 + (size_t)1024 * (((size_t)m->p * rf_walks(l2)) | 1) * 4 <= LDS_MAX) {
                *log2r = l2; *stride = st;
                return true;
            }
    }
    return false;
}

// key-space node records of the forest for this grid (see rf_walk_db_kernel); cached per geometry and key type
template <typename KT>
static int build_rf_nodes_t(mhs_model *m, const mhs_grid &grid, int C, int log2r, int form) {
    const bool big = form == RF_BIG;
    const size_t nn = m->rf_thr.size();
    std::vector<KT> tkey(nn, (KT)0);
    std::vector<std::vector<KT>> sorted((size_t)m->p);
    for (size_t k = 0; k < nn; ++k) {
        const unsigned v = m->rf_var[k];
        if (v == 0xFFFFu) continue;
        const KT tk = split_tkey<KT, true>((int)v, C, m->rf_thr[k], grid);
        tkey[k] = tk;
        sorted[(size_t)v].push_back(tk);
    }
    std::vector<int> off;
    std::vector<KT> flat;
    sort_unique(sorted, off, flat);
    for (int v = 0; v < m->p; ++v)
        if (sorted[(size_t)v].size() >= ((size_t)1 << 24)) { set_error("randomForest: too many distinct split values"); return MHS_ERR_INVALID; }
    if (int rc = publish_axis_ranks(m, sorted, C, grid)) return rc;
    std::vector<unsigned long long> rec((nn ? nn : 1) + 3200, 0ull);     // + 25 600 bytes: rf_walk_ld_kernel's loaders read whole strides
    const unsigned R = (unsigned)rf_walks(log2r);
    if (form == RF_COMPACT) {
        // Records of the SPLIT nodes only, in node order, then one all-zero record at byte address D = 8 * splits.
        // A child field holds the LDS byte address of a split child's record or, for a terminal child, D + its number
        // among the tree's terminals (node order): the walk reads record min(state, D), picks the child field and keeps
        // max(child, state) -- a split node's children come after it and every terminal code is >= D, so a terminal state
        // stays what it is.  clval holds the terminals' predictions in that numbering, tree after tree (tree t's first:
        // tree_off[t] - (coff[t] - t), its nodes less its split nodes before it).
        rec.clear();
        // csub: per record, the split nodes in the subtree below it (itself included) and that subtree's first terminal.
        // mhs_rf_load numbers nodes in pre-order, so the subtree of node k is the node range [k, end(k)): its split records are
        // consecutive, and so are its terminals (rf_walk_cbs_kernel stages both ranges).
        std::vector<int> coff(1, 0), newid, csub, before, end;
        std::vector<double> clval;
        for (int t = 0; t < m->n_trees; ++t) {
            const int o = m->rf_off[(size_t)t], cnt = m->rf_off[(size_t)t + 1] - o;
            newid.assign((size_t)cnt, 0);
            before.assign((size_t)cnt + 1, 0);
            end.assign((size_t)cnt, 0);
            unsigned splits = 0;
            for (int k = 0; k < cnt; ++k) {
                before[(size_t)k] = (int)splits;
                if (m->rf_var[(size_t)(o + k)] != 0xFFFFu) newid[(size_t)k] = (int)splits++;
                else clval.push_back(m->rf_thr[(size_t)(o + k)]);       // a terminal's rf_thr is its prediction (mhs_rf_load)
            }
            before[(size_t)cnt] = (int)splits;
            for (int k = cnt - 1; k >= 0; --k)
                end[(size_t)k] = m->rf_var[(size_t)(o + k)] == 0xFFFFu ? k + 1 : end[(size_t)m->rf_right[(size_t)(o + k)]];
            for (int k = 0; k < cnt; ++k)
                if (m->rf_var[(size_t)(o + k)] != 0xFFFFu) {
                    csub.push_back(before[(size_t)end[(size_t)k]] - before[(size_t)k]);
                    csub.push_back(k - before[(size_t)k]);
                }
            csub.push_back(0); csub.push_back(0);
            const unsigned D = 8u * splits;
            auto code = [&](unsigned k) { return m->rf_var[(size_t)o + k] != 0xFFFFu ? 8u * (unsigned)newid[k] : D + (k - (unsigned)before[k]); };
            for (int k = 0; k < cnt; ++k) {
                const unsigned v = m->rf_var[(size_t)(o + k)];
                if (v == 0xFFFFu) continue;
                const std::vector<KT> &sv = sorted[(size_t)v];
                const unsigned j = (unsigned)(std::lower_bound(sv.begin(), sv.end(), tkey[(size_t)(o + k)]) - sv.begin());
                const unsigned left = m->rf_left[(size_t)(o + k)], right = m->rf_right[(size_t)(o + k)];
                const unsigned children = code(left) | (code(right) << 16);
                rec.push_back(((unsigned long long)children << 32) | ((j << 8) | (v * R * 4u)));
            }
            rec.push_back(0ull);
            coff.push_back((int)rec.size());
        }
        rec.push_back(0ull); rec.push_back(0ull);                 // rf_walk_cbs_kernel reads 16-byte units: one word past a range's end
        clval.push_back(0.0); clval.push_back(0.0);
        if (int rc = publish(m, coff, &m->rf_coff)) return rc;
        if (int rc = publish(m, clval, &m->rf_clval)) return rc;
        if (int rc = publish(m, csub, &m->rf_csub)) return rc;
    } else
    for (size_t k = 0; k < nn; ++k) {
        const unsigned v = m->rf_var[k];
        const unsigned left = m->rf_left[k], right = m->rf_right[k];   // node indices within the tree (terminal: its own index)
        const unsigned unit = big ? 1u : 8u;   // children as node indices or as LDS byte addresses
        unsigned node0 = 0, children;
        if (v == 0xFFFFu) {
            children = (left * unit) | ((left * unit) << 16);
            if (!big) node0 = RF_LEAF_WORD;      // above every key: the compare is false, both children are the node itself
        } else {
            const std::vector<KT> &sv = sorted[(size_t)v];
            const unsigned j = (unsigned)(std::lower_bound(sv.begin(), sv.end(), tkey[k]) - sv.begin());
            node0 = (j << 8) | (v * R * 4u);
            children = (left * unit) | ((right * unit) << 16);
        }
        rec[k] = ((unsigned long long)children << 32) | node0;
    }
    if (int rc = publish(m, flat, (KT **)&m->lut_sorted)) return rc;
    if (int rc = publish(m, off, &m->lut_sorted_off)) return rc;
    return publish(m, rec, &m->rf_nodes);
}

static int build_rf_nodes(mhs_model *m, const mhs_grid &grid, int C, int log2r, int form, int key64, TreeTables *tt) {
    std::lock_guard<std::mutex> lk(m->mu);
    if (!(m->rf_nodes && m->rf_log2r == log2r && m->rf_form == form && same_meta(m, grid, C, key64))) {
        if (int rc = key64 ? build_rf_nodes_t<double>(m, grid, C, log2r, form) : build_rf_nodes_t<float>(m, grid, C, log2r, form)) return rc;
        m->meta_grid = grid; m->meta_C = C; m->meta_key64 = key64;
        m->rf_log2r = log2r; m->rf_form = form;
    }
    *tt = TreeTables{m->lut_sorted, m->lut_sorted_off, nullptr, m->rf_nodes, form == RF_COMPACT ? m->rf_coff : nullptr, nullptr, nullptr, nullptr,
                     m->axis_rank, m->axis_ncol, form == RF_COMPACT ? m->rf_csub : nullptr, form == RF_COMPACT ? m->rf_clval : nullptr};
    return MHS_OK;
}

// the loader-wave kernel where it applies (or is asked for), else the double-buffered one; *launched = false when neither does
static int launch_rf_walk(const mhs_model *m, const StackDev &s, const PredGeom &g, const mhs_grid &grid,
                          double w, int acc, double *out, hipStream_t st, int64_t total, bool *launched) {
    const int key64 = s.dtype == MHS_F64, pick = rf_pick();
    *launched = false;
    (void)total;
    int sb_l2 = 0, sb_stride = 0;
    if (pick == RF_PICK_SUB && rf_walk_ld_config(m, &sb_l2, &sb_stride) && m->n_trees <= 64 * RF_ENTRY_BATCHES) {
        // block-level subtrees in three shared buffers, one loader wave, 15 walker waves (same LDS budget as the loader-wave kernel)
        TreeTables tt;
        if (int rc = build_rf_nodes(const_cast<mhs_model *>(m), grid, s.C, sb_l2, RF_SMALL, key64, &tt)) return rc;
        const int R = rf_walks(sb_l2);
        const int tiles = rf_strips(g, R);
        unsigned blocks;
        if (tiles) {
            const int64_t tiles_x = (g.nc + 15) / 16, tiles_y = (g.nr + 4 * R - 1) / (4 * R);
            blocks = (unsigned)(((tiles_x + 4) / 5) * ((tiles_y + 2) / 3));                      // 5 x 3 wave tiles per block
        } else blocks = (unsigned)((rf_lane_count(g, R, 0) + 64 * 15 - 1) / (64 * 15));
        const int *dmin = rf_plain() ? nullptr : m->rf_dmin;
        const size_t tbytes = (size_t)3 * sb_stride + 64 + (size_t)1024 * (((size_t)m->p * R) | 1) * 4;
#define MHS_SB(L2, ST) (key64 ? rf_walk_sub_kernel<L2, true, ST> : rf_walk_sub_kernel<L2, false, ST>)
        auto tk = sb_stride == 16384 ? (sb_l2 == 3 ? MHS_SB(3, 16384) : MHS_SB(2, 16384))
                : sb_stride == 24576 ? (sb_l2 == 3 ? MHS_SB(3, 24576) : MHS_SB(2, 24576))
                                     : MHS_SB(2, 25600);
#undef MHS_SB
        if (tbytes <= LDS_MAX) {
            MHS_HIP(hipFuncSetAttribute((const void *)tk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tbytes));
            hipLaunchKernelGGL(tk, dim3(blocks), dim3(1024), tbytes, st, (const uint2 *)tt.rf_nodes, m->rf_lval, m->tree_off,
                               m->rf_depth, tt.sorted, tt.sorted_off, m->n_trees, m->p, s, g, w, acc, out, dmin, tiles,
                               (int)(tiles && dmin), tt.axis_rank, tt.axis_ncol);
            *launched = true;
            return MHS_OK;
        }
    }
    int ld_l2 = 0, ld_stride = 0;
    if (pick != RF_PICK_DB && rf_walk_ld_config(m, &ld_l2, &ld_stride)) {      // three node buffers, one loader wave, 15 walker waves
        TreeTables tt;
        if (int rc = build_rf_nodes(const_cast<mhs_model *>(m), grid, s.C, ld_l2, RF_SMALL, key64, &tt)) return rc;
        const int R = rf_walks(ld_l2);
        const int strips = rf_strips(g, R);
        const int64_t per_block = 64 * 15;
        unsigned blocks = (unsigned)((rf_lane_count(g, R, strips) + per_block - 1) / per_block);
        if (strips) blocks = (unsigned)((((int64_t)(g.nc + 15) / 16) * ((g.nr + 4 * R - 1) / (4 * R)) + 15 - 1) / 15);   // 16 x 4R-cell wave tiles
        const int *dmin = rf_plain() ? nullptr : m->rf_dmin;
        const size_t tbytes = (size_t)3 * ld_stride + 32 + (size_t)1024 * (((size_t)m->p * R) | 1) * 4;
#define MHS_LD(L2, ST) (key64 ? rf_walk_ld_kernel<L2, true, ST> : rf_walk_ld_kernel<L2, false, ST>)
        auto tk = ld_stride == 16384 ? (ld_l2 == 3 ? MHS_LD(3, 16384) : MHS_LD(2, 16384))
                : ld_stride == 24576 ? (ld_l2 == 3 ? MHS_LD(3, 24576) : MHS_LD(2, 24576))
                                     : MHS_LD(2, 25600);
#undef MHS_LD
        MHS_HIP(hipFuncSetAttribute((const void *)tk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tbytes));
        hipLaunchKernelGGL(tk, dim3(blocks), dim3(1024), tbytes, st, (const uint2 *)tt.rf_nodes, m->rf_lval, m->tree_off,
                           m->rf_depth, tt.sorted, tt.sorted_off, m->n_trees, m->p, s, g, w, acc, out, dmin, strips ? 2 : 0,
                           (int)(strips && dmin), tt.axis_rank, tt.axis_ncol);
        *launched = true;
        return MHS_OK;
    }
    const int log2r = rf_walk_db_log2r(m);
    if (log2r < 0) return MHS_OK;
    TreeTables tt;
    if (int rc = build_rf_nodes(const_cast<mhs_model *>(m), grid, s.C, log2r, RF_SMALL, key64, &tt)) return rc;
    const int R = rf_walks(log2r);
    const size_t dbytes = rf_walk_db_lds(m, log2r);
    // grids: a lane's walks on R adjacent rows and the early exit per wave
    const int strips = rf_strips(g, R);
    const unsigned blocks = (unsigned)((rf_lane_count(g, R, strips) + 1023) / 1024);
    const int *dmin = rf_plain() ? nullptr : m->rf_dmin;
    auto dk = log2r == 3 ? (key64 ? rf_walk_db_kernel<3, true> : rf_walk_db_kernel<3, false>)
            : log2r == 2 ? (key64 ? rf_walk_db_kernel<2, true> : rf_walk_db_kernel<2, false>)
                         : (key64 ? rf_walk_db_kernel<1, true> : rf_walk_db_kernel<1, false>);
    MHS_HIP(hipFuncSetAttribute((const void *)dk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dbytes));
    hipLaunchKernelGGL(dk, dim3(blocks), dim3(1024), dbytes, st, (const uint2 *)tt.rf_nodes, m->rf_lval, m->tree_off,
                       m->rf_depth, tt.sorted, key64, tt.sorted_off, m->n_trees, m->rf_max_nodes, m->p, s, g, w, acc, out, dmin, strips,
                       strips && dmin);
    *launched = true;
    return MHS_OK;
}

// rf_walk_compact_kernel: the most threads (whole waves, at least 8) whose keys fit beside one tree's split records,
// 0 = the form does not apply (terminal codes beyond 16 bits, too many predictors, no room)
static int rf_compact_threads(const mhs_model *m) {
    if (!m->rf_compact_ok || m->p * 16 > 255) return 0;
    const size_t tree_bytes = std::max((size_t)m->rf_cmax * 8, (size_t)RF_COARSE_BYTES);
    if (tree_bytes >= LDS_MAX) return 0;
    const size_t per_lane = (((size_t)m->p * 4) | 1) * 4;
    int nt = (int)std::min<size_t>(1024, (LDS_MAX - tree_bytes) / per_lane) / 64 * 64;
    if (nt < 512 || (size_t)nt * 8 < (size_t)m->rf_cmax) return 0;     // PF <= 8 records per thread
    return nt;
}

static int launch_rf_compact(const mhs_model *m, const StackDev &s, const PredGeom &g, const mhs_grid &grid,
                             double w, int acc, double *out, hipStream_t st, int64_t total, int nt) {
    const int key64 = s.dtype == MHS_F64;
    TreeTables tt;
    if (int rc = build_rf_nodes(const_cast<mhs_model *>(m), grid, s.C, 2, RF_COMPACT, key64, &tt)) return rc;
    const size_t bytes = std::max((size_t)m->rf_cmax * 8, (size_t)RF_COARSE_BYTES) + (size_t)nt * (((size_t)m->p * 4) | 1) * 4;
    const int strips = rf_strips(g, 4);
    const unsigned blocks = (unsigned)((rf_lane_count(g, 4, strips) + nt - 1) / nt);
    const int *dmin = rf_plain() ? nullptr : m->rf_dmin;
    const bool pf4 = (size_t)nt * 4 >= (size_t)m->rf_cmax;
    auto k = pf4 ? (key64 ? rf_walk_compact_kernel<4, true> : rf_walk_compact_kernel<4, false>)
                 : (key64 ? rf_walk_compact_kernel<8, true> : rf_walk_compact_kernel<8, false>);
    MHS_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL(k, dim3(blocks), dim3((unsigned)nt), bytes, st, (const uint2 *)tt.rf_nodes, tt.rf_clval, m->tree_off, tt.rf_coff,
                       m->rf_depth, tt.sorted, tt.sorted_off, m->n_trees, m->rf_cmax, m->p, s, g, w, acc, out, dmin, strips,
                       strips && dmin);
    return MHS_OK;
}


// rf_walk_cbs_kernel: threads and wave tiles (across x down) of a block, and the bytes of LDS left for subtrees beside the keys;
// false = does not apply (the compact form's own limits, more predictors or trees than the prefixes hold, a window too low for tiles)
static bool rf_cbs_config(const mhs_model *m, const PredGeom &g, int *nt, int *tw, int *th, unsigned *region) {
    if (!m->rf_compact_ok || m->p * 16 > 255 || m->p > RF_PREFIX_MAX_P || m->n_trees > 64 * RF_CBS_BATCHES || !rf_strips(g, 4)) return false;
    const size_t per_lane = (((size_t)m->p * 4) | 1) * 4;
    const size_t need = std::max(((size_t)m->rf_cmax + 1) * 8, (size_t)RF_COARSE_BYTES);     // a whole tree's records + the parity slot
    static const int shapes[][3] = {{1024, 4, 4}, {960, 5, 3}, {768, 4, 3}, {640, 5, 2}, {512, 4, 2}};
    for (const auto &sh : shapes) {
        if ((size_t)sh[0] * per_lane + need > LDS_MAX || need > ((size_t)sh[0] * 8 - 256) * 8) continue;   // ... and within one fetch turn
        if ((size_t)sh[0] * 8 < (size_t)m->rf_cmax) continue;     // the whole-tree loop: 8 records a thread
        *nt = sh[0]; *tw = sh[1]; *th = sh[2];
        *region = (unsigned)((LDS_MAX - (size_t)sh[0] * per_lane) & ~(size_t)7);
        return true;
    }
    return false;
}

static int launch_rf_cbs(const mhs_model *m, const StackDev &s, const PredGeom &g, const mhs_grid &grid,
                         double w, int acc, double *out, hipStream_t st, int nt, int tw, int th, unsigned region) {
    const int key64 = s.dtype == MHS_F64;
    TreeTables tt;
    if (int rc = build_rf_nodes(const_cast<mhs_model *>(m), grid, s.C, 2, RF_COMPACT, key64, &tt)) return rc;
    const size_t bytes = (size_t)region + (size_t)nt * (((size_t)m->p * 4) | 1) * 4;
    const int64_t tiles_x = (g.nc + 15) / 16, tiles_y = ((int64_t)g.nr + 15) / 16;
    const unsigned blocks = (unsigned)(((tiles_x + tw - 1) / tw) * ((tiles_y + th - 1) / th));
    const int *dmin = rf_plain() ? nullptr : m->rf_dmin;
    // blocks whose subtrees average more LDS slots than this take the whole-tree loop (MHS_RF_CBS_ROUGH: the measurements' knob)
    const char *re = getenv("MHS_RF_CBS_ROUGH");
    const unsigned rough = re ? (unsigned)atoi(re) : 3000u;
    auto k = key64 ? rf_walk_cbs_kernel<true> : rf_walk_cbs_kernel<false>;
    MHS_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL(k, dim3(blocks), dim3((unsigned)nt), bytes, st, (const uint2 *)tt.rf_nodes, tt.rf_clval, m->tree_off, tt.rf_coff,
                       tt.rf_csub, m->rf_depth, tt.sorted, tt.sorted_off, m->n_trees, region, m->p, s, g, w, acc, out, dmin,
                       tt.axis_rank, tt.axis_ncol, tw, th, (int)(dmin != nullptr), rough);
    return MHS_OK;
}

int launch_forest(const mhs_model *m, const StackDev &s, const PredGeom &g, const mhs_grid *grid, double weight, int accumulate,
                  double *out, hipStream_t st, int64_t total, bool *launched) {
    *launched = false;
    if (!(grid && m->rf_fast && !s.all_from_planes && !getenv("MHS_TREES_GENERIC"))) return MHS_OK;
    const int pick = rf_pick();
    int cnt = 0, ctw = 0, cth = 0;
    unsigned cregion = 0;
    if (pick == RF_PICK_CBS && rf_cbs_config(m, g, &cnt, &ctw, &cth, &cregion)) {      // pinned: whatever the trees' size
        if (int rc = launch_rf_cbs(m, s, g, *grid, weight, accumulate, out, st, cnt, ctw, cth, cregion)) return rc;
        *launched = true;
        return MHS_OK;
    }
    // trees of 3 201 .. 4 095 nodes fit neither three times in LDS (the loader-wave kernel) nor need the compact form, and used
    // to take the double-buffered kernel: the block-subtree kernel is 1.6 x faster there on the 8d planes (124 -> 77 ms per 1e8
    // cells and 500 trees), 1.3 x with 1 % noise, 0.9 x with 10 % (tools/r06_forest_small.py 8000 6000)
    int ld_l2 = 0, ld_stride = 0;
    if (pick == RF_PICK_AUTO && !rf_walk_ld_config(m, &ld_l2, &ld_stride) && rf_cbs_config(m, g, &cnt, &ctw, &cth, &cregion)) {
        if (int rc = launch_rf_cbs(m, s, g, *grid, weight, accumulate, out, st, cnt, ctw, cth, cregion)) return rc;
        *launched = true;
        return MHS_OK;
    }
    if (pick != RF_PICK_COMPACT)
        if (int rc = launch_rf_walk(m, s, g, *grid, weight, accumulate, out, st, total, launched)) return rc;
    if (*launched) return MHS_OK;
    if (pick != RF_PICK_COMPACT && rf_cbs_config(m, g, &cnt, &ctw, &cth, &cregion)) {   // trees beyond 4 095 nodes: the block's subtrees
        if (int rc = launch_rf_cbs(m, s, g, *grid, weight, accumulate, out, st, cnt, ctw, cth, cregion)) return rc;
        *launched = true;
        return MHS_OK;
    }
    const int nt = rf_compact_threads(m);      // ... or whole trees, split nodes only
    if (nt > 0) {
        if (int rc = launch_rf_compact(m, s, g, *grid, weight, accumulate, out, st, total, nt)) return rc;
        *launched = true;
    }
    return MHS_OK;
}

}  // namespace mhs

using namespace mhs;

extern "C" {

int mhs_rf_load(int64_t n_trees, const int64_t *tree_offsets, const int32_t *left, const int32_t *right,
                const int32_t *status, const int32_t *best_var, const double *split,
                const double *node_pred, int p, mhs_model **out) {
    if (int rc = check_common(p, out)) return rc;
    MHS_REQUIRE(tree_offsets && left && right && status && best_var && split && node_pred, "NULL rf array");
    MHS_REQUIRE(n_trees >= 1 && n_trees < (1LL << 30) && tree_offsets[0] == 0, "bad tree offsets");
    const int64_t nn = tree_offsets[n_trees];
    MHS_REQUIRE(nn < (1LL << 31), "too many nodes");
    std::vector<Node> nodes((size_t)nn);
    std::vector<int> off((size_t)n_trees + 1);
    for (int64_t t = 0; t <= n_trees; ++t) off[t] = (int)tree_offsets[t];
    for (int64_t t = 0; t < n_trees; ++t) {
        const int64_t o = tree_offsets[t], cnt = tree_offsets[t + 1] - o;
        MHS_REQUIRE(cnt >= 1 && cnt <= 65535, "a randomForest tree must have 1..65535 nodes");
        for (int64_t k = 0; k < cnt; ++k) {
            Node &nd = nodes[(size_t)(o + k)];
            if (status[o + k] == -1) {
                nd.val = node_pred[o + k]; nd.var = -1; nd.left = nd.right = nd.missing = 0;
            } else {
                MHS_REQUIRE(best_var[o + k] >= 1 && best_var[o + k] <= p, "rf bestvar out of range");
                MHS_REQUIRE(left[o + k] >= 1 && left[o + k] <= cnt && right[o + k] >= 1 && right[o + k] <= cnt,
                            "rf daughter index out of range");
                nd.val = split[o + k]; nd.var = (short)(best_var[o + k] - 1);
                nd.left = (unsigned short)(left[o + k] - 1); nd.right = (unsigned short)(right[o + k] - 1);
                nd.missing = 0;
            }
        }
    }
    // Nodes renumbered in PRE-ORDER (round 5): a subtree is then a contiguous range of node ids [k, k + size(k)), which is what
    // lets the forest kernel stage, per block of cells, only the subtree its cells can reach (rf_walk_ring_kernel).  The tree is
    // the same tree -- same splits, same leaves, same predictions; randomForest's own numbering (breadth-first, right daughter =
    // left daughter + 1) is not needed by any kernel.  A malformed tree (a node reached twice or never, a cycle) keeps its
    // numbering and takes the generic walk.
    bool paired = true;
    {
        std::vector<Node> ordered(nodes.size());
        std::vector<int> newid, stack;
        for (int64_t t = 0; t < n_trees && paired; ++t) {
            const int o = off[t], cnt = off[t + 1] - off[t];
            newid.assign((size_t)cnt, -1);
            stack.assign(1, 0);
            int next = 0;
            while (!stack.empty() && paired) {
                const int k = stack.back();
                stack.pop_back();
                if (k < 0 || k >= cnt || newid[(size_t)k] >= 0) { paired = false; break; }
                newid[(size_t)k] = next++;
                const Node &nd = nodes[(size_t)(o + k)];
                if (nd.var >= 0) { stack.push_back(nd.right); stack.push_back(nd.left); }      // left first: left child = parent + 1
            }
            if (next != cnt) paired = false;
            for (int k = 0; k < cnt && paired; ++k) {
                Node nd = nodes[(size_t)(o + k)];
                if (nd.var >= 0) { nd.left = (unsigned short)newid[nd.left]; nd.right = (unsigned short)newid[nd.right]; }
                ordered[(size_t)(o + newid[(size_t)k])] = nd;
            }
        }
        if (paired) nodes.swap(ordered);
    }
    mhs_model *m = new mhs_model();
    m->kind = K_RF; m->p = p; m->n_trees = (int)n_trees;
    if (int rc = finish_trees(m, nodes, off)) { mhs_model_free(m); return rc; }
    if (paired) {
        m->rf_thr.resize(nodes.size());
        m->rf_left.resize(nodes.size());
        m->rf_right.resize(nodes.size());
        m->rf_var.resize(nodes.size());
        std::vector<double> lval(nodes.size());
        std::vector<int> depth((size_t)n_trees, 0), shallow((size_t)n_trees, 0), lev;
        int max_nodes = 0;
        for (int64_t t = 0; t < n_trees && paired; ++t) {
            const int o = off[t], cnt = off[t + 1] - off[t];
            if (cnt > 65535) { paired = false; break; }   // node indices within a tree are 16-bit in the walk kernels
            max_nodes = std::max(max_nodes, cnt);
            m->rf_off.push_back(o);
            lev.assign((size_t)cnt, -1);
            lev[0] = 0;
            int dmax = 0, dlow = 1 << 30;
            for (int k = 0; k < cnt; ++k) {  // randomForest numbers children after their parent
                const Node &nd = nodes[(size_t)(o + k)];
                if (lev[k] < 0) { paired = false; break; }  // unreachable or out-of-order node
                m->rf_thr[(size_t)(o + k)] = nd.val;
                lval[(size_t)(o + k)] = nd.var < 0 ? nd.val : 0.0;
                if (nd.var >= 0) {
                    if (nd.left != k + 1 || nd.right <= nd.left || nd.right >= cnt) { paired = false; break; }      // pre-order
                    m->rf_left[(size_t)(o + k)] = nd.left;
                    m->rf_right[(size_t)(o + k)] = nd.right;
                    m->rf_var[(size_t)(o + k)] = (unsigned short)nd.var;
                    lev[nd.left] = lev[nd.right] = lev[k] + 1;
                    dmax = std::max(dmax, lev[k] + 1);
                } else {
                    m->rf_left[(size_t)(o + k)] = m->rf_right[(size_t)(o + k)] = (unsigned short)k;  // self loop
                    m->rf_var[(size_t)(o + k)] = 0xFFFFu;
                    dlow = std::min(dlow, lev[k]);
                }
            }
            depth[(size_t)t] = dmax;
            m->rf_max_depth = std::max(m->rf_max_depth, dmax);
            shallow[(size_t)t] = std::min(dlow, dmax);
        }
        if (paired) {
            m->rf_off.push_back((int)nodes.size());
            m->rf_compact_ok = 1;
            for (int64_t t = 0; t < n_trees; ++t) {
                int splits = 0;
                for (int k = m->rf_off[(size_t)t]; k < m->rf_off[(size_t)t + 1]; ++k) splits += m->rf_var[(size_t)k] != 0xFFFFu;
                m->rf_cmax = std::max(m->rf_cmax, splits + 1);
                if (8 * splits + (m->rf_off[(size_t)t + 1] - m->rf_off[(size_t)t]) > 65535) m->rf_compact_ok = 0;
            }
            m->rf_fast = true;
            m->rf_max_nodes = max_nodes;
            int rc = to_device(lval.data(), lval.size(), &m->rf_lval);
            if (!rc) rc = to_device(depth.data(), depth.size(), &m->rf_depth);
            if (!rc) rc = to_device(shallow.data(), shallow.size(), &m->rf_dmin);
            if (rc) { mhs_model_free(m); return rc; }
        }
    }
    m->slot = current_slot(); m->device = ctx().device;
    m->reload = [n_trees, to = std::vector<int64_t>(tree_offsets, tree_offsets + n_trees + 1), l = std::vector<int32_t>(left, left + nn),
                 r = std::vector<int32_t>(right, right + nn), st = std::vector<int32_t>(status, status + nn),
                 bv = std::vector<int32_t>(best_var, best_var + nn), sp = std::vector<double>(split, split + nn),
                 np_ = std::vector<double>(node_pred, node_pred + nn), p](mhs_model **o) {
        return mhs_rf_load(n_trees, to.data(), l.data(), r.data(), st.data(), bv.data(), sp.data(), np_.data(), p, o);
    };
    *out = m;
    return MHS_OK;
}

}  // extern "C"
