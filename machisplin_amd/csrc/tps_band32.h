// Round-4 GCV route of the spline fit: 32-column panels, GCV on the band on the GPU (tps_band32.hip)
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>
#include "common.h"

namespace mhs {
constexpr int B32_NB = 32;               // band width of the reduction
constexpr int B32_MAXPART = 16;          // partial sums a kernel loads in ONE batch
constexpr int B32_MAXBLK = 32;           // row blocks (partial sums) per panel: two batches past 16 (panels of more than 4 096 rows)
constexpr int B32_BT_MAXBLK = 128;       // row blocks of the back-transform (up to 32 768 unknowns)
constexpr int B32_PANEL_REC = 2 * B32_NB * B32_NB;      // doubles a panel leaves for the back-transform: T, then the top block of V
constexpr int B32_MIN_M = 320;           // smallest order the route is used for (below: tps_fit.hip's 8-column route)
constexpr int B32_MAX_M = B32_BT_MAXBLK * 256;
constexpr int B32_MAXLAM = 1024;         // lambdas per search round
constexpr int B32_PAIR_SPLIT = 768;      // eval_pair: slots of the first batch (the second takes the rest)

// device work space of one fit on this route, carved from the lane's arena
struct Band32Ws {
    double *Gp1, *Gp2, *R1, *Qtop, *aux, *Zc[2], *Vr[2], *Yp, *Wh, *Mp, *sgp, *Tall, *ab, *abF, *abR, *win, *res, *lamd, *out, *Lbuf, *ybuf, *btpart;
    int *flags;
};
size_t band32_workspace_bytes(int m, int64_t n);
void band32_carve(Band32Ws &w, char *base, int m, int64_t n);
int band32_npanels(int m);

// Stage 1.  A: n x n projected matrix (column-major, ld; B = A[3:, 3:], row 3 of every column 16-byte aligned), g_dev: Q2'y
// (m entries, rotated in place to Q'g).  Leaves the reflectors below the band, T factors in ws.Tall, the band in ws.ab
// (device, ab[j * 33 + d]).  *breakdown = 1 if a panel's Cholesky met a non-positive pivot (the caller falls back).
int band32_reduce(FitLane &L, hipStream_t s, hipStream_t s2, double *A, int64_t ld, int m, int64_t vs, double *g_dev, Band32Ws &ws,
                  int *breakdown);
// g <- Q'g for another right-hand side with a finished reduction (the reduction cache), bit for bit what band32_reduce did
int band32_qt(hipStream_t s, const double *A, int64_t ld, int m, const double *Tall, double *g_dev, double *sgp);
// r <- Q r
int band32_backtransform(hipStream_t s, const double *A, int64_t ld, int m, const double *Tall, double *r_dev, double *btpart);

// lambda by GCV on the band (device evaluations, host-driven rounds), then q = (Bb + lambda I)^-1 g (host vector, m).
// ab_host: the band on the host (m x 33).  Returns MHS_OK or an error code (error text set).
struct Band32Search {
    hipStream_t s = nullptr, s_aux = nullptr;      // s_aux: an idle second stream for a batch beside the main one (optional)
    const double *ab_dev = nullptr, *g_dev = nullptr;
    const double *ab_host = nullptr, *g_host = nullptr;
    int m = 0;
    int64_t n = 0, N = 0;
    double pure_ss = 0.0;
    Band32Ws *ws = nullptr;
    double *pin = nullptr;            // pinned host buffer: B32_MAXLAM lambdas, then 4 doubles per lambda of results
    int rounds = 0;                   // evaluation rounds made (diagnostic)
    bool packed = false;              // ws->abF / abR hold this band and right-hand side
    int pack();
    // per lambda: eigenvalues of Bb below -lambda (inertia), tr (Bb + lambda I)^-1, g'(Bb + lambda I)^-2 g
    int eval_batch(const double *lam, int count, bool deriv, double *neg, double *tr, double *q2);
    int eval_pair(const double *lamA, int nA, double *negA, const double *lamB, int nB, double *negB, double *trB);
    int enqueue(hipStream_t st, int off, const double *lam, int nl, bool deriv);
    void collect(int off, int nl, double *neg, double *tr, double *q2) const;
    void report(int off, int nl, bool deriv) const;
    int find_lambda(int mode, double *lam_out);
    int solve(double lam, double *gcv, double *eff_df, double *q_host);
    void gcv_from_terms(double lam, double tr_inv, double qq, double *gcv, double *tra) const;
};
int band32_pinned(FitLane &L, double **out);      // the lane's pinned buffer (allocated on first use)
}  // namespace mhs
