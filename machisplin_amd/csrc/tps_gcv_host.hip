// Host side of the TPS fit: everything that is O(n) per evaluation once the GPU has
// reduced B = Q2' K Q2 to the symmetric tridiagonal T = P' B P (diag a, off-diag b) and
// rotated the data, g = P' Q2' y.
//
//   GCV(l) = (RSS(l)/n + pure_ss/(N-n)) / (1 - trA(l)/n)^2          (fields' Krig.fgcv)
//   RSS(l) = l^2 |(T + l I)^-1 g|^2 ,  trA(l) = 3 + m - l tr((T + l I)^-1) ,  m = n - 3
//
// which equals the eigenvalue form  RSS = sum (l z_i/(e_i+l))^2, trA = 3 + sum e_i/(e_i+l)
// that fields evaluates, without needing the spectrum: a tridiagonal solve and the
// diagonal of a tridiagonal inverse are O(m).  The lambda grid and golden-section search
// follow gcv.Krig / Krig.find.gcvmin / golden.section.search.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include "common.h"
#include "tps_host.h"

namespace mhs {

// number of eigenvalues of the tridiagonal (a, b) strictly below x (Sturm sequence)
static int sturm_count(const double *a, const double *b, int64_t m, double x) {
    int cnt = 0;
    double q = a[0] - x;
    if (q < 0) ++cnt;
    for (int64_t i = 1; i < m; ++i) {
        const double den = (q != 0.0) ? q : 1e-300;
        q = a[i] - x - b[i - 1] * b[i - 1] / den;
        if (q < 0) ++cnt;
    }
    return cnt;
}

// k-th smallest eigenvalue (0-based) by bisection on Gershgorin bounds
static double eig_kth(const double *a, const double *b, int64_t m, int64_t k) {
    double lo = a[0], hi = a[0];
    for (int64_t i = 0; i < m; ++i) {
        const double r = (i > 0 ? fabs(b[i - 1]) : 0.0) + (i + 1 < m ? fabs(b[i]) : 0.0);
        lo = std::min(lo, a[i] - r);
        hi = std::max(hi, a[i] + r);
    }
    for (int it = 0; it < 200; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (mid <= lo || mid >= hi) break;
        if (sturm_count(a, b, m, mid) > k) hi = mid; else lo = mid;
    }
    return 0.5 * (lo + hi);
}

void TridiagGcv::eval(double lam, double *gcv, double *tra, double *q_out) const {
    // M = T + lam I ; forward pivots dp, backward pivots dm
    std::vector<double> &dp = work_dp, &dm = work_dm, &q = work_q;
    dp.resize(m); dm.resize(m); q.resize(m);
    dp[0] = a[0] + lam;
    for (int64_t i = 1; i < m; ++i) dp[i] = a[i] + lam - b[i - 1] * b[i - 1] / dp[i - 1];
    dm[m - 1] = a[m - 1] + lam;
    for (int64_t i = m - 2; i >= 0; --i) dm[i] = a[i] + lam - b[i] * b[i] / dm[i + 1];
    double tr_inv = 0.0;
    for (int64_t i = 0; i < m; ++i) tr_inv += 1.0 / (dp[i] + dm[i] - (a[i] + lam));
    // Thomas solve M q = g with the forward pivots
    q[0] = g[0];
    for (int64_t i = 1; i < m; ++i) q[i] = g[i] - b[i - 1] / dp[i - 1] * q[i - 1];
    q[m - 1] /= dp[m - 1];
    for (int64_t i = m - 2; i >= 0; --i) q[i] = (q[i] - b[i] * q[i + 1]) / dp[i];
    double qq = 0.0;
    for (int64_t i = 0; i < m; ++i) qq += q[i] * q[i];
    const double rss = lam * lam * qq;
    const double tr = 3.0 + (double)m - lam * tr_inv;
    double mse = rss / (double)n;
    if (N - n > 0) mse += pure_ss / (double)(N - n);
    const double den = 1.0 - tr / (double)n;
    if (gcv) *gcv = den > 0 ? mse / (den * den) : NAN;
    if (tra) *tra = tr;
    if (q_out) std::copy(q.begin(), q.end(), q_out);
}

static double golden_section(const TridiagGcv &t, double ax, double bx, double cx, double tol) {
    const double r = 0.61803399, con = 1.0 - r;
    double x0 = ax, x3 = cx, x1, x2, f1, f2;
    if (fabs(cx - bx) > fabs(bx - ax)) { x1 = bx; x2 = bx + con * (cx - bx); }
    else { x2 = bx; x1 = bx - con * (bx - ax); }
    t.eval(x1, &f1, nullptr, nullptr);
    t.eval(x2, &f2, nullptr, nullptr);
    for (int k = 0; k < 25; ++k) {
        if (f2 < f1) {
            x0 = x1; x1 = x2; x2 = r * x1 + con * x3;
            f1 = f2; t.eval(x2, &f2, nullptr, nullptr);
        } else {
            x3 = x2; x2 = x1; x1 = r * x2 + con * x0;
            f2 = f1; t.eval(x1, &f1, nullptr, nullptr);
        }
        if (fabs(f2 - f1) < tol) break;
    }
    (void)x0; (void)x3;
    return f1 < f2 ? x1 : x2;
}

double TridiagGcv::find_lambda(int mode) const {
    const double emax = eig_kth(a, b, m, m - 1);
    const double emin = std::max(eig_kth(a, b, m, 0), 1e-300);
    double tr;
    double l1 = emax;
    for (int k = 0; k < 20; ++k) {
        eval(l1, nullptr, &tr, nullptr);
        if (tr < 3.0 + 0.05) break;
        l1 *= 4.0;
    }
    double l2 = emin;
    for (int k = 0; k < 20; ++k) {
        eval(l2, nullptr, &tr, nullptr);
        if (tr > 0.95 * (double)n) break;
        l2 /= 4.0;
    }
    const int nstep = 200;
    std::vector<double> grid, gv;
    const double la = log(l2), lb = log(l1);
    for (int i = 0; i < nstep; ++i) {
        const double lam = exp(la + (lb - la) * (double)i / (double)(nstep - 1));
        double gcv;
        eval(lam, &gcv, nullptr, nullptr);
        if (!std::isnan(gcv)) { grid.push_back(lam); gv.push_back(gcv); }
    }
    if (grid.empty()) return NAN;
    size_t il = 0;
    for (size_t i = 1; i < gv.size(); ++i) if (gv[i] < gv[il]) il = i;
    if (il == 0 || il + 1 == gv.size()) return grid[il];
    if (mode == MHS_GCV_FIELDS) return golden_section(*this, grid[il - 1], grid[il], grid[il + 1], 0.01 * gv[il]);
    // converged: golden section on log(lambda)
    double lo = log(grid[il - 1]), hi = log(grid[il + 1]);
    const double r = 0.5 * (sqrt(5.0) - 1.0);
    double x1 = hi - r * (hi - lo), x2 = lo + r * (hi - lo), f1, f2;
    eval(exp(x1), &f1, nullptr, nullptr);
    eval(exp(x2), &f2, nullptr, nullptr);
    for (int it = 0; it < 200; ++it) {
        if (f1 < f2) { hi = x2; x2 = x1; f2 = f1; x1 = hi - r * (hi - lo); eval(exp(x1), &f1, nullptr, nullptr); }
        else { lo = x1; x1 = x2; f1 = f2; x2 = lo + r * (hi - lo); eval(exp(x2), &f2, nullptr, nullptr); }
        if (fabs(hi - lo) < 1e-13) break;
    }
    return exp(0.5 * (lo + hi));
}

// ------------------------------------------------------------------ banded form --
bool BandGcv::eval(double lam, double *gcv, double *tra, double *q_out, Work &wk) const {
    const int64_t w = bw + 1;
    std::vector<double> &L = wk.L, &Z = wk.Z, &q = wk.q;
    L.assign(ab, ab + w * m);
    for (int64_t j = 0; j < m; ++j) L[w * j] += lam;
    // banded Cholesky, right-looking
    for (int64_t j = 0; j < m; ++j) {
        const double d = L[w * j];
        if (!(d > 0.0)) return false;
        const double ljj = sqrt(d), inv = 1.0 / ljj;
        L[w * j] = ljj;
        const int64_t kmax = std::min<int64_t>(bw, m - 1 - j);
        for (int64_t i = 1; i <= kmax; ++i) L[i + w * j] *= inv;
        for (int64_t k = 1; k <= kmax; ++k) {
            const double lk = L[k + w * j];
            double *col = &L[w * (j + k)];
            for (int64_t i = k; i <= kmax; ++i) col[i - k] -= L[i + w * j] * lk;
        }
    }
    // solve (L L') q = g
    q.assign(g, g + m);
    for (int64_t j = 0; j < m; ++j) {
        q[j] /= L[w * j];
        const int64_t kmax = std::min<int64_t>(bw, m - 1 - j);
        for (int64_t i = 1; i <= kmax; ++i) q[j + i] -= L[i + w * j] * q[j];
    }
    for (int64_t j = m - 1; j >= 0; --j) {
        const int64_t kmax = std::min<int64_t>(bw, m - 1 - j);
        double sum = q[j];
        for (int64_t i = 1; i <= kmax; ++i) sum -= L[i + w * j] * q[j + i];
        q[j] = sum / L[w * j];
    }
    double qq = 0.0;
    for (int64_t i = 0; i < m; ++i) qq += q[i] * q[i];
    // trace of the inverse: Takahashi recurrence on M = Lt D Lt', Lt = L diag(1/L_jj) unit lower,
    // D_j = L_jj^2; Z = M^-1 restricted to the band, built from the last column backwards
    Z.assign(w * m, 0.0);
    double tr_inv = 0.0;
    double lt[64];
    for (int64_t j = m - 1; j >= 0; --j) {
        const int64_t kmax = std::min<int64_t>(bw, m - 1 - j);
        const double inv = 1.0 / L[w * j];
        for (int64_t k = 1; k <= kmax; ++k) lt[k] = L[k + w * j] * inv;
        for (int64_t i = j + kmax; i > j; --i) {  // off-diagonal entries of column j
            double sum = 0.0;
            for (int64_t k = j + 1; k <= j + kmax; ++k)
                sum += lt[k - j] * ((i >= k) ? Z[(i - k) + w * k] : Z[(k - i) + w * i]);
            Z[(i - j) + w * j] = -sum;
        }
        double sum = 0.0;
        for (int64_t k = 1; k <= kmax; ++k) sum += lt[k] * Z[k + w * j];
        Z[w * j] = inv * inv - sum;
        tr_inv += Z[w * j];
    }
    const double rss = lam * lam * qq;
    const double tr = 3.0 + (double)m - lam * tr_inv;
    double mse = rss / (double)n;
    if (N - n > 0) mse += pure_ss / (double)(N - n);
    const double den = 1.0 - tr / (double)n;
    if (gcv) *gcv = den > 0 ? mse / (den * den) : NAN;
    if (tra) *tra = tr;
    if (q_out) std::copy(q.begin(), q.end(), q_out);
    return true;
}

int BandGcv::inertia_below(double x, Work &wk) const {
    const int64_t w = bw + 1;
    std::vector<double> &L = wk.L;
    L.assign(ab, ab + w * m);
    int cnt = 0;
    for (int64_t j = 0; j < m; ++j) {
        double d = L[w * j] - x;
        if (d == 0.0) d = -1e-300;
        if (d < 0.0) ++cnt;
        const double inv = 1.0 / d;
        const int64_t kmax = std::min<int64_t>(bw, m - 1 - j);
        for (int64_t k = 1; k <= kmax; ++k) {
            const double lk = L[k + w * j] * inv;
            double *col = &L[w * (j + k)];
            for (int64_t i = k; i <= kmax; ++i) col[i - k] -= L[i + w * j] * lk;
        }
    }
    return cnt;
}

namespace {
// Minimal persistent worker pool for the independent GCV evaluations (no R / HIP API is touched
// from the workers).  Threads are created once per search, not once per parallel region: a
// bisection round is ~0.3 ms of work, thread creation on a many-core host costs about as much.
class Pool {
public:
    explicit Pool(int n) : n_(std::max(1, n)) {
        for (int t = 1; t < n_; ++t) workers_.emplace_back([this, t]() { loop(t); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto &th : workers_) th.join();
    }
    int size() const { return n_; }
    template <typename F>
    void run(int count, F f) {  // f(index, thread id)
        if (n_ == 1 || count <= 1) { for (int i = 0; i < count; ++i) f(i, 0); return; }
        job_ = [&f](int i, int t) { f(i, t); };
        count_ = count;
        next_.store(0);
        pending_.store(n_ - 1);
        { std::lock_guard<std::mutex> lk(mu_); ++gen_; }
        cv_.notify_all();
        for (int i = next_++; i < count_; i = next_++) job_(i, 0);
        while (pending_.load(std::memory_order_acquire) > 0) std::this_thread::yield();
    }
private:
    void loop(int t) {
        unsigned long seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return gen_ != seen; }); seen = gen_; if (stop_) return; }
            for (int i = next_++; i < count_; i = next_++) job_(i, t);
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }
    int n_;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    unsigned long gen_ = 0;
    bool stop_ = false;
    std::function<void(int, int)> job_;
    int count_ = 0;
    std::atomic<int> next_{0}, pending_{0};
};
int pick_threads(int requested) {
    if (requested > 0) return requested;
    if (const char *e = getenv("MHS_GCV_THREADS")) { const int v = atoi(e); if (v > 0) return std::min(v, 256); }
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(16u, hc ? hc : 1u));
}
}  // namespace

static double band_eig_kth(const BandGcv &B, int64_t k, Pool &pool, std::vector<BandGcv::Work> &wk) {
    const double *ab = B.ab; const int bw = B.bw; const int64_t m = B.m;
    const int64_t w = bw + 1;
    double lo = ab[0], hi = ab[0];
    for (int64_t j = 0; j < m; ++j) {  // Gershgorin
        double r = 0.0;
        for (int64_t d = 1; d <= bw; ++d) {
            if (j + d < m) r += fabs(ab[d + w * j]);
            if (j - d >= 0) r += fabs(ab[d + w * (j - d)]);
        }
        lo = std::min(lo, ab[w * j] - r);
        hi = std::max(hi, ab[w * j] + r);
    }
    // multi-section: P interior points per round, evaluated concurrently
    const int nt = pool.size(), P = std::max(1, nt - 1);
    std::vector<double> xs((size_t)P);
    std::vector<int> cnt((size_t)P);
    for (int it = 0; it < 200; ++it) {
        bool distinct = true;
        for (int i = 0; i < P; ++i) {
            xs[i] = lo + (hi - lo) * (double)(i + 1) / (double)(P + 1);
            if (xs[i] <= lo || xs[i] >= hi) distinct = false;
        }
        if (!distinct) {
            const double mid = 0.5 * (lo + hi);
            if (mid <= lo || mid >= hi) break;
            if (B.inertia_below(mid, wk[0]) > k) hi = mid; else lo = mid;
            continue;
        }
        pool.run(P, [&](int i, int t) { cnt[i] = B.inertia_below(xs[i], wk[t]); });
        double nlo = lo, nhi = hi;
        for (int i = 0; i < P; ++i) { if (cnt[i] > k) { nhi = xs[i]; break; } nlo = xs[i]; }
        lo = nlo; hi = nhi;
    }
    return 0.5 * (lo + hi);
}

double BandGcv::eig_kth(int64_t k) const {
    Pool pool(pick_threads(threads));
    std::vector<Work> wk((size_t)pool.size());
    return band_eig_kth(*this, k, pool, wk);
}

double BandGcv::find_lambda(int mode) const {
    Pool pool(pick_threads(threads));
    const int nt = pool.size();
    std::vector<Work> wk((size_t)nt);
    const double emax = band_eig_kth(*this, m - 1, pool, wk);
    const double emin = std::max(band_eig_kth(*this, 0, pool, wk), 1e-300);
    // gcv.Krig's bracket: l1 = emax * 4^k until trA < nt + .05, l2 = emin / 4^k until trA > .95 n;
    // the 2 x 20 candidates are evaluated concurrently, the first that qualifies is taken
    double tr1[20], tr2[20];
    bool ok1[20], ok2[20];
    pool.run(40, [&](int i, int t) {
        if (i < 20) ok1[i] = eval(emax * pow(4.0, i), nullptr, &tr1[i], nullptr, wk[t]);
        else ok2[i - 20] = eval(emin / pow(4.0, i - 20), nullptr, &tr2[i - 20], nullptr, wk[t]);
    });
    double l1 = emax, l2 = emin;
    for (int k = 0; k < 20; ++k) {
        if (!ok1[k]) return NAN;
        if (tr1[k] < 3.0 + 0.05) break;
        l1 *= 4.0;
    }
    for (int k = 0; k < 20; ++k) {
        if (!ok2[k]) break;  // not SPD at this tiny shift: stop shrinking
        if (tr2[k] > 0.95 * (double)n) break;
        l2 /= 4.0;
    }
    const int nstep = 200;
    std::vector<double> lamv(nstep), gcvv(nstep);
    const double la = log(l2), lb = log(l1);
    pool.run(nstep, [&](int i, int t) {
        lamv[i] = exp(la + (lb - la) * (double)i / (double)(nstep - 1));
        double gcv = NAN;
        if (!eval(lamv[i], &gcv, nullptr, nullptr, wk[t])) gcv = NAN;
        gcvv[i] = gcv;
    });
    std::vector<double> grid, gv;
    for (int i = 0; i < nstep; ++i) if (!std::isnan(gcvv[i])) { grid.push_back(lamv[i]); gv.push_back(gcvv[i]); }
    if (grid.empty()) return NAN;
    size_t il = 0;
    for (size_t i = 1; i < gv.size(); ++i) if (gv[i] < gv[il]) il = i;
    if (il == 0 || il + 1 == gv.size()) return grid[il];
    auto f = [&](double lam) { double gcv = NAN; if (!eval(lam, &gcv, nullptr, nullptr, wk[0])) return (double)NAN; return gcv; };
    if (mode == MHS_GCV_FIELDS) {  // golden.section.search, tol = 0.01 * GCVmin
        const double r = 0.61803399, con = 1.0 - r, tol = 0.01 * gv[il];
        const double ax = grid[il - 1], bx = grid[il], cx = grid[il + 1];
        double x0 = ax, x3 = cx, x1, x2;
        if (fabs(cx - bx) > fabs(bx - ax)) { x1 = bx; x2 = bx + con * (cx - bx); }
        else { x2 = bx; x1 = bx - con * (bx - ax); }
        double f1 = f(x1), f2 = f(x2);
        for (int k = 0; k < 25; ++k) {
            if (f2 < f1) { x0 = x1; x1 = x2; x2 = r * x1 + con * x3; f1 = f2; f2 = f(x2); }
            else { x3 = x2; x2 = x1; x1 = r * x2 + con * x0; f2 = f1; f1 = f(x1); }
            if (fabs(f2 - f1) < tol) break;
        }
        (void)x0; (void)x3;
        return f1 < f2 ? x1 : x2;
    }
    double lo = log(grid[il - 1]), hi = log(grid[il + 1]);
    const double r = 0.5 * (sqrt(5.0) - 1.0);
    double x1 = hi - r * (hi - lo), x2 = lo + r * (hi - lo), f1 = f(exp(x1)), f2 = f(exp(x2));
    for (int it = 0; it < 200; ++it) {
        if (f1 < f2) { hi = x2; x2 = x1; f2 = f1; x1 = hi - r * (hi - lo); f1 = f(exp(x1)); }
        else { lo = x1; x1 = x2; f1 = f2; x2 = lo + r * (hi - lo); f2 = f(exp(x2)); }
        if (fabs(hi - lo) < 1e-13) break;
    }
    return exp(0.5 * (lo + hi));
}

// ---- Householder QR of the n x 3 polynomial matrix (host, O(n)) -------------------------
void qr_n3(std::vector<double> &T /* n x 3 column-major, overwritten */, int64_t n,
           std::vector<double> v[3], double tau[3], double R[9]) {
    for (int i = 0; i < 9; ++i) R[i] = 0.0;
    for (int k = 0; k < 3; ++k) {
        double *x = &T[(size_t)k * n];
        const double alpha = x[k];
        double xn = 0.0;
        for (int64_t i = k + 1; i < n; ++i) xn += x[i] * x[i];
        xn = sqrt(xn);
        v[k].assign((size_t)n, 0.0);
        if (xn == 0.0) {
            tau[k] = 0.0;
            v[k][k] = 1.0;
            R[k + 3 * k] = alpha;
        } else {
            const double beta = -copysign(hypot(alpha, xn), alpha);
            tau[k] = (beta - alpha) / beta;
            const double s = 1.0 / (alpha - beta);
            v[k][k] = 1.0;
            for (int64_t i = k + 1; i < n; ++i) v[k][i] = x[i] * s;
            R[k + 3 * k] = beta;
        }
        for (int j = 0; j < k; ++j) R[j + 3 * k] = x[j];  // rows above the diagonal are final
        // apply H_k to the remaining columns
        for (int j = k + 1; j < 3; ++j) {
            double *y = &T[(size_t)j * n];
            double s = 0.0;
            for (int64_t i = k; i < n; ++i) s += v[k][i] * y[i];
            s *= tau[k];
            for (int64_t i = k; i < n; ++i) y[i] -= s * v[k][i];
        }
    }
}

void apply_reflector(const std::vector<double> &v, double tau, double *x, int64_t n) {
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += v[i] * x[i];
    s *= tau;
    for (int64_t i = 0; i < n; ++i) x[i] -= s * v[i];
}

}  // namespace mhs

using namespace mhs;

extern "C" int mhs_host_gcv_tridiag(const double *diag, const double *offdiag, const double *g,
                                    int64_t m, int64_t n_unique, int64_t n_obs, double pure_ss,
                                    double lambda, int gcv_mode, double *lambda_out,
                                    double *gcv_out, double *eff_df_out, double *q_out) {
    MHS_REQUIRE(diag && offdiag && g && m >= 1, "NULL or empty input");
    MHS_REQUIRE(n_unique == m + 3 && n_obs >= n_unique, "n_unique must equal m + 3 and n_obs >= n_unique");
    TridiagGcv t;
    t.a = diag; t.b = offdiag; t.g = g; t.m = m; t.n = n_unique; t.N = n_obs; t.pure_ss = pure_ss;
    double lam = lambda;
    if (std::isnan(lam)) lam = t.find_lambda(gcv_mode);
    if (std::isnan(lam) || lam < 0) { set_error("GCV search failed"); return MHS_ERR_NUMERIC; }
    double gcv, tra;
    t.eval(lam, &gcv, &tra, q_out);
    if (lambda_out) *lambda_out = lam;
    if (gcv_out) *gcv_out = gcv;
    if (eff_df_out) *eff_df_out = tra;
    return MHS_OK;
}

extern "C" int mhs_host_gcv_band(const double *ab, int bw, const double *g, int64_t m, int64_t n_unique,
                                 int64_t n_obs, double pure_ss, double lambda, int gcv_mode,
                                 double *lambda_out, double *gcv_out, double *eff_df_out, double *q_out) {
    MHS_REQUIRE(ab && g && m >= 1 && bw >= 1 && bw < m + 1, "NULL or empty input");
    MHS_REQUIRE(n_unique == m + 3 && n_obs >= n_unique, "n_unique must equal m + 3 and n_obs >= n_unique");
    BandGcv t;
    t.ab = ab; t.g = g; t.m = m; t.n = n_unique; t.N = n_obs; t.bw = bw; t.pure_ss = pure_ss;
    double lam = lambda;
    if (std::isnan(lam)) lam = t.find_lambda(gcv_mode);
    if (std::isnan(lam) || lam < 0) { set_error("GCV search failed"); return MHS_ERR_NUMERIC; }
    double gcv, tra;
    BandGcv::Work wk;
    if (!t.eval(lam, &gcv, &tra, q_out, wk)) { set_error("band matrix + lambda I is not positive definite"); return MHS_ERR_NUMERIC; }
    if (lambda_out) *lambda_out = lam;
    if (gcv_out) *gcv_out = gcv;
    if (eff_df_out) *eff_df_out = tra;
    return MHS_OK;
}
