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
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
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
// ---- fixed bandwidth fast path (the GPU reduction always hands over bandwidth 8) -------------------------------
// Every trip count is a compile-time constant and nothing tests for the end of the matrix: the factor lives in
// m + B zero-padded columns, so the updates of the last columns fall into the padding.  LDL' instead of LL'
// takes the square root off the column-to-column dependency chain (divide, multiply, fma: the chain IS the
// run time of a band this narrow); the Takahashi window is the dense one described below.
namespace {
// in-place LDL' of the band in L (column j: d_j, then the B unit-lower entries); returns the number of
// negative pivots, or -1 at the first non-positive pivot when require_pd
template <int B>
int ldl_fixed(double *L, int64_t m, bool require_pd) {
    constexpr int W = B + 1;
    int neg = 0;
    for (int64_t j = 0; j < m; ++j) {
        double *c = L + W * j;
        double d = c[0];
        if (require_pd) { if (!(d > 0.0)) return -1; }
        else { if (d == 0.0) d = -1e-300; if (d < 0.0) ++neg; }
        const double inv = 1.0 / d;
        double a[B], l[B];
        for (int i = 0; i < B; ++i) { a[i] = c[i + 1]; l[i] = a[i] * inv; }
        for (int k = 1; k <= B; ++k) {
            double *ck = c + W * k;
            const double lk = l[k - 1];
            for (int dd = 0; dd <= B - k; ++dd) ck[dd] -= a[k - 1 + dd] * lk;
        }
        c[0] = d;
        for (int i = 0; i < B; ++i) c[i + 1] = l[i];
    }
    return neg;
}
template <int B>
void load_band(std::vector<double> &L, const double *ab, int64_t m, double shift) {
    constexpr int W = B + 1;
    L.resize((size_t)(W * (m + B)));
    for (int64_t j = 0; j < m; ++j) {
        L[W * j] = ab[W * j] + shift;
        for (int i = 1; i <= B; ++i) L[i + W * j] = j + i < m ? ab[i + W * j] : 0.0;
    }
    for (int64_t e = W * m; e < W * (m + B); ++e) L[e] = 0.0;
}
template <int B>
bool eval_fixed(const BandGcv &G, double lam, double *qq_out, double *tr_inv_out, double *q_out, BandGcv::Work &wk) {
    constexpr int W = B + 1;
    const int64_t m = G.m;
    std::vector<double> &Lv = wk.L, &qv = wk.q;
    load_band<B>(Lv, G.ab, m, lam);
    double *L = Lv.data();
    if (ldl_fixed<B>(L, m, true) < 0) return false;
    // L D L' q = g
    qv.assign((size_t)(m + B), 0.0);
    double *q = qv.data();
    for (int64_t i = 0; i < m; ++i) q[i] = G.g[i];
    for (int64_t j = 0; j < m; ++j) {
        const double qj = q[j];
        const double *c = L + W * j;
        for (int i = 1; i <= B; ++i) q[j + i] -= c[i] * qj;
    }
    for (int64_t e = m; e < m + B; ++e) q[e] = 0.0;   // what the last columns pushed into the padding
    double qq = 0.0;
    for (int64_t j = m - 1; j >= 0; --j) {
        const double *c = L + W * j;
        double s0 = q[j] / c[0], s1 = 0.0;
        for (int i = 1; i <= B; i += 2) { s0 -= c[i] * q[j + i]; if (i + 1 <= B) s1 -= c[i + 1] * q[j + i + 1]; }
        q[j] = s0 + s1;
        qq += q[j] * q[j];
    }
    // trace of the inverse (Takahashi), dense sliding window: see the general version below
    double Zc[2 * B + 1][2 * B + 1];
    for (int a = 0; a < 2 * B + 1; ++a) for (int b = 0; b < 2 * B + 1; ++b) Zc[a][b] = 0.0;
    double tr_inv = 0.0;
    int o = B + 1;
    for (int64_t j = m - 1; j >= 0; --j) {
        const double *c = L + W * j;
        double lt[B], y[B];
        for (int k = 0; k < B; ++k) lt[k] = c[k + 1];
        double dot = 0.0;
        for (int a = 0; a < B; ++a) {
            const double *row = &Zc[o + a][o];
            double sum = 0.0;
            for (int b = 0; b < B; ++b) sum += row[b] * lt[b];
            y[a] = sum;
            dot += lt[a] * sum;
        }
        const double zjj = 1.0 / c[0] + dot;
        tr_inv += zjj;
        if (o == 0) {
            for (int a = B - 2; a >= 0; --a) for (int b = B - 2; b >= 0; --b) Zc[B + 2 + a][B + 2 + b] = Zc[a][b];
            o = B + 2;
        }
        --o;
        Zc[o][o] = zjj;
        for (int b = 0; b + 1 < B; ++b) { Zc[o][o + 1 + b] = -y[b]; Zc[o + 1 + b][o] = -y[b]; }
    }
    *qq_out = qq; *tr_inv_out = tr_inv;
    if (q_out) std::copy(q, q + m, q_out);
    return true;
}
}  // namespace

bool BandGcv::eval(double lam, double *gcv, double *tra, double *q_out, Work &wk) const {
    const int64_t w = bw + 1;
    if (bw == 8) {
        double qq, tr_inv;
        if (!eval_fixed<8>(*this, lam, &qq, &tr_inv, q_out, wk)) return false;
        return finish(lam, qq, tr_inv, gcv, tra);
    }
    std::vector<double> &L = wk.L, &q = wk.q;
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
    // trace of the inverse: Takahashi recurrence on M = Lt D Lt', Lt = L diag(1/L_jj) unit lower, D_j = L_jj^2.
    // Column j of Z = M^-1 needs only the bw x bw block of Z on rows/columns j+1..j+bw, which lies inside the
    // band: it is kept as a dense symmetric window (a square that slides up the diagonal of a 2 bw scratch array
    // and is moved back every bw columns), so a column is one small matrix-vector product with no index tests.
    double tr_inv = 0.0;
    if (bw <= 16) {
        constexpr int MAXB = 16;
        double Zc[2 * MAXB + 1][2 * MAXB + 1];
        for (int a = 0; a < 2 * MAXB + 1; ++a) for (int b = 0; b < 2 * MAXB + 1; ++b) Zc[a][b] = 0.0;
        double lt[MAXB], y[MAXB];
        const int B = bw;
        int o = B + 1;   // window = Zc[o .. o+B-1][o .. o+B-1] <-> rows/columns j+1 .. j+B
        for (int64_t j = m - 1; j >= 0; --j) {
            const int64_t kmax = std::min<int64_t>(bw, m - 1 - j);
            const double inv = 1.0 / L[w * j];
            for (int k = 0; k < B; ++k) lt[k] = k < kmax ? L[(k + 1) + w * j] * inv : 0.0;
            double dot = 0.0;
            for (int a = 0; a < B; ++a) {
                const double *row = &Zc[o + a][o];
                double sum = 0.0;
                for (int b = 0; b < B; ++b) sum += row[b] * lt[b];
                y[a] = sum;
                dot += lt[a] * sum;
            }
            const double zjj = inv * inv + dot;
            tr_inv += zjj;
            if (o == 0) {   // move the window (minus its last row and column, which slide out) back down
                for (int a = B - 2; a >= 0; --a) for (int b = B - 2; b >= 0; --b) Zc[B + 2 + a][B + 2 + b] = Zc[a][b];
                o = B + 2;
            }
            --o;
            Zc[o][o] = zjj;
            for (int b = 0; b + 1 < B; ++b) { Zc[o][o + 1 + b] = -y[b]; Zc[o + 1 + b][o] = -y[b]; }
        }
    } else {
        std::vector<double> &Z = wk.Z;
        Z.assign(w * m, 0.0);
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
    }
    if (q_out) std::copy(q.begin(), q.end(), q_out);
    return finish(lam, qq, tr_inv, gcv, tra);
}

bool BandGcv::finish(double lam, double qq, double tr_inv, double *gcv, double *tra) const {
    const double rss = lam * lam * qq;
    const double tr = 3.0 + (double)m - lam * tr_inv;
    double mse = rss / (double)n;
    if (N - n > 0) mse += pure_ss / (double)(N - n);
    const double den = 1.0 - tr / (double)n;
    if (gcv) *gcv = den > 0 ? mse / (den * den) : NAN;
    if (tra) *tra = tr;
    return true;
}

int BandGcv::inertia_below(double x, Work &wk) const {
    const int64_t w = bw + 1;
    std::vector<double> &L = wk.L;
    if (bw == 8) {
        load_band<8>(L, ab, m, -x);
        return ldl_fixed<8>(L.data(), m, false);
    }
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
// Worker pool for the independent GCV evaluations (no R / HIP API is touched from the workers).  A search is
// 30-40 rounds of 0.1-0.4 ms, so what matters is the latency of a round: between begin() and end() the workers
// SPIN on a ticket word (a condition-variable wake-up costs as much as a round); outside a search they sleep.
// A round ends when its ITEMS are done, not when every worker has reported: a worker that is late (still
// waking up, descheduled) simply finds the ticket of a later round.  The ticket is (round << 32 | next item);
// an item is claimed by a compare-and-swap on the whole word, so a claim can only succeed while that round is
// still open, and the round's job fields are rewritten only after all of its items are done.
class Pool {
public:
    explicit Pool(int n) : n_(std::max(1, n)) {
        for (int t = 1; t < n_; ++t) workers_.emplace_back([this, t]() { loop(t); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_.store(true, std::memory_order_release); }
        cv_.notify_all();
        for (auto &th : workers_) th.join();
    }
    int size() const { return n_; }
    void begin() {
        { std::lock_guard<std::mutex> lk(mu_); active_.store(true, std::memory_order_release); }
        cv_.notify_all();
    }
    void end() { active_.store(false, std::memory_order_release); }
    template <typename F>
    void run(int count, F f) {  // f(index, thread id); between begin() and end()
        if (n_ == 1 || count <= 1) { for (int i = 0; i < count; ++i) f(i, 0); return; }
        fn_ = [](void *c, int i, int t) { (*static_cast<F *>(c))(i, t); };
        ctx_ = &f;
        count_ = count;
        done_.store(0, std::memory_order_relaxed);
        const uint64_t round = (ticket_.load(std::memory_order_relaxed) >> 32) + 1;
        ticket_.store(round << 32, std::memory_order_release);
        work(round, 0);
        while (done_.load(std::memory_order_acquire) < count) __builtin_ia32_pause();
    }
private:
    void work(uint64_t round, int t) {
        for (;;) {
            uint64_t v = ticket_.load(std::memory_order_acquire);
            if ((v >> 32) != round) return;
            const int i = (int)(v & 0xffffffffu);
            if (i >= count_) return;
            if (!ticket_.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel)) continue;
            fn_(ctx_, i, t);
            done_.fetch_add(1, std::memory_order_release);
        }
    }
    void loop(int t) {
        uint64_t seen = 0;
        unsigned idle = 0;
        for (;;) {
            { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return stop_.load() || active_.load(); }); if (stop_.load()) return; }
            while (active_.load(std::memory_order_acquire) && !stop_.load(std::memory_order_acquire)) {
                const uint64_t round = ticket_.load(std::memory_order_acquire) >> 32;
                if (round == seen) {
                    // mostly spin, but yield now and then: a freshly woken worker may have been placed on the
                    // core of the thread that drives the search
                    if ((++idle & 63) == 0) std::this_thread::yield(); else __builtin_ia32_pause();
                    continue;
                }
                seen = round; idle = 0;
                work(round, t);
            }
        }
    }
    int n_;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<bool> active_{false};
    std::atomic<bool> stop_{false};
    void (*fn_)(void *, int, int) = nullptr;
    void *ctx_ = nullptr;
    int count_ = 0;
    std::atomic<uint64_t> ticket_{0};
    std::atomic<int> done_{0};
};
}  // namespace
// CPUs this process may keep busy: the hardware threads, capped by the cgroup CPU quota (cpu.max of cgroup v2,
// cfs_quota_us / cfs_period_us of v1).  Spinning workers beyond the quota get the whole cgroup throttled --
// including the thread that feeds the GPU.
int cpu_budget() {
    const unsigned hc = std::thread::hardware_concurrency();
    int n = (int)std::max(1u, hc ? hc : 1u);
    long long quota = -1, period = 100000;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if (FILE *f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(f1, "%lld", &quota) != 1) quota = -1;
        fclose(f1);
        if (FILE *f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(f2, "%lld", &period) != 1) period = 100000;
            fclose(f2);
        }
    }
    if (quota > 0 && period > 0) n = (int)std::min<long long>(n, std::max<long long>(1, quota / period));
    return n;
}
namespace {
int auto_threads() {
    if (const char *e = getenv("MHS_GCV_THREADS")) { const int v = atoi(e); if (v > 0) return std::min(v, 256); }
    return std::min(64, cpu_budget());
}
std::mutex g_shared_mu;
Pool &shared_pool() { static Pool pool(auto_threads()); return pool; }
}  // namespace

// the pool of one search: the process-wide one if it is free and no thread count was requested
struct GcvPool {
    std::unique_lock<std::mutex> lock;
    std::unique_ptr<Pool> own;
    Pool *pool;
    explicit GcvPool(int requested) : lock(g_shared_mu, std::defer_lock) {
        if (requested <= 0 && lock.try_lock()) pool = &shared_pool();
        else { own.reset(new Pool(requested > 0 ? requested : std::min(16, auto_threads()))); pool = own.get(); }
        pool->begin();
    }
    ~GcvPool() { pool->end(); }
};
GcvPool *gcv_pool_lease(int threads) { return new GcvPool(threads); }
void gcv_pool_release(GcvPool *p) { delete p; }
namespace {
struct PoolLease {   // the caller's lease, or one for the duration of this search
    GcvPool *mine = nullptr;
    Pool *pool;
    PoolLease(GcvPool *given, int threads) { if (!given) mine = given = gcv_pool_lease(threads); pool = given->pool; }
    ~PoolLease() { gcv_pool_release(mine); }
};
}  // namespace

static void gershgorin(const BandGcv &B, double *lo_out, double *hi_out) {
    const double *ab = B.ab; const int bw = B.bw; const int64_t m = B.m, w = bw + 1;
    double lo = ab[0], hi = ab[0];
    for (int64_t j = 0; j < m; ++j) {
        double r = 0.0;
        for (int64_t d = 1; d <= bw; ++d) {
            if (j + d < m) r += fabs(ab[d + w * j]);
            if (j - d >= 0) r += fabs(ab[d + w * (j - d)]);
        }
        lo = std::min(lo, ab[w * j] - r);
        hi = std::max(hi, ab[w * j] + r);
    }
    *lo_out = lo; *hi_out = hi;
}

// Eigenvalues number k[0..nk-1] (0-based, ascending) by multi-section on the inertia count: every round the pool
// evaluates its points shared out over the nk brackets at once, each bracket shrinking by (points + 1) per round,
// down to a relative width of 1e-13 (the eigenvalues only place the ends of the lambda grid).
static void band_eig_multi(const BandGcv &B, const int64_t *k, int nk, Pool &pool, std::vector<BandGcv::Work> &wk,
                           double *out) {
    double glo, ghi;
    gershgorin(B, &glo, &ghi);
    std::vector<double> lo((size_t)nk, glo), hi((size_t)nk, ghi);
    std::vector<char> done((size_t)nk, 0);
    // Points per bracket per round: a CONSTANT, not a function of the pool size -- the sequence of brackets, hence
    // the last bits of the eigenvalue, the ends of the lambda grid and finally lambda itself must not depend on how
    // many host threads happen to serve the search (the same fit on a lane of mhs_tps_surface, on another box or
    // under another CPU quota has to give the same bits).
    const int P = 8;
    std::vector<double> xs((size_t)(P * nk));
    std::vector<int> cnt((size_t)(P * nk));
    for (int it = 0; it < 400; ++it) {
        int live = 0;
        for (int e = 0; e < nk; ++e) {
            if (done[e]) continue;
            if (hi[e] - lo[e] <= 1e-13 * std::max(fabs(lo[e]), fabs(hi[e]))) { done[e] = 1; continue; }
            bool distinct = true;
            for (int i = 0; i < P; ++i) {
                xs[e * P + i] = lo[e] + (hi[e] - lo[e]) * (double)(i + 1) / (double)(P + 1);
                if (xs[e * P + i] <= lo[e] || xs[e * P + i] >= hi[e]) distinct = false;
            }
            if (!distinct) {   // down to neighbouring doubles: plain bisection until it stalls
                const double mid = 0.5 * (lo[e] + hi[e]);
                if (mid <= lo[e] || mid >= hi[e]) { done[e] = 1; continue; }
                for (int i = 0; i < P; ++i) xs[e * P + i] = mid;
            }
            ++live;
        }
        if (!live) break;
        pool.run(P * nk, [&](int i, int t) { if (!done[i / P]) cnt[i] = B.inertia_below(xs[i], wk[t]); });
        for (int e = 0; e < nk; ++e) {
            if (done[e]) continue;
            double nlo = lo[e], nhi = hi[e];
            for (int i = 0; i < P; ++i) { if (cnt[e * P + i] > k[e]) { nhi = xs[e * P + i]; break; } nlo = xs[e * P + i]; }
            lo[e] = nlo; hi[e] = nhi;
        }
    }
    for (int e = 0; e < nk; ++e) out[e] = 0.5 * (lo[e] + hi[e]);
}

double BandGcv::eig_kth(int64_t k) const {
    PoolLease lease(pool, threads);
    std::vector<Work> wk((size_t)lease.pool->size());
    double v;
    band_eig_multi(*this, &k, 1, *lease.pool, wk, &v);
    return v;
}

double BandGcv::find_lambda(int mode) const {
    const bool timing = getenv("MHS_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[gcv m=%lld] %-24s %8.3f ms\n", (long long)m, what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    PoolLease lease(this->pool, threads);
    Pool &pool = *lease.pool;
    const int nt = pool.size();
    std::vector<Work> wk((size_t)nt);
    const int64_t ends[2] = {m - 1, 0};
    double ev[2];
    band_eig_multi(*this, ends, 2, pool, wk, ev);
    const double emax = ev[0], emin = std::max(ev[1], 1e-300);
    lap("extreme eigenvalues");
    // gcv.Krig's bracket: l1 = emax * 4^k until trA < nt + .05, l2 = emin / 4^k until trA > .95 n;
    // the 2 x 20 candidates are evaluated concurrently, the first that qualifies is taken
    double tr1[20], tr2[20];
    bool ok1[20], ok2[20];
    pool.run(40, [&](int i, int t) {
        if (i < 20) ok1[i] = eval(emax * pow(4.0, i), nullptr, &tr1[i], nullptr, wk[t]);
        else ok2[i - 20] = eval(emin / pow(4.0, i - 20), nullptr, &tr2[i - 20], nullptr, wk[t]);
    });
    lap("bracket (40 evals)");
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
    lap("grid (200 evals)");
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
        lap("golden section");
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
