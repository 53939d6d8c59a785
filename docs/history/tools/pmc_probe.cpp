// torch-free driver of the C ABI for PMC collection (rocprofv3 --pmc crashes under the torch-hosted bench)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "machisplin_hip.h"
#define CK(x) do { int rc_ = (x); if (rc_) { printf("fail %d: %s\n", rc_, mhs_last_error()); return 1; } } while (0)
int main(int argc, char **argv) {
    const int n = 5000, side = argc > 1 ? atoi(argv[1]) : 4000;
    CK(mhs_init(0));
    std::vector<double> kn(2 * n), c(n);
    srand(1);
    for (int i = 0; i < n; ++i) { kn[i] = rand() / (double)RAND_MAX; kn[n + i] = rand() / (double)RAND_MAX; c[i] = rand() / (double)RAND_MAX - 0.5; }
    double d3[3] = {0.1, 0.2, 0.3}, ce[2] = {-78.0, -5.0 - side / 1200.0}, sc[2] = {side / 1200.0, side / 1200.0};
    mhs_tps *t = nullptr;
    CK(mhs_tps_from_coef(kn.data(), c.data(), d3, n, 1e-3, ce, sc, &t));
    mhs_grid g = {-78.0, -5.0, 1.0 / 1200, 1.0 / 1200, side, side};
    double *out = nullptr;
    if (hipMalloc((void **)&out, sizeof(double) * (size_t)side * side) != hipSuccess) return 2;
    for (int rep = 0; rep < 2; ++rep) CK(mhs_tps_predict_grid_dev(t, &g, 0, side, 0, side, out, side, nullptr));
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    // lm member over 3 float planes: an HBM-bound kernel for calibration
    float *planes = nullptr;
    if (hipMalloc((void **)&planes, sizeof(float) * 3 * (size_t)side * side) != hipSuccess) return 4;
    (void)hipMemset(planes, 0, sizeof(float) * 3 * (size_t)side * side);
    double coef[6] = {1, 2, 3, 4, 5, 6};
    mhs_model *lm = nullptr;
    CK(mhs_lm_load(coef, 5, &lm));
    mhs_stack st = {planes, 3, MHS_F32, (int64_t)side * side, side, NAN};
    for (int rep = 0; rep < 2; ++rep) CK(mhs_predict_dev(lm, &g, &st, 0, side, 0, side, 1.0, 0, out, side, nullptr));
    if (hipDeviceSynchronize() != hipSuccess) return 5;
    {   // gbm: 10000 random 5-split trees (gbm layout: 16 nodes, children of split s at 1+3s, 2+3s, 3+3s)
        const int nt = 10000, npt = 16, p = 5;
        std::vector<int64_t> off(nt + 1);
        std::vector<int32_t> var(nt * npt, -1), left(nt * npt, 0), right(nt * npt, 0), miss(nt * npt, 0);
        std::vector<double> val(nt * npt);
        for (int t = 0; t <= nt; ++t) off[t] = (int64_t)t * npt;
        for (int t = 0; t < nt; ++t) {
            int leaves[16], nl = 1; leaves[0] = 0;
            for (int k = 0; k < npt; ++k) val[t * npt + k] = (rand() / (double)RAND_MAX - 0.5) * 1e-3;
            for (int sidx = 0; sidx < 5; ++sidx) {
                const int pick = rand() % nl, node = leaves[pick], v = rand() % p;
                var[t * npt + node] = v;
                val[t * npt + node] = v < 3 ? rand() / (double)RAND_MAX : (v == 3 ? -78.0 + (rand() / (double)RAND_MAX) * side / 1200.0 : -5.0 - (rand() / (double)RAND_MAX) * side / 1200.0);
                left[t * npt + node] = 1 + 3 * sidx; right[t * npt + node] = 2 + 3 * sidx; miss[t * npt + node] = 3 + 3 * sidx;
                leaves[pick] = 1 + 3 * sidx; leaves[nl++] = 2 + 3 * sidx;
            }
        }
        mhs_model *gb = nullptr;
        CK(mhs_gbm_load(0.0, nt, off.data(), var.data(), val.data(), left.data(), right.data(), miss.data(), p, &gb));
        std::vector<float> hp((size_t)3 * side * side);
        for (auto &x : hp) x = rand() / (float)RAND_MAX;
        (void)hipMemcpy(planes, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) CK(mhs_predict_dev(gb, &g, &st, 0, side, 0, side, 1.0, 1, out, side, nullptr));
        if (hipDeviceSynchronize() != hipSuccess) return 6;
        // ksvm: 3000 support vectors
        const int nsv = 3000;
        std::vector<double> alpha(nsv), sv((size_t)nsv * p), xc(p, 0.0), xs(p, 1.0);
        for (int i = 0; i < nsv; ++i) { alpha[i] = rand() / (double)RAND_MAX - 0.5; for (int j = 0; j < p; ++j) sv[(size_t)i * p + j] = rand() / (double)RAND_MAX; }
        xc[3] = -78.0; xc[4] = -5.0 - side / 1200.0; xs[3] = xs[4] = side / 1200.0;
        mhs_model *sm = nullptr;
        CK(mhs_svr_load(alpha.data(), sv.data(), nsv, p, 0.1, 0.3, xc.data(), xs.data(), 0.0, 1.0, &sm));
        for (int rep = 0; rep < 2; ++rep) CK(mhs_predict_dev(sm, &g, &st, 0, side, 0, side, 1.0, 1, out, side, nullptr));
        if (hipDeviceSynchronize() != hipSuccess) return 7;
    }
    {   // randomForest: 50 complete depth-11 trees in randomForest's node layout (daughters 2k+1, 2k+2, 1-based ids)
        const int nt = 50, depth = 11, nn = (1 << (depth + 1)) - 1, p = 5;
        std::vector<int64_t> off(nt + 1);
        std::vector<int32_t> left((size_t)nt * nn), right((size_t)nt * nn), status((size_t)nt * nn), bvar((size_t)nt * nn);
        std::vector<double> split((size_t)nt * nn), pred((size_t)nt * nn);
        for (int t = 0; t <= nt; ++t) off[t] = (int64_t)t * nn;
        for (int t = 0; t < nt; ++t)
            for (int k = 0; k < nn; ++k) {
                const size_t i = (size_t)t * nn + k;
                const bool leaf = k >= (1 << depth) - 1;
                const int v = rand() % p;
                left[i] = leaf ? 0 : 2 * k + 2; right[i] = leaf ? 0 : 2 * k + 3; status[i] = leaf ? -1 : -3; bvar[i] = leaf ? 0 : v + 1;
                split[i] = leaf ? 0.0 : (v < 3 ? rand() / (double)RAND_MAX : (v == 3 ? -78.0 + (rand() / (double)RAND_MAX) * side / 1200.0 : -5.0 - (rand() / (double)RAND_MAX) * side / 1200.0));
                pred[i] = rand() / (double)RAND_MAX;
            }
        mhs_model *rf = nullptr;
        CK(mhs_rf_load(nt, off.data(), left.data(), right.data(), status.data(), bvar.data(), split.data(), pred.data(), p, &rf));
        mhs_stack st2 = {planes, 3, MHS_F32, (int64_t)side * side, side, NAN};
        for (int rep = 0; rep < 2; ++rep) CK(mhs_predict_dev(rf, &g, &st2, 0, side, 0, side, 1.0, 1, out, side, nullptr));
        if (hipDeviceSynchronize() != hipSuccess) return 8;
    }
    printf("ok side=%d cells=%lld\n", side, (long long)side * side);
    return 0;
}
