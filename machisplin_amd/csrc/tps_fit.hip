#include "common.h"
extern "C" MHS_API int mhs_tps_fit(const double *xy, const double *y, int64_t N, double lambda, int gcv_mode, mhs_tps **out) {
    mhs::set_error("mhs_tps_fit: not implemented yet");
    return MHS_ERR_INVALID;
}
