// TEMPORARY stub while the BA kernels are being written
#include "common.hpp"
extern "C" {
void sfmhip_ba_default_options(sfm_ba_options* o) { memset(o, 0, sizeof *o); }
int sfmhip_ba_solve(sfmhip_ctx*, double*, double*, int, double*, int, const int32_t*, const int32_t*, const double*, int, const sfm_ba_options*, sfm_ba_summary*) { return SFMHIP_E_ARG; }
int sfmhip_ba_create(sfmhip_ctx*, const double*, const double*, int, const double*, int, const int32_t*, const int32_t*, const double*, int, const sfm_ba_options*, sfmhip_ba**) { return SFMHIP_E_ARG; }
void sfmhip_ba_destroy(sfmhip_ba*) {}
int sfmhip_ba_set_allreduce(sfmhip_ba*, sfmhip_allreduce_fn, void*) { return SFMHIP_E_ARG; }
int sfmhip_ba_run(sfmhip_ba*, sfm_ba_summary*) { return SFMHIP_E_ARG; }
int sfmhip_ba_iterate(sfmhip_ba*, int, sfm_ba_summary*) { return SFMHIP_E_ARG; }
int sfmhip_ba_reset(sfmhip_ba*) { return SFMHIP_E_ARG; }
int sfmhip_ba_get_params(sfmhip_ba*, double*, double*, double*) { return SFMHIP_E_ARG; }
int sfmhip_ba_reduced_system(sfmhip_ba*, double, double*, double*, int*, double*) { return SFMHIP_E_ARG; }
int sfmhip_ba_phase_ms(sfmhip_ba*, double*) { return SFMHIP_E_ARG; }
}
