// TEMPORARY stub
#include "common.hpp"
extern "C" int sfmhip_estimate_normals(sfmhip_ctx*, const double*, int, int, double*) { return SFMHIP_E_ARG; }
