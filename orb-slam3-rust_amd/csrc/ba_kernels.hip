// placeholder — replaced by the BA kernels
#include "orbx_internal.hpp"
int ba_solve_visual(orbx_handle* h, const orbx_camera*, const orbx_ba_config*, int, const double*, int,
                    const double*, int, double*, int, const orbx_ba_obs*, orbx_should_stop_fn, void*,
                    double*, int*, double*, double*) {
  return orbx_fail(h, ORBX_ERR_INVALID, "BA not built yet");
}
