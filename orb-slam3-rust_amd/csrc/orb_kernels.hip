// placeholder — replaced by the extractor kernels
#include "orbx_internal.hpp"
int orb_prepare_geometry(orbx_handle* h, int, int) { return orbx_fail(h, ORBX_ERR_INVALID, "extractor not built yet"); }
int launch_orb_extract(orbx_handle* h, const uint8_t*, int, int, int, size_t, orbx_keypoint*, uint8_t*, int*, int) {
  return orbx_fail(h, ORBX_ERR_INVALID, "extractor not built yet");
}
