// Zonal energy spectrum (placeholder TU; the rocFFT implementation replaces this file).
#include "wbx_common.hpp"
namespace wbx {
void spectrum_release(wbx_ctx*) {}
}  // namespace wbx
extern "C" int wbx_zonal_spectrum(wbx_ctx*, const float*, const int64_t*, const int32_t*, const double*, int64_t,
                                  int32_t, int32_t, double*) {
  return wbx::fail(WBX_ERR_INVALID, "wbx_zonal_spectrum: not built yet");
}
