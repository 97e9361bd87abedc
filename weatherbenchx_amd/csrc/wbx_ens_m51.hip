// Ensemble kernels specialised for M == 51 (IFS ENS: 50 perturbed members, 51 with the control).
#include "wbx_ens_impl.hpp"
namespace wbx {
int launch_ens_m51(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a, int algo, bool map) {
  return launch_ens_bucket<51, true>(ctx, plan, a, algo, map);
}
}  // namespace wbx
