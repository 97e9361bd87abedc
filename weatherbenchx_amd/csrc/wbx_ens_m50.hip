// Ensemble kernels specialised for M == 50 (IFS ENS: 50 perturbed members, 51 with the control).
#include "wbx_ens_atoms.hpp"
#include "wbx_ens_impl.hpp"
namespace wbx {
int launch_ens_m50(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a, int algo, bool map) {
  return launch_ens_bucket<50, true>(ctx, plan, a, algo, map);
}
int launch_ens_atoms_m50(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a, const EnsBinnedCall& c) {
  return launch_ens_atoms<50, true>(ctx, plan, a, c);
}
}  // namespace wbx
