// Ensemble kernels specialised for M == 51 (IFS ENS: 50 perturbed members, 51 with the control).
#include "wbx_ens_atoms.hpp"
#include "wbx_ens_impl.hpp"
namespace wbx {
int launch_ens_m51(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a, int algo, bool map) {
  return launch_ens_bucket<51, true>(ctx, plan, a, algo, map);
}
int launch_ens_atoms_m51(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a, const EnsBinnedCall& c) {
  return launch_ens_atoms<51, true>(ctx, plan, a, c);
}
}  // namespace wbx
