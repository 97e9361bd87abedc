// Ensemble kernels for the M <= 64 register bucket (runtime M, +inf padding).
#include "wbx_ens_atoms.hpp"
#include "wbx_ens_impl.hpp"
namespace wbx {
int launch_ens_m64(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a, int algo, bool map) {
  return launch_ens_bucket<64, false>(ctx, plan, a, algo, map);
}
int launch_ens_atoms_m64(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a, const EnsBinnedCall& c) {
  return launch_ens_atoms<64, false>(ctx, plan, a, c);
}
}  // namespace wbx
