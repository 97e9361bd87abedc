// Ensemble entry points + the generic (any M, fp32/fp64) ensemble op.
// The register-bucket fast paths live in wbx_ens_m*.hip (see wbx_ens_impl.hpp).
#include "wbx_ens_atoms.hpp"
#include "wbx_ens_impl.hpp"

namespace wbx {

static int ens_common(wbx_ctx* ctx, const wbx_s1_plan* plan, int dtype, int M, int64_t member_stride, int algo,
                      const void* p, const void* t, const uint8_t* mask, double* out, bool map, int lane) {
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  if (int rc = check_plan(plan)) return rc;
  WBX_REQUIRE(plan->vec == 1, "ensemble kernels use vec=1");
  if (plan->flags & WBX_FLAG_MASKED) WBX_REQUIRE(mask != nullptr, "WBX_FLAG_MASKED set but mask is NULL");
  WBX_REQUIRE(M >= 1, "ensemble size must be >= 1 (got %d)", M);
  WBX_REQUIRE(algo == WBX_ENS_SORT || algo == WBX_ENS_PAIRWISE || (algo == WBX_ENS_DIAG_LOADONLY && (M == 50 || M == 51)) ||
                  (algo == WBX_ENS_DIAG_PAIRWISE_LDS && M == 51),
              "unknown ensemble algorithm %d", algo);
  const bool empty = plan->nkey * plan->ndepth * plan->nx == 0;
  WBX_REQUIRE(empty || (p != nullptr && t != nullptr), "predictions/targets pointer is NULL");
  WBX_REQUIRE(out != nullptr || plan->nkey == 0, "output pointer is NULL");
  WBX_HIP(hipSetDevice(ctx->device));
  S1Args a;
  fill_args(plan, a);
  a.in[0] = p;
  a.in[1] = t;
  a.in[3] = mask;
  a.out = out;
  a.M = M;
  a.mstride = member_stride;
  a.lane = lane;
  if (dtype == WBX_F64) return launch_ens_op<EnsOpGeneric<double>>(ctx, plan, a, map);
  WBX_REQUIRE(dtype == WBX_F32, "unknown dtype %d", dtype);
  if (plan->flags & WBX_FLAG_SKIPNA_ENS) {
    // per-point ensemble sizes: the register-resident rank form over the valid members (WBX_ENS_SKIPNA_SORT, wbx_ens_impl.hpp);
    // WBX_ENS_SKIPNA_GENERIC=1 pins the generic from-memory operator (A/B timing, tests)
    static const bool generic = getenv("WBX_ENS_SKIPNA_GENERIC") && atoi(getenv("WBX_ENS_SKIPNA_GENERIC")) != 0;
    if (generic || M > 64) return launch_ens_op<EnsOpGeneric<float>>(ctx, plan, a, map);
    algo = WBX_ENS_SKIPNA_SORT;
  }
  if (M == 51) return launch_ens_m51(ctx, plan, a, algo, map);
  if (M == 50) return launch_ens_m50(ctx, plan, a, algo, map);
  if (M <= 4) return launch_ens_m4(ctx, plan, a, algo, map);
  if (M <= 8) return launch_ens_m8(ctx, plan, a, algo, map);
  if (M <= 16) return launch_ens_m16(ctx, plan, a, algo, map);
  if (M <= 32) return launch_ens_m32(ctx, plan, a, algo, map);
  if (M <= 64) return launch_ens_m64(ctx, plan, a, algo, map);
  return launch_ens_op<EnsOpGeneric<float>>(ctx, plan, a, map);
}

}  // namespace wbx

extern "C" int wbx_ens_partial(wbx_ctx* ctx, const wbx_s1_plan* plan, int dtype, int M, int64_t member_stride,
                               int algo, const void* p, const void* t, const uint8_t* mask, double* partial_out) {
  return wbx::ens_common(ctx, plan, dtype, M, member_stride, algo, p, t, mask, partial_out, false, 0);
}

extern "C" int wbx_ens_map(wbx_ctx* ctx, const wbx_s1_plan* plan, int dtype, int M, int64_t member_stride, int algo,
                           int lane, const void* p, const void* t, double* out) {
  if (lane < 0 || lane >= WBX_ENS_LANES) return wbx::fail(WBX_ERR_INVALID, "ensemble lane %d out of range", lane);
  return wbx::ens_common(ctx, plan, dtype, M, member_stride, algo, p, t, nullptr, out, true, lane);
}

// ---- the ensemble family with weights, bins and mask in one pass (wbx_ens_atoms.hpp) -------------------------------------
namespace wbx {

static int ens_binned_geometry(const wbx_s1_plan* plan, int64_t nA, int64_t nBk, int64_t nBr, int32_t w_on_x, BinnedArgs& g) {
  if (int rc = check_plan(plan)) return rc;
  WBX_REQUIRE(nA >= 1 && nBk >= 1 && nBr >= 1 && plan->nx >= 1 && plan->ndepth >= 1, "empty geometry");
  patch_geometry(g, nA * nBk, nBk, nBr, (w_on_x & WBX_BINNED_W_ON_X) ? plan->nx : 1, plan->ndepth, plan->nx, ens_atoms_rows(), ens_atoms_taper());
  return 0;
}

}  // namespace wbx

extern "C" int wbx_ens_binned_atoms_size(const wbx_s1_plan* plan, int64_t nA, int64_t nBk, int64_t nBr, int32_t w_on_x,
                                         int64_t* bytes_out) {
  using namespace wbx;
  WBX_REQUIRE(bytes_out != nullptr, "bytes_out is NULL");
  BinnedArgs g;
  if (int rc = ens_binned_geometry(plan, nA, nBk, nBr, w_on_x, g)) return rc;
  *bytes_out = (int64_t)atoms_carve(g, nullptr);
  return 0;
}

extern "C" int wbx_ens_binned_atoms(wbx_ctx* ctx, const wbx_s1_plan* plan, int64_t nA, int64_t nBk, int64_t nBr,
                                    int32_t w_on_x, const uint64_t* bits, void* atoms_out, int64_t* overflow_out) {
  using namespace wbx;
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  WBX_REQUIRE(bits != nullptr && atoms_out != nullptr, "NULL pointer");
  BinnedArgs g;
  if (int rc = ens_binned_geometry(plan, nA, nBk, nBr, w_on_x, g)) return rc;
  WBX_HIP(hipSetDevice(ctx->device));
  atoms_carve(g, atoms_out);
  if (int rc = atoms_launch(ctx, g, bits, plan->ndepth, plan->nx, true)) return rc;
  if (overflow_out) {  // patches with more than ATOM_MAX distinct membership words: wbx_ens_binned cannot take these bins
    const size_t n = (size_t)g.nBk * g.nrs * g.nxt;
    int32_t* host = (int32_t*)malloc(n * sizeof(int32_t));
    WBX_REQUIRE(host != nullptr, "out of host memory");
    hipError_t err = hipMemcpyAsync(host, g.nwords, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
    int64_t bad = 0;
    for (size_t i = 0; i < n && err == hipSuccess; ++i) bad += host[i] < 0;
    free(host);
    WBX_HIP(err);
    *overflow_out = bad;
  }
  return 0;
}

extern "C" int wbx_ens_binned(wbx_ctx* ctx, const wbx_s1_plan* plan, int dtype, int M, int64_t member_stride, int algo,
                              const void* p, const void* t, const uint8_t* mask, const double* wt, const uint64_t* bits,
                              int64_t nA, int64_t nBk, int64_t nBr, int32_t w_on_x, int32_t nbin, const void* atoms,
                              double* out) {
  using namespace wbx;
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  if (int rc = check_plan(plan)) return rc;
  WBX_REQUIRE(dtype == WBX_F32, "wbx_ens_binned takes float32 members (got dtype %d)", dtype);
  WBX_REQUIRE(algo == WBX_ENS_SORT, "wbx_ens_binned runs the rank form (WBX_ENS_SORT) only");
  WBX_REQUIRE(M >= 2 && M <= 64, "wbx_ens_binned handles 2..64 members (got %d)", M);
  WBX_REQUIRE(!(plan->flags & WBX_FLAG_SKIPNA_ENS), "wbx_ens_binned does not take skipna_ensemble (per-point member counts)");
  WBX_REQUIRE(nbin >= 1 && nbin <= 64, "wbx_ens_binned handles 1..64 bins (got %d)", nbin);
  WBX_REQUIRE(nA >= 0 && nBk >= 0 && nBr >= 0 && nA * nBk * nBr == plan->nkey, "nA*nBk*nBr must equal plan->nkey");
  WBX_REQUIRE((w_on_x & ~63) == 0 && (w_on_x & 6) != 6, "w_on_x: unknown or contradictory WBX_BINNED_* flags (%d)", w_on_x);
  WBX_REQUIRE(wt == nullptr || (w_on_x & (WBX_BINNED_WT_X_ONLY | WBX_BINNED_WT_ROW_ONLY)),
              "wbx_ens_binned takes factored weights only (WBX_BINNED_WT_X_ONLY / WBX_BINNED_WT_ROW_ONLY), or wt = NULL");
  WBX_REQUIRE(!(w_on_x & WBX_BINNED_WT_X_ONLY) || (w_on_x & WBX_BINNED_W_ON_X), "WBX_BINNED_WT_X_ONLY needs WBX_BINNED_W_ON_X");
  const bool twin_out = (plan->flags & WBX_FLAG_MASKED) && (w_on_x & WBX_BINNED_TWIN_MASK);
  const bool skipna = (plan->flags & WBX_FLAG_SKIPNA) != 0;
  const int64_t nout = nA * nBk * (skipna ? (twin_out ? 20 : 10) : (twin_out ? ENS_ATOMS_NOUT2 : ENS_ATOMS_NOUT)) * nbin;
  if (nout == 0) return 0;
  WBX_REQUIRE(out != nullptr, "out is NULL");
  WBX_HIP(hipSetDevice(ctx->device));
  if (nBr * plan->ndepth * plan->nx == 0) {
    if (!(w_on_x & WBX_BINNED_ACCUMULATE)) WBX_HIP(hipMemsetAsync(out, 0, (size_t)nout * sizeof(double), ctx->stream));
    return 0;
  }
  WBX_REQUIRE(p != nullptr && t != nullptr && bits != nullptr, "p/t/bits is NULL");
  if (plan->flags & WBX_FLAG_MASKED) {
    WBX_REQUIRE(mask != nullptr, "WBX_FLAG_MASKED set but mask is NULL");
    WBX_REQUIRE(w_on_x & WBX_BINNED_W_ON_X, "wbx_ens_binned with a validity mask needs W indexed per x (WBX_BINNED_W_ON_X)");
    WBX_REQUIRE(plan->xstride[3] >= 0, "the mask's x stride must be non-negative");
  }
  WBX_REQUIRE(plan->xstride[0] >= 0 && plan->xstride[1] >= 0 && (double)plan->nx * (double)plan->xstride[0] * 4.0 < 4294967296.0 &&
                  (double)plan->nx * (double)plan->xstride[1] * 4.0 < 4294967296.0,
              "x offsets of the members and of the targets must be non-negative and fit 32 bits");
  S1Args a;
  fill_args(plan, a);
  a.in[0] = p;
  a.in[1] = t;
  a.in[3] = mask;
  a.M = M;
  a.mstride = member_stride;
  EnsBinnedCall c;
  c.wt = wt;
  c.bits = bits;
  c.nA = nA;
  c.nBk = nBk;
  c.nBr = nBr;
  c.nj = (w_on_x & WBX_BINNED_W_ON_X) ? plan->nx : 1;
  c.nbin = nbin;
  c.w_on_x = w_on_x;
  c.prepared = atoms;
  c.out = out;
  if (M == 51) return launch_ens_atoms_m51(ctx, plan, a, c);
  if (M == 50) return launch_ens_atoms_m50(ctx, plan, a, c);
  if (M <= 4) return launch_ens_atoms_m4(ctx, plan, a, c);
  if (M <= 8) return launch_ens_atoms_m8(ctx, plan, a, c);
  if (M <= 16) return launch_ens_atoms_m16(ctx, plan, a, c);
  if (M <= 32) return launch_ens_atoms_m32(ctx, plan, a, c);
  return launch_ens_atoms_m64(ctx, plan, a, c);
}
