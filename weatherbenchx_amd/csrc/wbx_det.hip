// Deterministic statistic families for the stage-1 skeleton (wbx_s1.hpp).
//
// Reference semantics restated (weatherbenchX/metrics/deterministic.py):
//   Error :91-100            e  = p - t
//   AbsoluteError :103-112   |e|
//   SquaredError :115-123    e^2
//   SquaredPredictionAnomaly :222-232   (p - c)^2
//   SquaredTargetAnomaly :235-245       (t - c)^2
//   AnomalyCovariance :248-259          (p - c)(t - c)
// and the mask / skipna pre-processing of Aggregator.aggregate_stat_var
// (weatherbenchX/aggregation.py:339-357): masked-out or (skipna) NaN statistic values
// become 0 and are counted out of sum_weights through the paired count lane.
//
// All arithmetic is fp64 on the widened inputs, so results follow the reference evaluated
// on float64-cast inputs to summation-order round-off.
#include <type_traits>

#include "wbx_s1.hpp"

namespace wbx {

#ifndef WBX_CLIM_STREAM
#define WBX_CLIM_STREAM true
#endif

// 4-wide vector types that only promise element alignment: gfx950 global loads may be unaligned, so rows
// that start at any element (e.g. 721-long latitude rows) still get one global_load_dwordx4 per lane.
template <typename T>
struct Vec4 {
  typedef T type __attribute__((ext_vector_type(4), aligned(sizeof(T))));
};

template <typename T, int V, bool STREAM = true>
__device__ __forceinline__ void load_x(const void* base, int64_t off, int64_t x, int64_t xs, T (&v)[V]) {
  const T* p = reinterpret_cast<const T*>(base) + off;
  if constexpr (V == 4 && sizeof(T) == 1) {
    // four mask bytes = one aligned dword (the planner only picks vec = 4 when every mask row starts 4-byte aligned)
    if (xs == 1) {
      const uint32_t q = *reinterpret_cast<const uint32_t*>(p + x);
      v[0] = (T)(q & 0xffu);
      v[1] = (T)((q >> 8) & 0xffu);
      v[2] = (T)((q >> 16) & 0xffu);
      v[3] = (T)(q >> 24);
    } else {
      T s = p[0];
      v[0] = v[1] = v[2] = v[3] = s;
    }
  } else if constexpr (V == 4) {
    if (xs == 1) {
      using V4 = typename Vec4<T>::type;
      V4 q;
      if constexpr (STREAM)
        q = __builtin_nontemporal_load(reinterpret_cast<const V4*>(p + x));  // see ld_stream (element-aligned type)
      else
        q = *reinterpret_cast<const V4*>(p + x);
      v[0] = q.x;
      v[1] = q.y;
      v[2] = q.z;
      v[3] = q.w;
    } else {  // xs == 0: broadcast along x
      T s = p[0];
      v[0] = v[1] = v[2] = v[3] = s;
    }
  } else {
#pragma unroll
    for (int k = 0; k < V; ++k) {
      if constexpr (sizeof(T) > 1 && STREAM)
        v[k] = ld_stream(p + (x + k) * xs);
      else
        v[k] = p[(x + k) * xs];  // mask bytes are re-read by every depth row: keep them cached
    }
  }
}

// FUNC: 0 = DET3 (p,t), 1 = DET6 (p,t,c), 2 = PASS1 (p).
// MM: 0 = no count lanes; 1 = mask only (ONE count lane shared by every value lane: validity does not depend on the
// statistic); 2 = skipna without a mask, 3 = skipna with a mask (one count lane per value lane: a NaN may hit some
// statistics only).  Whether the mask is read is a template parameter, not a run-time flag: a load under a branch
// makes the compiler drain the whole load queue (s_waitcnt vmcnt(0)) instead of waiting for the oldest load.
template <typename T, int FUNC, int MM>
struct DetOp {
  static constexpr int NIN = FUNC == WBX_DET6 ? 3 : (FUNC == WBX_DET3 ? 2 : 1);
  static constexpr int NLANE = FUNC == WBX_DET6 ? 6 : (FUNC == WBX_DET3 ? 3 : 1);
  static constexpr int NACC = NLANE + (MM == 1 ? 1 : (MM >= 2 ? NLANE : 0));
  static constexpr bool HAS_MASK = MM == 1 || MM == 3;
  static constexpr bool MROW_OK = HAS_MASK;
  static constexpr int XR_UNROLL = 2, XK_UNROLL = 4, MIN_WAVES = 1;

  __device__ __forceinline__ static void lanes(double p, double t, double c, double (&val)[NLANE]) {
    if constexpr (FUNC == WBX_PASS1) {
      val[0] = p;
    } else {
      const double e = p - t;
      val[0] = e;
      val[1] = fabs(e);
      val[2] = e * e;
      if constexpr (FUNC == WBX_DET6) {
        const double pa = p - c, ta = t - c;
        val[3] = pa * pa;
        val[4] = ta * ta;
        val[5] = pa * ta;
      }
    }
  }

  __device__ __forceinline__ static void values(const S1Args& a, const int64_t (&ro)[WBX_MAX_INPUTS], int64_t x,
                                                double (&val)[NLANE]) {
    T p[1], t[1] = {0}, c[1] = {0};
    load_x<T, 1>(a.in[0], ro[0], x, a.xstride[0], p);
    if constexpr (NIN > 1) load_x<T, 1>(a.in[1], ro[1], x, a.xstride[1], t);
    if constexpr (NIN > 2) load_x<T, 1>(a.in[2], ro[2], x, a.xstride[2], c);
    lanes((double)p[0], (double)t[0], (double)c[0], val);
  }

  template <int V, bool XK>
  __device__ __forceinline__ static void accum(const S1Args& a, const int64_t (&ro)[WBX_MAX_INPUTS], int64_t x,
                                               double (&acc)[XK ? V : 1][NACC]) {
    uint8_t m[V];
    if constexpr (HAS_MASK) load_x<uint8_t, V>(a.in[3], ro[3], x, a.xstride[3], m);
    fold<V, XK>(a, ro, x, acc, m);
  }

  // mask row staged in LDS by the kernel (unit x stride)
  template <int V>
  __device__ __forceinline__ static void accum_mrow(const S1Args& a, const int64_t (&ro)[WBX_MAX_INPUTS], int64_t x,
                                                    double (&acc)[1][NACC], const uint8_t* mrow) {
    uint8_t m[V];
    if constexpr (V == 4) {
      const uint32_t q = *reinterpret_cast<const uint32_t*>(mrow + x);
      m[0] = (uint8_t)(q & 0xffu);
      m[1] = (uint8_t)((q >> 8) & 0xffu);
      m[2] = (uint8_t)((q >> 16) & 0xffu);
      m[3] = (uint8_t)(q >> 24);
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) m[k] = mrow[x + k];
    }
    fold<V, false>(a, ro, x, acc, m);
  }

  template <int V, bool XK>
  __device__ __forceinline__ static void fold(const S1Args& a, const int64_t (&ro)[WBX_MAX_INPUTS], int64_t x,
                                              double (&acc)[XK ? V : 1][NACC], const uint8_t (&m)[V]) {
    T p[V], t[V] = {}, c[V] = {};
    load_x<T, V>(a.in[0], ro[0], x, a.xstride[0], p);
    if constexpr (NIN > 1) load_x<T, V>(a.in[1], ro[1], x, a.xstride[1], t);
    if constexpr (NIN > 2) load_x<T, V, WBX_CLIM_STREAM>(a.in[2], ro[2], x, a.xstride[2], c);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      bool valid = true;
      if constexpr (HAS_MASK) {
        // a masked-out point contributes 0 to every lane whatever its values (aggregation.py:339-357): zeroing the
        // three INPUTS (p = t = c = 0 -> every statistic is exactly 0) costs 3 selects instead of 2 per fp64 lane
        valid = m[k] != 0;
        p[k] = valid ? p[k] : T(0);
        if constexpr (NIN > 1) t[k] = valid ? t[k] : T(0);
        if constexpr (NIN > 2) c[k] = valid ? c[k] : T(0);
      }
      double val[NLANE];
      lanes((double)p[k], NIN > 1 ? (double)t[k] : 0.0, NIN > 2 ? (double)c[k] : 0.0, val);
      double(&A)[NACC] = acc[XK ? k : 0];
      if constexpr (MM == 0) {
#pragma unroll
        for (int l = 0; l < NLANE; ++l) A[l] += val[l];
      } else if constexpr (MM == 1) {
#pragma unroll
        for (int l = 0; l < NLANE; ++l) A[l] += val[l];
        A[NLANE] += valid ? 1.0 : 0.0;
      } else {
#pragma unroll
        for (int l = 0; l < NLANE; ++l) {
          const bool fin = !(val[l] != val[l]);
          A[l] += fin ? val[l] : 0.0;
          A[NLANE + l] += (fin && valid) ? 1.0 : 0.0;
        }
      }
    }
  }
};

template <typename T, int FUNC>
static int dispatch_partial(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a) {
  if (plan->x_weights != nullptr && plan->plane_rows > 0) {  // folded weights over contiguous planes: flat float4 sweep
    if constexpr (std::is_same<T, float>::value) {
      WBX_REQUIRE(!(plan->flags & WBX_FLAG_SKIPNA), "flat x-weighted mode does not take the skipna flag");
      for (int i = 0; i < DetOp<T, FUNC, 0>::NIN; ++i)
        WBX_REQUIRE(plan->xstride[i] == 1 && (((uintptr_t)a.in[i]) & 15) == 0,
                    "flat x-weighted mode needs unit x stride and 16-byte aligned inputs");
      if (plan->flags & WBX_FLAG_MASKED) {
        WBX_REQUIRE(plan->xstride[3] == 1 && (((uintptr_t)a.in[3]) & 3) == 0,
                    "flat x-weighted mode needs a unit-stride, 4-byte aligned mask");
        if (plan->nkey == 0 || plan->ndepth == 0 || plan->nx == 0) return launch_partial<DetOp<T, FUNC, 1>, 1>(ctx, plan, a);
        return launch_flat_weighted<DetOp<T, FUNC, 1>>(ctx, plan, a);
      }
      if (plan->nkey == 0 || plan->ndepth == 0 || plan->nx == 0) return launch_partial<DetOp<T, FUNC, 0>, 1>(ctx, plan, a);
      return launch_flat_weighted<DetOp<T, FUNC, 0>>(ctx, plan, a);
    } else {
      return fail(WBX_ERR_INVALID, "flat x-weighted mode is fp32 only");
    }
  }
  WBX_REQUIRE(plan->x_weights == nullptr, "x_weights needs the flat mode (plane_rows > 0, fp32, no mask)");
  if (plan->plane_rows > 0) {
    if constexpr (std::is_same<T, float>::value) {
      WBX_REQUIRE(!(plan->flags & WBX_FLAG_SKIPNA), "plane mode does not take the skipna flag");
      for (int i = 0; i < DetOp<T, FUNC, 0>::NIN; ++i)
        WBX_REQUIRE(plan->xstride[i] == 1 && (((uintptr_t)a.in[i]) & 15) == 0,
                    "plane mode needs unit x stride and 16-byte aligned inputs");
      if (plan->flags & WBX_FLAG_MASKED) {
        WBX_REQUIRE(plan->xstride[3] == 1 && (((uintptr_t)a.in[3]) & 3) == 0,
                    "plane mode needs a unit-stride, 4-byte aligned mask");
        if (plan->nkey == 0 || plan->ndepth == 0 || plan->nx == 0) return launch_partial<DetOp<T, FUNC, 1>, 1>(ctx, plan, a);
        return launch_plane<DetOp<T, FUNC, 1>>(ctx, plan, a);
      }
      if (plan->nkey == 0 || plan->ndepth == 0 || plan->nx == 0) return launch_partial<DetOp<T, FUNC, 0>, 1>(ctx, plan, a);
      return launch_plane<DetOp<T, FUNC, 0>>(ctx, plan, a);
    } else {
      return fail(WBX_ERR_INVALID, "plane mode is fp32 only");
    }
  }
  const bool masked = plan->flags & WBX_FLAG_MASKED;
  if (plan->flags & WBX_FLAG_SKIPNA) {
    if (masked) {
      if (plan->vec == 4) return launch_partial<DetOp<T, FUNC, 3>, 4>(ctx, plan, a);
      return launch_partial<DetOp<T, FUNC, 3>, 1>(ctx, plan, a);
    }
    if (plan->vec == 4) return launch_partial<DetOp<T, FUNC, 2>, 4>(ctx, plan, a);
    return launch_partial<DetOp<T, FUNC, 2>, 1>(ctx, plan, a);
  }
  if (masked) {
    // (lat, lon) mask under a reduction over init_time: same mask row for every depth row of a key -> LDS
    const bool mrow = !plan->x_kept && plan->depth_off[3] == nullptr && plan->xstride[3] == 1 &&
                      plan->nx <= WBX_MROW_MAX && plan->ndepth > 1;
    if (plan->vec == 4) return launch_partial<DetOp<T, FUNC, 1>, 4>(ctx, plan, a, mrow);
    return launch_partial<DetOp<T, FUNC, 1>, 1>(ctx, plan, a, mrow);
  }
  if (plan->vec == 4) return launch_partial<DetOp<T, FUNC, 0>, 4>(ctx, plan, a);
  return launch_partial<DetOp<T, FUNC, 0>, 1>(ctx, plan, a);
}

template <typename T>
static int dispatch_func(wbx_ctx* ctx, const wbx_s1_plan* plan, int func, S1Args& a, bool map) {
  switch (func) {
    case WBX_DET3:
      return map ? launch_map<DetOp<T, WBX_DET3, 0>>(ctx, plan, a) : dispatch_partial<T, WBX_DET3>(ctx, plan, a);
    case WBX_DET6:
      return map ? launch_map<DetOp<T, WBX_DET6, 0>>(ctx, plan, a) : dispatch_partial<T, WBX_DET6>(ctx, plan, a);
    case WBX_PASS1:
      return map ? launch_map<DetOp<T, WBX_PASS1, 0>>(ctx, plan, a) : dispatch_partial<T, WBX_PASS1>(ctx, plan, a);
  }
  return fail(WBX_ERR_INVALID, "unknown deterministic family %d", func);
}

static int det_common(wbx_ctx* ctx, const wbx_s1_plan* plan, int func, int dtype, const void* p, const void* t,
                      const void* c, const uint8_t* mask, double* out, bool map, int lane) {
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  if (int rc = check_plan(plan)) return rc;
  WBX_REQUIRE(out != nullptr || plan->nkey == 0, "output pointer is NULL");
  WBX_REQUIRE(p != nullptr || plan->nkey * plan->ndepth * plan->nx == 0, "predictions pointer is NULL");
  if (func != WBX_PASS1) WBX_REQUIRE(t != nullptr || plan->nkey * plan->ndepth * plan->nx == 0, "targets pointer is NULL");
  if (func == WBX_DET6) WBX_REQUIRE(c != nullptr || plan->nkey * plan->ndepth * plan->nx == 0, "climatology pointer is NULL");
  if (plan->flags & WBX_FLAG_MASKED) WBX_REQUIRE(mask != nullptr, "WBX_FLAG_MASKED set but mask is NULL");
  WBX_HIP(hipSetDevice(ctx->device));
  S1Args a;
  fill_args(plan, a);
  a.in[0] = p;
  a.in[1] = t;
  a.in[2] = c;
  a.in[3] = mask;
  a.out = out;
  a.lane = lane;
  if (dtype == WBX_F32) return dispatch_func<float>(ctx, plan, func, a, map);
  if (dtype == WBX_F64) return dispatch_func<double>(ctx, plan, func, a, map);
  return fail(WBX_ERR_INVALID, "unknown dtype %d", dtype);
}

}  // namespace wbx

extern "C" int wbx_det_partial(wbx_ctx* ctx, const wbx_s1_plan* plan, int func, int dtype, const void* p,
                               const void* t, const void* c, const uint8_t* mask, double* partial_out) {
  return wbx::det_common(ctx, plan, func, dtype, p, t, c, mask, partial_out, false, 0);
}

extern "C" int wbx_det_map(wbx_ctx* ctx, const wbx_s1_plan* plan, int func, int dtype, int lane, const void* p,
                           const void* t, const void* c, double* out) {
  const int nl = func == WBX_DET6 ? 6 : (func == WBX_DET3 ? 3 : 1);
  if (lane < 0 || lane >= nl) return wbx::fail(WBX_ERR_INVALID, "lane %d out of range for family %d", lane, func);
  return wbx::det_common(ctx, plan, func, dtype, p, t, c, nullptr, out, true, lane);
}

extern "C" int wbx_s1_partial_len(const wbx_s1_plan* plan, int lanes, int64_t* n_out) {
  if (!plan || !n_out || lanes <= 0) return wbx::fail(WBX_ERR_INVALID, "bad arguments to wbx_s1_partial_len");
  const int64_t nj = plan->x_kept ? plan->nx : 1;
  const int64_t nl = (plan->flags & WBX_FLAG_SKIPNA) ? 2 * (int64_t)lanes
                     : ((plan->flags & WBX_FLAG_MASKED) ? (int64_t)lanes + 1 : (int64_t)lanes);
  *n_out = plan->nkey * plan->nchunk * nl * nj;
  return 0;
}
