// wbx_chunk_replay: a recorded sequence of entry-point calls run again with this chunk's pointers (include/wbx.h).
// Host code only.  Every entry is dispatched through the entry point's OWN prototype: the parameter types are deduced from
// the function, each 64-bit pattern is converted to the parameter it stands for, and a record whose argument count does not
// match the prototype is refused.
#include <type_traits>
#include <utility>

#include "wbx_common.hpp"

namespace wbx {

template <typename T>
static inline T replay_arg(uint64_t v) {
  if constexpr (std::is_pointer_v<T>)
    return reinterpret_cast<T>(static_cast<uintptr_t>(v));
  else
    return static_cast<T>(static_cast<int64_t>(v));
}

template <typename... A, size_t... I>
static inline int replay_invoke(int (*fn)(A...), const uint64_t* a, std::index_sequence<I...>) {
  return fn(replay_arg<A>(a[I])...);
}

template <typename... A>
static inline int replay_call(int (*fn)(A...), const wbx_call& c, const char* name) {
  static_assert(sizeof...(A) <= WBX_CALL_MAX_ARGS, "WBX_CALL_MAX_ARGS too small");
  WBX_REQUIRE(c.nargs == (int32_t)sizeof...(A), "wbx_chunk_replay: %s takes %d arguments, the record holds %d", name, (int)sizeof...(A),
              (int)c.nargs);
  return replay_invoke(fn, c.args, std::index_sequence_for<A...>{});
}

}  // namespace wbx

extern "C" int wbx_chunk_replay(wbx_call* calls, int32_t ncalls, const wbx_reloc* relocs, int32_t nrelocs, const uint64_t* slots,
                                int32_t nslots) {
  using namespace wbx;
  WBX_REQUIRE(ncalls >= 0 && nrelocs >= 0 && nslots >= 0, "negative count");
  WBX_REQUIRE(ncalls == 0 || calls != nullptr, "calls is NULL");
  WBX_REQUIRE(nrelocs == 0 || (relocs != nullptr && slots != nullptr), "relocs / slots is NULL");
  for (int32_t j = 0; j < nrelocs; ++j) {
    const wbx_reloc& r = relocs[j];
    WBX_REQUIRE(r.call >= 0 && r.call < ncalls && r.slot >= 0 && r.slot < nslots && r.arg >= 0 && r.arg < calls[r.call].nargs &&
                    r.arg < WBX_CALL_MAX_ARGS,
                "wbx_chunk_replay: relocation %d out of range (call %d, arg %d, slot %d)", (int)j, (int)r.call, (int)r.arg, (int)r.slot);
    calls[r.call].args[r.arg] = slots[r.slot] + (uint64_t)r.offset;
  }
#define WBX_REPLAY_CASE(ID, FN) \
  case ID:                      \
    rc = replay_call(FN, c, #FN); \
    break;
  for (int32_t i = 0; i < ncalls; ++i) {
    const wbx_call& c = calls[i];
    int rc = 0;
    switch (c.fn) {
      WBX_REPLAY_CASE(WBX_FN_DET_PARTIAL, wbx_det_partial)
      WBX_REPLAY_CASE(WBX_FN_ENS_PARTIAL, wbx_ens_partial)
      WBX_REPLAY_CASE(WBX_FN_ENS2_PARTIAL, wbx_ens2_partial)
      WBX_REPLAY_CASE(WBX_FN_CAT_PARTIAL, wbx_cat_partial)
      WBX_REPLAY_CASE(WBX_FN_CAT_EXCEED_FIELD, wbx_cat_exceed_field)
      WBX_REPLAY_CASE(WBX_FN_CONTRACT, wbx_contract)
      WBX_REPLAY_CASE(WBX_FN_CONTRACT_BITS, wbx_contract_bits)
      WBX_REPLAY_CASE(WBX_FN_DET_BINNED, wbx_det_binned)
      WBX_REPLAY_CASE(WBX_FN_ENS_BINNED, wbx_ens_binned)
      WBX_REPLAY_CASE(WBX_FN_ZONAL_SPECTRUM, wbx_zonal_spectrum)
      WBX_REPLAY_CASE(WBX_FN_ZONAL_SPECTRUM_SLABS, wbx_zonal_spectrum_slabs)
      WBX_REPLAY_CASE(WBX_FN_DET_SPECTRUM, wbx_det_spectrum)
      WBX_REPLAY_CASE(WBX_FN_DET_SPECTRUM_SLABS, wbx_det_spectrum_slabs)
      WBX_REPLAY_CASE(WBX_FN_DET_SPECTRUM_FOLDED, wbx_det_spectrum_folded)
      WBX_REPLAY_CASE(WBX_FN_ACC_ADD, wbx_acc_add)
      WBX_REPLAY_CASE(WBX_FN_MEMSET, wbx_memset)
      WBX_REPLAY_CASE(WBX_FN_MEMCPY_D2D, wbx_memcpy_d2d)
      WBX_REPLAY_CASE(WBX_FN_CTX_WAIT_FENCE, wbx_ctx_wait_fence)
      WBX_REPLAY_CASE(WBX_FN_FENCE_RECORD, wbx_fence_record)
      default:
        WBX_REQUIRE(false, "wbx_chunk_replay: call %d names entry point %d, which cannot be part of a record", (int)i, (int)c.fn);
    }
    if (rc != 0) return rc;  // (wbx_last_error holds the callee's message)
  }
#undef WBX_REPLAY_CASE
  return 0;
}
