// Zonal energy spectrum: batched 1-D R2C rocFFT along longitude + HIP |F|^2 reduction.
//
// No reference implementation exists in this snapshot -> "parity unpinned"; the definition is
// fixed in include/wbx.h and pinned by analytic tests (Parseval, single sinusoid, constant field).
// Rows are transformed in place through rocFFT's strided layout (no gather copy), in tiles of rows so the
// complex scratch stays bounded; each tile's |F_k|^2 is scaled and added to its group's spectrum.
#include <rocfft/rocfft.h>

#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "wbx_common.hpp"
#include "wbx_s1.hpp"

namespace wbx {

struct FftPlan {
  rocfft_plan plan = nullptr;
  rocfft_execution_info info = nullptr;
  void* work = nullptr;
  size_t work_size = 0;
};

// ---- order-independent sums (r5) ---------------------------------------------------------------------------------------
// The spectra of a group are summed over rows that many teams (waves, blocks) share.  Rounds 1-4 let every team ADD its sums to
// power[group][k] with fp64 atomics: correct, but the order of the additions -- and with it the last bits of 9-14 % of the
// outputs -- differed from run to run: the one order-dependent sum of the library.  Now a team's sums for one group over one
// run of its rows are a RECORD: {group, key = team << 24 | sequence number} + the values, written with plain stores into a slot
// taken from a counter, and spec_close_kernel adds the records of every group in KEY order -- which is fixed by the launch
// geometry, not by who arrived first.  No atomic touches a value; the slot counter is the only atomic left.
//   * capacity: a team opens a record when its group changes and when it is done, and one more for the second row of a pair
//     that straddles two groups: teams + 2 x (group changes along the rows) for teams that walk contiguous rows, stated per
//     kernel by its launcher.  The group changes are counted once per group table (a tiny kernel + one read-back, cached by the
//     table's device address; every copy / memset through this library into that address drops the entry);
//   * a launch that still runs out of slots (a table rewritten behind the library's back) turns its outputs NaN, never wrong;
//   * the closing kernel also does what the memset in front of the transform did (accumulate = 0 starts from zero).
struct SpecRecs {
  double* data;             // [capacity + 1][width]: width = wavenumbers x fields; the last row takes what does not fit
  int32_t* group;           // [capacity]
  unsigned long long* key;  // [capacity]
  unsigned int* count;      // [0], [1]: slots taken (launches alternate: this one counts in count[parity], its closing kernel
                            // clears the other); [2 + parity]: this launch ran out of slots
  unsigned int capacity;    // slots in all: [0, nstatic) are spoken for by the launch geometry (a team's last record, a step's table:
  unsigned int nstatic;     // no counter, no waiting for one), [nstatic, capacity) are dealt out by count[parity]
  int32_t width, parity;
};

struct FftState {
  SpecRecs recs = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
  size_t recs_bytes = 0;  // of recs.data
  std::map<std::tuple<int, int64_t, int64_t, int64_t>, FftPlan> plans;  // (nlon, lon_stride, row_stride, batch)
  void* scratch = nullptr;  // complex tile
  std::map<int, void*> twiddles;  // nlon -> device float2[n/2] + float2[n/2 + 1] of the fused path
  std::map<std::vector<int64_t>, void*> slab_offsets;  // slab offset lists of latitude-fastest fields, on the device
  std::map<std::pair<const void*, size_t>, int> occupancy;  // (fused kernel, dynamic LDS bytes) -> resident blocks per CU
  size_t scratch_size = 0;
  bool setup = false;
};

static void destroy_plan(FftPlan& p) {
  if (p.info) rocfft_execution_info_destroy(p.info);
  if (p.plan) rocfft_plan_destroy(p.plan);
  if (p.work) (void)hipFree(p.work);
  p = FftPlan();
}

void spectrum_release(wbx_ctx* ctx) {
  auto* st = reinterpret_cast<FftState*>(ctx->fft_state);
  if (!st) return;
  for (auto& kv : st->plans) destroy_plan(kv.second);
  if (st->scratch) (void)hipFree(st->scratch);
  if (st->recs.data) (void)hipFree(st->recs.data);
  if (st->recs.group) (void)hipFree(st->recs.group);
  if (st->recs.key) (void)hipFree(st->recs.key);
  if (st->recs.count) (void)hipFree(st->recs.count);
  for (auto& kv : st->twiddles) (void)hipFree(kv.second);
  for (auto& kv : st->slab_offsets) (void)hipFree(kv.second);
  if (st->setup) rocfft_cleanup();
  delete st;
  ctx->fft_state = nullptr;
}

#define WBX_FFT(expr)                                                                              \
  do {                                                                                             \
    rocfft_status _s = (expr);                                                                     \
    if (_s != rocfft_status_success) return fail(WBX_ERR_FFT, "%s failed with rocfft_status %d", #expr, (int)_s); \
  } while (0)

static int get_plan(wbx_ctx* ctx, FftState* st, int nlon, int64_t lon_stride, int64_t row_stride, int64_t batch,
                    FftPlan** out) {
  auto key = std::make_tuple(nlon, lon_stride, row_stride, batch);
  auto it = st->plans.find(key);
  if (it != st->plans.end()) {
    *out = &it->second;
    return 0;
  }
  FftPlan p;
  rocfft_plan_description desc = nullptr;
  WBX_FFT(rocfft_plan_description_create(&desc));
  const size_t in_stride = (size_t)lon_stride, out_stride = 1;
  const size_t nk = (size_t)nlon / 2 + 1;
  WBX_FFT(rocfft_plan_description_set_data_layout(desc, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved,
                                                  nullptr, nullptr, 1, &in_stride, (size_t)row_stride, 1, &out_stride, nk));
  const size_t length = (size_t)nlon;
  WBX_FFT(rocfft_plan_create(&p.plan, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                             rocfft_precision_single, 1, &length, (size_t)batch, desc));
  WBX_FFT(rocfft_plan_description_destroy(desc));
  WBX_FFT(rocfft_execution_info_create(&p.info));
  WBX_FFT(rocfft_plan_get_work_buffer_size(p.plan, &p.work_size));
  if (p.work_size) {
    WBX_HIP(hipMalloc(&p.work, p.work_size));
    WBX_FFT(rocfft_execution_info_set_work_buffer(p.info, p.work, p.work_size));
  }
  WBX_FFT(rocfft_execution_info_set_stream(p.info, ctx->stream));
  st->plans[key] = p;
  *out = &st->plans[key];
  return 0;
}

// A record for (group g, key): called by every lane of a wave (lane = its lane id), one slot per call; -> where the wave
// stores the record's `width` values.
__device__ __forceinline__ double* spec_rec_open(const SpecRecs& R, int32_t g, unsigned long long key, int lane) {
  unsigned int r = 0;
  if (lane == 0) {
    r = R.nstatic + atomicAdd(R.count + R.parity, 1u);
    if (r < R.capacity) {
      R.group[r] = g;
      R.key[r] = key;
    } else {
      R.count[2 + R.parity] = 1u;
      r = R.capacity;
    }
  }
  r = (unsigned int)__builtin_amdgcn_readfirstlane((int)r);
  return R.data + (size_t)r * (size_t)R.width;
}
__device__ __forceinline__ unsigned long long spec_key(int64_t team, unsigned int seq) {
  return ((unsigned long long)team << 24) | (unsigned long long)(seq & 0xffffffu);
}
// The slot the launch geometry reserves for this team / step (slot < R.nstatic): no counter to wait for.  One thread writes the
// header; g < 0 marks a reserved slot that stays empty (every reserved slot gets a header from exactly one thread per launch).
__device__ __forceinline__ double* spec_rec_static(const SpecRecs& R, unsigned int slot, int32_t g, unsigned long long key, bool writer) {
  if (writer) {
    R.group[slot] = g;
    R.key[slot] = key;
  }
  return R.data + (size_t)slot * (size_t)R.width;
}
// The same for a team that is a whole BLOCK (every thread calls; `slot` is a word of the block's LDS; two block barriers).
__device__ __forceinline__ double* spec_rec_open_block(const SpecRecs& R, int32_t g, unsigned long long key, unsigned int* slot) {
  if (threadIdx.x == 0) {
    unsigned int r = R.nstatic + atomicAdd(R.count + R.parity, 1u);
    if (r < R.capacity) {
      R.group[r] = g;
      R.key[r] = key;
    } else {
      R.count[2 + R.parity] = 1u;
      r = R.capacity;
    }
    *slot = r;
  }
  __syncthreads();
  const unsigned int r = *slot;
  __syncthreads();
  return R.data + (size_t)r * (size_t)R.width;
}

// power[f][g][k] (+)= the records of group g added in key order; one block per (group, 256 values).  The block scans the slot
// headers (coalesced, a few KB), collects its group's (key, slot) pairs in the LDS, sorts them, and every thread adds its
// value down the sorted list, quarter by quarter (below).  More records than the LDS list holds: further rounds, each taking the next LIST keys in
// order (the sum still runs in key order).
constexpr int SPEC_CLOSE_LIST = 2048;
constexpr int SPEC_CLOSE_BATCH = 16;  // headers / record values a thread asks for before it looks at any of them
__global__ void __launch_bounds__(256) spec_close_kernel(SpecRecs R, int32_t ngroup, int32_t nk, int32_t nfield, double* __restrict__ power0,
                                                        double* __restrict__ power1, int32_t accumulate, double* __restrict__ tail_out = nullptr,
                                                        int32_t ntail = 0, int32_t ntail_out = 0) {
  __shared__ unsigned long long keys[SPEC_CLOSE_LIST];
  __shared__ unsigned int slots[SPEC_CLOSE_LIST];
  __shared__ unsigned int n_list, n_more;
  const int32_t g = (int32_t)blockIdx.x;
  const int tid = (int)threadIdx.x;
  const unsigned int dyn = R.count[R.parity];
  const unsigned int taken = R.nstatic + dyn < R.capacity ? R.nstatic + dyn : R.capacity;
  const bool overflow = R.count[2 + R.parity] != 0u;
  // (r6) `ntail` more values behind the fields of a record (wbx_det_spectrum_folded: the deterministic sums of the record's rows), the
  // first `ntail_out` of them go to tail_out[g][..]; the padding is added like everything else and dropped
  const int nval = nk * nfield + ntail;
  __shared__ double tail_sink[256];
  auto out_of = [&](int v) -> double* {
    if (v >= nk * nfield) {
      const int j = v - nk * nfield;
      return j < ntail_out ? tail_out + (int64_t)g * ntail_out + j : tail_sink + threadIdx.x;
    }
    return (v < nk ? power0 : power1) + (int64_t)g * nk + (v < nk ? v : v - nk);
  };
  // the slots whose header names this group and whose key is in [floor, below): appended to the LDS list (past LIST entries they
  // are only counted).  The headers are read BATCH at a time into registers -- the loads of a batch are in flight together.
  auto collect = [&](unsigned long long floor_key, unsigned long long below) {
    for (unsigned int base = 0; base < taken; base += 256 * SPEC_CLOSE_BATCH) {
      int32_t gr[SPEC_CLOSE_BATCH];
#pragma unroll
      for (int i = 0; i < SPEC_CLOSE_BATCH; ++i) {
        const unsigned int r = base + tid + 256u * i;
        gr[i] = r < taken ? R.group[r] : -1;
      }
#pragma unroll
      for (int i = 0; i < SPEC_CLOSE_BATCH; ++i) {
        if (gr[i] == g) {
          const unsigned int r = base + tid + 256u * i;
          const unsigned long long k = R.key[r];
          if (k >= floor_key && k < below) {
            const unsigned int at = atomicAdd(&n_list, 1u);
            if (at < SPEC_CLOSE_LIST) {
              keys[at] = k;
              slots[at] = r;
            } else {
              atomicAdd(&n_more, 1u);
            }
          }
        }
      }
    }
  };
  unsigned long long floor_key = 0ull;  // keys below it have been added in earlier rounds
  bool first_round = true;
  while (true) {
    if (tid == 0) {
      n_list = 0u;
      n_more = 0u;
    }
    __syncthreads();
    collect(floor_key, ~0ull);
    __syncthreads();
    const unsigned int more = n_more;
    unsigned int n = n_list < SPEC_CLOSE_LIST ? n_list : SPEC_CLOSE_LIST;
    unsigned long long next_floor = ~0ull;
    if (more) {
      // (rare: one group holds more than LIST records)  This round takes the keys in [floor, T): T = the largest bound with at
      // most LIST keys below it, found by bisection over the key space; the list collected above is discarded.
      unsigned long long lo = floor_key, hi = ~0ull;
      while (hi - lo > 1ull) {
        const unsigned long long mid = lo + (hi - lo) / 2ull;
        __syncthreads();
        if (tid == 0) {
          n_list = 0u;
          n_more = 0u;
        }
        __syncthreads();
        collect(floor_key, mid);
        __syncthreads();
        if (n_more == 0u) lo = mid; else hi = mid;
      }
      next_floor = lo;
      __syncthreads();
      if (tid == 0) {
        n_list = 0u;
        n_more = 0u;
      }
      __syncthreads();
      collect(floor_key, lo);
      __syncthreads();
      n = n_list;
    }
    if (n <= 256u) {
      // (the usual case) rank sort: thread i counts the keys below its own (keys are unique) and puts its slot there
      const unsigned long long mine = tid < (int)n ? keys[tid] : 0ull;
      const unsigned int my_slot = tid < (int)n ? slots[tid] : 0u;
      unsigned int rank = 0u;
      if (tid < (int)n) {
        for (unsigned int j = 0; j < n; ++j) rank += keys[j] < mine ? 1u : 0u;
      }
      __syncthreads();
      if (tid < (int)n) slots[rank] = my_slot;
      __syncthreads();
    } else {
      // bitonic sort of (keys, slots)[0, n) by key, padded to a power of two with the largest key
      unsigned int np = 1u;
      while (np < n) np <<= 1;
      for (unsigned int i = n + tid; i < np; i += 256) {
        keys[i] = ~0ull;
        slots[i] = 0u;
      }
      __syncthreads();
      for (unsigned int size = 2u; size <= np; size <<= 1) {
        for (unsigned int stride = size >> 1; stride > 0u; stride >>= 1) {
          for (unsigned int i = tid; i < np; i += 256) {
            const unsigned int j = i ^ stride;
            if (j > i) {
              const bool up = (i & size) == 0u;
              const unsigned long long a = keys[i], b = keys[j];
              if ((a > b) == up) {
                keys[i] = b;
                keys[j] = a;
                const unsigned int t = slots[i];
                slots[i] = slots[j];
                slots[j] = t;
              }
            }
          }
          __syncthreads();
        }
      }
    }
    // The block's 256 values (blockIdx.y), one per thread.  The sum of a value is a fixed function of the record SET: the sorted
    // list is cut in four quarters [n w / 4, n (w + 1) / 4), each quarter is added in list order (BATCH records' values asked for at
    // a time), and the four partial sums are added in quarter order.  (Rounds 5-6 gave a block 64 values and each of its four waves
    // one quarter -- the same association, so the same bits --; with four times the values per block the slot headers are scanned
    // by a quarter of the blocks: configs[4]'s close, 740 groups x 1450 values, 17 020 -> 4 440 blocks.)  Between rounds the running
    // sums live in the output (this block is their only reader / writer).
    {
      const int v = (int)blockIdx.y * 256 + tid;
      double* const dst = v < nval ? out_of(v) : nullptr;
      if (dst) {
        double total = (first_round && !accumulate) ? 0.0 : *dst;
#pragma unroll 1
        for (int wv = 0; wv < 4; ++wv) {
          const unsigned int q_lo = (unsigned int)(((unsigned long long)n * (unsigned)wv) >> 2);
          const unsigned int q_hi = (unsigned int)(((unsigned long long)n * (unsigned)(wv + 1)) >> 2);
          double sum = 0.0;
          for (unsigned int q0 = q_lo; q0 < q_hi; q0 += SPEC_CLOSE_BATCH) {
            double x[SPEC_CLOSE_BATCH];
#pragma unroll
            for (int i = 0; i < SPEC_CLOSE_BATCH; ++i) x[i] = q0 + i < q_hi ? R.data[(size_t)slots[q0 + i] * (size_t)R.width + v] : 0.0;
#pragma unroll
            for (int i = 0; i < SPEC_CLOSE_BATCH; ++i) sum += x[i];  // (+ 0.0 past the end of the list leaves every sum as it is)
          }
          total += sum;
        }
        *dst = (overflow && !more) ? __builtin_nan("") : total;
      }
      __syncthreads();  // the list is rebuilt by the next round
    }
    if (!more) break;
    floor_key = next_floor;
    first_round = false;
  }
  // the other parity's counters are the next launch's: clear them (nobody reads them now; stream order puts this in front of
  // that launch)
  if (g == 0 && blockIdx.y == 0 && tid == 0) {
    R.count[1 - R.parity] = 0u;
    R.count[2 + (1 - R.parity)] = 0u;
  }
}

// group changes along rows [0, n): #{i >= 1 : group[i] != group[i - 1]}
__global__ void __launch_bounds__(256) spec_changes_kernel(const int32_t* __restrict__ group, int64_t n, unsigned long long* __restrict__ out) {
  unsigned long long mine = 0ull;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x + 1; i < n; i += (int64_t)gridDim.x * 256)
    mine += group[i] != group[i - 1] ? 1ull : 0ull;
  if (mine) atomicAdd(out, mine);
}

static std::mutex g_changes_lock;
static std::map<std::pair<const void*, int64_t>, int64_t> g_changes;  // (group table, rows) -> group changes

// Every write through this library into [dst, dst + bytes) drops the cached counts of tables inside it (wbx_ctx.hip calls it).
void spectrum_note_write(const void* dst, size_t bytes) {
  std::lock_guard<std::mutex> hold(g_changes_lock);
  if (g_changes.empty()) return;
  const char* lo = reinterpret_cast<const char*>(dst);
  for (auto it = g_changes.begin(); it != g_changes.end();) {
    const char* p = reinterpret_cast<const char*>(it->first.first);
    if (p + (size_t)it->first.second * sizeof(int32_t) > lo && p < lo + bytes)
      it = g_changes.erase(it);
    else
      ++it;
  }
}

static int spec_group_changes(wbx_ctx* ctx, const int32_t* group, int64_t nrows, int64_t* out) {
  {
    std::lock_guard<std::mutex> hold(g_changes_lock);
    auto it = g_changes.find(std::make_pair((const void*)group, nrows));
    if (it != g_changes.end()) {
      *out = it->second;
      return 0;
    }
  }
  unsigned long long* d = nullptr;
  WBX_HIP(hipMalloc(reinterpret_cast<void**>(&d), sizeof(unsigned long long)));
  WBX_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream));
  const unsigned blocks = (unsigned)((nrows + 255) / 256 < 1024 ? (nrows + 255) / 256 : 1024);
  if (nrows > 1) hipLaunchKernelGGL(spec_changes_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, ctx->stream, group, nrows, d);
  unsigned long long h = 0;
  WBX_HIP(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  WBX_HIP(hipStreamSynchronize(ctx->stream));
  WBX_HIP(hipFree(d));
  *out = (int64_t)h;
  std::lock_guard<std::mutex> hold(g_changes_lock);
  if (g_changes.size() > 256) g_changes.clear();
  g_changes[std::make_pair((const void*)group, nrows)] = (int64_t)h;
  return 0;
}

// The launch's record store: `teams` one-record-at-the-end teams (or blocks), `extra` further records the kernel may open
// (2 x group changes for contiguous walkers).  Grows the context's buffers when needed; alternates the counter parity.
static int spec_recs_prepare(wbx_ctx* ctx, FftState* st, int64_t nstatic, int64_t dynamic, int32_t width, SpecRecs* out) {
  const int64_t cap = nstatic + dynamic + 64;
  WBX_REQUIRE(cap < ((int64_t)1 << 31), "too many spectrum records (%lld)", (long long)cap);
  SpecRecs& R = st->recs;
  const size_t need = (size_t)(cap + 1) * (size_t)width * sizeof(double);
  if (!R.count) {
    WBX_HIP(hipMalloc(reinterpret_cast<void**>(&R.count), 4 * sizeof(unsigned int)));
    WBX_HIP(hipMemsetAsync(R.count, 0, 4 * sizeof(unsigned int), ctx->stream));
  }
  if (R.capacity < (unsigned int)cap || st->recs_bytes < need) {
    WBX_HIP(hipStreamSynchronize(ctx->stream));
    if (R.data) (void)hipFree(R.data);
    if (R.group) (void)hipFree(R.group);
    if (R.key) (void)hipFree(R.key);
    const int64_t grown = (cap > (int64_t)R.capacity ? cap : (int64_t)R.capacity) + cap / 4;
    const size_t bytes = (need > st->recs_bytes ? need : st->recs_bytes) + need / 4;
    R.data = nullptr;
    R.group = nullptr;
    R.key = nullptr;
    WBX_HIP(hipMalloc(reinterpret_cast<void**>(&R.data), bytes));
    WBX_HIP(hipMalloc(reinterpret_cast<void**>(&R.group), (size_t)grown * sizeof(int32_t)));
    WBX_HIP(hipMalloc(reinterpret_cast<void**>(&R.key), (size_t)grown * sizeof(unsigned long long)));
    st->recs_bytes = bytes;
    R.capacity = (unsigned int)grown;
  }
  R.parity ^= 1;
  *out = R;
  out->width = width;
  out->nstatic = (unsigned int)nstatic;
  out->capacity = (unsigned int)cap;  // (the headers hold `grown` >= cap slots, the values at least (cap + 1) x width doubles)
  return 0;
}

static int spec_close(wbx_ctx* ctx, const SpecRecs& R, int32_t ngroup, int32_t nk, int32_t nfield, double* power0, double* power1,
                      int32_t accumulate, double* tail_out = nullptr, int32_t ntail = 0, int32_t ntail_out = 0) {
  if (ngroup <= 0) return 0;
  hipLaunchKernelGGL(spec_close_kernel, dim3((unsigned)ngroup, (unsigned)((nk * nfield + ntail + 255) / 256)), dim3(256), 0, ctx->stream, R,
                     ngroup, nk, nfield, power0, power1, accumulate, tail_out, ntail, ntail_out);
  WBX_HIP(hipGetLastError());
  return 0;
}

// One block = a run of rows x ALL wavenumbers (256 at a time); consecutive rows of the same group are summed in registers and
// stored as one record per (group change) of the block.  F is [rows][nk] interleaved complex (coalesced along k).
__global__ void __launch_bounds__(256) power_kernel(const float2* __restrict__ F, int64_t row0, int64_t nrows_tile,
                                                    int rows_per_block, int nk, int nlon,
                                                    const int32_t* __restrict__ group, const double* __restrict__ scale,
                                                    SpecRecs recs, int64_t team_base) {
  __shared__ unsigned int rec_slot;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r_end = r_begin + rows_per_block < nrows_tile ? r_begin + rows_per_block : nrows_tile;
  const int64_t team = team_base + blockIdx.x;
  if (r_begin >= r_end) {  // (block-uniform)
    spec_rec_static(recs, (unsigned int)team, -1, 0ull, threadIdx.x == 0);
    return;
  }
  unsigned int seq = 0;
  // runs of rows of one group: [ra, rb) -> one record, all wavenumbers; the block's LAST run goes into its reserved slot
  int64_t ra = r_begin;
  while (ra < r_end) {
    const int32_t cur = group[row0 + ra];
    int64_t rb = ra + 1;
    while (rb < r_end && group[row0 + rb] == cur) ++rb;
    double* const rec = rb == r_end ? spec_rec_static(recs, (unsigned int)team, cur, spec_key(team, 0xffffffu), threadIdx.x == 0)
                                    : spec_rec_open_block(recs, cur, spec_key(team, seq++), &rec_slot);
    for (int k = (int)threadIdx.x; k < nk; k += 256) {
      const double norm = 1.0 / ((double)nlon * (double)nlon) * (k == 0 ? 1.0 : 2.0);
      double acc = 0.0;
      for (int64_t r = ra; r < rb; ++r) {
        const float2 f = F[r * nk + k];
        const double re = (double)f.x, im = (double)f.y;
        acc += (re * re + im * im) * norm * scale[row0 + r];
      }
      rec[k] = acc;
    }
    ra = rb;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Fused path: row -> LDS -> mixed-radix FFT -> |F|^2 * scale -> per-group sums, reading the field ONCE.
//
// The rocFFT route moves 5x the field through HBM (R2C = complex FFT of length n/2 writing a temporary + a
// post-processing kernel writing the Hermitian output, then power_kernel reading it back).  Here ONE WAVE owns one
// row at a time: the n reals are viewed as n/2 complex numbers in a wave-private LDS buffer and transformed by
// Stockham passes of radix 4 / 2 / 5 / 3.  In a pass every lane first reads the inputs of all its butterflies into
// registers, then writes the outputs back into the SAME buffer -- LDS executes a wave's instructions in order, so no
// barrier and no ping-pong buffer are needed (the 4 waves of a block only share the twiddle tables).  The n/2 + 1
// Hermitian coefficients come from the usual even/odd split, and |F_k|^2 * (k ? 2 : 1) / n^2 * scale[row] is added to
// fp64 accumulators in registers (lane l owns wavenumbers l, l + 64, ...).  Consecutive rows of a wave normally belong
// to one group (lead, level): the accumulators are flushed with fp64 atomics only when the group changes.  Used when
// rows are contiguous (lon_stride == 1, 8-byte aligned), n <= 2048 and n / 2 has no prime factor above 5; everything
// else takes the rocFFT route.
struct FusedSpec {
  int n, n2, npass;
  int radix[16];
  // per pass, worked out on the host (the device has no integer divide: each n2 / (ns R) or ns / R in the pass loop was a
  // ~30-instruction sequence, as much as the butterfly itself): input stride, 1 / ns, and where the pass's twiddles
  // start in the packed table: exp(-2 pi i k t / (ns R)) sits at toff + (t - 1) ns + k, so the lanes of a wave
  // (consecutive k) read consecutive entries -- the one shared exp(-2 pi i m / n2) table was read at stride t k n2/(ns R),
  // an up to 8-way bank conflict per twiddle load.  The passes need n2 - 1 entries in all.
  int ns[16], toff[16];
  float inv_ns[16];
};

// A wave transforms TWO rows at once: every quantity is a pair (row A, row B) in one 64-bit register pair, so the
// butterflies and twiddle products compile to packed fp32 instructions (v_pk_add / v_pk_mul / v_pk_fma_f32) and both rows
// share the twiddle loads, the index arithmetic and the LDS instructions (one ds_read_b128 / ds_write_b128 per point).
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
struct C2 {
  v2 re, im;  // (row A, row B)
};
__device__ __forceinline__ C2 cadd(C2 a, C2 b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ C2 csub(C2 a, C2 b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ C2 mul_mi(C2 a) { return {a.im, -a.re}; }  // a * (-i)
__device__ __forceinline__ C2 mul_pi(C2 a) { return {-a.im, a.re}; }  // a * (+i)
__device__ __forceinline__ C2 cscale(C2 a, float s) { return {a.re * s, a.im * s}; }
// (r6) Where TWO products meet in one sum, which of them the compiler fuses with the addition (a rounding each way) depended on
// the shape of the surrounding code: grouping a stage's LDS reads changed the last fp32 bit of the fused sweep's spectra against
// the plain kernel's.  The product that is to STAY a product is formed by `prod` (no contraction allowed on it); the other one
// then is the only candidate and fuses.  Sums with a single product were never ambiguous.
__device__ __forceinline__ v2 prod(v2 a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ v2 prod(v2 a, v2 b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ C2 ctw(C2 a, float2 w) {  // a * (w.x + i w.y), the same twiddle for both rows
  return {a.re * w.x - prod(a.im, w.y), a.im * w.x + prod(a.re, w.y)};
}
__device__ __forceinline__ v2 norm2(C2 x) { return x.re * x.re + prod(x.im, x.im); }  // |x|^2 of both rows
__device__ __forceinline__ C2 ld_c2(const v4* p) {
  const v4 q = *p;
  return {{q.x, q.y}, {q.z, q.w}};
}
__device__ __forceinline__ void st_c2(v4* p, C2 a) { *p = (v4){a.re.x, a.re.y, a.im.x, a.im.y}; }

// WBX_SPECTRUM_DEMEAN (what it is for: wbx_zspec1440.hpp): rows are shifted by an estimate of their mean in front of the fp32
// transform and F_0 is restored in fp64 -- the 1440-point kernels, and the one-wave teams (G == 64: rows of up to 256 points,
// i.e. the 64- and 240-point grids of the public configs) of the generic kernel below, where the team's mean is one DPP
// reduction; teams of two or four waves (257..2048-point rows other than 1440) transform the rows as they are.
#ifndef WBX_SPECTRUM_DEMEAN
#define WBX_SPECTRUM_DEMEAN 1
#endif

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_moved_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}

__device__ __forceinline__ float wave_sum_uniform_f32(float v) {  // all 64 lanes active; steps as wave_sum_lane63 (wbx_common.hpp)
  v += dpp_moved_f32<0xB1, 0xf>(v);
  v += dpp_moved_f32<0x4E, 0xf>(v);
  v += dpp_moved_f32<0x124, 0xf>(v);
  v += dpp_moved_f32<0x128, 0xf>(v);
  v += dpp_moved_f32<0x142, 0xa>(v);
  v += dpp_moved_f32<0x143, 0xc>(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <int R>
__device__ __forceinline__ void butterfly(C2 (&v)[R]) {
  if constexpr (R == 2) {
    const C2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  } else if constexpr (R == 3) {
    const C2 t1 = cadd(v[1], v[2]);
    const C2 t2 = csub(v[0], cscale(t1, 0.5f));
    const C2 t3 = cscale(csub(v[1], v[2]), 0.8660254037844386f);
    v[0] = cadd(v[0], t1);
    v[1] = cadd(t2, mul_mi(t3));
    v[2] = cadd(t2, mul_pi(t3));
  } else if constexpr (R == 4) {
    const C2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]), t3 = csub(v[1], v[3]);
    v[0] = cadd(t0, t2);
    v[2] = csub(t0, t2);
    v[1] = cadd(t1, mul_mi(t3));
    v[3] = cadd(t1, mul_pi(t3));
  } else {  // R == 5
    constexpr float c1 = 0.30901699437494745f, c2 = -0.8090169943749475f, s1 = 0.9510565162951535f,
                    s2 = 0.5877852522924731f;
    const C2 a = v[0];
    const C2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    v[0] = cadd(a, cadd(t1, t2));
    // (a + c2 t2) + c1 t1: two fused multiply-adds, each with ONE product (r6; it was a + (c1 t1 + c2 t2): a product, a fused
    // multiply-add whose pairing was the compiler's, and an addition)
    const C2 m1 = cadd(cscale(t1, c1), cadd(cscale(t2, c2), a));
    const C2 m2 = cadd(cscale(t1, c2), cadd(cscale(t2, c1), a));
    const C2 n1 = {t3.re * s1 + prod(t4.re, s2), t3.im * s1 + prod(t4.im, s2)};   // s1 t3 + s2 t4
    const C2 n2v = {t3.re * s2 - prod(t4.re, s1), t3.im * s2 - prod(t4.im, s1)};  // s2 t3 - s1 t4
    v[1] = cadd(m1, mul_mi(n1));
    v[4] = cadd(m1, mul_pi(n1));
    v[2] = cadd(m2, mul_mi(n2v));
    v[3] = cadd(m2, mul_pi(n2v));
  }
}

// G threads (one wave, or a block of 2 / 4 waves) work on one pair of rows at a time.
template <int G>
__device__ __forceinline__ void team_sync() {
  if constexpr (G == 64)
    __builtin_amdgcn_wave_barrier();  // same wave: LDS keeps program order
  else
    __syncthreads();
}

// Sum of a per-thread (row A, row B) pair over the team; every thread gets it.  Teams of 2 / 4 waves exchange their wave sums
// through the first bytes of the team's transform buffer, which is idle between the last read of the previous row pair (a
// team_sync closes it) and the first pass's stores: two more block barriers per row pair.  All threads of the team must call.
template <int G>
__device__ __forceinline__ v2 team_total_f32(v2 s, v4* buf, int tid) {
  const v2 w = {wave_sum_uniform_f32(s.x), wave_sum_uniform_f32(s.y)};
  if constexpr (G == 64) {
    return w;
  } else {
    float2* sc = reinterpret_cast<float2*>(buf);
    if ((tid & 63) == 0) sc[tid >> 6] = make_float2(w.x, w.y);
    __syncthreads();
    v2 tot = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < G / 64; ++i) {
      const float2 q = sc[i];
      tot.x += q.x;
      tot.y += q.y;
    }
    __syncthreads();  // ... before anybody stores the first pass's outputs over them
    return tot;
  }
}

// One in-place pass of radix R over the team's n2 points (two rows each); thread t owns butterflies t, t + G, ... (at
// most NB of them).  The passes are the TRANSPOSED Stockham flow graph (the DFT matrix is symmetric, so running the
// transposed passes in reverse order is the same transform): butterfly j = q * ns + k gathers its inputs at stride ns,
// (j - k) * R + k + t * ns, multiplies OUTPUT t by exp(-2 pi i k t / (ns R)) and stores it at j + t * nb -- consecutive
// lanes store consecutive points.  LDS stores are the expensive direction on CDNA4 (ds_write_b128 ~13 cycles against 4 for
// a read, and an N-way bank conflict multiplies that), so the strided side is the read.  ns shrinks from n2 / R to 1.
// FIRST (ns == nb): the inputs k + t * nb are taken straight from the two rows in global memory (coalesced), the raw
// rows never visit the LDS.  All LDS reads precede all writes.
// FIRST with WBX_SPECTRUM_DEMEAN: the rows are shifted by the mean of their even points (the real parts of the packed row:
// every one of them is in some thread's registers here; teams of 2 / 4 waves add their wave sums through the LDS,
// team_total_f32) before the first butterfly; the shift comes back in *msh for the k = 0 term of the unpack.
template <int R, int NB, int G, bool FIRST>
__device__ __forceinline__ void team_pass(v4* __restrict__ buf, const float2* __restrict__ tw, int n2, int ns, int toff,
                                          float inv_ns, int tid, const v2* __restrict__ rowa,
                                          const v2* __restrict__ rowb, bool two, v2* msh) {
  const int nb = n2 / R;  // exp(-2 pi i t k / (ns R)) = tw[toff + (t - 1) ns + k]
  C2 v[NB][R];
  int kk[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int j = tid + G * i;
    kk[i] = j;
    if (j < nb) {
      int k = j, base = j;
      if constexpr (!FIRST) {
        const int q = (int)(((float)j + 0.5f) * inv_ns);  // j / ns, exact for j < 2^22
        k = j - q * ns;
        base = (j - k) * R + k;
      }
      kk[i] = k;
#pragma unroll
      for (int t = 0; t < R; ++t) {
        if constexpr (FIRST) {
          const v2 a = __builtin_nontemporal_load(rowa + j + t * nb);
          const v2 b = two ? __builtin_nontemporal_load(rowb + j + t * nb) : (v2){0.f, 0.f};
          v[i][t] = {{a.x, b.x}, {a.y, b.y}};
        } else {
          v[i][t] = ld_c2(buf + base + t * ns);
        }
      }
    }
  }
  if constexpr (FIRST && WBX_SPECTRUM_DEMEAN) {
    v2 sum = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (tid + G * i < nb) {
#pragma unroll
        for (int t = 0; t < R; ++t) sum += v[i][t].re;
      }
    }
    const float inv_n2 = 1.0f / (float)n2;
    const v2 m = team_total_f32<G>(sum, buf, tid) * inv_n2;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (tid + G * i < nb) {
#pragma unroll
        for (int t = 0; t < R; ++t) {
          v[i][t].re -= m;
          v[i][t].im -= m;
        }
      }
    }
    *msh = m;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int j = tid + G * i;
    if (j < nb) {
      const int k = kk[i];
      butterfly<R>(v[i]);
      if (ns > 1) {
        const float2* twk = tw + toff + k;
#pragma unroll
        for (int t = 1; t < R; ++t) v[i][t] = ctw(v[i][t], twk[(t - 1) * ns]);
      }
    }
  }
  if constexpr (!FIRST) team_sync<G>();  // reads above, writes below
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int j = tid + G * i;
    if (j < nb) {
#pragma unroll
      for (int t = 0; t < R; ++t) st_c2(buf + j + t * nb, v[i][t]);
    }
  }
  team_sync<G>();
}

template <int R, int G, bool FIRST>
__device__ __forceinline__ void team_pass_any(v4* buf, const float2* tw, int n2, int ns, int toff, float inv_ns, int tid,
                                              const v2* rowa, const v2* rowb, bool two, v2* msh) {
  // fused_factor() admits at most 256 butterflies per pass
  constexpr int NBMAX = 256 / G;
  const int nbl = (n2 / R + G - 1) / G;  // butterflies per thread
  if constexpr (NBMAX >= 4) {
    if (nbl > 3) return team_pass<R, 4, G, FIRST>(buf, tw, n2, ns, toff, inv_ns, tid, rowa, rowb, two, msh);
    if (nbl > 2) return team_pass<R, 3, G, FIRST>(buf, tw, n2, ns, toff, inv_ns, tid, rowa, rowb, two, msh);
  }
  if constexpr (NBMAX >= 2) {
    if (nbl > 1) return team_pass<R, 2, G, FIRST>(buf, tw, n2, ns, toff, inv_ns, tid, rowa, rowb, two, msh);
  }
  team_pass<R, 1, G, FIRST>(buf, tw, n2, ns, toff, inv_ns, tid, rowa, rowb, two, msh);
}

template <int G, bool FIRST>
__device__ __forceinline__ void team_pass_radix(const FusedSpec& fs, int p, v4* buf, const float2* tw, int tid,
                                                const v2* rowa, const v2* rowb, bool two, v2* msh = nullptr) {
  const int rdx = fs.radix[p], n2 = fs.n2, ns = fs.ns[p], toff = fs.toff[p];
  const float inv_ns = fs.inv_ns[p];
  if (rdx == 4)
    team_pass_any<4, G, FIRST>(buf, tw, n2, ns, toff, inv_ns, tid, rowa, rowb, two, msh);
  else if (rdx == 2)
    team_pass_any<2, G, FIRST>(buf, tw, n2, ns, toff, inv_ns, tid, rowa, rowb, two, msh);
  else if (rdx == 3)
    team_pass_any<3, G, FIRST>(buf, tw, n2, ns, toff, inv_ns, tid, rowa, rowb, two, msh);
  else
    team_pass_any<5, G, FIRST>(buf, tw, n2, ns, toff, inv_ns, tid, rowa, rowb, two, msh);
}

// tw_pass = the packed per-pass twiddles (n2 - 1 entries, FusedSpec::toff);  tw_real[k] = exp(-2 pi i k / n), k <= n2 / 2 (only those are copied to the LDS).  KPT >= ceil((n2 / 2 + 1) / G).
// G = 64: a block is 4 independent one-wave teams sharing the twiddle tables; G = 128 / 256: the block IS the team
// (its __syncthreads are team barriers), sweeping its own run of row pairs.
// R0 > 0 (G == 256 only: one first-pass butterfly per thread): the first pass has radix R0 and its inputs -- the raw
// rows -- are fetched into registers one row pair AHEAD, right after the previous pair's first pass has consumed them, so
// the HBM latency runs under the remaining passes instead of in front of every pair.
template <int KPT, int G, int R0>
__global__ void __launch_bounds__(256) zspec_fused_kernel(const float* __restrict__ field, int64_t row_stride, int64_t nrows,
                                                          int rows_per_team, FusedSpec fs,
                                                          const float2* __restrict__ tw_pass_g,
                                                          const float2* __restrict__ tw_real_g,
                                                          const int32_t* __restrict__ group,
                                                          const double* __restrict__ scale, SpecRecs recs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ unsigned int rec_slot;  // (teams of a whole block: the record slot thread 0 took, see spec_rec_open_block)
  constexpr int NTEAM = G == 64 ? 4 : 1;
  const int n2 = fs.n2;
  float2* tw_pass = reinterpret_cast<float2*>(lds_raw);
  float2* tw_real = tw_pass + n2;
  // team index in an SGPR: the row bookkeeping (group / scale look-ups, row pointers) then compiles to scalar loads on
  // lgkmcnt, which do not drain the vector-memory counter the prefetched rows are in flight on
  const int tid = G == 64 ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
  const int team = G == 64 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  const int ntr = n2 / 2 + 1;  // the mirrored unpack needs k <= n2 / 2 only
  v4* buf = reinterpret_cast<v4*>(tw_real + ntr + (ntr & 1)) + (int64_t)team * n2;
  for (int i = threadIdx.x; i < n2; i += blockDim.x) tw_pass[i] = tw_pass_g[i];
  for (int i = threadIdx.x; i < ntr; i += blockDim.x) tw_real[i] = tw_real_g[i];
  __syncthreads();
  const int64_t w = (int64_t)blockIdx.x * NTEAM + team;
  const int64_t r0 = w * rows_per_team;
  int64_t r1 = r0 + rows_per_team < nrows ? r0 + rows_per_team : nrows;
  if (r0 >= r1) {  // G == 64: only wave-level syncs follow; G > 64: the whole block leaves together
    spec_rec_static(recs, (unsigned int)w, -1, 0ull, tid == 0);  // (the team's reserved slot stays empty)
    return;
  }
  const double inv_nn = 1.0 / ((double)fs.n * (double)fs.n);
  const int nh = n2 / 2;  // a thread owns wavenumbers k = tid + G i <= nh and their mirrors n2 - k
  double acc[KPT], accm[KPT];
#pragma unroll
  for (int i = 0; i < KPT; ++i) acc[i] = accm[i] = 0.0;
  int32_t cur = group[r0];
  unsigned int seq = 0;  // the team's records carry the key (team, sequence number): spec_close_kernel adds them in that order
  auto open = [&](int32_t g) -> double* {
    if constexpr (G == 64)
      return spec_rec_open(recs, g, spec_key(w, seq++), tid);
    else
      return spec_rec_open_block(recs, g, spec_key(w, seq++), &rec_slot);
  };
  auto flush = [&](int32_t next, bool last = false) {  // (the team's LAST record sits in the slot the launch reserved for it)
    double* const rec = last ? spec_rec_static(recs, (unsigned int)w, cur, spec_key(w, 0xffffffu), tid == 0) : open(cur);
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int k = tid + G * i;
      if (k <= nh) {  // every wavenumber of the record is written once (k and its mirror n2 - k)
        rec[k] = k == 0 ? acc[i] : 2.0 * acc[i];
        if (n2 - k != k) rec[n2 - k] = 2.0 * accm[i];
      }
      acc[i] = accm[i] = 0.0;
    }
    cur = next;
  };
  constexpr int RP = R0 > 0 ? R0 : 1;
  v2 pa[RP], pb[RP];  // prefetched inputs of this thread's first-pass butterfly (rows A and B)
  const int nb0 = R0 > 0 ? n2 / RP : 0;
  auto fetch = [&](int64_t r) {
    const bool two = r + 1 < r1;
    const v2* rowa = reinterpret_cast<const v2*>(field + r * row_stride);
    const v2* rowb = reinterpret_cast<const v2*>(field + (two ? r + 1 : r) * row_stride);
    const int j = tid < nb0 ? tid : nb0 - 1;  // idle threads re-read a valid point: no branch around the loads
#pragma unroll
    for (int t = 0; t < RP; ++t) {
      pa[t] = __builtin_nontemporal_load(rowa + j + t * nb0);
      pb[t] = __builtin_nontemporal_load(rowb + j + t * nb0);
    }
  };
  if constexpr (R0 > 0) fetch(r0);
  for (int64_t r = r0; r < r1; r += 2) {
    const bool two = r + 1 < r1;  // team-uniform; a missing second row is a row of zeros with scale 0
    const int32_t ga = group[r], gb = two ? group[r + 1] : ga;
    const v2* rowa = reinterpret_cast<const v2*>(field + r * row_stride);
    const v2* rowb = reinterpret_cast<const v2*>(field + (two ? r + 1 : r) * row_stride);
    v2 msh = {0.f, 0.f};  // the shift of rows A and B (WBX_SPECTRUM_DEMEAN), team-uniform
    if constexpr (R0 > 0) {
      C2 v[RP];
#pragma unroll
      for (int t = 0; t < RP; ++t) v[t] = {{pa[t].x, two ? pb[t].x : 0.f}, {pa[t].y, two ? pb[t].y : 0.f}};
      if (r + 2 < r1) fetch(r + 2);
      if constexpr (WBX_SPECTRUM_DEMEAN) {  // the rows shifted by the mean of their even points, as in team_pass
        v2 sum = {0.f, 0.f};
        if (tid < nb0) {
#pragma unroll
          for (int t = 0; t < RP; ++t) sum += v[t].re;
        }
        msh = team_total_f32<G>(sum, buf, tid) * (1.0f / (float)n2);
#pragma unroll
        for (int t = 0; t < RP; ++t) {
          v[t].re -= msh;
          v[t].im -= msh;
        }
      }
      if (tid < nb0) {
        butterfly<RP>(v);
        if (nb0 > 1) {  // ns == nb0, k == tid; the first pass's twiddles open the packed table
#pragma unroll
          for (int t = 1; t < RP; ++t) v[t] = ctw(v[t], tw_pass[(t - 1) * nb0 + tid]);
        }
#pragma unroll
        for (int t = 0; t < RP; ++t) st_c2(buf + tid + t * nb0, v[t]);
      }
      team_sync<G>();
    } else {
      team_pass_radix<G, true>(fs, 0, buf, tw_pass, tid, rowa, rowb, two, &msh);
    }
    for (int p = 1; p < fs.npass; ++p) team_pass_radix<G, false>(fs, p, buf, tw_pass, tid, rowa, rowb, two);
    // Hermitian unpack of the half-length transform Z: X_k = E_k + W^k O_k with W = exp(-2 pi i / n),
    // E_k = (Z_k + conj Z_{n2-k}) / 2, O_k = (Z_k - conj Z_{n2-k}) / (2i), k = 0..n2 (Z_{n2} = Z_0).  The mirrored
    // coefficient comes from the same two points: X_{n2-k} = conj(E_k - W^k O_k), so a thread takes k <= n2 / 2 and
    // n2 - k together (half the LDS reads and twiddle products).  |X|^2 in packed fp32 (the transform itself is fp32),
    // widened once per row for the fp64 sums; the factor 2 of k > 0 is applied when the sums are flushed.
    const double sca = scale[r] * inv_nn, scb = two ? scale[r + 1] * inv_nn : 0.0;
    if (ga != cur) flush(ga);  // team-uniform
    const bool split = gb != ga;  // the pair straddles a group boundary (rare): row B's values are a record of their own
    double* const rec_b = split ? open(gb) : nullptr;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int k = tid + G * i;
      if (k <= nh) {
        const int km = n2 - k;
        const C2 zk = ld_c2(buf + k);
        const C2 zc = ld_c2(buf + (k == 0 ? 0 : km));
        const C2 e = {(zk.re + zc.re) * 0.5f, (zk.im - zc.im) * 0.5f};
        const C2 o = {(zk.im + zc.im) * 0.5f, (zc.re - zk.re) * 0.5f};
        const C2 wo = ctw(o, tw_real[k]);
        const C2 x = cadd(e, wo), xm = csub(e, wo);
        const v2 p = norm2(x), pm = norm2(xm);  // (row A, row B)
        double pxd = (double)p.x, pyd = (double)p.y;
        if constexpr (WBX_SPECTRUM_DEMEAN) {
          if (k == 0) {  // x.re = F'_0 of the shifted rows, x.im = 0: F_0 = F'_0 + n m, formed and squared in fp64
            const double fa = (double)x.re.x + (double)fs.n * (double)msh.x, fb = (double)x.re.y + (double)fs.n * (double)msh.y;
            pxd = fa * fa;
            pyd = fb * fb;
          }
        }
        const bool mirror = km != k;
        if (split) {
          acc[i] = fma(pxd, sca, acc[i]);
          rec_b[k] = pyd * scb * (k == 0 ? 1.0 : 2.0);
          if (mirror) {
            accm[i] = fma((double)pm.x, sca, accm[i]);
            rec_b[km] = (double)pm.y * scb * 2.0;
          }
        } else {
          acc[i] = fma(pxd, sca, fma(pyd, scb, acc[i]));
          if (mirror) accm[i] = fma((double)pm.x, sca, fma((double)pm.y, scb, accm[i]));
        }
      }
    }
    team_sync<G>();  // buf is overwritten by the next pair of rows
  }
  flush(cur, true);
}

#include "wbx_zspec1440.hpp"
#include "wbx_zspec_det.hpp"
#include "wbx_zspec_det_latfast.hpp"

// Diagnostic instantiations -- WBX_SPECTRUM_KNOCK: kernels whose RESULTS ARE WRONG by design (timing only);
// WBX_SPECTRUM_PROF=<file>: the phase-stamped kernels, counters appended to a file -- exist only in a library built with
// -DWBX_DIAGNOSTICS (`make diag`: libwbx_hip_diag.so, what tools/kbench_spectrum_raw.py and tools/spec_phase_profile.py load
// through WBX_LIBRARY_PATH).  The shipped library ignores both variables and says so once: a stray environment variable in a
// production job must not be able to corrupt spectra or write files.
#ifdef WBX_DIAGNOSTICS
static int spectrum_knock() {
  static const int knock = [] {
    const char* e = getenv("WBX_SPECTRUM_KNOCK");
    const int k = e ? atoi(e) : 0;
    if (k != 0)
      fprintf(stderr, "libwbx_hip (diagnostic build): WBX_SPECTRUM_KNOCK=%d -- diagnostic spectrum kernels are in use, their results are not valid\n", k);
    return k;
  }();
  return knock;
}
static const char* spectrum_prof_path() {
  static const char* path = getenv("WBX_SPECTRUM_PROF");
  return path;
}
#else
static void spectrum_diag_ignored() {
  static const bool said = [] {
    if (getenv("WBX_SPECTRUM_KNOCK") || getenv("WBX_SPECTRUM_PROF"))
      fprintf(stderr, "libwbx_hip: WBX_SPECTRUM_KNOCK / WBX_SPECTRUM_PROF are ignored: this library was built without -DWBX_DIAGNOSTICS (make diag)\n");
    return true;
  }();
  (void)said;
}
static int spectrum_knock() {
  spectrum_diag_ignored();
  return 0;
}
static const char* spectrum_prof_path() { return nullptr; }
#endif

// Launches zspec1440_kernel: as many one-wave teams per block as the LDS holds (12: tables + 12 x 11.4 KB), one block per
// CU, and -- every team takes the same time -- a grid of exactly `rounds` resident sets.
#ifndef WBX_SPECTRUM_SKEW_DEFAULT
#define WBX_SPECTRUM_SKEW_DEFAULT 100  // per mille of a team's rows (WBX_SPECTRUM_SKEW overrides; 0 / 40 / 70 / 100 / 130 / 200 / 250: 0.249 / 0.245 / 0.242 / 0.239 / 0.241 / 0.242 / 0.248 ms)
#endif

static int launch_1440(wbx_ctx* ctx, FftState* st, const float* field, int64_t row_stride, int64_t nrows,
                       const int32_t* group, const double* scale, double* power_out, int32_t ngroup, int32_t accumulate) {
  void*& tab = st->twiddles[-Z14_N];
  if (!tab) {
    std::vector<float2> host;
    zspec1440_tables(host);
    WBX_HIP(hipMalloc(&tab, host.size() * sizeof(float2)));
    WBX_HIP(hipMemcpyAsync(tab, host.data(), host.size() * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    WBX_HIP(hipStreamSynchronize(ctx->stream));
  }
  static const int teams_env = getenv("WBX_SPECTRUM_1440_TEAMS") ? atoi(getenv("WBX_SPECTRUM_1440_TEAMS")) : 12;
  const int nteam = teams_env < 1 ? 1 : (teams_env > 12 ? 12 : teams_env);
  const size_t lds = (size_t)Z14_TABLES * sizeof(float2) + (size_t)nteam * Z14_BUF * sizeof(v4);
  // WBX_SPECTRUM_PROF=<file>: the phase-stamped instantiation; the eight counters are appended to the file per launch
  const char* prof_path = spectrum_prof_path();
  const void* fn = prof_path ? reinterpret_cast<const void*>(&zspec1440_kernel<true, 0>) : reinterpret_cast<const void*>(&zspec1440_kernel<false, 0>);
  int& per_cu = st->occupancy[std::make_pair(fn, lds)];
  if (per_cu == 0) {
    if (lds > 48 * 1024) WBX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    WBX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * nteam, lds));
    if (per_cu <= 0) per_cu = 1;
  }
  int rounds = 1;
  if (const char* e = getenv("WBX_SPECTRUM_ROUNDS")) rounds = atoi(e) > 0 ? atoi(e) : 1;
  if (getenv("WBX_SPECTRUM_DEBUG")) fprintf(stderr, "zspec1440: teams/block=%d lds=%zu blocks/CU=%d\n", nteam, lds, per_cu);
  int64_t teams = (int64_t)per_cu * ctx->num_cus * nteam * rounds;
  if (teams > (nrows + 1) / 2) teams = (nrows + 1) / 2;
  int rows_per_team = (int)((nrows + teams - 1) / teams);
  rows_per_team += rows_per_team & 1;  // whole pairs
  teams = (nrows + rows_per_team - 1) / rows_per_team;
  const unsigned blocks = (unsigned)((teams + nteam - 1) / nteam);
  // rows the oldest four waves of a block take more / the youngest four less (whole pairs), see the kernel
  static const int skew_permille = getenv("WBX_SPECTRUM_SKEW") ? atoi(getenv("WBX_SPECTRUM_SKEW")) : WBX_SPECTRUM_SKEW_DEFAULT;
  int skew = 0;
  if (nteam == 12 && rows_per_team >= 16) skew = 2 * (int)(((int64_t)rows_per_team * skew_permille + 1000) / 2000);
  if (skew >= rows_per_team) skew = 0;
  // the records of this launch (teams walk contiguous rows: one record per team + one per group change + one per straddling pair)
  int64_t changes = 0;
  if (int rc = spec_group_changes(ctx, group, nrows, &changes)) return rc;
  SpecRecs recs;
  if (int rc = spec_recs_prepare(ctx, st, (int64_t)blocks * nteam, 2 * changes, Z14_N2 + 1, &recs)) return rc;
  if (prof_path) {
    static unsigned long long* prof = nullptr;  // (diagnostic path: one device, never freed)
    unsigned long long host[26];
    if (!prof) WBX_HIP(hipMalloc(reinterpret_cast<void**>(&prof), sizeof(host)));
    const bool timing_only = prof_path[0] == '-';  // "-": the stamped kernel back to back, no read-back (event timing)
    if (!timing_only) {
      for (int i = 0; i < 26; ++i) host[i] = i == 11 ? ~0ull : 0ull;
      WBX_HIP(hipMemcpyAsync(prof, host, sizeof(host), hipMemcpyHostToDevice, ctx->stream));
      WBX_HIP(hipStreamSynchronize(ctx->stream));
    }
    hipLaunchKernelGGL((zspec1440_kernel<true, 0>), dim3(blocks), dim3(64 * nteam), lds, ctx->stream, field, row_stride, nrows,
                       rows_per_team, skew, reinterpret_cast<const float2*>(tab), group, scale, recs, prof);
    WBX_HIP(hipGetLastError());
    if (int rc = spec_close(ctx, recs, ngroup, Z14_N2 + 1, 1, power_out, nullptr, accumulate)) return rc;
    if (timing_only) return 0;
    WBX_HIP(hipMemcpyAsync(host, prof, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
    WBX_HIP(hipStreamSynchronize(ctx->stream));
    if (FILE* f = fopen(prof_path, "a")) {
      for (int i = 0; i < 26; ++i) fprintf(f, "%llu%c", host[i], i == 25 ? '\n' : ' ');
      fclose(f);
    }
    return 0;
  }
  const int knock = spectrum_knock();  // diagnostic, wrong results
#define WBX_Z14_LAUNCH(KN)                                                                                                 \
  hipLaunchKernelGGL((zspec1440_kernel<false, KN>), dim3(blocks), dim3(64 * nteam), lds, ctx->stream, field, row_stride, nrows, \
                     rows_per_team, skew, reinterpret_cast<const float2*>(tab), group, scale, recs,                        \
                     static_cast<unsigned long long*>(nullptr))
  switch (knock) {
    case 1: WBX_Z14_LAUNCH(1); break;
    case 2: WBX_Z14_LAUNCH(2); break;
    case 6: WBX_Z14_LAUNCH(6); break;
    case 7: WBX_Z14_LAUNCH(7); break;
    case 8: WBX_Z14_LAUNCH(8); break;
    case 14: WBX_Z14_LAUNCH(14); break;
    case 15: WBX_Z14_LAUNCH(15); break;
    case 17: hipLaunchKernelGGL((zspec1440_kernel<false, 0, true, true>), dim3(blocks), dim3(64 * nteam), lds, ctx->stream, field, row_stride, nrows, rows_per_team, skew, reinterpret_cast<const float2*>(tab), group, scale, recs, static_cast<unsigned long long*>(nullptr)); break;  // loads in front of pass 1
    case 16: hipLaunchKernelGGL((zspec1440_kernel<false, 0, false>), dim3(blocks), dim3(64 * nteam), lds, ctx->stream, field, row_stride, nrows, rows_per_team, skew, reinterpret_cast<const float2*>(tab), group, scale, recs, static_cast<unsigned long long*>(nullptr)); break;  // no priority rotation
    default: WBX_Z14_LAUNCH(0); break;
  }
#undef WBX_Z14_LAUNCH
  WBX_HIP(hipGetLastError());
  return spec_close(ctx, recs, ngroup, Z14_N2 + 1, 1, power_out, nullptr, accumulate);
}

// Launches zspec1440_latfast_kernel over nslab slabs of rps adjacent rows (row_stride 1, longitude strided).
#ifndef WBX_SPECTRUM_LF_PRIO_DEFAULT
#define WBX_SPECTRUM_LF_PRIO_DEFAULT 1  // (0 / 1 / 2: 0.3673 / 0.3630 / 0.3674 ms back to back on one box) user priority of a block's waves by age (WBX_SPECTRUM_LF_PRIO overrides), see the kernel
#endif

static int launch_1440_latfast(wbx_ctx* ctx, FftState* st, const float* field, int64_t lon_stride, const int64_t* d_slab_off,
                               int64_t rps, int64_t nslab, const int32_t* group, const double* scale, double* power_out,
                               int32_t ngroup, int32_t accumulate) {
  void*& tab = st->twiddles[-Z14_N];
  if (!tab) {
    std::vector<float2> host;
    zspec1440_tables(host);
    WBX_HIP(hipMalloc(&tab, host.size() * sizeof(float2)));
    WBX_HIP(hipMemcpyAsync(tab, host.data(), host.size() * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    WBX_HIP(hipStreamSynchronize(ctx->stream));
  }
  const size_t lds = (size_t)Z14_TABLES * sizeof(float2) + (size_t)Z14_TEAMS * Z14_BUFL * sizeof(v4) + (size_t)(Z14_N2 + 2) * sizeof(double);
  const char* prof_path = spectrum_prof_path();
  const void* fn = prof_path ? reinterpret_cast<const void*>(&zspec1440_latfast_kernel<true, 0>) : reinterpret_cast<const void*>(&zspec1440_latfast_kernel<false, 0>);
  int& per_cu = st->occupancy[std::make_pair(fn, lds)];
  if (per_cu == 0) {
    WBX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    WBX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * Z14_TEAMS, lds));
    if (per_cu <= 0) per_cu = 1;
  }
  int nlocal = per_cu * ctx->num_cus / 8;  // blocks per XCD
  if (nlocal < 1) nlocal = 1;
  // runs per slab: 22 rows = 11 of the 12 teams.  Measured on configs[3] (721 rows per slab, 32 blocks per XCD), ms per
  // field for 31 / 32 / 33 / 34 / 36 / 40 / 48 runs: 0.396 / 0.418 / 0.389 / 0.425 / 0.393 / 0.421 / 0.512 -- 32 runs, where
  // the 32 blocks of an XCD walk the same longitudes of ONE slab in lockstep, is the slowest of the neighbours (FETCH 1.06 x
  // algorithmic against 1.09 x for 33: the shared lines are found in L2 either way)
  int64_t runs = (rps + 21) / 22;
  if (const char* e = getenv("WBX_SPECTRUM_LF_RUNS")) runs = atoi(e) >= (rps + Z14_RUN - 1) / Z14_RUN ? atoi(e) : runs;  // A/B timing
  static const int lf_prio = getenv("WBX_SPECTRUM_LF_PRIO") ? atoi(getenv("WBX_SPECTRUM_LF_PRIO")) : WBX_SPECTRUM_LF_PRIO_DEFAULT;
  const int64_t per_xcd = ((nslab + 7) / 8) * runs;  // (slab, run) pairs of the busiest XCD
  if (per_xcd < nlocal) nlocal = (int)per_xcd;
  // the records of this launch: the block's table goes out when the step's group differs from the last one's (at most once per
  // step, and once when the block is done); a team whose rows are not of its step's group -- a run that crosses a group change --
  // and the second row of a straddling pair write records of their own: at most 13 per group change, and never more than two
  // per row
  int64_t changes = 0;
  if (int rc = spec_group_changes(ctx, group, rps * nslab, &changes)) return rc;
  const int64_t own = 13 * changes < 2 * rps * nslab ? 13 * changes : 2 * rps * nslab;
  SpecRecs recs;
  if (int rc = spec_recs_prepare(ctx, st, nslab * runs, own, Z14_N2 + 1, &recs)) return rc;  // static: one slot per step
  if (prof_path) {
    static unsigned long long* prof = nullptr;  // (diagnostic path, see launch_1440)
    unsigned long long host[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!prof) WBX_HIP(hipMalloc(reinterpret_cast<void**>(&prof), sizeof(host)));
    WBX_HIP(hipMemcpyAsync(prof, host, sizeof(host), hipMemcpyHostToDevice, ctx->stream));
    WBX_HIP(hipStreamSynchronize(ctx->stream));
    if (spectrum_knock() == 3)
      hipLaunchKernelGGL((zspec1440_latfast_kernel<true, 3>), dim3(8 * nlocal), dim3(64 * Z14_TEAMS), lds, ctx->stream, field,
                         lon_stride, d_slab_off, rps, nslab, (nslab + 7) / 8, (int)runs, (int)(rps / runs), (int)(rps % runs), lf_prio,
                         reinterpret_cast<const float2*>(tab), group, scale, recs, prof);
    else
      hipLaunchKernelGGL((zspec1440_latfast_kernel<true, 0>), dim3(8 * nlocal), dim3(64 * Z14_TEAMS), lds, ctx->stream, field,
                         lon_stride, d_slab_off, rps, nslab, (nslab + 7) / 8, (int)runs, (int)(rps / runs), (int)(rps % runs), lf_prio,
                         reinterpret_cast<const float2*>(tab), group, scale, recs, prof);
    if (int rc = spec_close(ctx, recs, ngroup, Z14_N2 + 1, 1, power_out, nullptr, accumulate)) return rc;
    WBX_HIP(hipMemcpyAsync(host, prof, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
    WBX_HIP(hipStreamSynchronize(ctx->stream));
    if (FILE* f = fopen(prof_path, "a")) {
      fprintf(f, "latfast");
      for (int i = 0; i < 6; ++i) fprintf(f, " %llu", host[i]);
      fprintf(f, "\n");
      fclose(f);
    }
    return 0;
  }
  const int knock = spectrum_knock();  // diagnostic, wrong results
#define WBX_Z14LF_LAUNCH(KN)                                                                                               \
  hipLaunchKernelGGL((zspec1440_latfast_kernel<false, KN>), dim3(8 * nlocal), dim3(64 * Z14_TEAMS), lds, ctx->stream, field,  \
                     lon_stride, d_slab_off, rps, nslab, (nslab + 7) / 8, (int)runs, (int)(rps / runs), (int)(rps % runs), lf_prio,           \
                     reinterpret_cast<const float2*>(tab), group, scale, recs, static_cast<unsigned long long*>(nullptr))
  if (knock == 1) WBX_Z14LF_LAUNCH(1);
  else if (knock == 2) WBX_Z14LF_LAUNCH(2);
  else if (knock == 3) WBX_Z14LF_LAUNCH(3);
  else if (knock == 9) hipLaunchKernelGGL((zspec1440_latfast_kernel<false, 0, 1>), dim3(8 * nlocal), dim3(64 * Z14_TEAMS), lds, ctx->stream, field, lon_stride, d_slab_off, rps, nslab, (nslab + 7) / 8, (int)runs, (int)(rps / runs), (int)(rps % runs), lf_prio, reinterpret_cast<const float2*>(tab), group, scale, recs, static_cast<unsigned long long*>(nullptr));  // a quarter of the loads in front of pass 1, none behind the unpack
  else if (knock == 8) hipLaunchKernelGGL((zspec1440_latfast_kernel<false, 0, 0>), dim3(8 * nlocal), dim3(64 * Z14_TEAMS), lds, ctx->stream, field, lon_stride, d_slab_off, rps, nslab, (nslab + 7) / 8, (int)runs, (int)(rps / runs), (int)(rps % runs), lf_prio, reinterpret_cast<const float2*>(tab), group, scale, recs, static_cast<unsigned long long*>(nullptr));  // the next run's loads in one burst

  else WBX_Z14LF_LAUNCH(0);
#undef WBX_Z14LF_LAUNCH
  WBX_HIP(hipGetLastError());
  return spec_close(ctx, recs, ngroup, Z14_N2 + 1, 1, power_out, nullptr, accumulate);
}

static bool fused_factor(int n, FusedSpec& fs) {
  if (n < 4 || (n & 1) || n > 2048) return false;
  int m = n / 2;
  fs.n = n;
  fs.n2 = m;
  fs.npass = 0;
  while (m % 4 == 0) { fs.radix[fs.npass++] = 4; m /= 4; }
  while (m % 2 == 0) { fs.radix[fs.npass++] = 2; m /= 2; }
  while (m % 5 == 0) { fs.radix[fs.npass++] = 5; m /= 5; }
  while (m % 3 == 0) { fs.radix[fs.npass++] = 3; m /= 3; }
  if (m != 1 || fs.npass > 16) return false;
  int ns = fs.n2, toff = 0;
  for (int p = 0; p < fs.npass; ++p) {
    if (fs.n2 / fs.radix[p] > 256) return false;  // a lane keeps <= 4 butterflies of a pass in registers
    ns /= fs.radix[p];
    fs.ns[p] = ns;
    fs.toff[p] = toff;
    if (ns > 1) toff += (fs.radix[p] - 1) * ns;
    fs.inv_ns[p] = 1.0f / (float)ns;
  }
  return true;
}

// Launches zspec_fused_kernel over `nrows` contiguous rows (row r at field + r * row_stride, unit longitude stride).
static int launch_fused(wbx_ctx* ctx, FftState* st, const FusedSpec& fs, const float* field, int64_t row_stride,
                        int64_t nrows, const int32_t* group, const double* scale, double* power_out, int32_t ngroup,
                        int32_t accumulate) {
  const int nlon = fs.n, n2 = fs.n2;
  static const bool use_1440 = getenv("WBX_SPECTRUM_1440") == nullptr || atoi(getenv("WBX_SPECTRUM_1440")) != 0;
  if (nlon == Z14_N && use_1440 && !getenv("WBX_SPECTRUM_TEAM"))  // (WBX_SPECTRUM_TEAM pins the generic kernel's team size)
    return launch_1440(ctx, st, field, row_stride, nrows, group, scale, power_out, ngroup, accumulate);
  int64_t changes = 0;
  if (int rc = spec_group_changes(ctx, group, nrows, &changes)) return rc;
  SpecRecs recs;
  void*& tw = st->twiddles[nlon];
  if (!tw) {
    std::vector<float2> host((size_t)n2 + n2 + 1);
    for (int p = 0; p < fs.npass; ++p) {  // packed per-pass twiddles (n2 - 1 entries, see FusedSpec)
      const int ns = fs.ns[p], R = fs.radix[p];
      if (ns <= 1) continue;
      for (int t = 1; t < R; ++t)
        for (int k = 0; k < ns; ++k) {
          const double a = -2.0 * M_PI * (double)k * (double)t / ((double)ns * (double)R);
          host[fs.toff[p] + (t - 1) * ns + k] = make_float2((float)cos(a), (float)sin(a));
        }
    }
    for (int k = 0; k <= n2; ++k) {
      const double a = -2.0 * M_PI * (double)k / (double)nlon;
      host[n2 + k] = make_float2((float)cos(a), (float)sin(a));
    }
    WBX_HIP(hipMalloc(&tw, host.size() * sizeof(float2)));
    WBX_HIP(hipMemcpyAsync(tw, host.data(), host.size() * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    WBX_HIP(hipStreamSynchronize(ctx->stream));
  }
  const float2* tw_pass = reinterpret_cast<const float2*>(tw);
  const float2* tw_real = tw_pass + n2;
  // threads per row pair: more threads = fewer registers per thread and more waves per LDS byte (n = 1440, configs[3]:
  // 64 / 128 / 256 threads -> 2.16 / 2.00 / 1.75 ms per step); short rows keep one wave busy
  int G = n2 <= 128 ? 64 : (n2 <= 256 ? 128 : 256);
  if (const char* e = getenv("WBX_SPECTRUM_TEAM")) G = atoi(e);  // tests drive every team size
  WBX_REQUIRE(G == 64 || G == 128 || G == 256, "WBX_SPECTRUM_TEAM must be 64, 128 or 256");
  const int nteam = G == 64 ? 4 : 1;
  const int ntr = n2 / 2 + 1;
  const size_t lds = (size_t)(n2 + ntr + (ntr & 1) + 2 * nteam * n2) * sizeof(float2);  // tables + one row PAIR per team
  const int kpt = (n2 / 2 + 1 + G - 1) / G;
  int rounds = 3;  // a few sets of blocks per resident slot: the last set's imbalance is a fraction of one set
  if (const char* e = getenv("WBX_SPECTRUM_ROUNDS")) rounds = atoi(e) > 0 ? atoi(e) : 1;
  // A team sweeps a contiguous run of rows (one group for most of it).  Blocks all take the same time, so the grid is a
  // whole number of resident sets (blocks per CU from the occupancy query: LDS and VGPRs both limit it): 2048 blocks on
  // 1280 resident slots ran as 1.6 rounds, the last one 60 % empty.
#define WBX_LAUNCH_FUSED_R(KPT, GG, RR)                                                                                 \
  do {                                                                                                              \
    const void* fn = reinterpret_cast<const void*>(&zspec_fused_kernel<KPT, GG, RR>);                               \
    int& per_cu = st->occupancy[std::make_pair(fn, lds)]; /* queried once per (kernel, LDS size) */                 \
    if (per_cu == 0) {                                                                                              \
      if (lds > 48 * 1024) WBX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
      WBX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, GG == 64 ? 256 : GG, lds));                 \
      if (per_cu <= 0) per_cu = 1;                                                                                  \
    }                                                                                                               \
    if (getenv("WBX_SPECTRUM_DEBUG")) fprintf(stderr, "zspec: G=%d lds=%zu blocks/CU=%d\n", GG, lds, per_cu);             \
    int64_t teams = (int64_t)per_cu * ctx->num_cus * nteam * rounds;                                                \
    if (teams > (nrows + 1) / 2) teams = (nrows + 1) / 2;                                                           \
    int rows_per_team = (int)((nrows + teams - 1) / teams);                                                         \
    rows_per_team += rows_per_team & 1; /* whole pairs */                                                           \
    teams = (nrows + rows_per_team - 1) / rows_per_team;                                                            \
    const unsigned blocks = (unsigned)((teams + nteam - 1) / nteam);                                                \
    if (int rc = spec_recs_prepare(ctx, st, (int64_t)blocks * nteam, 2 * changes, n2 + 1, &recs)) return rc;         \
    hipLaunchKernelGGL((zspec_fused_kernel<KPT, GG, RR>), dim3(blocks), dim3(GG == 64 ? 256 : GG), lds, ctx->stream, field, row_stride, \
                       nrows, rows_per_team, fs, tw_pass, tw_real, group, scale, recs);                             \
  } while (0)
#define WBX_LAUNCH_FUSED(KPT, GG) WBX_LAUNCH_FUSED_R(KPT, GG, 0)
#define WBX_LAUNCH_FUSED_256(KPT)                                  \
  do {                                                             \
    if (!prefetch) WBX_LAUNCH_FUSED_R(KPT, 256, 0);                \
    else if (fs.radix[0] == 4) WBX_LAUNCH_FUSED_R(KPT, 256, 4);    \
    else if (fs.radix[0] == 2) WBX_LAUNCH_FUSED_R(KPT, 256, 2);    \
    else if (fs.radix[0] == 5) WBX_LAUNCH_FUSED_R(KPT, 256, 5);    \
    else WBX_LAUNCH_FUSED_R(KPT, 256, 3);                          \
  } while (0)
  static const bool prefetch = getenv("WBX_SPECTRUM_PREFETCH") == nullptr || atoi(getenv("WBX_SPECTRUM_PREFETCH")) != 0;
  if (G == 256) {
    if (kpt <= 2) WBX_LAUNCH_FUSED_256(2); else WBX_LAUNCH_FUSED_256(3);
  } else if (G == 128) {
    if (kpt <= 3) WBX_LAUNCH_FUSED(3, 128); else WBX_LAUNCH_FUSED(5, 128);
  } else {
    if (kpt <= 2) WBX_LAUNCH_FUSED(2, 64);
    else if (kpt <= 4) WBX_LAUNCH_FUSED(4, 64);
    else if (kpt <= 6) WBX_LAUNCH_FUSED(6, 64);
    else WBX_LAUNCH_FUSED(9, 64);
  }
#undef WBX_LAUNCH_FUSED_256
#undef WBX_LAUNCH_FUSED_R
#undef WBX_LAUNCH_FUSED
  WBX_HIP(hipGetLastError());
  return spec_close(ctx, recs, ngroup, n2 + 1, 1, power_out, nullptr, accumulate);
}

// scratch[(row - row0) * nlon + j] = field[slab_off[row / rps] + (row % rps) + j * lon_stride] for rows of unit row
// stride (latitude-fastest fields: the rows of a slab are adjacent, longitude is strided): 64 x 64 tiles through LDS,
// reads coalesced along the rows, writes coalesced along longitude.
__global__ void __launch_bounds__(256) transpose_rows_kernel(const float* __restrict__ field, int64_t lon_stride,
                                                             const int64_t* __restrict__ slab_off, int64_t rps,
                                                             int64_t row0, int64_t nrows, int nlon,
                                                             float* __restrict__ out) {
  __shared__ float tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  const int64_t rb = (int64_t)blockIdx.y * 64;             // first row of the tile, relative to row0
  const int jb = blockIdx.x * 64;
  for (int dj = ty; dj < 64; dj += 4) {
    const int64_t r = rb + tx;
    const int j = jb + dj;
    if (r < nrows && j < nlon) {
      const int64_t gr = row0 + r;
      const int64_t slab = gr / rps;
      tile[dj][tx] = __builtin_nontemporal_load(field + (slab_off ? slab_off[slab] : 0) + (gr - slab * rps) + (int64_t)j * lon_stride);
    }
  }
  __syncthreads();
  for (int dr = ty; dr < 64; dr += 4) {
    const int64_t r = rb + dr;
    const int j = jb + tx;
    if (r < nrows && j < nlon) out[r * nlon + j] = tile[tx][dr];
  }
}

// rocFFT route for one slab: batched strided R2C + power_kernel, in row tiles of <= 256 MiB of complex scratch.
static int rocfft_route(wbx_ctx* ctx, FftState* st, const float* field, int64_t lon_stride, int64_t row_stride,
                        int64_t nrows, int32_t nlon, const int32_t* group, const double* scale, double* power_out, int32_t ngroup,
                        int32_t accumulate) {
  const int nk = nlon / 2 + 1;
  const int rows_per_block = 64;
  int64_t changes = 0;
  if (int rc = spec_group_changes(ctx, group, nrows, &changes)) return rc;
  SpecRecs recs;  // one record per 64-row block (+ one per tile for its ragged last block) + one per group change inside a block
  int64_t team_base = 0;
  // tile of rows: <= 256 MiB of complex scratch
  int64_t tile = ((int64_t)256 << 20) / ((int64_t)nk * 8);
  if (tile < 1) tile = 1;
  if (tile > nrows) tile = nrows;
  int64_t nblocks = 0;  // every block of every tile has a reserved slot and writes its header: the count has to be exact
  for (int64_t r0 = 0; r0 < nrows; r0 += tile) nblocks += ((r0 + tile <= nrows ? tile : nrows - r0) + rows_per_block - 1) / rows_per_block;
  if (int rc = spec_recs_prepare(ctx, st, nblocks, changes, nk, &recs)) return rc;
  // a strided batch only tiles cleanly when rows are uniformly spaced, which they are by construction
  const size_t need = (size_t)tile * nk * 8;
  if (st->scratch_size < need) {
    if (st->scratch) {
      WBX_HIP(hipStreamSynchronize(ctx->stream));
      WBX_HIP(hipFree(st->scratch));
    }
    WBX_HIP(hipMalloc(&st->scratch, need));
    st->scratch_size = need;
  }
  for (int64_t r0 = 0; r0 < nrows; r0 += tile) {
    const int64_t n = r0 + tile <= nrows ? tile : nrows - r0;
    FftPlan* plan = nullptr;
    if (int rc = get_plan(ctx, st, nlon, lon_stride, row_stride, n, &plan)) return rc;
    void* in = const_cast<float*>(field + r0 * row_stride);
    void* out = st->scratch;
    WBX_FFT(rocfft_execute(plan->plan, &in, &out, plan->info));
    const unsigned blocks = (unsigned)((n + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(power_kernel, dim3(blocks), dim3(256), 0, ctx->stream, reinterpret_cast<const float2*>(st->scratch), r0, n,
                       rows_per_block, nk, nlon, group, scale, recs, team_base);
    WBX_HIP(hipGetLastError());
    team_base += blocks;
  }
  return spec_close(ctx, recs, ngroup, nk, 1, power_out, nullptr, accumulate);
}

// ---- spectra of (p, t) + the deterministic lanes of the same rows in one sweep (wbx_zspec_det.hpp) --------------------------
static int launch_1440_det(wbx_ctx* ctx, FftState* st, const wbx_s1_plan* plan, bool has_c, const void* p, const void* t,
                           const void* c, const int32_t* group, const double* scale, double* partial_out, double* power_p,
                           double* power_t, int32_t ngroup, const double* det_scale = nullptr, double* det_out = nullptr) {
  const bool fold = det_scale != nullptr;
  void*& tab = st->twiddles[-Z14_N];
  if (!tab) {
    std::vector<float2> host;
    zspec1440_tables(host);
    WBX_HIP(hipMalloc(&tab, host.size() * sizeof(float2)));
    WBX_HIP(hipMemcpyAsync(tab, host.data(), host.size() * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    WBX_HIP(hipStreamSynchronize(ctx->stream));
  }
  S1Args a;
  fill_args(plan, a);
  a.in[0] = p;
  a.in[1] = t;
  a.in[2] = c;
  a.out = partial_out;
  const size_t lds = (size_t)Z14_TABLES * sizeof(float2) + (size_t)ZD_TEAMS * Z14_BUF * sizeof(v4) +
                     ((has_c && !WBX_ZD_C_IN_REGISTERS) ? (size_t)ZD_TEAMS * 24 * 64 * sizeof(float) : 0);  // + the climatology staging slots
  const void* fn = fold ? (has_c ? reinterpret_cast<const void*>(&zspec1440_det_kernel<true, true>) : reinterpret_cast<const void*>(&zspec1440_det_kernel<false, true>))
                        : (has_c ? reinterpret_cast<const void*>(&zspec1440_det_kernel<true>) : reinterpret_cast<const void*>(&zspec1440_det_kernel<false>));
  int& per_cu = st->occupancy[std::make_pair(fn, lds)];
  if (per_cu == 0) {
    if (lds > 48 * 1024) WBX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    WBX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * ZD_TEAMS, lds));
    if (per_cu <= 0) per_cu = 1;
  }
  const int64_t nrows = plan->nkey;
  int64_t teams = (int64_t)per_cu * ctx->num_cus * ZD_TEAMS;  // one resident set: every team takes the same time
  int64_t rows_per_team = (nrows + teams - 1) / teams;
  if (rows_per_team < 1) rows_per_team = 1;
  const int64_t blocks = (nrows + rows_per_team * ZD_TEAMS - 1) / (rows_per_team * ZD_TEAMS);
  WBX_REQUIRE(blocks < (int64_t)1 << 31 && rows_per_team < (int64_t)1 << 31, "launch too large");
  // records of 2 x 721 values (the predictions' spectrum, then the targets'): one per team + one per group change (a pair is
  // the p and the t row of ONE location: no straddling)
  int64_t changes = 0;
  if (int rc = spec_group_changes(ctx, group, nrows, &changes)) return rc;
  SpecRecs recs;
  if (int rc = spec_recs_prepare(ctx, st, blocks * ZD_TEAMS, changes, 2 * (Z14_N2 + 1) + (fold ? ZD_TAIL : 0), &recs)) return rc;
  if (fold && has_c)
    hipLaunchKernelGGL((zspec1440_det_kernel<true, true>), dim3((unsigned)blocks), dim3(64 * ZD_TEAMS), lds, ctx->stream, a, nrows,
                       (int)rows_per_team, reinterpret_cast<const float2*>(tab), group, scale, recs, det_scale);
  else if (fold)
    hipLaunchKernelGGL((zspec1440_det_kernel<false, true>), dim3((unsigned)blocks), dim3(64 * ZD_TEAMS), lds, ctx->stream, a, nrows,
                       (int)rows_per_team, reinterpret_cast<const float2*>(tab), group, scale, recs, det_scale);
  else if (has_c)
    hipLaunchKernelGGL((zspec1440_det_kernel<true>), dim3((unsigned)blocks), dim3(64 * ZD_TEAMS), lds, ctx->stream, a, nrows,
                       (int)rows_per_team, reinterpret_cast<const float2*>(tab), group, scale, recs, (const double*)nullptr);
  else
    hipLaunchKernelGGL((zspec1440_det_kernel<false>), dim3((unsigned)blocks), dim3(64 * ZD_TEAMS), lds, ctx->stream, a, nrows,
                       (int)rows_per_team, reinterpret_cast<const float2*>(tab), group, scale, recs, (const double*)nullptr);
  WBX_HIP(hipGetLastError());
  if (fold) return spec_close(ctx, recs, ngroup, Z14_N2 + 1, 2, power_p, power_t, 0, det_out, ZD_TAIL, has_c ? 6 : 3);
  return spec_close(ctx, recs, ngroup, Z14_N2 + 1, 2, power_p, power_t, 0);
}

// ---- the same for latitude-fastest fields (wbx_zspec_det_latfast.hpp) -------------------------------------------------------
static int launch_1440_det_latfast(wbx_ctx* ctx, FftState* st, const wbx_s1_plan* plan, bool has_c, const void* p, const void* t,
                                   const void* c, int64_t rps, const int32_t* group, const double* scale, double* partial_out,
                                   double* power_p, double* power_t, int32_t ngroup) {
  void*& tab = st->twiddles[-Z14_N];
  if (!tab) {
    std::vector<float2> host;
    zspec1440_tables(host);
    WBX_HIP(hipMalloc(&tab, host.size() * sizeof(float2)));
    WBX_HIP(hipMemcpyAsync(tab, host.data(), host.size() * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    WBX_HIP(hipStreamSynchronize(ctx->stream));
  }
  S1Args a;
  fill_args(plan, a);
  a.in[0] = p;
  a.in[1] = t;
  a.in[2] = c;
  a.out = partial_out;
  const size_t lds = (size_t)Z14_TABLES * sizeof(float2) + (size_t)ZL_TEAMS * WBX_ZL_BUFL * sizeof(v4) +
                     (size_t)2 * (Z14_N2 + 2) * sizeof(double) + (size_t)ZL_GROUPS * 4 * 2 * 6 * sizeof(double);
  const void* fn = has_c ? reinterpret_cast<const void*>(&zspec1440_det_latfast_kernel<true>)
                         : reinterpret_cast<const void*>(&zspec1440_det_latfast_kernel<false>);
  int& per_cu = st->occupancy[std::make_pair(fn, lds)];
  if (per_cu == 0) {
    WBX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    WBX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, ZL_THREADS, lds));
    if (per_cu <= 0) per_cu = 1;
  }
  const int64_t nslab = plan->nkey / rps;
  int nlocal = per_cu * ctx->num_cus / 8;  // blocks per XCD
  if (nlocal < 1) nlocal = 1;
  // Runs of exactly eight rows from the start of the slab (the last one short): a slab of a [.., 1440, nlat] array starts on a
  // 128-byte line (1440 = 32 x 45), so a run's 32-byte segment never straddles two lines.  The L1 keeps ~64 line requests in
  // flight whatever part of a line is used (pmc: TCP_PENDING_STALL 58 % of the cycles, 5.2 x the algorithmic bytes requested
  // from L2 with runs of 7-8 rows at arbitrary offsets): every line a CU asks for twice is bandwidth lost.
  // WBX_SPECTRUM_ZL_EVEN=1: runs of equal length instead (A/B timing)
  int64_t runs = (rps + ZL_TEAMS - 1) / ZL_TEAMS;
  static const bool even_runs = getenv("WBX_SPECTRUM_ZL_EVEN") && atoi(getenv("WBX_SPECTRUM_ZL_EVEN"));
  const int run_base = even_runs ? (int)(rps / runs) : ZL_TEAMS, run_rem = even_runs ? (int)(rps % runs) : 0;
  const int64_t per_xcd = ((nslab + 7) / 8) * runs;  // (slab, run) pairs of the busiest XCD
  if (per_xcd < nlocal) nlocal = (int)per_xcd;
  WBX_REQUIRE(runs < (int64_t)1 << 30, "launch too large");
  // records of 2 x 721 values: the block's tables once per step at most (+ once at the end); a team whose row is of another group
  // than its step's -- a run that crosses a group change -- writes its own: at most 8 per group change, one per row
  int64_t changes = 0;
  if (int rc = spec_group_changes(ctx, group, rps * nslab, &changes)) return rc;
  SpecRecs recs;
  if (int rc = spec_recs_prepare(ctx, st, nslab * runs, 8 * changes < rps * nslab ? 8 * changes : rps * nslab,
                                 2 * (Z14_N2 + 1), &recs))
    return rc;
#ifdef WBX_DIAGNOSTICS  // (knock-out instantiations, wrong results by design: `make diag` only, like WBX_SPECTRUM_KNOCK)
  static const int knock = getenv("WBX_ZL_KNOCK") ? atoi(getenv("WBX_ZL_KNOCK")) : 0;  // diagnostic, wrong results
#define WBX_ZL_LAUNCH(KN)                                                                                                      \
  hipLaunchKernelGGL((zspec1440_det_latfast_kernel<true, KN>), dim3(8 * nlocal), dim3(ZL_THREADS), lds, ctx->stream, a, rps, nslab, \
                     (nslab + 7) / 8, (int)runs, run_base, run_rem, reinterpret_cast<const float2*>(tab), group, \
                     scale, recs)
  if (has_c && knock) {
    switch (knock) {
      case 1: WBX_ZL_LAUNCH(1); break;
      case 2: WBX_ZL_LAUNCH(2); break;
      case 3: WBX_ZL_LAUNCH(3); break;
      case 4: WBX_ZL_LAUNCH(4); break;
      case 8: WBX_ZL_LAUNCH(8); break;
      default: WBX_ZL_LAUNCH(11); break;
    }
    WBX_HIP(hipGetLastError());
    return spec_close(ctx, recs, ngroup, Z14_N2 + 1, 2, power_p, power_t, 0);
  }
#undef WBX_ZL_LAUNCH
#endif
  if (has_c)
    hipLaunchKernelGGL((zspec1440_det_latfast_kernel<true>), dim3(8 * nlocal), dim3(ZL_THREADS), lds, ctx->stream, a, rps, nslab,
                       (nslab + 7) / 8, (int)runs, run_base, run_rem, reinterpret_cast<const float2*>(tab), group,
                       scale, recs);
  else
    hipLaunchKernelGGL((zspec1440_det_latfast_kernel<false>), dim3(8 * nlocal), dim3(ZL_THREADS), lds, ctx->stream, a, rps, nslab,
                       (nslab + 7) / 8, (int)runs, run_base, run_rem, reinterpret_cast<const float2*>(tab), group,
                       scale, recs);
  WBX_HIP(hipGetLastError());
  return spec_close(ctx, recs, ngroup, Z14_N2 + 1, 2, power_p, power_t, 0);
}

}  // namespace wbx

static int det_spectrum_entry(const char* who, wbx_ctx* ctx, const wbx_s1_plan* plan, int func, int dtype, const void* p, const void* t,
                              const void* c, const int32_t* group, const double* scale, const double* det_scale, int64_t ngroup,
                              double* det_out, double* power_p, double* power_t, bool fold) {
  using namespace wbx;
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  if (int rc = check_plan(plan)) return rc;
  WBX_REQUIRE(func == WBX_DET3 || func == WBX_DET6, "%s takes WBX_DET3 or WBX_DET6 (got %d)", who, func);
  WBX_REQUIRE(dtype == WBX_F32, "%s takes float32 fields", who);
  WBX_REQUIRE(plan->nx == Z14_N && !plan->x_kept && plan->ndepth == 1 && plan->nchunk == 1,
              "%s needs rows of %d points summed along x, one depth row and one chunk per key", who, Z14_N);
  WBX_REQUIRE(!(plan->flags & (WBX_FLAG_MASKED | WBX_FLAG_SKIPNA)) && plan->x_weights == nullptr, "no masks / folded weights here");
  const int nin = func == WBX_DET6 ? 3 : 2;
  for (int i = 0; i < nin; ++i) {
    WBX_REQUIRE(plan->xstride[i] == 1, "input %d is not contiguous along x", i);
  }
  WBX_REQUIRE(ngroup >= 1, "ngroup must be >= 1");
  if (plan->nkey == 0) return 0;
  WBX_REQUIRE(p && t && (func == WBX_DET3 || c) && group && scale && det_out && power_p && power_t && (!fold || det_scale), "NULL pointer");
  WBX_REQUIRE((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(c)) % 8 == 0,
              "fields must be 8-byte aligned (row offsets must be even: the caller's plan)");
  WBX_HIP(hipSetDevice(ctx->device));
  auto* st = reinterpret_cast<FftState*>(ctx->fft_state);
  if (!st) {
    st = new FftState();
    ctx->fft_state = st;
    WBX_FFT(rocfft_setup());  // (the state is shared with wbx_zonal_spectrum, which may take the rocFFT route later)
    st->setup = true;
  }
  // (no memset: the closing kernel of the records starts both spectra -- and the folded sums -- from zero)
  if (fold)
    return launch_1440_det(ctx, st, plan, func == WBX_DET6, p, t, c, group, scale, nullptr, power_p, power_t, (int32_t)ngroup, det_scale, det_out);
  return launch_1440_det(ctx, st, plan, func == WBX_DET6, p, t, c, group, scale, det_out, power_p, power_t, (int32_t)ngroup);
}

extern "C" int wbx_det_spectrum(wbx_ctx* ctx, const wbx_s1_plan* plan, int func, int dtype, const void* p, const void* t,
                                const void* c, const int32_t* group, const double* scale, int64_t ngroup, double* partial_out,
                                double* power_p, double* power_t) {
  return det_spectrum_entry("wbx_det_spectrum", ctx, plan, func, dtype, p, t, c, group, scale, nullptr, ngroup, partial_out, power_p,
                            power_t, false);
}

extern "C" int wbx_det_spectrum_folded(wbx_ctx* ctx, const wbx_s1_plan* plan, int func, int dtype, const void* p, const void* t,
                                       const void* c, const int32_t* group, const double* scale, const double* det_scale,
                                       int64_t ngroup, double* det_out, double* power_p, double* power_t) {
  return det_spectrum_entry("wbx_det_spectrum_folded", ctx, plan, func, dtype, p, t, c, group, scale, det_scale, ngroup, det_out,
                            power_p, power_t, true);
}


extern "C" int wbx_det_spectrum_slabs(wbx_ctx* ctx, const wbx_s1_plan* plan, int func, int dtype, const void* p, const void* t,
                                      const void* c, int64_t rows_per_slab, const int32_t* group, const double* scale,
                                      int64_t ngroup, double* partial_out, double* power_p, double* power_t) {
  using namespace wbx;
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  if (int rc = check_plan(plan)) return rc;
  WBX_REQUIRE(func == WBX_DET3 || func == WBX_DET6, "wbx_det_spectrum_slabs takes WBX_DET3 or WBX_DET6 (got %d)", func);
  WBX_REQUIRE(dtype == WBX_F32, "wbx_det_spectrum_slabs takes float32 fields");
  WBX_REQUIRE(plan->nx == Z14_N && !plan->x_kept && plan->ndepth == 1 && plan->nchunk == 1,
              "wbx_det_spectrum_slabs needs rows of %d points summed along x, one depth row and one chunk per key", Z14_N);
  WBX_REQUIRE(!(plan->flags & (WBX_FLAG_MASKED | WBX_FLAG_SKIPNA)) && plan->x_weights == nullptr, "no masks / folded weights here");
  WBX_REQUIRE(rows_per_slab >= 1 && plan->nkey % rows_per_slab == 0, "nkey = %lld is not a whole number of slabs of %lld rows",
              (long long)plan->nkey, (long long)rows_per_slab);
  const int nin = func == WBX_DET6 ? 3 : 2;
  for (int i = 0; i < nin; ++i) {
    WBX_REQUIRE(plan->xstride[i] >= rows_per_slab, "input %d: the longitude stride %lld is smaller than a slab's %lld adjacent rows", i,
                (long long)plan->xstride[i], (long long)rows_per_slab);
  }
  WBX_REQUIRE(ngroup >= 1, "ngroup must be >= 1");
  if (plan->nkey == 0) return 0;
  WBX_REQUIRE(p && t && (func == WBX_DET3 || c) && group && scale && partial_out && power_p && power_t, "NULL pointer");
  WBX_REQUIRE((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(c)) % 4 == 0, "fields must be 4-byte aligned");
  WBX_HIP(hipSetDevice(ctx->device));
  auto* st = reinterpret_cast<FftState*>(ctx->fft_state);
  if (!st) {
    st = new FftState();
    ctx->fft_state = st;
    WBX_FFT(rocfft_setup());  // (the state is shared with wbx_zonal_spectrum, which may take the rocFFT route later)
    st->setup = true;
  }
  return launch_1440_det_latfast(ctx, st, plan, func == WBX_DET6, p, t, c, rows_per_slab, group, scale, partial_out, power_p, power_t,
                                 (int32_t)ngroup);
}


static int zonal_spectrum_impl(wbx_ctx* ctx, const float* field, int64_t lon_stride, int64_t row_stride, int64_t rps,
                               int64_t nslab, const int64_t* h_slab_offsets, int32_t nlon, const int32_t* group,
                               const double* scale, int32_t ngroup, int32_t accumulate, double* power_out) {
  using namespace wbx;
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  WBX_REQUIRE(nlon >= 2 && rps >= 0 && nslab >= 0 && ngroup >= 0, "bad spectrum extents (nlon=%d, rows=%lld x %lld)", nlon,
              (long long)nslab, (long long)rps);
  WBX_REQUIRE(lon_stride >= 1 && row_stride >= 1, "strides must be positive");
  WBX_REQUIRE(nslab <= 1 || h_slab_offsets != nullptr, "slab offsets are NULL");
  const int nk = nlon / 2 + 1;
  const int64_t nrows = rps * nslab;
  WBX_REQUIRE(power_out != nullptr || ngroup == 0, "power_out is NULL");
  WBX_HIP(hipSetDevice(ctx->device));
  if (nrows == 0) {  // (with rows, the closing kernel of the first launch starts the sums from zero: no memset in front)
    if (!accumulate && ngroup > 0) WBX_HIP(hipMemsetAsync(power_out, 0, (size_t)ngroup * nk * sizeof(double), ctx->stream));
    return 0;
  }
  WBX_REQUIRE(field && group && scale, "field/group/scale is NULL");
  auto* st = reinterpret_cast<FftState*>(ctx->fft_state);
  if (!st) {
    st = new FftState();
    ctx->fft_state = st;
    WBX_FFT(rocfft_setup());
    st->setup = true;
  }
  FusedSpec fs;
  const char* force = getenv("WBX_SPECTRUM_PATH");  // "rocfft" pins the library route (A/B timing, tests)
  const bool fused_ok = !(force && force[0] == 'r') && fused_factor(nlon, fs);
  if (fused_ok && lon_stride == 1 && (row_stride % 2) == 0) {
    // contiguous rows: straight into the fused kernel, slab by slab (a slab = rows at a uniform stride)
    bool aligned = (((uintptr_t)field) & 7) == 0;
    for (int64_t o = 0; o < nslab && aligned; ++o) aligned = ((nslab > 1 ? h_slab_offsets[o] : 0) % 2) == 0;
    if (aligned) {
      for (int64_t o = 0; o < nslab; ++o) {
        const int64_t off = nslab > 1 ? h_slab_offsets[o] : (h_slab_offsets ? h_slab_offsets[0] : 0);
        // (every launch closes its own records; launches after the first add to what the earlier ones left)
        if (int rc = launch_fused(ctx, st, fs, field + off, row_stride, rps, group + o * rps, scale + o * rps, power_out, ngroup,
                                  o == 0 ? accumulate : 1))
          return rc;
      }
      return 0;
    }
  }
  if (fused_ok && row_stride == 1 && lon_stride > 1) {
    // latitude-fastest fields (rows adjacent, longitude strided): transpose row tiles into a contiguous scratch, then the
    // fused kernel -- 3 passes over the field instead of one strided rocFFT batch per slab (configs[3], 296 slabs of
    // 721 rows: 27.6 ms -> see DESIGN.md)
    // (Scratch tiles small enough to stay in the 256 MB Infinity Cache between the transpose and the FFT were measured
    // SLOWER: 16 / 32 / 64 / 128 / 256 MB -> 8.1 / 6.1 / 4.1 / 3.3 / 3.0 ms per configs[3] step; the short launches do
    // not fill the chip.)  WBX_SPECTRUM_TILE_MB: A/B timing.
    static const int tile_mb = getenv("WBX_SPECTRUM_TILE_MB") ? atoi(getenv("WBX_SPECTRUM_TILE_MB")) : 256;
    int64_t tile = ((int64_t)(tile_mb > 0 ? tile_mb : 256) << 20) / ((int64_t)nlon * 4);
    tile -= tile & 1;
    if (tile < 2) tile = 2;
    if (tile > nrows) tile = nrows;
    int64_t* d_off = nullptr;
    if (nslab > 1 || (h_slab_offsets && h_slab_offsets[0] != 0)) {
      // the slab offsets of a chunk layout repeat from chunk to chunk: kept on the device per content, so that the steady
      // state has no host-blocking copy in front of the kernels
      std::vector<int64_t> key(h_slab_offsets, h_slab_offsets + nslab);
      auto it = st->slab_offsets.find(key);
      if (it == st->slab_offsets.end()) {
        if (st->slab_offsets.size() > 16) {
          WBX_HIP(hipStreamSynchronize(ctx->stream));
          for (auto& kv : st->slab_offsets) (void)hipFree(kv.second);
          st->slab_offsets.clear();
        }
        void* d = nullptr;
        WBX_HIP(hipMalloc(&d, (size_t)nslab * sizeof(int64_t)));
        WBX_HIP(hipMemcpyAsync(d, h_slab_offsets, (size_t)nslab * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
        WBX_HIP(hipStreamSynchronize(ctx->stream));  // the host array may go away after the call
        it = st->slab_offsets.emplace(std::move(key), d).first;
      }
      d_off = reinterpret_cast<int64_t*>(it->second);
    }
    static const bool use_latfast = (getenv("WBX_SPECTRUM_1440") == nullptr || atoi(getenv("WBX_SPECTRUM_1440")) != 0) &&
                                    (getenv("WBX_SPECTRUM_LATFAST") == nullptr || atoi(getenv("WBX_SPECTRUM_LATFAST")) != 0);
    if (nlon == Z14_N && use_latfast && !getenv("WBX_SPECTRUM_TEAM"))
      return launch_1440_latfast(ctx, st, field, lon_stride, d_off, rps, nslab, group, scale, power_out, ngroup, accumulate);
    const size_t need = (size_t)tile * nlon * sizeof(float);
    if (st->scratch_size < need) {
      if (st->scratch) {
        WBX_HIP(hipStreamSynchronize(ctx->stream));
        WBX_HIP(hipFree(st->scratch));
        st->scratch = nullptr;
        st->scratch_size = 0;
      }
      WBX_HIP(hipMalloc(&st->scratch, need));
      st->scratch_size = need;
    }
    float* rows = reinterpret_cast<float*>(st->scratch);
    for (int64_t r0 = 0; r0 < nrows; r0 += tile) {
      const int64_t n = r0 + tile <= nrows ? tile : nrows - r0;
      dim3 grid((unsigned)((nlon + 63) / 64), (unsigned)((n + 63) / 64));
      hipLaunchKernelGGL(transpose_rows_kernel, grid, dim3(256), 0, ctx->stream, field, lon_stride, d_off, rps, r0, n, (int)nlon, rows);
      WBX_HIP(hipGetLastError());
      if (int rc = launch_fused(ctx, st, fs, rows, nlon, n, group + r0, scale + r0, power_out, ngroup, r0 == 0 ? accumulate : 1)) return rc;
    }
    return 0;
  }
  // library route, one strided batch per slab
  for (int64_t o = 0; o < nslab; ++o) {
    const int64_t off = h_slab_offsets ? h_slab_offsets[o] : 0;
    if (int rc = rocfft_route(ctx, st, field + off, lon_stride, row_stride, rps, nlon, group + o * rps, scale + o * rps, power_out, ngroup,
                              o == 0 ? accumulate : 1))
      return rc;
  }
  return 0;
}

extern "C" int wbx_zonal_spectrum(wbx_ctx* ctx, const float* field, int64_t lon_stride, int64_t row_stride,
                                  int64_t nrows, int32_t nlon, const int32_t* group, const double* scale,
                                  int32_t ngroup, int32_t accumulate, double* power_out) {
  return zonal_spectrum_impl(ctx, field, lon_stride, row_stride, nrows, 1, nullptr, nlon, group, scale, ngroup, accumulate,
                             power_out);
}

extern "C" int wbx_zonal_spectrum_slabs(wbx_ctx* ctx, const float* field, int64_t lon_stride, int64_t row_stride,
                                        int64_t rows_per_slab, int64_t nslab, const int64_t* h_slab_offsets, int32_t nlon,
                                        const int32_t* group, const double* scale, int32_t ngroup, int32_t accumulate,
                                        double* power_out) {
  return zonal_spectrum_impl(ctx, field, lon_stride, row_stride, rows_per_slab, nslab, h_slab_offsets, nlon, group, scale,
                             ngroup, accumulate, power_out);
}
