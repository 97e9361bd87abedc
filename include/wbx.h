/*
 * wbx.h -- C ABI of libwbx_hip.so, the MI355X (gfx950) engine behind the
 * WeatherBench-X scoring hot path (Statistic.compute -> Aggregator reduce).
 *
 * The reference (google-research/weatherbenchX) is pure Python: the "FFI" a
 * maintainer would bind is the set of whole-array xarray/NumPy calls inside
 * the plugin classes.  Every entry point below names the reference call
 * site(s) it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *  - plain C types only; all data pointers are DEVICE pointers unless the
 *    name says host (`h_`); the caller owns every buffer.
 *  - every function returns 0 on success or a negative wbx_status; the
 *    message is retrievable with wbx_last_error() (thread local).
 *  - all work is enqueued on the context's HIP stream and is asynchronous;
 *    wbx_ctx_synchronize() (or reading through wbx_memcpy_d2h) orders it.
 *  - a context is single-threaded; distinct contexts are independent.
 *
 * Two-stage reduction:
 *   stage 1  (HBM-bound)  per-point statistics fused with an UNWEIGHTED partial
 *            sum over every reduced dimension that weights/bin masks do not
 *            depend on -> fp64 partial[key][chunk][lane][j]
 *   stage 2  (tiny)       partial (x) W, W = product of weights and bin masks,
 *            -> sum_weighted_statistics / sum_weights accumulators.
 */
#ifndef WBX_H_
#define WBX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WBX_ABI_VERSION 13

typedef enum wbx_status {
  WBX_OK = 0,
  WBX_ERR_INVALID = -1,    /* bad argument / unsupported combination */
  WBX_ERR_HIP = -2,        /* a HIP runtime call failed              */
  WBX_ERR_NO_DEVICE = -3,  /* no gfx950 device visible               */
  WBX_ERR_FFT = -4,        /* rocFFT failure                         */
  WBX_ERR_RCCL = -5        /* RCCL missing or a collective failed    */
} wbx_status;

typedef struct wbx_ctx wbx_ctx; /* opaque: device id, stream, timers, scratch */

/* element type of the statistic inputs */
typedef enum wbx_dtype { WBX_F32 = 0, WBX_F64 = 1 } wbx_dtype;

/* ---- fused per-point statistic families ---------------------------------
 * Lane order is part of the ABI.
 * WBX_DET3  inputs (p,t):    e=p-t, |e|, e^2
 *           replaces deterministic.py:91-123 (Error, AbsoluteError, SquaredError)
 * WBX_DET6  inputs (p,t,c):  e, |e|, e^2, (p-c)^2, (t-c)^2, (p-c)(t-c)
 *           + deterministic.py:222-259 (SquaredPredictionAnomaly,
 *           SquaredTargetAnomaly, AnomalyCovariance); c is gathered through
 *           the plan's gather table (metrics/base.py:382-406 `.sel(dayofyear,hour)`)
 * WBX_PASS1 input (p):       p          (any already-materialised statistic;
 *           replaces the xr.dot of aggregation.py:335 for user-defined stats)
 */
typedef enum wbx_det_func { WBX_DET3 = 0, WBX_DET6 = 1, WBX_PASS1 = 2 } wbx_det_func;
#define WBX_DET3_LANES 3
#define WBX_DET6_LANES 6
#define WBX_PASS1_LANES 1

/* ensemble lanes (probabilistic.py:116-336, wrappers.py:116-148):
 *  0 CRPSSkill            mean_m |p_m - t|
 *  1 CRPSSpread           sum_{m,m'} |p_m - p_m'| / (M (M - fair))
 *  2 EnsembleVariance     var_m(p, ddof=1)
 *  3 UnbiasedEnsembleMeanSquaredError  (mean_m p - t)^2 - var/M
 *  4 SquaredError of the ensemble mean (mean_m p - t)^2   (EnsembleMean + SquaredError)
 * Accuracy per point against the float64 restatement of the float32 members (tests/test_gpu_round4.py holds these on outputs
 * of 64 points -- one tile -- and on whole fields; geopotential-like 5.5e4 +- 30, spread = 0.05 x error, a target at the
 * ensemble mean): the rank-form fp32 kernels for M = 50 / 51 (ens_pipe_kernel, ens_atoms_kernel) sum e = x - median in fp32
 * chains of <= 8 terms, every chain of non-negative terms:
 *   lanes 0, 1, 4   relative: <= 9 x 2^-24 = 5.4e-7 worst case, ~1e-7 typical
 *   lane 2          relative to S = sum e^2 / (M - 1) <= 3 var: |d var| <= 5.4e-7 S (the subtraction (sum e)^2 / M is fp64)
 *   lane 3          a DIFFERENCE, so its bound is absolute, in units of the two terms it is the difference of:
 *                   |d lane3| <= 1e-6 (lane 4 + lane 2 / M); relative 1e-6 wherever lane 3 is not itself a cancellation
 * Every other ensemble kernel (padded buckets, masked / skipna wrappers, pair form, generic) sums in fp64 on the widened
 * members: ~1e-15 (pair form: ~1e-7, its |x_i - x_j| row sums are fp32). */
#define WBX_ENS_LANES 5
typedef enum wbx_ens_algo {
  WBX_ENS_SORT = 0,     /* rank / sorting-network form, probabilistic.py:214-240 (use_sort=True)  */
  WBX_ENS_PAIRWISE = 1  /* O(M^2) pairwise form,        probabilistic.py:241-247 (use_sort=False) */
} wbx_ens_algo;

#define WBX_FLAG_MASKED 1u /* input 3 (uint8 mask, nonzero = valid) is present: aggregation.py:339-352 */
#define WBX_FLAG_SKIPNA 2u /* NaN statistic values are dropped and counted out: aggregation.py:353-355 */
#define WBX_FLAG_FAIR   4u /* ensemble: fair CRPS spread (divide by M(M-1)):   probabilistic.py:239,247 */
#define WBX_FLAG_SKIPNA_ENS 8u /* ensemble: NaN members are missing members (skipna_ensemble=True), per-point
                                  ensemble size = count of non-NaN members: probabilistic.py:143-145,206-216,303-336.
                                  float32, M <= 64: members in registers, NaN -> +inf, sorted, rank form over the first n
                                  (either `algo`: the same number); otherwise the pair form over members re-read from memory */

#define WBX_MAX_INPUTS 4 /* 0 = predictions, 1 = targets, 2 = climatology, 3 = mask(uint8) */

/* Stage-1 plan.  The host flattens the statistic's index space into
 *   key   : every (key) index gets its own partial sum      [nkey]
 *   depth : summed inside stage 1, split in nchunk chunks   [ndepth]
 *   x     : the innermost, thread-mapped dimension          [nx]
 * and hands the kernel per-input ELEMENT offset tables, so any dim order /
 * any stride (the loaders' layout is arbitrary, data_loaders/xarray_loaders.py:185-188)
 * is consumed in place, without a transposed copy.
 * element address of input i at (key k, depth d, x):
 *     in[i] + key_off[i][k] + depth_off[i][d] + x * xstride[i]
 *     (+ gather_tab[gather_key[k] * n_gather_depth + gather_depth[d]] for i == 2)
 * All table pointers are device pointers; a NULL table means all zeros.
 */
typedef struct wbx_s1_plan {
  int64_t nkey;
  int64_t ndepth;
  int64_t nx;
  int32_t x_kept;        /* 1: keep one partial per x (nj = nx); 0: sum x away (nj = 1) */
  int32_t nchunk;        /* depth is cut into nchunk pieces of depth_chunk rows          */
  int64_t depth_chunk;
  int64_t xstride[WBX_MAX_INPUTS];
  const int64_t* key_off[WBX_MAX_INPUTS];   /* [nkey]   */
  const int64_t* depth_off[WBX_MAX_INPUTS]; /* [ndepth] */
  const int32_t* gather_key;   /* [nkey]   or NULL */
  const int32_t* gather_depth; /* [ndepth] or NULL */
  const int64_t* gather_tab;   /* [n_gather_key * n_gather_depth] or NULL */
  int32_t n_gather_depth;
  uint32_t flags;        /* WBX_FLAG_* */
  int32_t block_threads; /* 64, 128 or 256 */
  int32_t vec;           /* 1 or 4: x elements per lane per load (4 needs 16-B alignment of every row) */
  int32_t plane_rows;    /* 0, or R > 0 ("plane mode", x kept, deterministic families, fp32): every group of R consecutive
                            depth rows starting at a multiple of R is ONE contiguous span of R*nx elements in every input
                            (latitude-fastest chunks: rows = longitudes). The span is fetched with aligned 16-B loads
                            through LDS instead of ragged per-row dword loads. depth_chunk must be a multiple of R. */
  int32_t reserved_;
  const double* x_weights; /* NULL, or DEVICE float64[nx]: every lane of the point at x is multiplied by x_weights[x] inside
                              stage 1 and x is SUMMED there (x_kept = 0) -- for weights that depend on the innermost dim
                              only and no bins (GridAreaWeighting on latitude-fastest data); count lanes (mask / skipna)
                              take the weight too.  Always with plane_rows = R > 0 (R contiguous depth rows form one span,
                              ndepth % R == 0) and fp32 inputs.  wbx_det_partial (no flag, or WBX_FLAG_MASKED with a mask
                              stored like the data): R % 4 == 0, 16-B aligned spans streamed as one flat float4 array
                              (element e takes x_weights[e mod nx]), depth_chunk % 4 == 0, nx <= 2045.  wbx_ens_partial
                              (any of FAIR / MASKED / SKIPNA, not SKIPNA_ENS): one point per lane walks the flat index,
                              any R, nx <= 2048. */
} wbx_s1_plan;

/* number of fp64 values stage 1 writes: nkey * nchunk * nlanes_total * nj, layout partial[key][chunk][lane][j].
 * Count lanes follow the value lanes: none without flags; ONE shared count lane with MASKED alone (validity is the
 * same for every statistic: nlanes_total = lanes + 1); one per value lane with SKIPNA (nlanes_total = 2 * lanes). */
int wbx_s1_partial_len(const wbx_s1_plan* plan, int lanes, int64_t* n_out);

/* ---- context ------------------------------------------------------------- */
int wbx_abi_version(void);
const char* wbx_last_error(void);
int wbx_device_count(int* n_out);
/* `hip_stream` may be NULL (library creates its own) or an existing hipStream_t
 * (e.g. torch.cuda.current_stream().cuda_stream) so launches order with the caller's work. */
int wbx_ctx_create(int device_id, void* hip_stream, wbx_ctx** out);
int wbx_ctx_destroy(wbx_ctx* ctx);
int wbx_ctx_synchronize(wbx_ctx* ctx);
int wbx_ctx_device_name(wbx_ctx* ctx, char* buf, size_t buflen);

/* device memory helpers for hosts without their own allocator */
int wbx_malloc(wbx_ctx* ctx, size_t bytes, void** dptr_out);
int wbx_free(wbx_ctx* ctx, void* dptr);
int wbx_memcpy_h2d(wbx_ctx* ctx, void* dptr, const void* h_src, size_t bytes);
int wbx_memcpy_d2h(wbx_ctx* ctx, void* h_dst, const void* dptr, size_t bytes); /* synchronises */
int wbx_memset(wbx_ctx* ctx, void* dptr, int value, size_t bytes);

/* Deferred results.  The reference's chunk loop (beam_pipeline.py:161-250 per chunk, CombinePerKey at :509) only
 * needs a chunk's sums when it combines them; reading them back synchronously after every chunk leaves the GPU idle
 * while the host prepares the next one.  With these the host enqueues the read-back of chunk k into pinned memory,
 * records a fence, launches chunk k+1 and waits on the fence of chunk k only when it combines it.
 *   wbx_host_alloc / wbx_host_free   page-locked host memory (hipHostMalloc)
 *   wbx_memcpy_d2h_async             enqueued on the context stream; `h_pinned` must come from wbx_host_alloc
 *   wbx_fence_*                      hipEvent (no timing) recorded on the context stream; wait blocks the host only */
typedef struct wbx_fence wbx_fence;
int wbx_host_alloc(wbx_ctx* ctx, size_t bytes, void** h_out);
int wbx_host_free(wbx_ctx* ctx, void* h_ptr);
int wbx_memcpy_d2h_async(wbx_ctx* ctx, void* h_pinned, const void* dptr, size_t bytes);
int wbx_fence_create(wbx_ctx* ctx, wbx_fence** out);
int wbx_fence_record(wbx_ctx* ctx, wbx_fence* fence);
int wbx_fence_wait(wbx_fence* fence);
int wbx_fence_destroy(wbx_fence* fence);
/* Work enqueued on `ctx` after this call starts only when `fence` (recorded on ANY context of the same process) has been
 * reached: hipStreamWaitEvent.  Orders a context's kernels behind another context's copies without blocking the host
 * (the chunk feeder's upload stream in front of the launch stream, beam_pipeline.py:69-116 -> :161-250). */
int wbx_ctx_wait_fence(wbx_ctx* ctx, wbx_fence* fence);
/* Asynchronous upload from page-locked memory (wbx_host_alloc) on the context stream: the staging half of the chunk
 * feeder's double buffer.  The source must stay untouched until a fence recorded afterwards has been reached. */
int wbx_memcpy_h2d_async(wbx_ctx* ctx, void* dptr, const void* h_pinned, size_t bytes);
int wbx_memcpy_d2d(wbx_ctx* ctx, void* dst, const void* src, size_t bytes); /* enqueued on the context stream */
/* Host side of the chunk feeder (ABI 13): dst[b][c][r] = src[b][r][c] for `batch` planes of rows x cols elements of 4 or 8
 * bytes, blocked (32 x 32 tiles, AVX2 8 x 8 blocks when the CPU has them).  The loaders hand on whatever dim order the store
 * has (weatherbenchX/data_loaders/xarray_loaders.py:185-188, 236-239: real archives are [.., longitude, latitude]); a file-backed
 * loader copies every chunk into page-locked memory anyway, and with `device_layout='lon_fastest'` that copy is this
 * transposition, so the DMA and the kernels see longitude-fastest fields (zonal transforms: 0.60 of the HBM peak instead of 0.38)
 * and no transposed copy is made on the device.  No context, no stream: plain host memory on both sides, which must not
 * overlap; thread-safe (callers split a chunk's planes over threads). */
int wbx_host_transpose(void* dst, const void* src, int64_t batch, int64_t rows, int64_t cols, int32_t elem_bytes);
/* The shader clock this device sustains with `blocks` x 256 threads of dependent fp32 FMAs running (ABI 13): s_memtime ticks
 * over s_memrealtime's constant 100 MHz, in MHz.  A measurement aid (boxes of one pool differ by 4-8 % on the same kernel):
 * bench.py prints it beside its numbers.  Synchronous; a few milliseconds. */
int wbx_clock_probe(wbx_ctx* ctx, int32_t blocks, double* shader_mhz_out);

/* ---- accumulators ------------------------------------------------------------------------------------------
 * The reference combines per-chunk AggregationStates on the host (beam.CombinePerKey(CombiningSum()),
 * beam_pipeline.py:509-510, beam_utils.py:30-50; AggregationState.sum, aggregation.py:84-110).  Here the stage-2 /
 * binned / spectrum outputs of a chunk stay in HBM and are added into a persistent float64 accumulator right behind
 * the kernel that produced them:
 *     acc[i] = (overwrite ? 0 : acc[i]) + src[i],  i < n          (stream ordered, deterministic, no atomics)
 * A rank's accumulators live in ONE device buffer that is all-reduced in place over RCCL and read back once per job. */
int wbx_acc_add(wbx_ctx* ctx, double* acc, const double* src, int64_t n, int32_t overwrite);

/* ---- accumulators across ranks (one process per GPU) ----------------------------------------------------------
 * The combine stage of the reference -- beam.CombinePerKey(CombiningSum()) over every (statistic, variable, offsets)
 * key of the job, beam_pipeline.py:509-519, beam_utils.py:30-50 -- as ONE collective on device memory:
 *     wbx_acc_allreduce:  acc[i] <- sum over ranks of acc[i], i < n      (ncclAllReduce(sum, double) in place, RCCL / xGMI,
 *                         enqueued on the context's stream: ordered behind the kernels and wbx_acc_add calls before it)
 *     wbx_acc_read:       the buffer on the host (waits for the stream)
 *     wbx_acc_reset:      acc[i] <- 0 (stream ordered)
 * Every rank lays its accumulators out identically (the caller's contract: same statistics and aggregators on every rank,
 * chunks of (init_time x lead_time) dealt round-robin); a slot only one rank wrote is zero on the others, so the
 * same sum also assembles results that keep init_time / lead_time (the reference's ConcatPerStatisticPerVariable,
 * beam_pipeline.py:253-319).
 * A communicator joins `nranks` processes: rank 0 calls wbx_comm_unique_id, hands the WBX_COMM_ID_BYTES bytes to the other
 * ranks by any means (a file, MPI, a torch.distributed store ...), and every rank calls wbx_comm_create on its own context.
 * RCCL is loaded on first use (librccl.so.1; WBX_RCCL_PATH overrides); without it these calls return WBX_ERR_RCCL. */
#define WBX_COMM_ID_BYTES 128
typedef struct wbx_comm wbx_comm; /* opaque: one RCCL communicator (rank, size, device) */
int wbx_comm_unique_id(void* id_out /* WBX_COMM_ID_BYTES */);
int wbx_comm_create(wbx_ctx* ctx, const void* unique_id, int32_t nranks, int32_t rank, wbx_comm** out);
int wbx_comm_destroy(wbx_comm* comm);
int wbx_comm_info(const wbx_comm* comm, int32_t* nranks_out, int32_t* rank_out, int64_t* collectives_out);
int wbx_acc_allreduce(wbx_ctx* ctx, wbx_comm* comm, double* acc, int64_t n);
int wbx_acc_read(wbx_ctx* ctx, const double* acc, int64_t n, double* host_out);
int wbx_acc_reset(wbx_ctx* ctx, double* acc, int64_t n);

/* valid_out[i] = !isnan(data[i]) (one byte per element, 1 = valid) over n contiguous elements: the `mask` coordinate
 * of data_loaders/base.py:25-56 (add_nan_mask_to_data) for payloads that are already in HBM -- the mask is built and
 * consumed (WBX_FLAG_MASKED, input 3) without ever visiting the host.  valid_out must be 4-byte aligned. */
int wbx_notnan_mask(wbx_ctx* ctx, const void* data, int dtype, int64_t n, uint8_t* valid_out);

/* HIP-event timer on the context stream (bench.py's roofline leg). */
int wbx_timer_start(wbx_ctx* ctx);
int wbx_timer_stop(wbx_ctx* ctx, float* ms_out); /* synchronises on the stop event */
/* Timing marks that do NOT synchronise: wbx_mark records a timing event on the context stream and returns its index;
 * wbx_mark_elapsed waits for mark i1 and returns the time between two marks of this context; wbx_marks_reset recycles
 * the events.  bench.py brackets every launch of its timed region with a pair of marks and reads them after the
 * region's closing synchronisation, so the durations it reports are those of the timed launches themselves. */
int wbx_mark(wbx_ctx* ctx, int* index_out);
int wbx_mark_elapsed(wbx_ctx* ctx, int i0, int i1, float* ms_out);
int wbx_marks_reset(wbx_ctx* ctx);

/* ---- stage 1: deterministic ----------------------------------------------
 * Replaces Statistic.compute + the stat-side einsum of Aggregator.aggregate_stat_var
 * (metrics/base.py:184-197, deterministic.py:91-123,222-259, aggregation.py:337-366). */
int wbx_det_partial(wbx_ctx* ctx, const wbx_s1_plan* plan, int func /*wbx_det_func*/,
                    int dtype /*wbx_dtype*/, const void* p, const void* t, const void* c,
                    const uint8_t* mask, double* partial_out);

/* ---- stage 1: ensemble ---------------------------------------------------
 * p has an extra member axis of length M and element stride `member_stride`
 * that is NOT part of key/depth/x.  Replaces probabilistic.py:116-336. */
int wbx_ens_partial(wbx_ctx* ctx, const wbx_s1_plan* plan, int dtype, int M,
                    int64_t member_stride, int algo /*wbx_ens_algo*/, const void* p,
                    const void* t, const uint8_t* mask, double* partial_out);

/* Ensemble-valued predictions AND targets (CRPSEnsembleDistance-style evaluations) with per-point member counts on both sides:
 * skipna_ensemble=True of probabilistic.py:133-145 (CRPSSkill over the non-NaN (prediction member, target member) pairs) and
 * :304-336 (UnbiasedEnsembleMeanSquaredError with both ensembles' own counts).  t has a member axis of length N and element
 * stride `target_member_stride`, like p's M / `member_stride`.  Lanes (partial layout, count lanes and flags as wbx_ens_partial):
 *   0  sum over valid pairs |p_i - t_j| / (n_p n_t)                    NaN without a valid pair
 *   1  (mean p - mean t)^2 - var(p) / n_p - var(t) / n_t  (ddof = 1)   NaN with fewer than two valid members on a side
 * WBX_FLAG_SKIPNA_ENS: NaN members are missing members; without it a NaN member makes both lanes NaN.  From-memory fp64
 * arithmetic for any M, N (O(M N) per point): a correctness path for a rare configuration, no roofline claim. */
#define WBX_ENS2_LANES 2
int wbx_ens2_partial(wbx_ctx* ctx, const wbx_s1_plan* plan, int dtype, int M, int64_t member_stride, int N,
                     int64_t target_member_stride, const void* p, const void* t, const uint8_t* mask, double* partial_out);

/* ---- stage 2: weighted / binned contraction --------------------------------
 * partial is viewed as [nA][nBk][nBr][nchunk][nlane][nj];  W as [nBk][nBr][nj][nbin].
 *   sum_j = 1: out[nA][nBk][nlane][nbin]      = sum_{Br,chunk,j} partial * W
 *   sum_j = 0: out[nA][nBk][nlane][nj][nbin]  = sum_{Br,chunk}   partial * W
 * Plain multiply-add with no zero skipping, so NaN * 0 = NaN poisons a bin exactly
 * as the reference's xr.dot does (aggregation.py:272-277,335). */
typedef struct wbx_s2_plan {
  int64_t nA, nBk, nBr, nchunk, nlane, nj, nbin;
  int32_t sum_j;
} wbx_s2_plan;
int wbx_contract(wbx_ctx* ctx, const wbx_s2_plan* plan, const double* partial,
                 const double* W, double* out);

/* Same contraction for W = wt[Bk][Br][j] * mask[Bk][Br][j][bin] with BOOLEAN masks and nbin <= 64 (Regions / LandSea
 * binning, binning.py:92-201, times GridAreaWeighting): `wt` is float64[nBk][nBr][nj]; bit b of bits[nBk][nBr][nj]
 * says whether the point belongs to bin b.  Result layout and NaN semantics are identical to wbx_contract. */
int wbx_contract_bits(wbx_ctx* ctx, const wbx_s2_plan* plan, const double* partial, const double* wt,
                      const uint64_t* bits, double* out);

/* ---- stage 1: indicator ("categorical") statistics --------------------------------------------------------
 * Per point a 0/1 (or member-fraction) vector along a NEW dimension of `ncat` categories:
 *   WBX_CAT_EXCEED  ErrorExceedance (deterministic.py:262-295) and, with M > 1 members at `member_stride`,
 *                   EnsembleErrorExceedance (probabilistic.py:836-861): lane k = fraction of non-NaN members with
 *                   |p_m - t| > thresholds[k]; NaN when every member's error is NaN.  `thresholds` is a DEVICE
 *                   float64[ncat]; NaN thresholds compare false (the host patches those lanes to NaN).
 *   WBX_CAT_RANK    RankHistogram (probabilistic.py:1306-1343): lane r = [#{m : p_m < t} == r], ncat = M + 1.
 * partial_out[nkey][nchunk][lanes_total][nj] exactly like wbx_det_partial, lanes_total = ncat (+1 with
 * WBX_FLAG_MASKED, x2 with WBX_FLAG_SKIPNA), so wbx_contract / wbx_contract_bits consume it unchanged.
 * Per-thread fp64 counters live in LDS columns (the category index is data dependent): lanes_total <= 128. */
typedef enum wbx_cat_func { WBX_CAT_EXCEED = 0, WBX_CAT_RANK = 1 } wbx_cat_func;
int wbx_cat_partial(wbx_ctx* ctx, const wbx_s1_plan* plan, int func, int dtype, int ncat, int M,
                    int64_t member_stride, const void* p, const void* t, const double* thresholds,
                    const uint8_t* mask, double* partial_out);
/* WBX_CAT_EXCEED against thresholds that depend on the statistic's own dims (one set per level, per latitude ...:
 * deterministic.py:262-295 compares |p - t| with any DataArray that broadcasts).  `threshold_field` is a DEVICE float64 array
 * addressed as input 2 of the plan: threshold k of the point (key, depth, x) is
 *     threshold_field[key_off[2][key] + depth_off[2][depth] + x * xstride[2] + k * category_stride]
 * (zero strides along the dims it does not depend on).  NaN thresholds give NaN indicators, exactly as above. */
int wbx_cat_exceed_field(wbx_ctx* ctx, const wbx_s1_plan* plan, int dtype, int ncat, int M, int64_t member_stride,
                         const void* p, const void* t, const double* threshold_field, int64_t category_stride,
                         const uint8_t* mask, double* partial_out);

/* ---- fused binned reduction (small depth, many boolean bins) -------------------------------------------------
 * Statistic, weight and bin membership in ONE pass over p, t, c -- for chunks where little is reduced before the
 * weight/bin-dependent dims, so that the stage-1 partials would be larger than the inputs (the public benchmark's
 * 1 init x 12 lead chunks with 34 region x land/sea bins, run_benchmark_evaluation.py:97-131,369-382).
 * The plan's keys must be ordered [nA][nBk][nBr] (as for wbx_contract); x is summed; `w_on_x` (WBX_BINNED_* flags) says
 * whether wt / bits are indexed [nBk][nBr][nx] (WBX_BINNED_W_ON_X: W depends on x) or [nBk][nBr][1], and whether the
 * weights come factored: WBX_BINNED_WT_X_ONLY -> wt[nBk][nx] (GridAreaWeighting, weighting.py:62-130, on
 * latitude-fastest chunks), WBX_BINNED_WT_ROW_ONLY -> wt[nBk][nBr] (the same on longitude-fastest chunks); `bits` keeps
 * its full index either way.  out[nA][nBk][lanes_total][nbin], lanes_total and NaN semantics exactly as
 * wbx_det_partial + wbx_contract_bits.  The plan's nchunk / x_kept / vec are ignored. */
#define WBX_BINNED_W_ON_X 1
#define WBX_BINNED_WT_X_ONLY 2
#define WBX_BINNED_WT_ROW_ONLY 4
#define WBX_BINNED_TWIN_MASK 16 /* wbx_ens_binned with WBX_FLAG_MASKED: 12 output lanes instead of 6 -- lanes 0-5 with the mask applied,
                                   lanes 6-11 the same statistics over ALL points (mask ignored) from the same pass over the members */
#define WBX_BINNED_ACCUMULATE 32 /* (ABI 12) out[i] += result[i] instead of out[i] = result[i]: `out` is an accumulator of a chunk loop
                                    (the caller's CombinePerKey(CombiningSum()), beam_pipeline.py:509-510) and the launch adds its sums into
                                    it itself -- one thread per element reads, adds and writes, in the kernel that forms the sums, so
                                    there is no wbx_acc_add launch (4 us + a dependency gap) behind a 0.3 ms chunk.  wbx_det_binned and
                                    wbx_ens_binned; an empty reduction (no rows) then leaves `out` as it is */
#define WBX_BINNED_MASK_ON_W 8 /* the validity mask (WBX_FLAG_MASKED) depends on the Bk / Br / x dims only (a (latitude, longitude)
                                  mask under Regions bins): its byte is folded into the atom-id byte, one load less per point.
                                  wbx_ens_binned without this flag: a per-point mask (strides along any dim), see there */
/* `atoms`: NULL, or the tables wbx_binned_atoms wrote for exactly this geometry (plan extents, nA, nBk, nBr, w_on_x) and
 * these `bits`.  The kernel works on a patch's "atoms" (= its distinct membership words: regions are boxes, so a patch
 * of 64 x ~150 rows sees a handful) and needs every point's atom index; the tables depend on the bins and the geometry
 * only, so a chunk loop computes them once.  With NULL they are recomputed inside every call. */
int wbx_det_binned(wbx_ctx* ctx, const wbx_s1_plan* plan, int func, int dtype, const void* p, const void* t,
                   const void* c, const uint8_t* mask, const double* wt, const uint64_t* bits, int64_t nA,
                   int64_t nBk, int64_t nBr, int32_t w_on_x, int32_t nbin, const void* atoms, double* out);
int wbx_binned_atoms_size(const wbx_s1_plan* plan, int64_t nA, int64_t nBk, int64_t nBr, int32_t w_on_x,
                          int64_t* bytes_out);
int wbx_binned_atoms(wbx_ctx* ctx, const wbx_s1_plan* plan, int64_t nA, int64_t nBk, int64_t nBr, int32_t w_on_x,
                     const uint64_t* bits, void* atoms_out /* device, wbx_binned_atoms_size bytes */);

/* ---- fused binned reduction of the ensemble family ---------------------------------------------------------------
 * The public benchmark's probabilistic configuration (public_benchmark/run_benchmark_evaluation.py:341-354, 365-382): CRPS /
 * spread-skill / ensemble-mean RMSE under Regions x land-sea bins, GridAreaWeighting and masked=True.  Replaces
 * probabilistic.py:116-336 + wrappers.py:116-148 + the two xr.dot of aggregation.py:337-366 for every ensemble statistic of a
 * (predictions, targets) pair: the M members of a point are read ONCE (no full-map partial is written), sorted in registers,
 * and the five lanes go straight into the bins.
 *   out[nA][nBk][6][nbin]: lanes 0-4 = the WBX_ENS_LANES of wbx_ens_partial, weighted and binned like wbx_det_binned does;
 *                          lane 5 = the sum of the weights of the valid points of the bin (the count lane, with and without mask).
 * plan / nA / nBk / nBr / bits / nbin / w_on_x as for wbx_det_binned, with these restrictions (WBX_ERR_INVALID otherwise; the
 * two-stage route wbx_ens_partial + wbx_contract_bits covers the rest):
 *   dtype WBX_F32, algo WBX_ENS_SORT (the pair form gives the same number), 2 <= M <= 64, plan->flags within
 *   FAIR | MASKED | SKIPNA;
 *   weights factored: WBX_BINNED_WT_X_ONLY (wt[nBk][nx]) or WBX_BINNED_WT_ROW_ONLY (wt[nBk][nBr]), or wt = NULL (ones);
 *   a mask (WBX_FLAG_MASKED) needs WBX_BINNED_W_ON_X.  It is addressed as input 3 of the plan (key_off[3] / depth_off[3] /
 *   xstride[3], one byte per element).  With WBX_BINNED_MASK_ON_W the caller says that it depends on the Bk / Br / x dims only (a
 *   (latitude, longitude) mask: zero strides along A and the depth dims).  WITHOUT that flag (ABI 11) it may have strides along
 *   any dim -- the per-point mask data_loaders/base.py:25-56 (add_nan_mask_to_data) builds: `mask = ~isnan(data)` over every dim
 *   of the data, which Aggregator(masked=True) consumes (aggregation.py:339-352).  Same single pass over the members; one more
 *   byte per point is read (209 instead of 208 for 51 members).
 * WBX_BINNED_TWIN_MASK (with a mask): out[nA][nBk][12][nbin] -- lanes 0-5 as above with the mask applied, lanes 6-11 the same six over
 * ALL points.  The reference masks the skill / unbiased-MSE / mean-MSE statistics of a variable whose targets carry a `mask`
 * coordinate but not its spread / variance (statistics of the predictions alone, probabilistic.py:165-273, aggregation.py:339-352):
 * both sets come out of ONE pass (masked-out points are accumulated under their atom's twin).
 * WBX_FLAG_SKIPNA (ABI 11; Aggregator(skipna=True), aggregation.py:343-344, 353-355: `mask & ~stat.isnull()` statistic by
 * statistic): out[nA][nBk][10][nbin] = the five value lanes with their NaN points left out, then five count lanes = the sum of the
 * weights of the points each statistic was valid at (the layout of wbx_ens_partial + wbx_contract_bits under SKIPNA).  Skill,
 * unbiased MSE and MSE of the mean are NaN where the target or a member is; spread and variance where a member is (an infinite
 * member counts as NaN here, as in every rank-form kernel of this library).  With WBX_BINNED_TWIN_MASK: out[nA][nBk][20][nbin] --
 * the ten over the valid points of the mask, then the ten over all points; of the first ten the spread / variance lanes (1, 2)
 * and of the second ten the target lanes (0, 3, 4) are NaN: they are not among the sums the pass forms, and the reference's
 * statistics never ask for them (a statistic of the predictions alone carries no mask).
 * `atoms`: NULL, or the tables wbx_ens_binned_atoms wrote for this geometry and these bits (its patches are shorter than
 * wbx_det_binned's, so the tables are its own).  *overflow_out = number of patches with more than 32 distinct membership
 * words (bins that are not boxes): when it is not zero use the two-stage route -- wbx_ens_binned would return NaN for the
 * cells of those patches.  wbx_ens_binned_atoms synchronises when overflow_out is not NULL.
 * Accuracy: as wbx_ens_partial (fp32 chain sums per point for M = 50 / 51, fp64 from the point's values on); lane 3 is formed
 * per (patch, bin) as lane 4 - lane 2 / M from the fp64 sums (the identity holds at every point). */
int wbx_ens_binned(wbx_ctx* ctx, const wbx_s1_plan* plan, int dtype, int M, int64_t member_stride, int algo, const void* p,
                   const void* t, const uint8_t* mask, const double* wt, const uint64_t* bits, int64_t nA, int64_t nBk,
                   int64_t nBr, int32_t w_on_x, int32_t nbin, const void* atoms, double* out);
int wbx_ens_binned_atoms_size(const wbx_s1_plan* plan, int64_t nA, int64_t nBk, int64_t nBr, int32_t w_on_x,
                              int64_t* bytes_out);
int wbx_ens_binned_atoms(wbx_ctx* ctx, const wbx_s1_plan* plan, int64_t nA, int64_t nBk, int64_t nBr, int32_t w_on_x,
                         const uint64_t* bits, void* atoms_out /* device, wbx_ens_binned_atoms_size bytes */,
                         int64_t* overflow_out /* host, or NULL */);

/* ---- materialisation of per-point statistics --------------------------------
 * Statistic.compute()'s full-resolution result (metrics/base.py:135-158) for callers
 * that really want it (unaggregated pipelines, beam_pipeline.py:563-595).
 * Uses the same plan tables with nchunk == 1, writes out[key][depth][x] (C order, fp64)
 * for one lane of the family. */
int wbx_det_map(wbx_ctx* ctx, const wbx_s1_plan* plan, int func, int dtype, int lane,
                const void* p, const void* t, const void* c, double* out);
int wbx_ens_map(wbx_ctx* ctx, const wbx_s1_plan* plan, int dtype, int M, int64_t member_stride,
                int algo, int lane, const void* p, const void* t, double* out);

/* ---- zonal energy spectrum ---------------------------------------------------
 * Not in the reference snapshot (no spectrum metric, no test) -> parity unpinned; definition of the
 * build (WeatherBench-2 lineage): F = rfft(row) / nlon, S_k = |F_k|^2 * (k == 0 ? 1 : 2), k = 0..nlon/2.
 * `field` holds nrows rows consumed IN PLACE: element i of row r is field[r * row_stride + i * lon_stride]
 * (lon-fastest: lon_stride 1, row_stride nlon; latitude-fastest real data: lon_stride nlat, row_stride 1).
 * Contiguous rows with 2/3/5-smooth length: one fused kernel (in-LDS FFT + |.|^2 reduction, the field is read once);
 * otherwise a batched 1-D R2C rocFFT along longitude, then a HIP |.|^2 reduction:
 *   power_out[group[r]][k] (+)= scale[r] * S_k(row r)
 * group[nrows] (int32 in [0, ngroup)) and scale[nrows] (float64, e.g. area weight / count) are device arrays;
 * power_out is float64[ngroup][nlon/2 + 1]; accumulate = 0 overwrites it, 1 adds to it.
 * Order of the sums (round 5; every route, also wbx_det_spectrum / wbx_det_spectrum_slabs): NO atomics on the result path.
 * A team of the transform kernel sums the rows it walks (a fixed share of the launch geometry, in row order) in registers
 * and stores the sums of each run of one group as a RECORD (group, key = (team, sequence number), nlon/2 + 1 values) in a
 * scratch store of the context; a closing kernel on the same stream then adds every group's records in KEY order -- the
 * result is a function of (inputs, launch geometry) only and bit-identical from run to run
 * (tests/test_gpu_round5.py::test_spectra_are_bit_reproducible, ::test_fused_det_spectra_are_bit_reproducible: 20 runs each).  `group` and `scale` are read by this call's
 * kernels in stream order; the number of group changes along `group` is cached per (pointer, nrows) to size the store and
 * the cache entry is dropped by every write this library makes into the table (wbx_memcpy_h2d*, wbx_memset, wbx_memcpy_d2d):
 * a table rewritten by anybody else's kernel must go through one of those calls first (a stale count that is too small does not
 * give wrong sums: the records that do not fit set every value of the call's result to NaN).
 * Cost against round 4's atomics on the same box: +3.1 % (lon-fastest), +2.7 % (lat-fastest), +3.7 % (fused det + spectra);
 * profiles/r05_spectra_records_ab.txt.
 * Accuracy (the one family that is not held to north_star's 1e-6 per value: the transform itself is fp32, only |F|^2 and the
 * sums over rows are fp64).  1440-point rows (0.25 degree grids, both layouts): a row is shifted by an estimate of its mean
 * before the transform and F_0 is restored in fp64, so the error does not scale with the field's mean --
 * |dS_k| <= 2e-6 S_k + 1e-6 sqrt(S'_max S_k), S'_max = max_{k >= 1} S_k, and S_0 to 1e-6, per row against float64 numpy.fft
 * (tests/test_spectra.py::bound_1440); measured on N(0, 1) and N(280, 1) rows: median 1.4e-7, and 1e-7 for every wavenumber
 * after a mean over 200 rows (profiles/r03_spectrum_demean_ab.txt; without the shift the N(280, 1) rows came out at 1e-5
 * in the median and 6e-5 after that mean).  Every other length the generic fused kernel takes (even, 2/3/5-smooth half length,
 * <= 2048 points: the 64- and 240-point grids of the public configs on one-wave teams, 360 / 720 / 1024 points on teams of two
 * or four waves) is shifted the same way and held to the same bound.  The rocFFT route (odd lengths, prime factors > 5)
 * transforms the rows as they are: |dS_k| <= 2e-5 S_k + 4e-7
 * sqrt(S_max S_k) with S_max including the mean. */
int wbx_zonal_spectrum(wbx_ctx* ctx, const float* field, int64_t lon_stride, int64_t row_stride, int64_t nrows,
                       int32_t nlon, const int32_t* group, const double* scale, int32_t ngroup,
                       int32_t accumulate, double* power_out);

/* The same over `nslab` slabs of `rows_per_slab` uniformly strided rows each: row i of slab o starts at
 * field + h_slab_offsets[o] + i * row_stride (h_slab_offsets is a HOST int64 array of element offsets; group / scale are
 * indexed by o * rows_per_slab + i).  A latitude-fastest field [lead, level, longitude, latitude] is lead x level slabs
 * of `latitude` adjacent rows (row_stride 1, lon_stride = nlat): those are transposed tile by tile into contiguous rows
 * and fed to the fused kernel in one call instead of one strided library batch per slab. */
int wbx_zonal_spectrum_slabs(wbx_ctx* ctx, const float* field, int64_t lon_stride, int64_t row_stride,
                             int64_t rows_per_slab, int64_t nslab, const int64_t* h_slab_offsets, int32_t nlon,
                             const int32_t* group, const double* scale, int32_t ngroup, int32_t accumulate,
                             double* power_out);

/* ---- spectra + deterministic lanes in one sweep ---------------------------------------------------------------------
 * configs[3] / configs[4]: "spectra of p and t + the full deterministic suite in the same sweep".  One launch
 * reads every row of p, t (and c for WBX_DET6) ONCE -- 12 B/point instead of 20 for wbx_det_partial + two wbx_zonal_spectrum
 * launches -- and produces
 *   partial_out[key][lane]      exactly what wbx_det_partial writes for this plan (nchunk = 1): the unweighted per-row sums of
 *                               e, |e|, e^2 (, (p-c)^2, (t-c)^2, (p-c)(t-c)); wbx_contract consumes it unchanged
 *   power_p / power_t[g][k]     what wbx_zonal_spectrum returns for the rows of p resp. t: sum over the rows of group g of
 *                               scale[row] * S_k(row), k = 0 .. 720; overwritten (ordered sums, see wbx_zonal_spectrum)
 * `plan` is the deterministic plan of (p, t[, c]) with x = longitude (nx = 1440, unit x strides, even row offsets), summed,
 * ndepth = 1, nchunk = 1, no mask: a row of the spectra = a key of the plan, group / scale are indexed by key.
 * Spectrum accuracy: as wbx_zonal_spectrum on 1440-point rows (fp32 transform of the mean-shifted rows); the deterministic lanes
 * are fp64 sums of the widened inputs like wbx_det_partial (1e-12).  Parity unpinned for the spectra. */
int wbx_det_spectrum(wbx_ctx* ctx, const wbx_s1_plan* plan, int func /* WBX_DET3 | WBX_DET6 */, int dtype /* WBX_F32 */,
                     const void* p, const void* t, const void* c, const int32_t* group, const double* scale, int64_t ngroup,
                     double* partial_out, double* power_p, double* power_t);

/* The same sweep with stage 2 of the deterministic lanes folded in (round 6).  Where the aggregation's W is a weight per ROW of
 * the plan and the rows of a group are exactly the rows stage 2 would sum into one output -- GridAreaWeighting over
 * (init_time, latitude, longitude), the spectra averaged over (init_time, latitude): configs[3] / configs[4] -- the kernel
 * multiplies every row's sums by det_scale[row] and adds them up with the spectra's records:
 *   det_out[g][lane] = sum over the rows of group g of det_scale[row] * (sum over the row of lane's statistic)
 * = what wbx_contract would make of wbx_det_spectrum's partial_out with that W: [ngroup][3 | 6], overwritten; ordered sums (the
 * records of a group are added in key order), fp64.  It saves the six fp64 wave sums and stores per row (6 % of the kernel) and
 * the 25 MB partial + its contraction.  `group`, `scale`, `det_scale` are device arrays indexed by key. */
int wbx_det_spectrum_folded(wbx_ctx* ctx, const wbx_s1_plan* plan, int func /* WBX_DET3 | WBX_DET6 */, int dtype /* WBX_F32 */,
                            const void* p, const void* t, const void* c, const int32_t* group, const double* scale,
                            const double* det_scale, int64_t ngroup, double* det_out, double* power_p, double* power_t);

/* The same for LATITUDE-FASTEST fields (the layout of the public ERA5 / WeatherBench archives, [.., longitude, latitude];
 * weatherbenchX/data_loaders/xarray_loaders.py:185-188 hands such chunks on as they are stored): `plan` is the deterministic
 * plan of (p, t[, c]) with x = longitude (nx = 1440, summed, ndepth = 1, nchunk = 1, no mask) whose keys are nkey /
 * rows_per_slab slabs of `rows_per_slab` rows each: key o * rows_per_slab + r is row r of slab o, and the rows of a slab are
 * ADJACENT elements of every input (row r + 1 = row r + 1 element; plan->xstride[i] = the longitude stride of input i,
 * >= rows_per_slab).  Outputs and accuracy as wbx_det_spectrum; the deterministic sums of a row are formed in a fixed order
 * (no atomics). */
int wbx_det_spectrum_slabs(wbx_ctx* ctx, const wbx_s1_plan* plan, int func /* WBX_DET3 | WBX_DET6 */, int dtype /* WBX_F32 */,
                           const void* p, const void* t, const void* c, int64_t rows_per_slab, const int32_t* group,
                           const double* scale, int64_t ngroup, double* partial_out, double* power_p, double* power_t);

/* ---- replayable chunk records (ABI 11) ---------------------------------------------------------------------------
 * The body of the reference's per-chunk stage (beam_pipeline.py:161-250: statistics of one chunk -> aggregation states ->
 * CombinePerKey) is, in steady state, the SAME sequence of calls into this library chunk after chunk: same plans, same
 * weights / bins / atom tables, same scratch and accumulator slots; only the inputs' device pointers (and, for statistics
 * against a climatology, the plan variant that carries the chunk's gather table) change.  A caller that has watched one such
 * chunk -- every call with its arguments as 64-bit patterns -- replays the following chunks with ONE call: the host side of
 * a chunk is then a few microseconds whatever the number of metrics, variables and aggregators (a public-benchmark chunk is
 * ~0.35 ms of kernel against 0.3-0.45 ms of Python bookkeeping through the per-call path).
 *   calls[i]   = {fn, nargs, args[]}: entry point WBX_FN_* with its arguments in declaration order, each as the 64-bit pattern
 *                of the value (pointers as addresses, integers sign-extended);
 *   relocs[j]  = {call, arg, slot, offset}: before the calls run, calls[call].args[arg] = slots[slot] + offset -- the
 *                arguments that follow the chunk (input pointers, a plan variant);
 *   slots[]    = this chunk's values.
 * The calls run in order, each on the context it names, exactly as if the caller had made them one by one; the first failure
 * stops the replay and is returned (wbx_last_error says which call).  Only entry points that enqueue work and neither allocate
 * nor synchronise can be part of a record (WBX_ERR_INVALID otherwise): */
typedef enum wbx_fn {
  WBX_FN_DET_PARTIAL = 1, WBX_FN_ENS_PARTIAL = 2, WBX_FN_ENS2_PARTIAL = 3, WBX_FN_CAT_PARTIAL = 4, WBX_FN_CAT_EXCEED_FIELD = 5,
  WBX_FN_CONTRACT = 6, WBX_FN_CONTRACT_BITS = 7, WBX_FN_DET_BINNED = 8, WBX_FN_ENS_BINNED = 9, WBX_FN_ZONAL_SPECTRUM = 10,
  WBX_FN_ZONAL_SPECTRUM_SLABS = 11, WBX_FN_DET_SPECTRUM = 12, WBX_FN_DET_SPECTRUM_SLABS = 13, WBX_FN_ACC_ADD = 14,
  WBX_FN_MEMSET = 15, WBX_FN_MEMCPY_D2D = 16, WBX_FN_CTX_WAIT_FENCE = 17, WBX_FN_FENCE_RECORD = 18, WBX_FN_DET_SPECTRUM_FOLDED = 19
} wbx_fn;
#define WBX_CALL_MAX_ARGS 20
typedef struct wbx_call {
  int32_t fn;     /* wbx_fn */
  int32_t nargs;  /* must equal the entry point's parameter count */
  uint64_t args[WBX_CALL_MAX_ARGS];
} wbx_call;
typedef struct wbx_reloc {
  int32_t call, arg, slot, reserved_;
  int64_t offset; /* bytes */
} wbx_reloc;
/* `calls` is patched in place (it belongs to the caller and is rewritten by every replay). */
int wbx_chunk_replay(wbx_call* calls, int32_t ncalls, const wbx_reloc* relocs, int32_t nrelocs, const uint64_t* slots,
                     int32_t nslots);

#ifdef __cplusplus
}
#endif
#endif /* WBX_H_ */
