"""Runs planned reductions on the device through the C ABI (the only place compute is launched).

reduce_statistics()  = stage 1 (wbx_det_partial / wbx_ens_partial) + stage 2 (wbx_contract)
materialise()        = wbx_det_map / wbx_ens_map (full-resolution statistic, on demand)

Inputs may be host numpy arrays (uploaded once and cached on the DataArray object) or torch
tensors already resident in HBM (consumed in place through their strides: no copy, no transpose).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import dataclasses
import os
from typing import Sequence

import numpy as np

from weatherbenchx_amd import _hip
from weatherbenchx_amd import planner
from weatherbenchx_amd import replay
from weatherbenchx_amd import xarray_lite as xr

_is_torch = xr._is_torch  # pylint: disable=protected-access


class _Dev:
  """A device-resident input: pointer + layout (+ whatever keeps the memory alive)."""

  def __init__(self, ptr, layout, dtype_code, keep, nbytes, fence=None):
    self.ptr, self.layout, self.dtype_code, self.keep, self.nbytes = ptr, layout, dtype_code, keep, nbytes
    self.fence = fence  # an asynchronous upload in flight: consumers make their stream wait on it (no host block)
    self.frozen = []    # host arrays made read-only while the DMA reads them (thaw() gives them back)

  def thaw(self):
    """The upload has been waited for and the device copy is being dropped: the loader may write its arrays again."""
    if self.frozen:
      if self.fence is not None:
        self.fence.wait()
      for arr in self.frozen:
        try:
          arr.flags.writeable = True
        except ValueError:
          pass
      self.frozen = []

  def __del__(self):
    # the DataArray that carried this copy is gone (a chunk that has been aggregated): its page-locked source is the
    # loader's to refill -- the DMA that read it is waited for first
    try:
      self.thaw()
    except Exception:  # pylint: disable=broad-except
      pass


def _order_uploads(ctx, devs):
  """Uploads still in flight on a feeder's copy stream: `ctx`'s stream waits on their events (stream-ordered; the
  host is not blocked)."""
  for d in devs:
    if d is not None and d.fence is not None:
      ctx.wait_fence(d.fence)


def _sync_torch_producers(arrays):
  for a in arrays:
    if a is not None and _is_torch(a) and a.is_cuda:
      import torch  # pylint: disable=g-import-not-at-top
      torch.cuda.current_stream(a.device).synchronize()
      return


def _common_dtype(datas) -> int:
  f64 = False
  for d in datas:
    if d is None:
      continue
    name = str(d.dtype).replace('torch.', '')
    if name != 'float32':
      f64 = True
  return _hip.F64 if f64 else _hip.F32


def _to_device(ctx: _hip.Context, da: xr.DataArray, dtype_code: int) -> _Dev:
  """Device view of a DataArray's payload in dtype `dtype_code` (cached per object and dtype)."""
  cache = da.__dict__.setdefault('_wbx_dev', {})
  if dtype_code in cache:
    return cache[dtype_code]
  data = da.data
  want = np.float32 if dtype_code == _hip.F32 else np.float64
  if _is_torch(data) and data.is_cuda:
    import torch  # pylint: disable=g-import-not-at-top
    tdt = torch.float32 if dtype_code == _hip.F32 else torch.float64
    t = data if data.dtype == tdt else data.to(tdt)
    if t is not data:
      torch.cuda.current_stream(t.device).synchronize()
    lay = planner.layout_of(t, da.dims)
    dev = _Dev(int(t.data_ptr()), lay, dtype_code, t, t.numel() * t.element_size())
  else:
    host = xr._to_numpy(data)  # pylint: disable=protected-access
    fence = None
    if host.dtype == want and host.flags['C_CONTIGUOUS'] and host.nbytes and _hip.is_loader_pinned(host):
      # page-locked source from pipeline.pinned_empty: pure DMA on the context stream, nobody waits on the host; the
      # array is kept with the device copy until both are dropped.  Contract (pinned_empty's docstring): a chunk's arrays
      # are filled once and then left alone -- the next chunk takes fresh ones.  The array is frozen here so that a loader
      # that re-uses it as its own double buffer fails loudly instead of racing the in-flight DMA; any other
      # page-locked memory (views of the read-back pool) goes through the synchronous copy below.
      buf = ctx.upload_async(host)
      fence = ctx.fence()
      keep = (buf, host)
      froze = []
      for arr in (host, data):
        if isinstance(arr, np.ndarray) and arr.flags.writeable:
          try:
            arr.flags.writeable = False
            froze.append(arr)
          except ValueError:
            pass
    else:
      host = np.ascontiguousarray(host, dtype=want)
      buf = keep = ctx.upload(host)
    st = [int(s // host.itemsize) for s in host.strides] if host.ndim else []
    lay = planner.InputLayout(strides=dict(zip(da.dims, st)), itemsize=host.itemsize, base_alignment=256)
    dev = _Dev(buf.ptr, lay, dtype_code, keep, host.nbytes, fence)
    if fence is not None:
      dev.frozen = froze
  cache[dtype_code] = dev
  return dev


def notnan_mask(tensor):
  """bool tensor (True = not NaN) for a float32 / float64 tensor in HBM, in the tensor's own memory layout, written by
  wbx_notnan_mask; layouts that are not one dense block fall back to torch's elementwise isnan."""
  import torch  # pylint: disable=g-import-not-at-top
  code = {torch.float32: _hip.F32, torch.float64: _hip.F64}.get(tensor.dtype)
  order = sorted(range(tensor.dim()), key=lambda i: (-tensor.stride(i), i))
  dense, expect = True, 1
  for i in reversed(order):
    if tensor.size(i) != 1 and tensor.stride(i) != expect:
      dense = False
    expect *= tensor.size(i)
  if code is None or not dense or tensor.numel() == 0:
    return ~torch.isnan(tensor)
  ctx = _hip.default_context(tensor.device.index)
  torch.cuda.current_stream(tensor.device).synchronize()  # whoever produced the data is done
  valid = torch.empty_strided(tuple(tensor.shape), tuple(tensor.stride()), dtype=torch.uint8, device=tensor.device)
  _hip.check(ctx.lib.wbx_notnan_mask(ctx.handle, C.c_void_p(tensor.data_ptr()), code, int(tensor.numel()),
                                     C.c_void_p(valid.data_ptr())), 'wbx_notnan_mask')
  ctx.synchronize()  # the mask may be consumed on another launch stream (ensemble statistics alternate over two)
  return valid.view(torch.bool)


def _mask_to_device(ctx, mask: xr.DataArray) -> _Dev:
  cache = mask.__dict__.setdefault('_wbx_dev', {})
  if 'u8' in cache:
    return cache['u8']
  data = mask.data
  if _is_torch(data) and data.is_cuda:  # built in HBM (data.add_nan_mask_to_data): consumed in place
    import torch  # pylint: disable=g-import-not-at-top
    t = data.view(torch.uint8) if data.dtype == torch.bool else (data != 0).view(torch.uint8)
    lay = planner.InputLayout(strides=dict(zip(mask.dims, [int(s) for s in t.stride()])), itemsize=1,
                              base_alignment=256 if t.data_ptr() % 256 == 0 else (16 if t.data_ptr() % 16 == 0 else 4))
    dev = cache['u8'] = _Dev(int(t.data_ptr()), lay, 'u8', t, t.numel())
    return dev
  host = np.ascontiguousarray(xr._to_numpy(mask.data).astype(bool).astype(np.uint8))  # pylint: disable=protected-access
  buf = ctx.upload(host)
  st = [int(s) for s in host.strides] if host.ndim else []
  dev = _Dev(buf.ptr, planner.InputLayout(strides=dict(zip(mask.dims, st)), itemsize=1, base_alignment=256),
             'u8', buf, host.nbytes)
  cache['u8'] = dev
  return dev


class _PlanOnDevice:
  """ctypes plan struct + the device copies of its tables."""

  def __init__(self, ctx: _hip.Context, plan: planner.S1Plan):
    self.keep = []
    s = _hip.S1PlanStruct()
    s.nkey, s.ndepth, s.nx = plan.nkey, plan.ndepth, plan.nx
    s.x_kept, s.nchunk, s.depth_chunk = int(plan.x_kept), plan.nchunk, plan.depth_chunk
    for i in range(_hip.MAX_INPUTS):
      s.xstride[i] = plan.xstride[i]
      s.key_off[i] = self._up(ctx, plan.key_off[i])
      s.depth_off[i] = self._up(ctx, plan.depth_off[i])
    s.gather_key = self._up(ctx, plan.gather_key)
    s.gather_depth = self._up(ctx, plan.gather_depth)
    s.gather_tab = self._up(ctx, plan.gather_tab)
    s.n_gather_depth = plan.n_gather_depth
    s.flags = plan.flags
    s.block_threads = plan.block_threads
    s.vec = plan.vec
    s.plane_rows = plan.plane_rows
    s.x_weights = self._up(ctx, plan.x_weights)
    self.struct = s

  def _up(self, ctx, tab):
    if tab is None:
      return None
    buf = ctx.upload(np.ascontiguousarray(tab))
    self.keep.append(buf)
    return buf.ptr

  def with_gather_table(self, ctx, gtab: np.ndarray) -> '_PlanOnDevice':
    """The same plan with another climatology gather table (same shape): every other table is shared.  The table is a few
    hundred bytes; it goes up through page-locked memory on the context stream, so the host does not wait for the kernels
    already enqueued (a blocking upload here would drain the chunk pipeline once per chunk)."""
    other = object.__new__(_PlanOnDevice)
    other.struct = type(self.struct).from_buffer_copy(self.struct)
    gtab = np.ascontiguousarray(gtab, dtype=np.int64)
    if hasattr(ctx, 'upload_async'):
      src = ctx.pinned_empty(gtab.shape, np.int64)
      src[...] = gtab
      buf = ctx.upload_async(src)
      other.keep = [self, src, buf]  # (src stays untouched for as long as this variant lives)
    else:
      buf = ctx.upload(gtab)
      other.keep = [self, buf]
    other.struct.gather_tab = buf.ptr
    return other


def _plan_signature(plan: planner.S1Plan):
  def h(t):
    return None if t is None else (t.shape, hash(t.tobytes()))
  return (plan.dims, tuple(plan.sizes.items()), plan.x_dim, plan.x_kept, plan.a_dims, plan.bk_dims, plan.br_dims,
          plan.depth_dims, plan.nchunk, plan.depth_chunk, tuple(plan.xstride), tuple(h(t) for t in plan.key_off),
          tuple(h(t) for t in plan.depth_off), h(plan.gather_key), h(plan.gather_depth), h(plan.gather_tab),
          plan.flags, plan.block_threads, plan.vec, plan.plane_rows, h(plan.x_weights))


_plan_cache: dict = {}
_w_cache: dict = {}


def _device_plan(ctx, plan: planner.S1Plan) -> _PlanOnDevice:
  key = (ctx.device_id, _plan_signature(plan))
  if key not in _plan_cache:
    if len(_plan_cache) > 64:
      _plan_cache.clear()
    _plan_cache[key] = _PlanOnDevice(ctx, plan)
  return _plan_cache[key]


class _WOnDevice:
  """Stage-2 operand: dense W, or (weights, membership bits) for boolean bin masks."""

  def __init__(self, kind, bufs, shape, bin_shape):
    self.kind, self.bufs, self.shape, self.bin_shape = kind, bufs, shape, bin_shape
    self.factored = None  # (WBX_BINNED_WT_* flag, device buffer) when the weights of a 'bits' operand separate
    self.atoms = {}       # launch geometry -> device buffer of wbx_binned_atoms tables


BITS_MIN_BINS = 5  # below this the dense contraction is just as cheap
SEPARABLE_BINNED_WEIGHTS = True  # False: always hand wbx_det_binned the dense wt[nBk][nBr][nj] (A/B timing and tests)


def pack_bits(plan: planner.S1Plan, weights, masks, bin_dims):
  """(wt float64[nBk][nBr][nj], bits uint64[nBk][nBr][nj], nbin, bin_shape) from the labeled factors."""
  wd = plan.bk_dims + plan.br_dims + ((plan.x_dim,) if plan.x_kept and plan.x_dim is not None else ())
  shape_w = [plan.sizes[d] for d in wd]
  sizes = {d: plan.sizes[d] for d in wd}
  if weights is None:
    wt = np.ones(shape_w, dtype=np.float64)
  else:
    extra = [d for d in weights.dims if d not in wd]
    if extra:
      raise ValueError(f'weights depend on dims {extra} that stage 1 does not keep')
    wt = np.broadcast_to(xr._bcast_data(weights.astype(np.float64), wd, sizes), shape_w)  # pylint: disable=protected-access
  bin_dims = tuple(bin_dims)
  bin_shape = tuple(masks.sizes[d] for d in bin_dims)
  nbin = int(np.prod(bin_shape, dtype=np.int64))
  extra = [d for d in masks.dims if d not in wd and d not in bin_dims]
  if extra:
    raise ValueError(f'bin masks depend on dims {extra} that stage 1 does not keep')
  all_dims = list(wd) + list(bin_dims)
  msizes = dict(sizes, **{d: masks.sizes[d] for d in bin_dims})
  m = np.broadcast_to(xr._bcast_data(masks, all_dims, msizes), shape_w + list(bin_shape))  # pylint: disable=protected-access
  m = m.reshape(shape_w + [nbin]).astype(bool)
  bits = np.zeros(shape_w, dtype=np.uint64)
  for b in range(nbin):
    bits |= m[..., b].astype(np.uint64) << np.uint64(b)
  shape4 = (plan.n(plan.bk_dims), plan.n(plan.br_dims), plan.nj, nbin)
  return (np.ascontiguousarray(wt.reshape(shape4[:3])), np.ascontiguousarray(bits.reshape(shape4[:3])), shape4, bin_shape)


def _device_w(ctx, plan: planner.S1Plan, w_da, bin_dims):
  """Stage-2 operand on the device, cached on the (aggregator-cached) labeled W object per stage-1 geometry."""
  sig = (ctx.device_id, plan.bk_dims, plan.br_dims, plan.x_dim if plan.x_kept else None, plan.nj,
         tuple(plan.sizes[d] for d in plan.bk_dims + plan.br_dims), tuple(bin_dims))
  store = _w_cache if w_da is None else w_da.__dict__.setdefault('_wbx_w', {})
  if sig not in store:
    if len(store) > 32:
      store.clear()
    factors = None if w_da is None else w_da.__dict__.get('_wbx_factors')
    nbin = int(np.prod([w_da.sizes[d] for d in bin_dims], dtype=np.int64)) if (w_da is not None and bin_dims) else 1
    if factors is not None and BITS_MIN_BINS <= nbin <= 64:
      wt, bits, shape4, bin_shape = pack_bits(plan, factors[0], factors[1], bin_dims)
      store[sig] = _WOnDevice('bits', (ctx.upload(wt), ctx.upload(bits)), shape4, bin_shape)
      # weights that depend on x only / on the rows only (GridAreaWeighting on latitude- / longitude-fastest data):
      # wbx_det_binned then keeps them in a register / resolves them per row instead of loading 8 bytes per point
      if SEPARABLE_BINNED_WEIGHTS and wt.shape[2] > 1 and np.array_equal(wt, np.broadcast_to(wt[:, :1, :], wt.shape)):
        store[sig].factored = (_hip.BINNED_WT_X_ONLY, ctx.upload(np.ascontiguousarray(wt[:, 0, :])))
      elif SEPARABLE_BINNED_WEIGHTS and np.array_equal(wt, np.broadcast_to(wt[:, :, :1], wt.shape)):
        store[sig].factored = (_hip.BINNED_WT_ROW_ONLY, ctx.upload(np.ascontiguousarray(wt[:, :, 0])))
    else:
      w, bin_shape = dense_w(plan, w_da, bin_dims)
      store[sig] = _WOnDevice('dense', (ctx.upload(w),), w.shape, bin_shape)
  return store[sig]


def _planned(ctx, kind, dims, sizes, layouts, reduce_dims, wdep, gather, flags, x_weights=None, one_wave=False, force_x=None):
  """_planned_inner + a note for the chunk recorder: a plan that carries a gather table follows the chunk's time labels."""
  hit = _planned_inner(ctx, kind, dims, sizes, layouts, reduce_dims, wdep, gather, flags, x_weights, one_wave, force_x)
  rec = replay.active()
  if rec is not None and gather is not None:
    rec.note_gather(hit[1], lambda g: _planned_inner(ctx, kind, dims, sizes, layouts, reduce_dims, wdep, g, flags, x_weights,
                                                     one_wave, force_x)[1])
  return hit


def _planned_inner(ctx, kind, dims, sizes, layouts, reduce_dims, wdep, gather, flags, x_weights=None, one_wave=False, force_x=None):
  """(plan, device plan) through a cheap signature, so steady-state chunks skip table building and uploads.

  The climatology gather table is the one table that follows the chunk's time labels (every chunk of a streamed evaluation
  has its own valid times): the signature holds its dims / shape / 16-byte alignment class only, and a chunk with new labels
  gets the cached plan with that table swapped (`_PlanOnDevice.with_gather_table`) -- rebuilding and uploading the per-key
  tables of a 20 lead x 37 level x 721 latitude chunk (533 540 keys) took 11-20 ms per chunk against 1.5 ms of kernels."""
  gsig = gbytes = None
  if gather is not None:
    gtable = np.ascontiguousarray(gather.table, dtype=np.int64)
    gsig = (tuple(gather.dims), gtable.shape, not bool(np.any(gtable % 4)))  # (vec = 4 needs every offset 16-B aligned)
    gbytes = gtable.tobytes()
    if not SWAP_GATHER_TABLES:
      gsig += (gbytes,)
  sig = (id(ctx), kind, bool(one_wave), force_x, tuple(dims), tuple(sizes[d] for d in dims),
         None if x_weights is None else hash(x_weights.tobytes()),
         tuple(None if l is None else (tuple(sorted(l.strides.items(), key=str)), l.itemsize, l.base_alignment % 16 == 0)
               for l in layouts),
         tuple(sorted(reduce_dims, key=str)), tuple(sorted(wdep, key=str)), flags, gsig)
  hit = _fast_plan_cache.get(sig)
  if hit is None:
    plan = planner.build_s1_plan(dims, sizes, layouts, reduce_dims, wdep_dims=wdep, gather=gather, flags=flags,
                                 allow_vec4=(kind == 'det' and x_weights is None and force_x is None), force_x_dim=force_x,
                                 fold_x=False if x_weights is None else
                                 (True if kind == 'det' else ('point64' if one_wave else 'point')))
    if x_weights is not None and plan.plane_rows > 0:
      assert not plan.x_kept and plan.vec == 1 and plan.nx == x_weights.size
      plan.x_weights = np.ascontiguousarray(x_weights, dtype=np.float64)
    if len(_fast_plan_cache) > 64:
      _fast_plan_cache.clear()
    hit = (plan, _device_plan(ctx, plan), {'built': gbytes})  # + variants of the plan by gather table contents
    _fast_plan_cache[sig] = hit
  plan, dplan, variants = hit
  if gather is None or gbytes == variants['built']:
    return plan, dplan
  var = variants.get(gbytes)
  if var is None:
    gtab = planner.gather_table(plan.key_dims, plan.depth_dims, gather)
    if len(variants) > GATHER_VARIANTS_MAX:
      # start a new generation; the old one is kept (unused) until the next turnover, so that the page-locked source of a
      # table whose upload is still queued on the stream is not handed out again under it
      retired = {k: v for k, v in variants.items() if k not in ('built', 'retired')}
      built = variants['built']
      variants.clear()
      variants['built'], variants['retired'] = built, retired
    plan_v = dataclasses.replace(plan, gather_tab=gtab)
    rec = replay.active()
    if rec is not None:  # (the table's upload is this chunk's own business: a replay makes its own, replay.ChunkRecord)
      rec.paused += 1
    try:
      var = variants[gbytes] = (plan_v, _swap_gather_table(ctx, dplan, plan_v))
    finally:
      if rec is not None:
        rec.paused -= 1
  return var


def _swap_gather_table(ctx, dplan: _PlanOnDevice, plan_v: planner.S1Plan) -> _PlanOnDevice:
  return dplan.with_gather_table(ctx, plan_v.gather_tab)


SWAP_GATHER_TABLES = True  # False: a plan per gather table, as before (A/B timing: tools/bench_new_labels.py)
GATHER_VARIANTS_MAX = 512  # climatology gather tables kept per cached plan (one per distinct set of time labels)
_fast_plan_cache: dict = {}
_thr_cache: dict = {}
_count_cache: dict = {}
# id(root array) -> root array: read-only host arrays that hold data-INDEPENDENT results (sums of weights per geometry).  A view
# of one of them means the same numbers every time it is met, whatever view object carries it.
_const_roots: dict = {}


def register_constant(arr: np.ndarray) -> np.ndarray:
  arr.flags.writeable = False
  _const_roots[id(arr)] = arr
  return arr


def _constant_view_key(a):
  """(root id, address, shape, strides) when `a` is a view of a registered constant, else None."""
  if not isinstance(a, np.ndarray) or a.flags.writeable:
    return None
  root = a
  while isinstance(root, np.ndarray) and root.base is not None:
    root = root.base
  if not isinstance(root, np.ndarray) or _const_roots.get(id(root)) is not root:
    return None
  return (id(root), a.__array_interface__['data'][0], a.shape, a.strides)
_scratch_bufs: dict = {}


def _scratch(ctx, slot: str, nbytes: int):
  """Grow-only scratch allocation per (context, slot): stage-1/2 outputs are consumed before the call returns
  (the final download synchronises the stream), so they can be reused by the next call."""
  key = (id(ctx), slot)
  buf = _scratch_bufs.get(key)
  if buf is None or buf.nbytes < nbytes:
    buf = ctx.alloc(int(nbytes * 1.25) + 256)
    _scratch_bufs[key] = buf
  return buf


def new_context() -> _hip.Context:
  """A second context (own HIP stream) on the default device -- the chunk feeder's copy stream."""
  return _hip.Context(_hip.default_context().device_id)


def stage_inputs(ctx, arrays: Sequence[xr.DataArray]):
  """Uploads the host payloads of `arrays` through `ctx` and caches the device copies on the DataArray objects, in the
  dtype reduce_statistics would pick for them together; device-resident payloads are left alone."""
  arrays = [a for a in arrays if a is not None]
  dtype_code = _common_dtype([a.data for a in arrays])
  for a in arrays:
    if not (_is_torch(a.data) and a.data.is_cuda):
      _to_device(ctx, a, dtype_code)


# Under deferred_results() consecutive ensemble / indicator reductions (the variables of a chunk, consecutive chunks) are
# independent launches: dealt in turn to two contexts (two HIP streams), the tail of one kernel -- its last, partly empty
# round of blocks; these kernels are VALU-bound and only fill the chip ~1.4 times -- overlaps the head of the next.
# Measured on six 51-member ensemble launches of 0.355 ms: 0.319 ms each (tools/bench_two_streams.py); configs[2] step
# 2.28 -> 2.10 ms, CRPS per region 0.63 -> 0.50 ms per chunk.  The HBM-bound deterministic kernels stay on one stream
# (two of them side by side only share the bandwidth; configs[1] measured 3.8 -> 4.0 ms).  Every reduction still runs
# start to end on ONE stream, scratch buffers are per context, cached operands are uploaded synchronously, and the
# state's fence covers every context that got work.
ALTERNATE_STREAMS = os.environ.get('WBX_ALTERNATE_STREAMS', '1') != '0'
# ... and, accumulating (chunk loops), consecutive chunks the other way round (Accumulation.next_chunk): chunk k + 1's kernel
# starts in the tail of chunk k's.  Public probabilistic chunk with chunk records on and two chunks in flight
# (tools/bench_replay.py, profiles/r05_replay.txt): 0.312 against 0.322 ms per chunk on one stream (mask coordinate 0.330 /
# 0.340, NaN mask 0.334 / 0.349; kernel alone 0.312 / 0.327 / 0.333).  (With ONE chunk in flight the pair ran dry for ~50 us
# while the host caught up and the same switch lost 5 %.)  0: every launch of a label on one stream (A/B timing).
ALTERNATE_CHUNKS = os.environ.get('WBX_ALTERNATE_CHUNKS', '1') != '0'
ENS_PIPE = os.environ.get('WBX_ENS_PIPE', '1') != '0'  # the library reads the same variable (csrc/wbx_ens_impl.hpp)
_stream_ring: list = []


def _launch_context(kind: str = 'det'):
  d = _deferred
  base = _hip.default_context()
  if d is None or not ALTERNATE_STREAMS or kind == 'det':
    return base
  if not _stream_ring or _stream_ring[0] is not base:
    _stream_ring[:] = [base, new_context()]
  if _accum is not None:
    # accumulating: the adds into an accumulator slot are ordered by ONE stream, so every stream has slots of its own
    # (Accumulation.accumulate) and a launch of chunk k + 1 may run on the other stream than the same launch of chunk k
    return _stream_ring[_accum.launch_turn(len(_stream_ring))]
  d.turn = (d.turn + 1) % len(_stream_ring)
  return _stream_ring[d.turn]


def launch_contexts():
  """The contexts kernels are launched on: the default one and, once it exists, the second launch stream."""
  base = _hip.default_context()
  return [base] + [c for c in _stream_ring if c is not base]


def known_contexts():
  """Every live context of this process (the default ones, the second launch stream, feeders' copy streams)."""
  return list(_hip.ALL_CONTEXTS or ())


def clear_caches():
  replay.invalidate_all()
  _stream_ring.clear()
  _plan_cache.clear()
  _w_cache.clear()
  _fast_plan_cache.clear()
  _count_cache.clear()
  _const_roots.clear()
  _scratch_bufs.clear()
  _thr_cache.clear()


# Spectra of p and t in the sweep of the deterministic launch (wbx_det_spectrum, csrc/wbx_zspec_det.hpp): the chunk loop
# registers a request per deterministic (p, t) pair whose zonal spectra are about to be aggregated too (pipeline._plan_fusion);
# the stage-1 launch of that pair then computes both spectra alongside its own lanes and leaves them with the source arrays,
# where spectra._run_spectrum finds them instead of launching.  Nothing is registered -> nothing changes.
_fusion_requests: dict = {}   # id(predictions DataArray) -> {'p', 't', 'entry', 'ngroup'}
_fusion_parks: list = []      # arrays that carry a fused spectrum nobody has asked for yet
FUSE_DET_SPECTRA = os.environ.get('WBX_FUSE_DET_SPECTRA', '1') != '0'
# The same fusion on latitude-fastest fields (wbx_det_spectrum_slabs, csrc/wbx_zspec_det_latfast.hpp) is correct but NOT faster
# than the three launches it replaces (configs[3] chunk on MI355X: 1.70 ms against 1.55 ms -- runs of eight adjacent rows use a
# quarter of every 128-byte line a CU asks its L2 for, and the L1 keeps only so many line requests in flight):
# opt-in (WBX_FUSE_DET_SPECTRA_LATFAST=1), the default stays the flat deterministic sweep + one staged-run launch per field.
FUSE_DET_SPECTRA_LATFAST = os.environ.get('WBX_FUSE_DET_SPECTRA_LATFAST', '0') == '1'


def request_det_spectra(p_da, t_da, entry, ngroup: int):
  """`entry`: spectra.LazySpectrum.rows_entry(...) of BOTH fields' spectra (group / scale per row of the fields' row dims)."""
  if FUSE_DET_SPECTRA:
    _fusion_requests[id(p_da)] = {'p': p_da, 't': t_da, 'entry': entry, 'ngroup': int(ngroup)}


def clear_det_spectra_requests():
  """End of a chunk: requests that no deterministic launch picked up, and spectra that were computed but never asked for
  (their buffers go back to the pool; a later, unrelated aggregation of the same array must not find them)."""
  _fusion_requests.clear()
  for da in _fusion_parks:
    da.__dict__.pop('_wbx_fused_spectrum', None)
  _fusion_parks.clear()


def _fused_rows(plan: planner.S1Plan, entry):
  """Key k of the deterministic plan = which row of the spectra's row numbering (C order over entry['row_dims'])?  None when the
  plan's rows are not exactly the fields' rows (a reduced dim that is summed inside stage 1 with more than one element, ...)."""
  row_dims, row_shape = entry['row_dims'], entry['row_shape']
  key_dims = plan.key_dims
  if plan.ndepth != 1 or set(key_dims) | set(plan.depth_dims) != set(row_dims):
    return None
  idx = np.zeros((), dtype=np.int64)
  strides = {}
  mult = 1
  for d, n in zip(reversed(row_dims), reversed(row_shape)):
    strides[d] = mult
    mult *= n
  for d in key_dims:
    idx = idx[..., None] + np.arange(plan.sizes[d], dtype=np.int64) * strides[d]
  return np.ascontiguousarray(np.asarray(idx).reshape(-1))


def _fused_slab_rows(plan: planner.S1Plan, entry, nin: int):
  """Latitude-fastest fields under a plan whose x is the (strided) longitude: rows per slab when the keys are whole slabs of
  ADJACENT rows in every input (wbx_det_spectrum_slabs), else None.  Cached with the spectra's row tables."""
  ckey = ('slabs', tuple(plan.key_dims), tuple(plan.sizes[d] for d in plan.key_dims), tuple(plan.xstride[:nin]))
  hit = entry['dev'].get(ckey)
  if hit is None:
    hit = False
    rows = _fused_rows(plan, entry)
    if rows is not None and rows.size == plan.nkey and plan.key_dims and plan.nchunk == 1 and not plan.x_kept:
      rps = int(plan.sizes[plan.key_dims[-1]])
      ok = rps >= 1 and all(plan.xstride[i] >= rps for i in range(nin))
      for i in range(nin):
        ko = plan.key_off[i]
        ok = ok and ko is not None and (rps == 1 or bool(np.all(np.diff(np.asarray(ko).reshape(-1, rps), axis=1) == 1)))
      if ok and plan.gather_key is not None and rps > 1:  # the climatology slice must not change inside a slab
        ok = bool(np.all(np.diff(np.asarray(plan.gather_key).reshape(-1, rps), axis=1) == 0))
      if ok:
        hit = rps
    entry['dev'][ckey] = hit
  return hit or None


def _fusion_latfast_plan(ctx, inputs, dims, sizes, layouts, reduce_dims, wdep, gather, func, dtype_code):
  """A pending request for the spectra of these (p, t) on fields whose LONGITUDE is strided (latitude-fastest archives): the
  deterministic plan with x = longitude -- rows = the spectra's rows -- when wbx_det_spectrum_slabs can run it, else None (the
  planner's own choice stays: x = the contiguous dim)."""
  if not FUSE_DET_SPECTRA_LATFAST:
    return None
  req = _fusion_requests.get(id(inputs[0]))
  if req is None or req['p'] is not inputs[0] or req['t'] is not inputs[1] or layouts[0] is None:
    return None
  lon = [d for d in dims if d not in req['entry']['row_dims']]
  if len(lon) != 1 or func not in (_hip.DET3, _hip.DET6) or dtype_code != _hip.F32 or ctx is not _hip.default_context():
    return None
  lon = lon[0]
  if (sizes[lon] != 1440 or lon not in set(reduce_dims) or lon in wdep or layouts[0].stride(lon) == 1
      or (gather is not None and lon in gather.dims)):
    return None
  nin = _hip.DET_INPUTS[func]
  if any(layouts[i] is None for i in range(nin)):
    return None
  plan, dplan = _planned(ctx, 'det', dims, sizes, layouts, reduce_dims, wdep, gather, 0, force_x=lon)
  if plan.ndepth != 1 or _fused_slab_rows(plan, req['entry'], nin) is None:
    return None
  return plan, dplan


class _FoldedS2:
  """What `_run_s1` returns when the fused launch has done stage 2 of the deterministic lanes itself (wbx_det_spectrum_folded):
  `res` = (device pointer, shape) of out[nA][nBk][lanes][nj_out][nbin], exactly what `_run_s2` would have returned."""

  def __init__(self, res):
    self.res = res


# Stage 2 of the deterministic lanes inside the fused det + spectra sweep where W is one weight per row of the plan
# (GridAreaWeighting over (init_time, latitude, longitude)) and the spectra's groups are the outputs stage 2 would form: the
# rows' sums are weighted and added up with the spectra's records instead of being stored row by row and contracted afterwards
# (-8 % of the sweep, no 25 MB partial, no contraction launch; tools/kbench_det_spectrum.py).  0: wbx_det_spectrum + wbx_contract.
FOLD_DET_SPECTRA = os.environ.get('WBX_FOLD_DET_SPECTRA', '1') != '0'


def _try_det_spectra(ctx, inputs, dplan, plan, devs, dtype_code, func, make_out, fold=None):
  """The fused launch when a request for these very inputs is pending and the plan qualifies: -> the partial's buffer
  (`make_out()`, written by the launch; both spectra parked with their source arrays) or a `_FoldedS2` (stage 2 done as well:
  `fold` = (s2 plan, W, bin dims, W on the device) of the reduction in progress), else False (the caller launches
  wbx_det_partial as usual)."""
  if not _fusion_requests or inputs is None:
    return False
  req = _fusion_requests.get(id(inputs[0]))
  if req is None or req['p'] is not inputs[0] or req['t'] is not inputs[1]:
    return False
  nin = _hip.DET_INPUTS[func]
  if (func not in (_hip.DET3, _hip.DET6) or dtype_code != _hip.F32 or plan.nx != 1440 or plan.x_kept or plan.ndepth != 1
      or plan.nchunk != 1 or plan.flags or plan.x_weights is not None or plan.x_dim is None or ctx is not _hip.default_context()):
    return False
  entry, ngroup = req['entry'], req['ngroup']
  rps = None
  if any(plan.xstride[i] != 1 for i in range(nin)):  # latitude-fastest fields: slabs of adjacent rows, longitude strided
    rps = _fused_slab_rows(plan, entry, nin)
    if rps is None:
      return False
  fkey = ('fused', tuple(plan.key_dims), tuple(plan.sizes[d] for d in plan.key_dims))
  bufs = entry['dev'].get(fkey)
  if bufs is None:
    rows = _fused_rows(plan, entry)
    if rows is None or rows.size != plan.nkey:
      entry['dev'][fkey] = bufs = False
    else:
      g = np.ascontiguousarray(np.asarray(entry['group'])[rows], dtype=np.int32)
      sc = np.ascontiguousarray(np.asarray(entry['scale'])[rows], dtype=np.float64)
      entry['dev'][fkey] = bufs = (ctx.upload(g), ctx.upload(sc), g)
  if bufs is False:
    return False
  # stage 2 folded in: W is a weight per row (nothing kept along x, no bins, no count lanes) and row (a, br)'s group IS a
  folded = None
  if FOLD_DET_SPECTRA and fold is not None and rps is None:
    s2, w_da, bin_dims, w_buf = fold
    nl = _hip.DET_LANES[func]
    if (s2.nBk == 1 and s2.nbin == 1 and s2.nj == 1 and not plan.x_kept and s2.nchunk == 1 and s2.nlane == nl and s2.nA == ngroup
        and s2.nA * s2.nBr == plan.nkey and w_buf.kind == 'dense'):
      key2 = fkey + ('fold', id(w_buf))
      folded = entry['dev'].get(key2)
      if folded is None:
        folded = False
        w, _ = dense_w(plan, w_da, bin_dims)
        if w.shape == (1, s2.nBr, 1, 1) and np.array_equal(bufs[2], np.repeat(np.arange(s2.nA, dtype=np.int32), s2.nBr)):
          folded = (ctx.upload(np.ascontiguousarray(np.tile(w[0, :, 0, 0], s2.nA))), w_buf)  # (w_buf: its id stays unique)
        entry['dev'][key2] = folded
      folded = folded or None
  del _fusion_requests[id(inputs[0])]
  replay.keep(bufs[:2], folded[0] if folded else None, dplan, devs[2] if nin > 2 else None)
  nk = 721
  # this launch's OWN result buffers (pooled device blocks): several variables of a chunk are launched before the spectra
  # pass reads the first of them, so one scratch slot per context would hand every variable the last variable's spectra
  pw_p = ctx.alloc(max(ngroup * nk, 1) * 8)
  pw_t = ctx.alloc(max(ngroup * nk, 1) * 8)
  ptr = lambda d: C.c_void_p(d.ptr) if d is not None else None

  det_out = out = None
  if folded:
    det_shape = fold[0].out_shape()
    det_out = _scratch(ctx, 's2out', int(np.prod(det_shape, dtype=np.int64)) * 8)
  else:
    out = make_out()

  def call():
    if folded:
      _hip.check(ctx.lib.wbx_det_spectrum_folded(ctx.handle, C.byref(dplan.struct), func, dtype_code, ptr(devs[0]), ptr(devs[1]),
                                                 ptr(devs[2]) if nin > 2 else None, C.c_void_p(bufs[0].ptr), C.c_void_p(bufs[1].ptr),
                                                 C.c_void_p(folded[0].ptr), int(ngroup), C.c_void_p(det_out.ptr),
                                                 C.c_void_p(pw_p.ptr), C.c_void_p(pw_t.ptr)), 'wbx_det_spectrum_folded')
    elif rps is None:
      _hip.check(ctx.lib.wbx_det_spectrum(ctx.handle, C.byref(dplan.struct), func, dtype_code, ptr(devs[0]), ptr(devs[1]),
                                          ptr(devs[2]) if nin > 2 else None, C.c_void_p(bufs[0].ptr), C.c_void_p(bufs[1].ptr),
                                          int(ngroup), C.c_void_p(out.ptr), C.c_void_p(pw_p.ptr), C.c_void_p(pw_t.ptr)),
                 'wbx_det_spectrum')
    else:
      _hip.check(ctx.lib.wbx_det_spectrum_slabs(ctx.handle, C.byref(dplan.struct), func, dtype_code, ptr(devs[0]), ptr(devs[1]),
                                                ptr(devs[2]) if nin > 2 else None, int(rps), C.c_void_p(bufs[0].ptr),
                                                C.c_void_p(bufs[1].ptr), int(ngroup), C.c_void_p(out.ptr), C.c_void_p(pw_p.ptr),
                                                C.c_void_p(pw_t.ptr)), 'wbx_det_spectrum_slabs')
  timed_launch(ctx, call, kind='det_spectrum', rows=int(plan.nkey), func=int(func), slab_rows=int(rps or 0), folded=bool(folded))
  for da, buf in ((req['p'], pw_p), (req['t'], pw_t)):
    # (the buffer object rides along: it is released -- stream ordered behind its consumer -- when the entry is dropped)
    da.__dict__['_wbx_fused_spectrum'] = {'ctx': ctx, 'ptr': buf.ptr, 'buf': buf, 'ngroup': ngroup, 'cache': entry['dev']}
    _fusion_parks.append(da)
  return _FoldedS2((det_out.ptr, det_shape)) if folded else out


def _run_s1(ctx, kind: str, dplan: _PlanOnDevice, plan: planner.S1Plan, devs: Sequence[_Dev | None], dtype_code: int,
            nlanes_total: int, func: int = 0, ens=None, cat=None, inputs=None, fold=None):
  """-> the stage-1 partial's device buffer, or a `_FoldedS2` when the launch has done stage 2 as well (`fold`: see
  _try_det_spectra)."""
  n = int(np.prod(plan.partial_shape(nlanes_total), dtype=np.int64))
  if kind == 'det' and _fusion_requests:
    hit = _try_det_spectra(ctx, inputs, dplan, plan, devs, dtype_code, func, lambda: _scratch(ctx, 'partial', n * 8), fold)
    if hit is not False:
      return hit  # the partial's buffer, or a _FoldedS2
  out = _scratch(ctx, 'partial', n * 8)
  ptr = lambda d: C.c_void_p(d.ptr) if d is not None else None

  def call():  # idempotent: a repetition overwrites the same partial buffer
    if kind == 'det':
      _hip.check(ctx.lib.wbx_det_partial(ctx.handle, C.byref(dplan.struct), func, dtype_code, ptr(devs[0]),
                                         ptr(devs[1]), ptr(devs[2]), ptr(devs[3]), C.c_void_p(out.ptr)), 'wbx_det_partial')
    elif kind == 'cat' and cat[5] is not None:  # thresholds that depend on the statistic's dims: input 2 of the plan
      cfunc, ncat, m, mstride, _, cstride = cat
      _hip.check(ctx.lib.wbx_cat_exceed_field(ctx.handle, C.byref(dplan.struct), dtype_code, int(ncat), int(m), int(mstride),
                                              ptr(devs[0]), ptr(devs[1]), ptr(devs[2]), int(cstride), ptr(devs[3]),
                                              C.c_void_p(out.ptr)), 'wbx_cat_exceed_field')
    elif kind == 'ens2':
      m, mstride, n_t, tstride = ens
      _hip.check(ctx.lib.wbx_ens2_partial(ctx.handle, C.byref(dplan.struct), dtype_code, int(m), int(mstride), int(n_t), int(tstride),
                                          ptr(devs[0]), ptr(devs[1]), ptr(devs[3]), C.c_void_p(out.ptr)), 'wbx_ens2_partial')
    elif kind == 'cat':
      cfunc, ncat, m, mstride, thr, _ = cat
      _hip.check(ctx.lib.wbx_cat_partial(ctx.handle, C.byref(dplan.struct), int(cfunc), dtype_code, int(ncat), int(m),
                                         int(mstride), ptr(devs[0]), ptr(devs[1]), ptr(thr), ptr(devs[3]),
                                         C.c_void_p(out.ptr)), 'wbx_cat_partial')
    else:
      m, mstride, algo = ens
      _hip.check(ctx.lib.wbx_ens_partial(ctx.handle, C.byref(dplan.struct), dtype_code, int(m), int(mstride),
                                         int(algo), ptr(devs[0]), ptr(devs[1]), ptr(devs[3]), C.c_void_p(out.ptr)), 'wbx_ens_partial')
  if S1_EVENT_LOG is None:
    call()
  else:
    timed_launch(ctx, call, kind=kind, vec=plan.vec, x_kept=plan.x_kept, plane_rows=plan.plane_rows,
                 x_weighted=plan.x_weights is not None, flat=plan.x_weights is not None and plan.plane_rows > 0,
                 grid=plan.nkey * plan.nchunk, block=plan.block_threads, algo=int(ens[2]) if kind == 'ens' else None,
                 flags=int(plan.flags))
  return out


# When set to a list, every stage-1 launch is bracketed by HIP events (synchronising) and logged here.
S1_EVENT_LOG = None
# launches per event pair in that mode (amortises the ~60 us event/dispatch overhead of a single launch)
S1_EVENT_REPEAT = 1
# True: the launches are bracketed by timing MARKS instead (wbx_mark: nothing waits, one launch per pair), so a pipelined
# loop runs exactly as it does untimed; `resolve_event_marks(log)` fills in the durations once the loop has been waited for.
S1_EVENT_MARKS = False


def timed_launch(ctx, call, **entry):
  """Runs `call` (one launch on ctx's stream); with S1_EVENT_LOG set, brackets it with HIP events and logs `entry`."""
  if S1_EVENT_LOG is None:
    call()
  elif S1_EVENT_MARKS:
    i0 = ctx.mark()
    call()
    S1_EVENT_LOG.append(dict(entry, ms=None, reps=1, marks=(ctx, i0, ctx.mark())))
  else:
    reps = max(1, int(S1_EVENT_REPEAT))
    ctx.timer_start()
    for _ in range(reps):
      call()
    S1_EVENT_LOG.append(dict(entry, ms=ctx.timer_stop() / reps, reps=reps))


def resolve_event_marks(log):
  """Durations of the entries logged under S1_EVENT_MARKS (waits for each entry's closing mark); recycles the marks."""
  ctxs = {}
  for e in log:
    if e.get('marks') is not None:
      ctx, i0, i1 = e.pop('marks')
      e['ms'] = ctx.mark_elapsed(i0, i1)
      ctxs[id(ctx)] = ctx
  for ctx in ctxs.values():
    ctx.marks_reset()
  return log


def _run_map(ctx, kind: str, dplan, plan: planner.S1Plan, devs, dtype_code: int, lane: int, func: int = 0,
             ens=None) -> np.ndarray:
  n = plan.nkey * plan.ndepth * plan.nx
  out = ctx.alloc(n * 8)
  ptr = lambda d: C.c_void_p(d.ptr) if d is not None else None
  if kind == 'det':
    _hip.check(ctx.lib.wbx_det_map(ctx.handle, C.byref(dplan.struct), func, dtype_code, lane, ptr(devs[0]),
                                   ptr(devs[1]), ptr(devs[2]), C.c_void_p(out.ptr)), 'wbx_det_map')
  else:
    m, mstride, algo = ens
    _hip.check(ctx.lib.wbx_ens_map(ctx.handle, C.byref(dplan.struct), dtype_code, int(m), int(mstride), int(algo),
                                   lane, ptr(devs[0]), ptr(devs[1]), C.c_void_p(out.ptr)), 'wbx_ens_map')
  return ctx.download(out.ptr, (n,), np.float64)


class DeferredResults:
  """Active inside `deferred_results()`: stage-2 / binned outputs are read back asynchronously into page-locked
  memory and `reduce_statistics` returns views whose CONTENT is valid only after the fence of `mark()` is waited on."""

  def __init__(self):
    self.ctxs = {}       # contexts with read-backs enqueued since the last mark()
    self.turn = 0        # whose turn it is among the launch contexts (_launch_context)
    self.keepalive = []  # payloads the enqueued kernels read (torch tensors, cached device buffers)

  def mark(self):
    """Fence covering every read-back enqueued so far (a no-op object when there was none)."""
    fences = [ctx.fence() for ctx in self.ctxs.values()]
    keep, self.ctxs, self.keepalive = self.keepalive, {}, []
    return ResultFence(fences, keep)


class ResultFence:
  """Holds the kernels' inputs until they have run: torch's caching allocator would otherwise hand a dropped
  tensor's memory to the next chunk while a kernel on the wbx stream is still reading it."""

  def __init__(self, fences=(), keepalive=()):
    self._fences = list(fences)
    self._keepalive = list(keepalive)

  def wait(self):
    for f in self._fences:
      f.wait()
    self._fences = []
    self._keepalive = []

  def __del__(self):
    # a state dropped without ever being looked at: its inputs must still outlive the kernels that read them
    try:
      self.wait()
    except Exception:  # pylint: disable=broad-except
      pass


_deferred: DeferredResults | None = None


@contextlib.contextmanager
def deferred_results():
  """Opt-in overlap of the host with the GPU: inside the block `Aggregator.aggregate_statistics` returns at once with
  an AggregationState that waits for its own results on first use (`state.wait()`, or any of its methods).  Reading
  the raw arrays of such a state before that is undefined.  pipeline.evaluate_chunks and bench.py use it to launch
  chunk k+1 before combining chunk k."""
  global _deferred
  prev, mine = _deferred, DeferredResults()
  _deferred = mine
  try:
    yield mine
  finally:
    _deferred = prev
    mine.mark().wait()


def deferred_active() -> DeferredResults | None:
  return _deferred


@contextlib.contextmanager
def synchronous_results():
  """Inside an enclosing `deferred_results()` / `accumulate_results()` block: results requested here are complete host
  arrays when the call returns (for callers that post-process the numbers on the host right away)."""
  global _deferred, _accum
  saved, _deferred = _deferred, None
  saved_acc, _accum = _accum, None
  try:
    yield
  finally:
    _deferred, _accum = saved, saved_acc


# Under deferred_results() (and no accumulation) a reduction whose LAST kernel writes every element of its result exactly
# once writes it straight into page-locked host memory: the separate device-to-host copy -- a blit kernel behind ~12 us
# of dependency latency, 5 % of a public-benchmark chunk -- disappears.
DIRECT_RESULTS = os.environ.get('WBX_DIRECT_RESULTS', '1') != '0'


class _HostResult:
  """A result that its kernel has been told to write into page-locked host memory (`view`, valid after the state's fence)."""

  def __init__(self, view):
    self.view = view


class _AccumulatedInPlace:
  """What _result_target hands to _deliver when the kernel adds its result into the accumulator slot itself."""

  def __init__(self, ptr):
    self.ptr = ptr


# WBX_FUSED_ACC_ADD=0: every chunk result goes through a scratch buffer + wbx_acc_add (A/B; the round-4 path)
FUSED_ACC_ADD = os.environ.get('WBX_FUSED_ACC_ADD', '1') != '0'


def _result_target(ctx, shape, scratch_name, can_accumulate=False):
  """(pointer the kernel writes its float64 `shape` result to, what to hand to _deliver[, add?]).  With `can_accumulate` (the
  launch understands WBX_BINNED_ACCUMULATE) a third value says whether the pointer is the chunk loop's accumulator slot of this
  result -- it exists from the second chunk on -- and the kernel must ADD into it: no scratch round trip, no wbx_acc_add launch."""
  n = int(np.prod(shape, dtype=np.int64))
  if can_accumulate:
    if FUSED_ACC_ADD and _accum is not None and n and S1_EVENT_LOG is None:  # (timed launches are repeated: they must not add)
      ptr = _accum.slot_pointer(ctx, n)
      if ptr is not None:
        return ptr, _AccumulatedInPlace(ptr), True
    return _result_target(ctx, shape, scratch_name) + (False,)
  if DIRECT_RESULTS and _deferred is not None and _accum is None and n:
    view = ctx.pinned_result(shape)
    return view.ctypes.data, _HostResult(view)
  out = _scratch(ctx, scratch_name, n * 8)
  return out.ptr, out.ptr


def _deliver(ctx, ptr, shape) -> np.ndarray:
  """Hands a finished device result (float64 `shape` at `ptr`) to the caller: added into the active accumulation
  (the returned view only carries the layout, see Accumulation), read back asynchronously under deferred_results(),
  or downloaded right away."""
  if isinstance(ptr, _HostResult):  # already on its way into page-locked memory (_result_target)
    _deferred.ctxs[id(ctx)] = ctx
    return ptr.view
  if _accum is not None:
    if _deferred is not None:
      _deferred.ctxs[id(ctx)] = ctx
    if isinstance(ptr, _AccumulatedInPlace):
      return _accum.accumulate(ctx, None, shape, in_place=ptr.ptr)
    return _accum.accumulate(ctx, ptr, shape)
  if _deferred is not None:
    _deferred.ctxs[id(ctx)] = ctx
    return ctx.download_async(ptr, shape)
  return ctx.download(ptr, shape, np.float64)


_download = _deliver  # (name kept for callers outside this module)


def _acc_add(ctx, dst_buf, dst_off: int, src_ptr, n: int, overwrite: bool):
  """acc[dst_off : dst_off + n] (+)= src on the context stream (wbx_acc_add)."""
  _hip.check(ctx.lib.wbx_acc_add(ctx.handle, C.c_void_p(dst_buf.ptr + 8 * int(dst_off)), C.c_void_p(src_ptr), int(n),
                                 int(bool(overwrite))), 'wbx_acc_add')


class _AccBlock:
  """A piece of the accumulator arena: device memory plus a (never read) host array of the same length whose only job
  is to give numpy views an address -- (offset, shape, strides) of every result view is recovered from it."""

  def __init__(self, ctx, nelem: int):
    self.ctx = ctx
    self.dev = ctx.alloc(nelem * 8)
    self.shadow = np.full(nelem, np.nan, dtype=np.float64)  # NaN: reading a layout-only view by mistake is loud
    self.base = self.shadow.__array_interface__['data'][0]
    self.cap, self.used = int(nelem), 0
    self.starts: list = []  # slot offsets in allocation order (ascending)
    self.keys: list = []


class Accumulation:
  """Device-resident accumulators of a chunk loop (the CombinePerKey(CombiningSum()) stage, beam_pipeline.py:509-510,
  kept in HBM).

  While it is active (`accumulate_results`), every reduction's device result is ADDED into a slot of this arena instead of
  being read back; the slot is identified by (label, ordinal of the result under that label), so the same launch of a
  later chunk with the same label lands on the same slot.  The arrays the Aggregator hands out meanwhile are views of a
  shadow array: they carry dims / shape / strides but no numbers.  `capture()` records where such a view lives;
  `distributed.reduce_accumulation` all-reduces the arena across ranks (one collective on the device buffer), reads it
  back once and rebuilds every captured array on the result."""

  BLOCK_ELEMS = 1 << 20  # 8 MB of accumulators per block; a larger slot gets a block of its own

  def __init__(self):
    self.blocks: list[_AccBlock] = []
    self.slots: dict = {}          # key -> (block, offset, n, ctx)
    self.label = None
    self._ordinal = 0
    self._launches = 0
    self.chunk_index = 0
    self._turns: dict = {}
    self.multi = False             # some slot has been added to more than once
    self._host_sum: dict = {}      # path -> DataArray summed on the host (results that never were on the device)
    self._host_const: dict = {}    # path -> [[read-only constant DataArray, multiplicity], ...]
    self.specs: dict = {}          # path -> [spec, ...]
    self.frames: dict = {}         # (path, index) -> (coords, name, attrs)
    self.ctxs: dict = {}
    self._recorder = None          # replay.ChunkRecorder while a chunk of this loop is being recorded

  # -- bookkeeping driven by the chunk loop ---------------------------------------------------------------------
  def set_label(self, label):
    """Names what is being aggregated next (aggregator, statistic, variable, chunk offsets that survive): results under
    the same label accumulate."""
    self.label, self._ordinal, self._launches = label, 0, 0

  def next_chunk(self):
    """The chunk loop moves on (pipeline._consume).  Consecutive chunks deal their ensemble launches to the launch streams the
    other way round: chunk k + 1's kernel then starts in the tail of chunk k's -- the last, partly empty round of blocks, ~80 us of
    a 0.32 ms ens_atoms_kernel -- instead of behind it and behind its accumulator add.  Each stream adds into slots of its own."""
    self.chunk_index += 1

  def launch_turn(self, n: int) -> int:
    key = (self.label, self._launches)
    self._launches += 1
    turn = self._turns.get(key)
    if turn is None:
      turn = self._turns[key] = len(self._turns) % n
    return (turn + (self.chunk_index if ALTERNATE_CHUNKS else 0)) % n

  # -- engine side ------------------------------------------------------------------------------------------------
  def _slot_key(self, ctx):
    key = (self.label, self._ordinal)
    turn = next((i for i, c in enumerate(_stream_ring) if c is ctx), 0)
    if turn:
      key += (turn,)  # this launch stream's own slot for the result: its adds are ordered by its stream alone
    return key

  def slot_pointer(self, ctx, n: int):
    """Device address of the slot the NEXT accumulate() on `ctx` will add `n` values into, or None while that slot does not
    exist (the first chunk creates it) -- for launches that add into the slot themselves (WBX_BINNED_ACCUMULATE)."""
    slot = self.slots.get(self._slot_key(ctx))
    if slot is None or slot[2] != n or slot[3] is not ctx:
      return None
    return slot[0].dev.ptr + 8 * int(slot[1])

  def accumulate(self, ctx, src_ptr, shape, in_place=None) -> np.ndarray:
    """Adds the float64 `shape` result at `src_ptr` into its slot (wbx_acc_add).  `in_place` = the slot address a launch was
    given by slot_pointer(): the kernel has added its result already, only the bookkeeping is left."""
    n = int(np.prod(shape, dtype=np.int64))
    key = self._slot_key(ctx)
    self._ordinal += 1
    slot = self.slots.get(key)
    if in_place is not None and (slot is None or slot[0].dev.ptr + 8 * int(slot[1]) != in_place or slot[2] != n):
      raise RuntimeError(f'accumulating {key}: a launch added into an address that is not this result\'s slot')
    if slot is None:
      blk = next((b for b in self.blocks if b.ctx is ctx and b.cap - b.used >= n), None)
      if blk is None:
        blk = _AccBlock(ctx, max(self.BLOCK_ELEMS, n))
        self.blocks.append(blk)
      slot = self.slots[key] = (blk, blk.used, n, ctx)
      blk.starts.append(blk.used)
      blk.keys.append(key)
      blk.used += n
      first = True
      if self._recorder is not None:
        self._recorder.refuse('an accumulator slot was created in the chunk (its first add overwrites)')
    else:
      if slot[2] != n:
        raise ValueError(f'accumulating {key}: this chunk produced {n} values where earlier chunks produced {slot[2]} '
                         '(chunks under one label must have the same result layout)')
      if slot[3] is not ctx:
        raise RuntimeError(f'accumulating {key}: launched on another stream than before')
      first = False
      self.multi = True
    blk, off = slot[0], slot[1]
    if n and in_place is None:
      _acc_add(ctx, blk.dev, off, src_ptr, n, first)
    self.ctxs[id(ctx)] = ctx
    return blk.shadow[off:off + n].reshape(shape)

  def locate(self, arr):
    """(slot key, offset inside the slot, shape, element strides) of a view handed out by accumulate(), else None."""
    if not isinstance(arr, np.ndarray) or arr.dtype != np.float64 or arr.size == 0:
      return None
    addr = arr.__array_interface__['data'][0]
    for blk in self.blocks:
      if blk.base <= addr < blk.base + 8 * blk.used:
        pos = (addr - blk.base) // 8
        import bisect  # pylint: disable=g-import-not-at-top
        i = bisect.bisect_right(blk.starts, pos) - 1
        key = blk.keys[i]
        n = self.slots[key][2]
        rel = pos - blk.starts[i]
        strides = tuple(int(st // 8) for st in arr.strides)
        last = rel + sum((sz - 1) * st for sz, st in zip(arr.shape, strides))
        if not 0 <= last < n:
          raise RuntimeError('a result view reaches outside its accumulator slot')
        return key, int(rel), tuple(int(x) for x in arr.shape), strides
    return None

  # -- result side ------------------------------------------------------------------------------------------------
  def capture(self, path, da, coeff: float = 1.0):
    """Remembers that the leaf `path` of the final result is (the sum over captures of) coeff * this array.  Arrays that
    are not views of the arena hold finished numbers: they are summed on the host under `path`."""
    terms = getattr(da, '_linear_terms', None)
    if terms is not None:  # a linear combination of pending arrays (aggregation._PendingLinear)
      for c, term in terms:
        self.capture(path, term, coeff * c)
      return
    loc = self.locate(da.data)
    if loc is None:
      data = da.data
      ckey = _constant_view_key(data)
      if ckey is not None or (isinstance(data, np.ndarray) and not data.flags.writeable and data.flags.owndata):
        # a cached constant (data-independent sums of weights handed out read-only, or ANY view of a registered one): met
        # again under the same path with the same frame it only bumps a multiplicity, folded into the host sums when they
        # are read (`host`)
        for index, term in enumerate(self._host_const.setdefault(path, [])):
          same = term[2] == ckey if ckey is not None else term[0].data is data
          if same and term[0].dims == da.dims and _same_coords(term[0], da):
            term[1] += coeff
            if self._recorder is not None:
              self._recorder.bumps.append((path, index, float(coeff)))
            return
        self._host_const[path].append([da, float(coeff), ckey])
        if self._recorder is not None:
          self._recorder.refuse('a constant term was met for the first time in the chunk')
        return
      self._host_add(self._host_sum, path, da if coeff == 1.0 else da * coeff)
      if self._recorder is not None:
        self._recorder.refuse('a result was summed on the host in the chunk')
      return
    spec = loc + (tuple(da.dims), float(coeff))
    lst = self.specs.setdefault(path, [])
    if spec not in lst:
      if self._recorder is not None:
        self._recorder.refuse('a result layout was met for the first time in the chunk')
      lst.append(spec)
      self.frames[(path, len(lst) - 1)] = (dict(da._coords), da.name, dict(da.attrs))  # pylint: disable=protected-access
    else:
      # the same slot again: its labels must be the ones it was created with -- a chunk whose bins carry other labels (bins over
      # the values of a coordinate of sparse data) would otherwise be added position by position under the first chunk's labels
      seen = self.frames[(path, lst.index(spec))][0]
      mine = da._coords  # pylint: disable=protected-access
      for k, (cdims, cvals) in mine.items():
        if k in da.dims and k in seen and seen[k][1] is not cvals and not np.array_equal(np.asarray(seen[k][1]), np.asarray(cvals)):
          raise ValueError(
              f'accumulating {path}: the labels of dimension {k!r} changed between chunks.  Results whose frame depends on the '
              "chunk's data cannot use the device accumulators: run beam_pipeline.define_pipeline(..., accumulate='host') or add "
              'the per-chunk AggregationStates yourself.')

  @staticmethod
  def _host_add(table, path, val):
    if path in table:
      a, b = xr.align(table[path], val, join='outer', fill_value=0)
      val = a + b
    table[path] = val

  @property
  def host(self) -> dict:
    """path -> DataArray: the results that never were on the device, summed over the captures."""
    out = dict(self._host_sum)
    for path, terms in self._host_const.items():
      for da, mult, _ in terms:
        self._host_add(out, path, da if mult == 1.0 else da * mult)
    return out

  def synchronize(self):
    for ctx in self.ctxs.values():
      ctx.synchronize()

  def slot_table(self):
    """[(key, n)] in allocation order."""
    return [(k, v[2]) for k, v in self.slots.items()]

  def download_slots(self) -> dict:
    """key -> float64 ndarray (one synchronous read-back per block)."""
    out = {}
    for blk in self.blocks:
      if not blk.used:
        continue
      host = blk.ctx.download(blk.dev.ptr, (blk.used,), np.float64)
      for start, key in zip(blk.starts, blk.keys):
        out[key] = host[start:start + self.slots[key][2]]
    return out


def _same_coords(a: xr.DataArray, b: xr.DataArray) -> bool:
  ca, cb = a._coords, b._coords  # pylint: disable=protected-access
  if ca.keys() != cb.keys():
    return False
  for k, (dims, vals) in ca.items():
    odims, ovals = cb[k]
    if dims != odims or (vals is not ovals and not np.array_equal(np.asarray(vals), np.asarray(ovals))):
      return False
  return True


_accum: Accumulation | None = None


@contextlib.contextmanager
def accumulate_results(acc: Accumulation):
  """Inside the block (which also is a `deferred_results()` block: nothing is waited for) every reduction adds its device
  result into `acc`; the states the Aggregator returns carry layout only.  Resolve with
  `distributed.reduce_accumulation(acc)` / `distributed.resolve_state(state, acc)` after the block."""
  global _accum
  prev = _accum
  with deferred_results() as d:
    _accum = acc
    try:
      yield d
    finally:
      _accum = prev


def accumulation_active() -> Accumulation | None:
  return _accum


def _run_s2(ctx, s2: planner.S2Plan, partial_ptr: int, w_buf):
  """-> (device pointer, shape) of out[nA][nBk][lanes][nj_out][nbin]."""
  st = _hip.S2PlanStruct(s2.nA, s2.nBk, s2.nBr, s2.nchunk, s2.nlane, s2.nj, s2.nbin, int(s2.sum_j))
  shape = s2.out_shape()
  n = int(np.prod(shape, dtype=np.int64))
  out = _scratch(ctx, 's2out', n * 8)
  if w_buf.kind == 'bits':
    _hip.check(ctx.lib.wbx_contract_bits(ctx.handle, C.byref(st), C.c_void_p(partial_ptr), C.c_void_p(w_buf.bufs[0].ptr),
                                         C.c_void_p(w_buf.bufs[1].ptr), C.c_void_p(out.ptr)), 'wbx_contract_bits')
  else:
    _hip.check(ctx.lib.wbx_contract(ctx.handle, C.byref(st), C.c_void_p(partial_ptr), C.c_void_p(w_buf.bufs[0].ptr),
                                    C.c_void_p(out.ptr)), 'wbx_contract')
  return out.ptr, shape


# 'auto': the fused binned kernel runs when the stage-1 partials would exceed BINNED_PARTIAL_RATIO x the input bytes
# (little is reduced before the weight/bin-dependent dims); 'never' / 'always' (whenever eligible) are for A/B timing.
# Measured on MI355X (34 region bins, 0.25 deg): fused ~0.43 ms per GB of inputs; two-stage ~0.2 ms per GB of inputs
# plus ~0.52 ms per GB of partials (written by stage 1, read back by the stage-2 patch kernel) -> break-even where
# the partials are ~45 % of the inputs (about 10 inits per chunk for the 6-lane family).
BINNED_MODE = 'auto'
PREPARED_ATOMS = True  # False: wbx_det_binned recomputes the atom tables in every call (A/B timing and tests)
FOLD_X_WEIGHTS = True  # False: keep x for stage 2 (plane mode / x-kept kernels), for A/B timing and tests
BINNED_PARTIAL_RATIO = 0.45


def _binned_eligible(kind, plan: planner.S1Plan, w_buf, devs, nl_total: int, nin: int) -> bool:
  if kind != 'det' or w_buf.kind != 'bits' or BINNED_MODE == 'never':
    return False
  if plan.x_kept and not plan.sum_j:  # x survives into the output: the two-stage path keeps it
    return False
  if BINNED_MODE == 'always':
    return True
  partial_bytes = int(np.prod(plan.partial_shape(nl_total), dtype=np.int64)) * 8
  input_bytes = plan.nkey * plan.ndepth * plan.nx * devs[0].layout.itemsize * nin
  return partial_bytes > BINNED_PARTIAL_RATIO * input_bytes


def _run_binned(ctx, dplan: _PlanOnDevice, plan: planner.S1Plan, devs, dtype_code: int, nl_total: int, func: int,
                w_buf):
  """wbx_det_binned -> (device pointer, shape) of out[nA][nBk][lanes][1][nbin] (like the sum_j stage-2 result)."""
  nA, nBk, nBr = plan.n(plan.a_dims), plan.n(plan.bk_dims), plan.n(plan.br_dims)
  nbin = w_buf.shape[-1]
  shape = (nA, nBk, nl_total, 1, nbin)
  out_ptr, handle, add = _result_target(ctx, shape, 's2out', can_accumulate=True)  # (the finish kernel writes every element once)
  ptr = lambda d: C.c_void_p(d.ptr) if d is not None else None
  w_flags = _hip.BINNED_W_ON_X if (plan.x_kept and plan.nj > 1) else 0
  wt_buf = w_buf.bufs[0]
  if w_buf.factored is not None and SEPARABLE_BINNED_WEIGHTS:
    w_flags |= w_buf.factored[0]
    wt_buf = w_buf.factored[1]
  # a validity mask that lives on the W dims only (a (latitude, longitude) land / NaN mask: zero stride along every A and
  # depth dim) is folded into the atom-id byte by the library: one vector-memory instruction less per row
  if devs[3] is not None and (w_flags & _hip.BINNED_W_ON_X):
    if all(devs[3].layout.stride(d) == 0 for d in tuple(plan.a_dims) + tuple(plan.depth_dims)):
      w_flags |= _hip.BINNED_MASK_ON_W
  # the atom tables (a patch's distinct membership words + every point's index) depend on the bins and the launch
  # geometry only: computed once per (W, geometry) and kept with the device copy of W
  akey = (id(ctx), nA, nBk, nBr, plan.ndepth, plan.nx, w_flags & _hip.BINNED_W_ON_X)
  atoms = w_buf.atoms.get(akey) if PREPARED_ATOMS else None
  if atoms is None and PREPARED_ATOMS:
    nbytes = C.c_int64(0)
    _hip.check(ctx.lib.wbx_binned_atoms_size(C.byref(dplan.struct), nA, nBk, nBr, w_flags & _hip.BINNED_W_ON_X,
                                             C.byref(nbytes)), 'wbx_binned_atoms_size')
    atoms = ctx.alloc(int(nbytes.value))
    _hip.check(ctx.lib.wbx_binned_atoms(ctx.handle, C.byref(dplan.struct), nA, nBk, nBr, w_flags & _hip.BINNED_W_ON_X,
                                        C.c_void_p(w_buf.bufs[1].ptr), C.c_void_p(atoms.ptr)), 'wbx_binned_atoms')
    if len(w_buf.atoms) > 8:
      w_buf.atoms.clear()
    w_buf.atoms[akey] = atoms
  def call():
    _hip.check(ctx.lib.wbx_det_binned(ctx.handle, C.byref(dplan.struct), func, dtype_code, ptr(devs[0]), ptr(devs[1]),
                                      ptr(devs[2]), ptr(devs[3]), C.c_void_p(wt_buf.ptr),
                                      C.c_void_p(w_buf.bufs[1].ptr), nA, nBk, nBr, w_flags | (_hip.BINNED_ACCUMULATE if add else 0), nbin,
                                      C.c_void_p(atoms.ptr) if atoms is not None else None,
                                      C.c_void_p(out_ptr)), 'wbx_det_binned')
  timed_launch(ctx, call, kind='det_binned', nbin=nbin, w_flags=w_flags)
  return handle, shape


# The ensemble family with weights, bins and mask in ONE pass (wbx_ens_binned, csrc/wbx_ens_atoms.hpp) whenever it applies:
# rank form, float32, 2..64 members, boolean bin masks, separable weights; (r5) any validity mask -- also the per-point one
# add_nan_mask_to_data builds, with strides along time / level -- and Aggregator(skipna=True).  False: the two-stage route
# (x-kept ensemble kernel + wbx_contract_bits), for A/B timing and tests.
ENS_BINNED = os.environ.get('WBX_ENS_BINNED', '1') != '0'
ENS_BINNED_LANES = 6  # the five ensemble lanes + the count lane, always
# With a mask, ONE launch yields the masked sums (lanes 0-5) and the sums over all points (lanes 6-11): the reference masks the
# skill / unbiased-MSE / mean-MSE statistics of such a variable but not its spread / variance (statistics of the predictions
# alone), which would otherwise be a second pass over the members.  The unmasked half is left with the predictions array
# ('_wbx_twin') for the member-only group of the same predictions to pick up.  False: one launch per mask setting (A/B, tests).
ENS_TWIN_MASK = os.environ.get('WBX_ENS_TWIN_MASK', '1') != '0'


def _mask_on_w_only(plan: planner.S1Plan, mask_dev) -> bool:
  return all(mask_dev.layout.stride(d) == 0 for d in tuple(plan.a_dims) + tuple(plan.depth_dims))


def _ens_binned_flags(plan: planner.S1Plan, w_buf, devs, twin_ok=True):
  """WBX_BINNED_* flags of a wbx_ens_binned call, or None when the weights / the mask rule it out."""
  w_flags = _hip.BINNED_W_ON_X if (plan.x_kept and plan.nj > 1) else 0
  if w_buf.factored is None or not SEPARABLE_BINNED_WEIGHTS:
    return None
  w_flags |= w_buf.factored[0]
  if (w_flags & _hip.BINNED_WT_X_ONLY) and not (w_flags & _hip.BINNED_W_ON_X):
    return None
  if devs[3] is not None:
    if not (w_flags & _hip.BINNED_W_ON_X) or plan.xstride[3] < 0:
      return None
    if _mask_on_w_only(plan, devs[3]):  # folded into the [bk][br][x] atom ids; else one id byte per point of the chunk
      w_flags |= _hip.BINNED_MASK_ON_W
    if ENS_TWIN_MASK and twin_ok:
      w_flags |= _hip.BINNED_TWIN_MASK
  return w_flags


def _ens_binned_atoms(ctx, dplan, plan: planner.S1Plan, w_buf, w_flags):
  """The atom tables of (bins, launch geometry) for wbx_ens_binned, computed once and kept with the device copy of W; None
  when some patch holds more distinct membership words than the kernel's table takes (arbitrary user masks)."""
  nA, nBk, nBr = plan.n(plan.a_dims), plan.n(plan.bk_dims), plan.n(plan.br_dims)
  akey = ('ens', ctx.device_id, nA, nBk, nBr, plan.ndepth, plan.nx, w_flags & _hip.BINNED_W_ON_X)
  hit = w_buf.atoms.get(akey)
  if hit is None:
    nbytes = C.c_int64(0)
    _hip.check(ctx.lib.wbx_ens_binned_atoms_size(C.byref(dplan.struct), nA, nBk, nBr, w_flags & _hip.BINNED_W_ON_X,
                                                 C.byref(nbytes)), 'wbx_ens_binned_atoms_size')
    atoms = ctx.alloc(int(nbytes.value))
    overflow = C.c_int64(0)
    _hip.check(ctx.lib.wbx_ens_binned_atoms(ctx.handle, C.byref(dplan.struct), nA, nBk, nBr, w_flags & _hip.BINNED_W_ON_X,
                                            C.c_void_p(w_buf.bufs[1].ptr), C.c_void_p(atoms.ptr), C.byref(overflow)),
               'wbx_ens_binned_atoms')
    if len(w_buf.atoms) > 8:
      w_buf.atoms.clear()
    hit = w_buf.atoms[akey] = (atoms, int(overflow.value))
  return hit[0] if hit[1] == 0 else None


def _ens_binned_route(ctx, kind, plan: planner.S1Plan, dplan, w_buf, devs, dtype_code, flags, ens, twin_ok=True):
  """(w_flags, atoms) when this reduction runs on wbx_ens_binned, else None."""
  if kind != 'ens' or not ENS_BINNED or w_buf.kind != 'bits' or dtype_code != _hip.F32:
    return None
  if ens['algo'] != _hip.ENS_SORT or not 2 <= ens['M'] <= 64 or (flags & _hip.FLAG_SKIPNA_ENS):
    return None
  if plan.x_kept and not plan.sum_j:  # x survives into the output: the two-stage path keeps it
    return None
  if plan.nkey * plan.ndepth * plan.nx == 0 or plan.xstride[0] < 0 or plan.xstride[1] < 0:
    return None
  if plan.nx * plan.xstride[0] * 4 >= 1 << 32 or plan.nx * plan.xstride[1] * 4 >= 1 << 32:  # 32-bit x byte offsets, both inputs
    return None
  w_flags = _ens_binned_flags(plan, w_buf, devs, twin_ok)
  if w_flags is None:
    return None
  atoms = _ens_binned_atoms(ctx, dplan, plan, w_buf, w_flags)
  if atoms is None:
    return None
  return w_flags, atoms


def _twin_key(dims, sizes, reduce_dims, w_da, bin_dims, ens, skipna=False):
  return (tuple(dims), tuple(sizes[d] for d in dims), tuple(sorted(reduce_dims, key=str)), id(w_da), tuple(bin_dims),
          ens['member_dim'], ens['M'], ens['algo'], bool(ens.get('fair', True)), bool(skipna))


def _twin_store(p_da, ctx, dims, sizes, reduce_dims, w_da, bin_dims, ens, result, skipna=False):
  p_da.__dict__.setdefault('_wbx_twin', {})[_twin_key(dims, sizes, reduce_dims, w_da, bin_dims, ens, skipna)] = (result, ctx, w_da)


def _twin_lookup(p_da, dims, sizes, reduce_dims, w_da, bin_dims, ens, skipna=False):
  """The unmasked half of a twin launch over these very predictions (same frame, reduction, weights / bins object and ensemble
  parameters), or None.  Only the statistics of the predictions ALONE are served from it (spread, variance: their group's
  companion target is a member of the predictions, lazy.ens_statistic(member_only=True)) -- the caller's lane picks them."""
  hit = p_da.__dict__.get('_wbx_twin', {}).get(_twin_key(dims, sizes, reduce_dims, w_da, bin_dims, ens, skipna))
  if hit is None:
    return None
  result, ctx, _ = hit
  if _deferred is not None:  # the reader's fence has to cover the launch that produced it
    _deferred.ctxs[id(ctx)] = ctx
  return result


def _run_ens_binned(ctx, dplan: _PlanOnDevice, plan: planner.S1Plan, devs, dtype_code: int, ens_args, w_buf, route):
  """wbx_ens_binned -> (device pointer, shape) of out[nA][nBk][6][1][nbin]: lanes 0-4 the ensemble family, lane 5 the sum of
  the weights of the valid points (the count lane)."""
  w_flags, atoms = route
  nA, nBk, nBr = plan.n(plan.a_dims), plan.n(plan.bk_dims), plan.n(plan.br_dims)
  nbin = w_buf.shape[-1]
  # six lanes (five values + the shared count), under skipna ten (five values + their five counts); twice that in twin mode
  per_set = 2 * _hip.ENS_LANES if (plan.flags & _hip.FLAG_SKIPNA) else ENS_BINNED_LANES
  shape = (nA, nBk, per_set * (2 if (w_flags & _hip.BINNED_TWIN_MASK) else 1), 1, nbin)
  out_ptr, handle, add = _result_target(ctx, shape, 's2out', can_accumulate=True)  # (level 3 writes every element once)
  ptr = lambda d: C.c_void_p(d.ptr) if d is not None else None
  m, mstride, algo = ens_args

  def call():
    _hip.check(ctx.lib.wbx_ens_binned(ctx.handle, C.byref(dplan.struct), dtype_code, int(m), int(mstride), int(algo), ptr(devs[0]),
                                      ptr(devs[1]), ptr(devs[3]), C.c_void_p(w_buf.factored[1].ptr), C.c_void_p(w_buf.bufs[1].ptr),
                                      nA, nBk, nBr, w_flags | (_hip.BINNED_ACCUMULATE if add else 0), nbin, C.c_void_p(atoms.ptr),
                                      C.c_void_p(out_ptr)), 'wbx_ens_binned')
  timed_launch(ctx, call, kind='ens_binned', nbin=nbin, w_flags=w_flags, flags=int(plan.flags))
  return handle, shape


def dense_w(plan: planner.S1Plan, w_da: xr.DataArray | None, bin_dims: Sequence) -> tuple[np.ndarray, tuple]:
  """W as float64 [nBk][nBr][nj][nbin] from the labeled product of weights and bin masks."""
  bin_dims = tuple(bin_dims)
  wd = plan.bk_dims + plan.br_dims + ((plan.x_dim,) if plan.x_kept and plan.x_dim is not None else ())
  if w_da is None:
    w_da = xr.DataArray(np.float64(1.0))
  extra = [d for d in w_da.dims if d not in wd and d not in bin_dims]
  if extra:
    raise ValueError(f'weights/bin masks depend on dims {extra} that stage 1 does not keep')
  shape_full = [plan.sizes[d] for d in wd] + [w_da.sizes[d] for d in bin_dims]
  present = [d for d in list(wd) + list(bin_dims) if d in w_da.dims]
  w = w_da.transpose(*present).values.astype(np.float64)
  idx = tuple(slice(None) if d in w_da.dims else None for d in list(wd) + list(bin_dims))
  w = np.broadcast_to(w[idx], shape_full)
  nbin = int(np.prod([w_da.sizes[d] for d in bin_dims], dtype=np.int64)) if bin_dims else 1
  w = np.ascontiguousarray(w.reshape(plan.n(plan.bk_dims), plan.n(plan.br_dims), plan.nj, nbin))
  return w, tuple(w_da.sizes[d] for d in bin_dims)


def reduce_statistics(kind: str, inputs: Sequence[xr.DataArray | None], dims: Sequence, sizes: dict, reduce_dims,
                      w_da: xr.DataArray | None, bin_dims: Sequence, *, func: int = 0, mask: xr.DataArray | None = None,
                      skipna: bool = False, gather: planner.GatherSpec | None = None, ens=None, cat=None,
                      ctx: _hip.Context | None = None):
  """Fused statistics + weighted/binned reduction.

  kind 'det' / 'ens' / 'cat' (indicator statistics: `cat` = {'func', 'ncat', 'thresholds' (float64 ndarray or None),
  'member_dim' (or None), 'M'}; one value lane per category).

  Returns (values, counts, out_dims): `values` is ONE array (lanes,) + out_dims, out_dims =
  (A dims..., Bk dims..., [x dim], bin dims...) -- a view of the kernel's output, so `values[lane]` is a view too;
  `counts` (same shape) is the matching sum of W over valid elements: per lane under skipna, one lane broadcast to all
  under a mask alone, else a data-independent constant broadcast to all.
  """
  global _deferred
  if (kind == 'ens' and mask is None and inputs[0] is not None
      and inputs[1] is inputs[0].__dict__.get('_wbx_member0', {}).get(ens['member_dim'])):
    # the member-only group alone (lazy.ens_statistic(member_only=True): its companion operand is lazy.first_member(p)) --
    # lanes 0, 3, 4 of a twin depend on the targets it was launched with, so a group with targets of its own never reads it
    twin = _twin_lookup(inputs[0], dims, sizes, reduce_dims, w_da, bin_dims, ens, skipna)
    if twin is not None:
      return twin
  ctx = ctx or _launch_context(kind)
  wdep = set(w_da.dims) - set(bin_dims) if w_da is not None else set()
  member_dim = ens['member_dim'] if ens else (cat.get('member_dim') if cat else None)
  datas = [i.data if i is not None else None for i in inputs]
  dtype_code = _common_dtype(datas)
  _sync_torch_producers(datas)
  devs = [(_to_device(ctx, i, dtype_code) if i is not None else None) for i in inputs]
  while len(devs) < 4:
    devs.append(None)
  thr_field = cat.get('thr_field') if cat else None
  if thr_field is not None:  # float64 whatever the inputs are: it is the kernel's own operand, not a statistic input
    _sync_torch_producers([thr_field.data])
    devs[2] = _to_device(ctx, thr_field, _hip.F64)
  flags = 0
  if mask is not None:
    flags |= _hip.FLAG_MASKED
    devs[3] = _mask_to_device(ctx, mask)
  if skipna:
    flags |= _hip.FLAG_SKIPNA
  if ens and ens.get('fair', True):
    flags |= _hip.FLAG_FAIR
  if ens and ens.get('skipna', False):
    flags |= _hip.FLAG_SKIPNA_ENS
  layouts = [d.layout if d is not None else None for d in devs]
  _order_uploads(ctx, devs)
  if _deferred is not None:
    _deferred.keepalive.append((datas, devs))
  # Weights that depend on the innermost (contiguous) dim ONLY and no bins -- GridAreaWeighting on latitude-fastest
  # data -- are folded into stage 1 (plan.x_weights): x is then summed there on the x-summed kernel instead of being
  # kept as nx partials per key for stage 2.
  x_weights = None
  hit = None
  if kind == 'det' and _fusion_requests and not flags and not bin_dims and inputs[0] is not None:
    # spectra of these very fields are about to be asked for and longitude is strided: rows = the spectra's rows
    hit = _fusion_latfast_plan(ctx, inputs, dims, sizes, layouts, reduce_dims, wdep, gather, func, dtype_code)
  fold_ok = (kind == 'det' and not (flags & ~_hip.FLAG_MASKED)) or (kind == 'ens' and not (flags & _hip.FLAG_SKIPNA_ENS))
  if hit is None and FOLD_X_WEIGHTS and fold_ok and w_da is not None and not bin_dims and len(w_da.dims) == 1:
    x_dim = planner.choose_x_dim(dims, sizes, layouts[0])
    if x_dim is not None and w_da.dims[0] == x_dim and x_dim in set(reduce_dims) and 1 < sizes[x_dim] <= 2045:
      xw = np.ascontiguousarray(w_da.values, dtype=np.float64)
      # the rank-form fp32 ensemble ops without mask / skipna run the pipelined one-wave sweep (ens_pipe_kernel<.., FLAT>)
      one_wave = (ENS_PIPE and kind == 'ens' and dtype_code == _hip.F32 and ens['algo'] == 0 and ens['M'] <= 64
                  and not (flags & (_hip.FLAG_MASKED | _hip.FLAG_SKIPNA | _hip.FLAG_SKIPNA_ENS)))
      hit = _planned(ctx, kind, dims, sizes, layouts, reduce_dims, set(), gather, flags, xw, one_wave=one_wave)
      if hit[0].plane_rows > 0 and dtype_code == _hip.F32:  # the flat float4 sweep applies (contiguous aligned planes)
        x_weights, wdep, w_da = xw, set(), None
      else:
        hit = None
  plan, dplan = hit if hit is not None else _planned(ctx, kind, dims, sizes, layouts, reduce_dims, wdep, gather, flags)
  nl = _hip.DET_LANES[func] if kind == 'det' else (int(cat['ncat']) if kind == 'cat' else
                                                   (_hip.ENS2_LANES if kind == 'ens2' else _hip.ENS_LANES))
  counted = bool(flags & 3)
  shared_count = counted and not (flags & _hip.FLAG_SKIPNA)  # mask only: one count lane for every statistic
  nl_total = nl + 1 if shared_count else nl * (2 if counted else 1)
  ens_args = cat_args = None
  if kind == 'ens':
    ens_args = (ens['M'], devs[0].layout.stride(member_dim), ens['algo'])
  if kind == 'ens2':
    ens_args = (ens['M'], devs[0].layout.stride(member_dim), ens['N'], devs[1].layout.stride(member_dim))
  if kind == 'cat':
    thr = None
    if cat.get('thresholds') is not None and thr_field is None:
      tkey = (ctx.device_id, np.asarray(cat['thresholds'], np.float64).tobytes())
      thr = _thr_cache.get(tkey)
      if thr is None:
        if len(_thr_cache) > 64:
          _thr_cache.clear()
        thr = _thr_cache[tkey] = ctx.upload(np.asarray(cat['thresholds'], np.float64))
    cat_args = (cat['func'], nl, cat.get('M', 1) if member_dim else 1,
                devs[0].layout.stride(member_dim) if member_dim else 0, thr,
                None if thr_field is None else devs[2].layout.stride(cat['cat_dim']))
  w_buf = _device_w(ctx, plan, w_da, bin_dims)
  # (a chunk that is being recorded: the record keeps what the launches below point at -- plan tables, weights / bins / atom
  #  tables, inputs that do not follow the chunk such as the climatology, threshold tables)
  # (input 2 only where it is the SAME array chunk after chunk: the climatology behind a gather table, a threshold field -- an
  #  aligned climatology materialised per chunk is left unaccounted for, and such a chunk is not recorded)
  replay.keep(dplan, w_buf, cat_args[4] if cat_args else None,
              devs[2] if (devs[2] is not None and (gather is not None or thr_field is not None)) else None)
  bin_shape = w_buf.bin_shape
  s2 = planner.build_s2_plan(plan, nl_total, w_buf.shape[-1])
  # Under skipna a twin launch does not form the masked spread / variance (wbx.h): fine while those are statistics of a
  # member-only group that reads the twin -- i.e. unless the predictions carry a mask coordinate themselves, in which case
  # lazy.ens_statistic keeps them in THIS group (lanes 1, 2) and no twin output is asked for.
  twin_ok = not (skipna and inputs[0] is not None and 'mask' in inputs[0].coords)
  ens_route = _ens_binned_route(ctx, kind, plan, dplan, w_buf, devs, dtype_code, flags, ens, twin_ok) if kind == 'ens' else None
  if ens_route is not None:
    res = _run_ens_binned(ctx, dplan, plan, devs, dtype_code, ens_args, w_buf, ens_route)
  elif _binned_eligible(kind, plan, w_buf, devs, nl_total, _hip.DET_INPUTS[func] if kind == 'det' else 2):
    res = _run_binned(ctx, dplan, plan, devs, dtype_code, nl_total, func, w_buf)
  else:
    partial = _run_s1(ctx, kind, dplan, plan, devs, dtype_code, nl_total, func=func, ens=ens_args, cat=cat_args, inputs=inputs,
                      fold=(s2, w_da, bin_dims, w_buf))
    # [nA][nBk][lanes][nj_out][nbin]
    res = partial.res if isinstance(partial, _FoldedS2) else _run_s2(ctx, s2, partial.ptr, w_buf)
  out = _deliver(ctx, *res)

  x_out = (plan.x_dim,) if (plan.x_kept and not plan.sum_j and plan.x_dim is not None) else ()
  out_dims = plan.a_dims + plan.bk_dims + x_out + tuple(bin_dims)
  lead_shape = [plan.sizes[d] for d in plan.a_dims] + [plan.sizes[d] for d in plan.bk_dims]
  tail_shape = [plan.sizes[d] for d in x_out] + list(bin_shape)

  def lanes_of(first, count):
    # (count,) + out_dims as a VIEW of out[nA][nBk][lane][nj_out][nbin]: lanes first, then the A / Bk / x / bin axes split
    # into their dims (splitting axes never copies), so the result can still be filled in later (deferred read-back) or
    # be located in an accumulator slot
    v = np.moveaxis(out[:, :, first:first + count], 2, 0).reshape([count] + lead_shape + tail_shape)
    if v.size and not np.may_share_memory(v, out):
      raise RuntimeError('internal: lane view of a reduction result was copied')
    return v

  if ens_route is not None and (ens_route[0] & _hip.BINNED_TWIN_MASK):
    # the second set: the same statistics over ALL points, for the statistics of these predictions that carry no mask
    if flags & _hip.FLAG_SKIPNA:  # [5 values | 5 counts] twice
      _twin_store(inputs[0], ctx, dims, sizes, reduce_dims, w_da, bin_dims, ens,
                  (lanes_of(2 * nl, nl), lanes_of(3 * nl, nl), out_dims), skipna=True)
    else:
      _twin_store(inputs[0], ctx, dims, sizes, reduce_dims, w_da, bin_dims, ens, (lanes_of(ENS_BINNED_LANES, nl),
                  np.broadcast_to(lanes_of(2 * ENS_BINNED_LANES - 1, 1), lanes_of(0, nl).shape), out_dims))
  values = lanes_of(0, nl)
  if counted:
    counts = np.broadcast_to(lanes_of(nl, 1), values.shape) if shared_count else lanes_of(nl, nl)
  else:
    # data-independent: (elements folded per partial) * sum of W, computed by the same stage-2 kernel
    ckey = (id(w_buf), s2.nBk, s2.nBr, s2.nj, s2.nbin, s2.sum_j, plan.reduced_count_per_partial())
    cnt = _count_cache.get(ckey)
    if cnt is None:
      ones = np.full((1, s2.nBk, s2.nBr, 1, 1, s2.nj), float(plan.reduced_count_per_partial()), dtype=np.float64)
      s2c = planner.S2Plan(nA=1, nBk=s2.nBk, nBr=s2.nBr, nchunk=1, nlane=1, nj=s2.nj, nbin=s2.nbin, sum_j=s2.sum_j)
      ones_buf = ctx.upload(ones)
      cptr, cshape = _run_s2(ctx, s2c, ones_buf.ptr, w_buf)  # [1][nBk][1][nj_out][nbin]
      cnt = ctx.download(cptr, cshape, np.float64)  # computed once per geometry and cached: read back synchronously
      cnt = np.array(cnt, dtype=np.float64)  # (own memory: the read-back pool recycles its blocks)
      if len(_count_cache) > 64:
        _count_cache.clear()
        _const_roots.clear()
      scaled = {}
      _count_cache[ckey] = (cnt, w_buf, scaled)  # keep w_buf alive so its id stays unique
      register_constant(cnt)
    else:
      cnt, scaled = cnt[0], cnt[2]
    if x_weights is not None:  # sum of the folded weights over the reduced elements: (others reduced) * sum_x w[x]
      factor = float(x_weights.sum()) / plan.nx
      if factor not in scaled:
        scaled[factor] = register_constant(cnt * factor)
      cnt = scaled[factor]
    # every view handed out below is a view of ONE registered read-only array: Accumulation.capture recognises it by its
    # root and only counts how often it met it -- no outer-join / full-size add per statistic, aggregator and chunk
    cnt = np.broadcast_to(cnt[:, :, 0], (s2.nA,) + cnt[:, :, 0].shape[1:]).reshape(lead_shape + tail_shape)
    counts = np.broadcast_to(cnt, values.shape)
  return values, counts, out_dims


def materialise(kind: str, inputs: Sequence[xr.DataArray | None], dims: Sequence, sizes: dict, lane: int, *,
                func: int = 0, gather: planner.GatherSpec | None = None, ens=None,
                ctx: _hip.Context | None = None) -> np.ndarray:
  """Full-resolution statistic (one lane) as a float64 ndarray over `dims`."""
  ctx = ctx or _hip.default_context()
  datas = [i.data if i is not None else None for i in inputs]
  dtype_code = _common_dtype(datas)
  _sync_torch_producers(datas)
  devs = [(_to_device(ctx, i, dtype_code) if i is not None else None) for i in inputs]
  while len(devs) < 4:
    devs.append(None)
  layouts = [d.layout if d is not None else None for d in devs]
  _order_uploads(ctx, devs)
  flags = _hip.FLAG_FAIR if (ens and ens.get('fair', True)) else 0
  if ens and ens.get('skipna', False):
    flags |= _hip.FLAG_SKIPNA_ENS
  plan = planner.build_s1_plan(dims, sizes, layouts, (), gather=gather, flags=flags, allow_vec4=False, map_mode=True)
  dplan = _device_plan(ctx, plan)
  ens_args = None
  if kind == 'ens':
    ens_args = (ens['M'], devs[0].layout.stride(ens['member_dim']), ens['algo'])
  flat = _run_map(ctx, kind, dplan, plan, devs, dtype_code, int(lane), func=func, ens=ens_args)
  order = plan.key_dims + plan.depth_dims + ((plan.x_dim,) if plan.x_dim is not None else ())
  arr = flat.reshape([plan.sizes[d] for d in order])
  return np.transpose(arr, [order.index(d) for d in dims]) if tuple(order) != tuple(dims) else arr
