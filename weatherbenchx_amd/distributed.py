"""Accumulator reduction across GPUs: the counterpart of the Beam combine stage.

The reference sums per-chunk accumulators with `beam.CombinePerKey(CombiningSum())` and concatenates the pieces that
belong to different chunk offsets afterwards (weatherbenchX/beam_pipeline.py:121-137, 253-319, 509-510,
weatherbenchX/beam_utils.py:30-50).  Here chunks of (init_time x lead_time) are sharded over one process per GPU; each
rank ADDS its chunks' results into device-resident accumulator slots (engine.Accumulation: a slot per
(aggregator, statistic, variable, surviving chunk offsets)).  At the end of the job

  1. the ranks exchange their slot tables once (a small pickled all-gather: names, sizes, result frames),
  2. every rank lays the UNION of all slots out in one float64 device buffer -- its own slots copied in, the others zero --
  3. ONE sum all-reduce of that buffer (RCCL over xGMI on the device pointer with backend 'nccl'; 'gloo' on CPU in tests)
     both adds the accumulators ranks share (a reduced time dim) and fills in the ones they own alone (a surviving time
     dim: disjoint offsets -> the reference's concat),
  4. the buffer is read back once and every result array is rebuilt as a view of it.

The buffer is KBs..MBs, so the collective is latency bound and is issued once per job / step, never per
chunk.  A `ReductionPlan` returned by the first call lets later calls with an unchanged layout skip step 1 (a bench
loop that all-reduces every step).
"""
from __future__ import annotations

import numpy as np

from weatherbenchx_amd import engine
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.aggregation import AggregationState, combining_sum


def _leaves(tree, prefix=()):
  if isinstance(tree, xr.DataArray):
    yield prefix, tree
  elif isinstance(tree, dict):
    for k in sorted(tree, key=str):
      yield from _leaves(tree[k], prefix + (k,))
  elif tree is not None:
    raise TypeError(f'unsupported leaf type {type(tree)}')


def shard_chunks(chunks: list, rank: int, world_size: int, mode: str = 'round_robin') -> list:
  """Assignment of time chunks to ranks: chunk i -> rank i mod n, or -- `mode='block'` -- contiguous blocks of
  the chunk list (consecutive init times stay on one rank: with a climatology behind a slab pool, climatology_cache.py,
  consecutive chunks share most of the (dayofyear, hour) slabs they name, chunks n days apart share none)."""
  if mode == 'block':
    n = len(chunks)
    lo, hi = rank * n // world_size, (rank + 1) * n // world_size
    return chunks[lo:hi]
  if mode != 'round_robin':
    raise ValueError(f"sharding mode {mode!r}: 'round_robin' or 'block'")
  return [c for i, c in enumerate(chunks) if i % world_size == rank]


def _group_info(group):
  """(world size, backend) of an initialised process group, (1, None) otherwise."""
  try:
    import torch.distributed as dist  # pylint: disable=g-import-not-at-top
  except ImportError:
    return 1, None
  if not dist.is_available() or not dist.is_initialized():
    return 1, None
  return dist.get_world_size(group), dist.get_backend(group)


class _DeviceView:
  """Exposes wbx device memory to torch through the CUDA array interface (zero copy): RCCL then reduces the accumulator
  buffer in place, no host hop."""

  def __init__(self, ptr: int, n: int):
    self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': '<f8', 'data': (int(ptr), False), 'version': 2,
                                     'strides': None}


def device_tensor(ptr: int, n: int, device_id: int):
  import torch  # pylint: disable=g-import-not-at-top
  return torch.as_tensor(_DeviceView(ptr, n), device=torch.device('cuda', device_id))


class ReductionPlan:
  """The global layout agreed on by the ranks: slot order / offsets and every leaf's views."""

  def __init__(self, local_sig, order, offsets, total, leaves, frames):
    self.local_sig, self.order, self.offsets, self.total = local_sig, order, offsets, total
    self.leaves, self.frames = leaves, frames
    self.collectives = 0  # sum all-reduces issued through this plan (tests / traces)


def _key_sort(key):
  return repr(key)


def _local_meta(acc: engine.Accumulation):
  """(slot sizes, host arrays, leaf specs, frames) of this rank -- read only: host leaves (results that never were on the
  device) get their specs and frames in LOCAL dicts, so resolving an accumulation twice, or after a host leaf changed its
  shape, can never leave a stale spec behind."""
  slots = dict(acc.slot_table())
  host = {}
  leaves = {path: list(specs) for path, specs in acc.specs.items()}
  frames = {(path, spec): acc.frames[(path, i)] for path, specs in leaves.items() for i, spec in enumerate(specs)}
  for path, da in acc.host.items():
    arr = np.array(da.values, dtype=np.float64, order='C')  # (ascontiguousarray would turn 0-d into 1-d)
    key = ('host', path)
    slots[key] = int(arr.size)
    host[key] = arr
    strides = tuple(int(s // 8) for s in arr.strides)
    spec = (key, 0, tuple(int(x) for x in arr.shape), strides, tuple(da.dims), 1.0)
    lst = leaves.setdefault(path, [])
    lst[:] = [sp for sp in lst if sp[0] != key]  # one spec per host leaf: the current one
    lst.append(spec)
    frames[(path, spec)] = (dict(da._coords), da.name, dict(da.attrs))  # pylint: disable=protected-access
  return slots, host, leaves, frames


def _frames_digest(frames) -> int:
  """A cheap hash of every result frame's coordinates (labels of surviving dims, names): a cached ReductionPlan is only
  reused while the arrays it would hand out carry the labels of THIS step (another latitude slice, other time labels ->
  re-plan)."""
  import zlib  # pylint: disable=g-import-not-at-top
  h = 0
  for key in sorted(frames, key=repr):
    coords, name, _ = frames[key]
    h = zlib.crc32(repr((key[0], name)).encode(), h)
    for cname in sorted(coords, key=str):
      dims, values = coords[cname]
      v = np.ascontiguousarray(np.asarray(values))
      h = zlib.crc32(repr((cname, tuple(dims), v.shape, str(v.dtype))).encode(), h)
      h = zlib.crc32(v.tobytes() if v.dtype != object else repr(v.tolist()).encode(), h)
  return h


class CabiCommunicator:
  """An RCCL communicator owned by libwbx_hip.so (wbx_comm_create): the payload collective of reduce_accumulation then is
  wbx_acc_allreduce on the library's own stream -- what a C / cgo / JNI host of the library would call; torch is not
  involved in moving the sums.  `CabiCommunicator.from_torch_group()` bootstraps it inside a torch.distributed job (the 128
  byte id travels through the group's object broadcast); a host without torch passes the id around itself."""

  def __init__(self, ctx, unique_id: bytes, nranks: int, rank: int):
    import ctypes as C  # pylint: disable=g-import-not-at-top
    from weatherbenchx_amd import _hip  # pylint: disable=g-import-not-at-top
    if len(unique_id) != _hip.COMM_ID_BYTES:
      raise ValueError(f'unique_id must be {_hip.COMM_ID_BYTES} bytes')
    self.ctx, self.nranks, self.rank = ctx, int(nranks), int(rank)
    handle = C.c_void_p()
    buf = C.create_string_buffer(bytes(unique_id), _hip.COMM_ID_BYTES)
    _hip.check(ctx.lib.wbx_comm_create(ctx.handle, buf, self.nranks, self.rank, C.byref(handle)), 'wbx_comm_create')
    self.handle = handle
    # what the payload collectives cost on the library's stream (event pair around wbx_acc_allreduce) and moved
    self.timings = {'collectives': 0, 'us_total': 0.0, 'us_last': None, 'bytes_last': 0}

  @staticmethod
  def new_unique_id() -> bytes:
    import ctypes as C  # pylint: disable=g-import-not-at-top
    from weatherbenchx_amd import _hip  # pylint: disable=g-import-not-at-top
    buf = C.create_string_buffer(_hip.COMM_ID_BYTES)
    _hip.check(_hip.load_library().wbx_comm_unique_id(buf), 'wbx_comm_unique_id')
    return buf.raw

  @classmethod
  def from_torch_group(cls, group=None, ctx=None):
    import torch.distributed as dist  # pylint: disable=g-import-not-at-top
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [cls.new_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return cls(ctx if ctx is not None else _reduction_context(), box[0], world, rank)

  @property
  def collectives(self) -> int:
    import ctypes as C  # pylint: disable=g-import-not-at-top
    n = C.c_int64()
    self.ctx.lib.wbx_comm_info(self.handle, None, None, C.byref(n))
    return int(n.value)

  def close(self):
    if self.handle is not None:
      self.ctx.lib.wbx_comm_destroy(self.handle)
      self.handle = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass


_reduce_ctx: dict = {}


def _reduction_context():
  from weatherbenchx_amd import _hip  # pylint: disable=g-import-not-at-top
  base = _hip.default_context()
  hit = _reduce_ctx.get(base.device_id)
  if hit is None or hit[0] is not base:
    hit = _reduce_ctx[base.device_id] = (base, engine.new_context())
  return hit[1]


def _build_plan(local_sig, slots, leaves, frames, group, exchange: bool, world: int) -> ReductionPlan:
  metas = [(slots, leaves, frames)]
  if exchange:
    import torch.distributed as dist  # pylint: disable=g-import-not-at-top
    gathered = [None] * world
    dist.all_gather_object(gathered, (slots, leaves, frames), group=group)
    metas = gathered
  union: dict = {}
  for sl, _, _ in metas:
    for key, n in sl.items():
      if union.setdefault(key, n) != n:
        raise ValueError(f'accumulator {key} has {n} values on one rank and {union[key]} on another: the ranks '
                         'do not run the same statistics / aggregators')
  order = sorted(union, key=_key_sort)
  offsets, total = {}, 0
  for key in order:
    offsets[key] = total
    total += union[key]
  all_leaves: dict = {}
  all_frames: dict = {}
  for _, lv, fr in metas:
    for path, specs in lv.items():
      have = all_leaves.setdefault(path, [])
      for spec in specs:
        if spec not in have:  # the same view of the same slot on several ranks is ONE array after the reduction
          have.append(spec)
          all_frames[(path, spec)] = fr[(path, spec)]
  return ReductionPlan(local_sig, order, offsets, total, all_leaves, all_frames)


def reduce_accumulation(acc: engine.Accumulation, group=None, *, all_reduce: bool = True, plan: ReductionPlan | None = None,
                        force: bool = False, fence=None, comm: CabiCommunicator | None = None):
  """-> ({path: DataArray}, ReductionPlan): every captured leaf of `acc`, summed over the ranks of `group`.

  `all_reduce=False` resolves the local accumulators only.  `force` runs the collective even in a one-rank group (the
  RCCL plumbing check on a single GPU).  Pass the returned plan back in while the local layout does not change to skip
  the layout exchange.  The payload buffer carries ONE extra element, "my layout or my result labels changed under the
  cached plan": it is summed with the payload, so a rank that has to re-plan makes EVERY rank re-plan after that collective
  (its own payload is zeros for that round) -- nobody is left waiting in a collective the others never enter, and the steady
  state stays at exactly one collective per call.  `fence`: wait for this fence (the state's own) instead of
  draining every launch stream -- a pipelined loop has the next step's kernels in flight.  `comm`: the payload collective goes
  through the library's own RCCL communicator (wbx_acc_allreduce) instead of torch.distributed; the layout exchange (once
  per job) still needs a torch group when there is more than one rank."""
  world, backend = _group_info(group)
  if comm is not None:
    if backend is not None and comm.nranks != world:
      raise ValueError(f'the C-ABI communicator has {comm.nranks} ranks, the torch group {world}')
    world = comm.nranks if backend is None else world
  collective = all_reduce and (world > 1 or (force and (backend is not None or comm is not None)))
  slots, host, leaves, frames = _local_meta(acc)
  local_sig = (tuple(sorted(slots.items(), key=lambda kv: _key_sort(kv[0]))),
               tuple(sorted(((p, tuple(s)) for p, s in leaves.items()), key=lambda kv: _key_sort(kv[0]))),
               _frames_digest(frames))
  exchange = collective and world > 1
  if exchange and backend is None:
    raise ValueError('more than one rank needs an initialised torch.distributed group for the layout exchange')
  stale = plan is not None and plan.local_sig != local_sig
  if plan is None or (stale and not exchange):
    plan, stale = _build_plan(local_sig, slots, leaves, frames, group, exchange, world), False

  # ---- the values: this rank's slots at their global offsets, zero elsewhere, then ONE sum over the ranks --------
  device_path = collective and (backend == 'nccl' or comm is not None)
  if fence is not None:
    fence.wait()     # the kernels (and accumulator adds) of exactly this state; later launches keep running
  else:
    acc.synchronize()
  ntot = plan.total + 1  # [payload ..., number of ranks whose cached plan went stale]
  if device_path:
    from weatherbenchx_amd import _hip  # pylint: disable=g-import-not-at-top
    import ctypes as C  # pylint: disable=g-import-not-at-top
    ctx = comm.ctx if comm is not None else _reduction_context()  # own stream: the copies do not queue behind the next step's kernels
    gbuf = ctx.alloc(ntot * 8)
    _hip.check(ctx.lib.wbx_memset(ctx.handle, C.c_void_p(gbuf.ptr), 0, ntot * 8), 'wbx_memset')
    if stale:
      one = np.ones(1, dtype=np.float64)
      _hip.check(ctx.lib.wbx_memcpy_h2d(ctx.handle, C.c_void_p(gbuf.ptr + 8 * plan.total), one.ctypes.data_as(C.c_void_p), 8),
                 'wbx_memcpy_h2d')
    else:
      for key, (blk, off, n, _) in acc.slots.items():
        if n:
          _hip.check(ctx.lib.wbx_memcpy_d2d(ctx.handle, C.c_void_p(gbuf.ptr + 8 * plan.offsets[key]),
                                            C.c_void_p(blk.dev.ptr + 8 * off), n * 8), 'wbx_memcpy_d2d')
      for key, arr in host.items():
        if arr.size:
          _hip.check(ctx.lib.wbx_memcpy_h2d(ctx.handle, C.c_void_p(gbuf.ptr + 8 * plan.offsets[key]),
                                            arr.ctypes.data_as(C.c_void_p), arr.nbytes), 'wbx_memcpy_h2d')
    if comm is not None:  # all of it is enqueued on the library's stream: copies -> ncclAllReduce -> read-back
      flat = np.empty(ntot, dtype=np.float64)
      i0 = ctx.mark()
      _hip.check(ctx.lib.wbx_acc_allreduce(ctx.handle, comm.handle, C.c_void_p(gbuf.ptr), ntot), 'wbx_acc_allreduce')
      i1 = ctx.mark()
      _hip.check(ctx.lib.wbx_acc_read(ctx.handle, C.c_void_p(gbuf.ptr), ntot, flat.ctypes.data_as(C.c_void_p)), 'wbx_acc_read')
      us = ctx.mark_elapsed(i0, i1) * 1e3  # (the read-back has waited for the stream)
      if ctx is not _hip.default_context():
        ctx.marks_reset()  # (the reduction context's marks are this function's alone)
      comm.timings.update(collectives=comm.timings['collectives'] + 1, us_total=comm.timings['us_total'] + us, us_last=us,
                          bytes_last=int(ntot) * 8)
    else:
      import torch  # pylint: disable=g-import-not-at-top
      import torch.distributed as dist  # pylint: disable=g-import-not-at-top
      ctx.synchronize()
      t = device_tensor(gbuf.ptr, ntot, ctx.device_id)
      dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)  # in place, on the device buffer: no host hop
      torch.cuda.current_stream(t.device).synchronize()
      del t
      flat = ctx.download(gbuf.ptr, (ntot,), np.float64)
    plan.collectives += 1
  else:
    flat = np.zeros(ntot, dtype=np.float64)
    if stale:
      flat[plan.total] = 1.0
    else:
      for key, arr in acc.download_slots().items():
        flat[plan.offsets[key]:plan.offsets[key] + arr.size] = arr
      for key, arr in host.items():
        flat[plan.offsets[key]:plan.offsets[key] + arr.size] = arr.reshape(-1)
    if collective:
      import torch  # pylint: disable=g-import-not-at-top
      import torch.distributed as dist  # pylint: disable=g-import-not-at-top
      dist.all_reduce(torch.from_numpy(flat), op=dist.ReduceOp.SUM, group=group)
      plan.collectives += 1
  if flat[plan.total] != 0.0:  # somebody's layout / labels changed: every rank re-plans, then the payload moves again
    before = plan.collectives
    out, plan = reduce_accumulation(acc, group, all_reduce=all_reduce, plan=None, force=force, fence=None, comm=comm)
    plan.collectives += before
    return out, plan

  # ---- the arrays: views of the reduced buffer ----------------------------------------------------------------
  out = {}
  for path, specs in plan.leaves.items():
    arrays = []
    for spec in specs:
      key, rel, shape, strides, dims, coeff = spec
      base = plan.offsets[key] + rel
      view = np.lib.stride_tricks.as_strided(flat[base:], shape=shape, strides=tuple(8 * s for s in strides),
                                             writeable=False)
      coords, name, attrs = plan.frames[(path, spec)]
      data = np.array(view, order='C') if coeff == 1.0 else np.array(view, order='C') * coeff
      arrays.append(xr.DataArray(data, dims=dims, coords=coords, name=name, attrs=attrs, _raw_coords=True))
    out[path] = arrays[0] if len(arrays) == 1 else combining_sum(arrays)
  return out, plan


def resolve_state(state: AggregationState, acc: engine.Accumulation, group=None, *, all_reduce: bool = True,
                  plan: ReductionPlan | None = None, force: bool = False, comm: CabiCommunicator | None = None):
  """An AggregationState produced under `engine.accumulate_results(acc)` -> (the same tree with real numbers, summed
  over the ranks of `group`; ReductionPlan).  Nothing is waited for before the collective: the state's sums never
  visit the host on their own."""
  fence = getattr(state, '_fence', None)
  for which, tree in (('sws', state.sum_weighted_statistics), ('sw', state.sum_weights)):
    for path, da in _leaves(tree):
      acc.capture((which,) + tuple(path), da)
  leaves, plan = reduce_accumulation(acc, group, all_reduce=all_reduce, plan=plan, force=force, fence=fence, comm=comm)
  state._fence = None  # pylint: disable=protected-access  (waited for inside: the chunk's inputs are released)
  trees = {'sws': {}, 'sw': {}}
  for path, da in leaves.items():
    which, rest = path[0], path[1:]
    if not rest:
      trees[which] = da
      continue
    node = trees[which]
    for k in rest[:-1]:
      node = node.setdefault(k, {})
    node[rest[-1]] = da
  return AggregationState(trees['sws'], trees['sw']), plan


def all_reduce_state(state: AggregationState, group=None, *, force: bool = False) -> AggregationState:
  """Sum of every rank's (already resolved, host-side) AggregationState: the state is packed into an accumulation as
  host leaves and goes through the same single collective.  Prefer `engine.accumulate_results` + `resolve_state`, which
  keeps the sums on the device."""
  world, backend = _group_info(group)
  if world == 1 and not (force and backend is not None):
    return state
  state.wait()
  acc = engine.Accumulation()
  out, _ = resolve_state(state, acc, group, force=force)
  return out
