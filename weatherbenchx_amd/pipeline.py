"""The per-chunk evaluation loop (counterpart of the Beam graph, weatherbenchX/beam_pipeline.py:121-250,
322-399, 446-537), without Beam: iterate time chunks, aggregate each on the local GPU, add accumulators whose
reduced time dims coincide, place the rest at their chunk offsets, optionally all-reduce across ranks.

Key semantics kept from `_AggregationKey` (beam_pipeline.py:121-137): a per-chunk accumulator is identified by
(type, statistic, variable, init offset if init_time survives, lead offset if lead_time survives, aggregator
name); equal keys are summed (CombiningSum), different offsets are concatenated along the surviving time dims.
"""
from __future__ import annotations

from typing import Callable, Mapping

import numpy as np

from weatherbenchx_amd import aggregation
from weatherbenchx_amd import climatology_cache
from weatherbenchx_amd import distributed
from weatherbenchx_amd import engine
from weatherbenchx_amd import replay
from weatherbenchx_amd import time_chunks as tc
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as metrics_base

LoadFn = Callable[[np.ndarray, object], tuple[Mapping, Mapping]]


def pinned_empty(shape, dtype=np.float32) -> np.ndarray:
  """Page-locked array for loaders (wbx_host_alloc, pooled per device): a `load_chunk` that decodes / reads its fields
  straight into such arrays gets them uploaded by pure DMA (hipMemcpyAsync on the feeder's copy stream, overlapping both
  the kernels of the previous chunk and the loader's work on the next one); the launch stream waits on the copy's
  event, the host never does.  Every chunk takes fresh arrays: a block returns to the pool when the chunk's DataArrays
  are dropped, i.e. after its kernels have run, so an array is never overwritten while it is being read (the pool is
  the double buffer).  CONTRACT: fill the array, hand it over, leave it alone -- the upload returns before the DMA has read
  it, and the array is made read-only at that point, so a loader that keeps one array as its own double buffer gets a
  ValueError on its next write instead of corrupting the chunk in flight.  Arrays from anywhere else are uploaded through
  the runtime's pageable path (measured 55 GB/s), synchronously."""
  from weatherbenchx_amd import _hip  # pylint: disable=g-import-not-at-top
  return _hip.default_context().pinned_empty(shape, dtype)


def _concat_pieces(pieces: dict) -> xr.DataArray:
  """pieces: {(init_offset|None, lead_offset|None): DataArray} -> one array, concatenated in offset order
  (ConcatPerStatisticPerVariable, beam_pipeline.py:253-319)."""
  keys = list(pieces)
  inits = sorted({k[0] for k in keys if k[0] is not None})
  leads = sorted({k[1] for k in keys if k[1] is not None})
  if not inits and not leads:
    return pieces[(None, None)]
  rows = []
  for i in (inits or [None]):
    row = [pieces[(i, l)] for l in (leads or [None])]
    rows.append(xr.concat(row, dim='lead_time') if leads else row[0])
  return xr.concat(rows, dim='init_time') if inits else rows[0]


class ChunkFeeder:
  """Loads and stages chunks one ahead of the consumer (the role of the reference's LoadPredictionsAndTargets stage,
  beam_pipeline.py:69-116, in front of the aggregation DoFn).

  A worker thread calls `load_chunk` for chunk k+1 and uploads its host arrays through its own context -- a second HIP
  stream, so the copies overlap the kernels of chunk k -- while the main thread aggregates chunk k.  ctypes releases the
  GIL during the copies, and the device buffers come from the context's free list, so steady state does no hipMalloc /
  hipFree.  Arrays that are already device resident pass through untouched."""

  def __init__(self, work, load_chunk: LoadFn, depth: int = 1):
    import queue  # pylint: disable=g-import-not-at-top
    import threading  # pylint: disable=g-import-not-at-top
    self._work, self._load = list(work), load_chunk
    self._queue = queue.Queue(maxsize=max(1, int(depth)))
    self._full = queue.Full
    self._stop = threading.Event()
    self._ctx = None
    self._thread = threading.Thread(target=self._run, name='wbx-chunk-feeder', daemon=True)
    self._thread.start()

  def _run(self):
    try:
      self._ctx = engine.new_context()
      for item in self._work:
        _, (init_chunk, lead_chunk) = item
        predictions, targets = self._load(init_chunk, lead_chunk)
        # (foreign labeled arrays are converted HERE and handed on converted: the staged device copy hangs on the DataArray)
        predictions, targets = metrics_base._converted(predictions), metrics_base._converted(targets)  # pylint: disable=protected-access
        for name in predictions.keys():
          if name in targets.keys():
            engine.stage_inputs(self._ctx, [xr.as_dataarray(predictions[name]), xr.as_dataarray(targets[name])])
        if not self._put((item, predictions, targets, None)):
          return
    except BaseException as e:  # pylint: disable=broad-except
      self._put((None, None, None, e))
      return
    self._put((None, None, None, None))

  def _put(self, entry) -> bool:
    while not self._stop.is_set():
      try:
        self._queue.put(entry, timeout=0.1)
        return True
      except self._full:
        continue
    return False

  def close(self):
    """Stops the worker (the consumer gave up early)."""
    self._stop.set()

  def __iter__(self):
    while True:
      item, predictions, targets, err = self._queue.get()
      if err is not None:
        raise err
      if item is None:
        return
      yield item[0], predictions, targets


def _offset_key(offsets, stat_dims, reduce_dims):
  """(init offset | None, lead offset | None): the chunk offsets of the time dims that survive the aggregation
  (`_AggregationKey`, beam_pipeline.py:121-137, :221-232)."""
  keep = lambda d: d in stat_dims and d not in reduce_dims
  return (offsets.init_time if keep('init_time') else None, offsets.lead_time if keep('lead_time') else None)


class _SharedLoads:
  """Passes that were given the SAME `load_chunk` callable share the loaded chunk: the loader runs once per chunk and every
  such pass sees the same DataArray objects -- which is also what lets statistics of different passes over the same fields
  fuse (_plan_fusion).  The streams advance in lockstep, so only the latest chunk of a loader is kept."""

  def __init__(self):
    self._last = {}

  def stream(self, work, load_chunk):
    for k, (offsets, (init_chunk, lead_chunk)) in enumerate(work):
      hit = self._last.get(id(load_chunk))
      if hit is None or hit[0] != k or hit[1] is not load_chunk:
        hit = self._last[id(load_chunk)] = (k, load_chunk, load_chunk(init_chunk, lead_chunk))
      yield (offsets, *hit[2])

  def tee(self, load_chunk, source):
    """The same for a chunk FEEDER: `source` (one iterator per loader, handed to every pass of that loader) is advanced by
    whichever pass asks for chunk k first; the others get that very item."""
    key = ('feeder', id(load_chunk))
    self._last.setdefault(key, [-1, None, source])
    k = 0
    while True:
      state = self._last[key]
      if state[0] < k:
        try:
          state[1] = next(state[2])
        except StopIteration:
          return
        state[0] = k
      yield state[1]
      k += 1


def _plan_fusion(group_stats):
  """Look ahead over ALL statistics of a chunk (every pass): a deterministic (p, t[, climatology]) group whose two fields also
  have their zonal spectra aggregated in this chunk -- the same DataArray objects, i.e. passes that share a loader -- gets a
  request for the fused launch (engine.request_det_spectra): its stage-1 kernel then produces both spectra in the same sweep
  (12 B/point for both families instead of 20; csrc/wbx_zspec_det.hpp).  Only the common, simple frame is fused: one
  aggregator for the two spectra, no bins, weights that do not depend on longitude / wavenumber; the deterministic side decides
  at launch time whether its plan qualifies.  Everything else runs as separate launches, as before."""
  from weatherbenchx_amd import lazy  # pylint: disable=g-import-not-at-top
  from weatherbenchx_amd import spectra  # pylint: disable=g-import-not-at-top
  engine.clear_det_spectra_requests()
  by_source, groups, first_pass = {}, {}, {}
  for ipass, (unique, aggregators) in enumerate(group_stats):
    for _, stats in unique:
      for stat in stats.values():
        if isinstance(stat, spectra.LazySpectrum) and stat.is_lazy:
          by_source.setdefault(id(stat._source), []).append((stat, aggregators, ipass))  # pylint: disable=protected-access
        elif isinstance(stat, lazy.LazyStatistic) and stat.is_lazy and stat._group.kind == 'det':  # pylint: disable=protected-access
          groups[id(stat._group)] = stat._group  # pylint: disable=protected-access
          first_pass.setdefault(id(stat._group), ipass)  # pylint: disable=protected-access
  for grp in groups.values():
    sp, st = by_source.get(id(grp.p)), by_source.get(id(grp.t))
    if not sp or not st or len(sp) != 1 or len(st) != 1:
      continue
    (spec_p, aggs_p, pass_p), (spec_t, aggs_t, pass_t) = sp[0], st[0]
    if aggs_p is not aggs_t or len(aggs_p) != 1:
      continue
    # the deterministic launch has to come FIRST: a spectra pass in front of it launches its own transforms, and the fused
    # launch afterwards would be the slower kernel for nothing (and leave two spectra nobody asks for)
    if not first_pass[id(grp)] < min(pass_p, pass_t):
      continue
    agg = next(iter(aggs_p.values()))
    entries = []
    for spec in (spec_p, spec_t):
      if spec._k_dim in set(agg.reduce_dims) or not set(agg.reduce_dims) <= set(spec.dims):  # pylint: disable=protected-access
        break
      wp = agg._cached_weight_product(spec)  # pylint: disable=protected-access
      if wp is None:
        break
      w_da, bin_dims = wp
      if bin_dims or (w_da is not None and (spec._k_dim in w_da.dims or spec._lon_dim in w_da.dims)):  # pylint: disable=protected-access
        break
      row_dims = [d for d in spec.dims if d != spec._k_dim]  # pylint: disable=protected-access
      kept = [d for d in row_dims if d not in set(agg.reduce_dims)]
      w = w_da if w_da is not None else xr.DataArray(np.float64(1.0))
      entries.append(spec.rows_entry(w, kept)[:2])
    if len(entries) == 2 and entries[0][0] is entries[1][0]:  # one (group, scale) table for both spectra
      engine.request_det_spectra(grp.p, grp.t, entries[0][0], entries[0][1])


_token_memo: dict = {}  # id(values array) -> (the array, its token): loaders hand the same latitude / longitude objects on


_volatile = [0]


def _coord_token(values):
  """Identity of a coordinate's values that survives a loader handing out fresh (equal) arrays: its CONTENT.  A chunk record
  made on one chunk runs the next with the recorded weights, bin masks and atom tables (GridAreaWeighting / Regions / LandSea
  build them from exactly these coordinates), so two chunks may only share a signature if their coordinates hold the same
  numbers -- an address says nothing: freed buffers come back at the same address, writeable arrays are rewritten in place
  (ADVICE r5).  The content hash of a large array is memoised on the OBJECT, and only for arrays nobody can rewrite
  (writeable = False; the memo keeps the object, so its id cannot come round again); a large coordinate that lives in HBM is
  not read back per chunk: such chunks get a token of their own each time, i.e. they are never replayed."""
  hit = _token_memo.get(id(values))
  if hit is not None and hit[0] is values:
    return hit[1]
  if xr._is_torch(values):  # pylint: disable=protected-access
    if values.numel() > 4096:
      _volatile[0] += 1
      return ('volatile', _volatile[0])
    values = values.detach().cpu().numpy()
  a = np.asarray(values)
  if a.dtype == object:
    token = (a.dtype.str, a.shape, hash(tuple(a.reshape(-1).tolist())))
  else:
    token = (a.dtype.str, a.shape, hash(np.ascontiguousarray(a).tobytes()))
  if isinstance(values, np.ndarray) and not values.flags.writeable:
    if len(_token_memo) > 256:
      _token_memo.clear()
    _token_memo[id(values)] = (values, token)
  return token


def _array_signature(da):
  """What the launch plan of a chunk array depends on: frame, memory layout, and every coordinate that does not follow the time
  chunk (those are looked at per chunk: alignment of predictions and targets, the climatology's gather table)."""
  da = xr.as_dataarray(da)
  data = da.data
  if xr._is_torch(data):  # pylint: disable=protected-access
    layout = ('torch', bool(data.is_cuda), str(data.dtype), tuple(int(x) for x in data.stride()), int(data.data_ptr()) % 256)
  else:
    a = np.asarray(data)
    staged = tuple(sorted((str(k), bool(getattr(v, 'fence', None) is not None)) for k, v in (da.__dict__.get('_wbx_dev') or {}).items()))
    layout = ('host', a.dtype.str, a.strides, staged)
  coords = []
  for k, (cd, cv) in da._coords.items():  # pylint: disable=protected-access
    if k == 'mask':
      coords.append((k, cd, 'torch' if xr._is_torch(cv) else 'host',  # pylint: disable=protected-access
                     tuple(int(x) for x in cv.stride()) if xr._is_torch(cv) else np.asarray(cv).strides))  # pylint: disable=protected-access
    elif set(cd) <= replay.TIME_DIMS and cd:
      coords.append((k, cd, 'time'))
    else:
      coords.append((k, cd, _coord_token(cv)))
  return (da.dims, tuple(da.shape), da.name, layout, tuple(coords))


def _time_aligned(p, t) -> bool:
  """The time labels predictions and targets share are equal (else the statistics align / join them: another plan)."""
  for d in ('init_time', 'lead_time'):
    if d in p._coords and d in t._coords:  # pylint: disable=protected-access
      a, b = np.asarray(p._coords[d][1]), np.asarray(t._coords[d][1])  # pylint: disable=protected-access
      if a.shape != b.shape or not np.array_equal(a, b):
        return False
  return True


# Chunks enqueued ahead of the one the GPU is working on: with 1 the host waits for chunk k - 1 before it enqueues chunk k + 1, and
# a launch stream runs dry for the ~50 us the host needs to get there whenever a chunk is shorter than expected; 2 keeps a chunk in
# hand (one more chunk's inputs are held in HBM).
CHUNKS_IN_FLIGHT = 2


# Weightings / binnings whose factors do not depend on a chunk's time labels (a record made on one chunk holds for the next)
_TIME_INDEPENDENT = ('GridAreaWeighting', 'Regions', 'LandSea')


class _Replayer:
  """Chunk records of one `_consume` loop (replay.py): a chunk signature met for the SECOND time is recorded while it takes the
  ordinary path (the first chunk builds plans, uploads tables and creates the accumulator slots -- not the steady state), and
  every later chunk with that signature is ONE library call."""

  def __init__(self, passes, acc):
    self.passes, self.acc = passes, acc
    self.seen, self.records = {}, {}
    self.ok = replay.ENABLED and engine.accumulation_active() is acc
    for _, _, aggregators in passes:
      for agg in aggregators.values():
        for part in list(agg.weigh_by or []) + list(agg.bin_by or []):
          if type(part).__name__ not in _TIME_INDEPENDENT and not getattr(part, 'time_independent', False):
            self.ok = False
    self.recorder = None
    self.sig = None

  def arrays(self, group):
    out = {}
    for ipass, (_, predictions, targets) in enumerate(group):
      for side, data in (('p', predictions), ('t', targets)):
        for name in data.keys():
          out[(ipass, side, str(name))] = xr.as_dataarray(data[name])
    return out

  def signature(self, group, arrays):
    sig = []
    for ipass, ((offsets, predictions, targets), (_, _, aggregators)) in enumerate(zip(group, self.passes)):
      keeps = tuple((name, None if 'init_time' in agg.reduce_dims else offsets.init_time,
                     None if 'lead_time' in agg.reduce_dims else offsets.lead_time) for name, agg in aggregators.items())
      names = tuple(str(n) for n in predictions.keys()), tuple(str(n) for n in targets.keys())
      per = tuple((role[1], role[2], _array_signature(da)) for role, da in arrays.items() if role[0] == ipass)
      aligned = all(_time_aligned(arrays[(ipass, 'p', n)], arrays[(ipass, 't', n)]) for n in names[0] if n in names[1])
      sig.append((keeps, names, per, aligned))
    sig.append(self.acc.chunk_index % 2 if engine.ALTERNATE_CHUNKS else 0)  # (which launch stream an ensemble launch takes)
    return tuple(sig)

  def try_replay(self, group):
    """-> the chunk's fences when it was replayed, else None (the caller runs it; `recording()` may then watch it)."""
    self.recorder = None
    if not self.ok:
      return None
    arrays = self.arrays(group)
    try:
      sig = self.signature(group, arrays)
      hash(sig)
    except TypeError:
      return None
    rec = self.records.get(sig)
    if rec is not None and rec is not False:
      done = rec.replay(arrays)
      if done is not None:
        return done
      self.records[sig] = False  # (its inputs cannot be addressed the way the recorded ones were: ordinary path from now on)
      return None
    n = self.seen[sig] = self.seen.get(sig, 0) + 1
    if rec is None and n >= 2:
      self.recorder = replay.ChunkRecorder(arrays, self.acc)
      self.sig = sig
    return None

  RECORDING_ATTEMPTS = 3  # (a chunk that still allocates -- pools filling up behind a record's pinned blocks -- is tried again)

  def finish(self, failed=False):
    if self.recorder is not None:
      rec = None if failed else self.recorder.finish()
      if rec is not None:
        self.records[self.sig] = rec
      elif failed or self.seen.get(self.sig, 0) > self.RECORDING_ATTEMPTS:
        self.records[self.sig] = False
      self.recorder = None


def _consume(chunk_streams, passes, acc):
  """Launches every chunk of every pass.  `chunk_streams[i]` yields (offsets, predictions, targets) for pass i; the streams
  advance in lockstep, so chunk k of every pass is enqueued before chunk k + 1 of any.  A chunk's results are ADDED to the
  device accumulators of `acc` right behind its kernels (slot = pass, aggregator, statistic, variable, surviving offsets);
  the host only records where each result lives.  The inputs of chunk k are released once chunk k+1 has been enqueued."""
  import collections  # pylint: disable=g-import-not-at-top
  in_flight = collections.deque()  # the states of the last CHUNKS_IN_FLIGHT chunks: their inputs are held until their kernels ran

  def retire(states):
    in_flight.append(states)
    while len(in_flight) > CHUNKS_IN_FLIGHT:
      for state in in_flight.popleft():
        state.wait()  # the fence of that chunk's kernels: lets go of its inputs (nothing is read back here)
  replayer = _Replayer(passes, acc)
  groups = iter(zip(*chunk_streams))
  group = next(groups, None)
  with climatology_cache.chunk_loop():
    while group is not None:
      group = _consume_one(group, groups, passes, acc, replayer, retire)
  while in_flight:
    for state in in_flight.popleft():
      state.wait()


def _consume_one(group, groups, passes, acc, replayer, retire):
  """Chunk `group` enqueued; -> the next group (loaded before this one's kernels have run)."""
  acc.next_chunk()
  # steady state: a chunk like one that has been recorded is ONE call into the library (replay.py; wbx_chunk_replay)
  done = replayer.try_replay(group)
  if done is not None:
    retire(done)
  else:
    if replayer.recorder is not None:
      with replayer.recorder:
        try:
          states = _run_chunk(group, passes, acc)
        except BaseException:
          replayer.finish(failed=True)
          raise
      replayer.finish()
    else:
      states = _run_chunk(group, passes, acc)
    retire(states)
  # climatology slab pools (climatology_cache.py): this chunk's slabs are stamped with fences behind its launches, and the
  # slabs of the NEXT chunk are asked for now -- one chunk ahead on the copy stream
  climatology_cache.chunk_enqueued()
  nxt = next(groups, None)
  if nxt is not None and climatology_cache.active():
    try:
      climatology_cache.prefetch_for(passes, nxt)
    except Exception:  # pylint: disable=broad-except  (the statistic itself raises it where the reference would)
      pass
  return nxt


def _run_chunk(group, passes, acc):
  """One chunk of every pass through the ordinary path: statistics -> fused launches -> adds into the accumulators."""
  states = []
  # Built-in statistics are lazy (no payload), so all of them can exist before the first launch: every statistic of a
  # (predictions, targets, climatology) triple then shares ONE fused launch (the reference generates and aggregates
  # them one at a time to bound the memory of materialised statistics, beam_pipeline.py:186-197)
  uniques = [list(metrics_base.generate_unique_statistics_for_all_metrics(metrics, predictions, targets))
             for (_, metrics, _), (_, predictions, targets) in zip(passes, group)]
  if len(passes) > 1:
    _plan_fusion([(u, aggs) for u, (_, _, aggs) in zip(uniques, passes)])
  for (pass_name, metrics, aggregators), (offsets, predictions, targets), unique in zip(passes, group, uniques):
    # (the spread lane of an ensemble group decides which kernel variant serves all its lanes: Aggregator.note_statistics)
    aggregation.Aggregator.note_statistics(dict(unique))
    for stat_name, stats in unique:
      for var_name, stat in stats.items():
        for agg_name, agg in aggregators.items():
          dims = getattr(stat, 'dims', ())
          key = _offset_key(offsets, dims, set(agg.reduce_dims))
          acc.set_label((pass_name, agg_name, stat_name, str(var_name), key))
          state = agg.aggregate_stat_var(stat)
          if state is None:
            continue
          got = state.sum_weighted_statistics.dims
          if ('init_time' in got, 'lead_time' in got) != (key[0] is not None, key[1] is not None):
            # e.g. ByExactCoord('lead_time') on station data: a BIN dim that is called like a chunked time dim, with labels that
            # depend on the chunk's data.  The device accumulators need one result frame per label for the whole loop.
            raise ValueError(
                f'statistic {stat_name!r} / {var_name!r}: the aggregated result has dims {got} but the chunk offsets say '
                f'init_time / lead_time survive = {key[0] is not None} / {key[1] is not None}.  Results whose frame depends on '
                "the chunk's data (bins over coordinates of sparse data) cannot use the device accumulators: run "
                "beam_pipeline.define_pipeline(..., accumulate='host') or add the per-chunk AggregationStates yourself.")
          for kind, da in (('sum_weighted_statistics', state.sum_weighted_statistics), ('sum_weights', state.sum_weights)):
            acc.capture((pass_name, agg_name, kind, stat_name, str(var_name), key), da)
          states.append(state)
  engine.clear_det_spectra_requests()
  return states


def evaluate_passes(times: tc.TimeChunks, passes, *, rank: int = 0, world_size: int = 1, all_reduce: bool = True,
                    prefetch: int = 0, group=None, force_collective: bool = False, comm=None, stats: dict | None = None,
                    sharding: str | None = None):
  """Several evaluations over the same time chunks -- `passes` = [(name, load_chunk, metrics, aggregator | {name: aggregator}),
  ...], e.g. a deterministic suite, zonal spectra under another aggregator and an ensemble suite from another loader -- as
  ONE job: the chunks of all passes run interleaved (chunk k of every pass before chunk k + 1), every accumulator of every
  pass lives in one engine.Accumulation, and the ranks combine EVERYTHING with ONE sum all-reduce of one device buffer at the
  end, like the reference's single CombinePerKey over all keys of the job (beam_pipeline.py:509-510).

  Returns {pass name: {aggregator name: AggregationState}} (aggregator key None for a single unnamed aggregator).
  `comm`: a distributed.CabiCommunicator -- the collective then goes through the library's own RCCL entry point
  (wbx_acc_allreduce) instead of torch.distributed.  `stats`, if given, receives {'collectives': n, 'accumulator_values': n}.
  `sharding`: 'round_robin' (chunk i -> rank i mod n) or 'block' (contiguous runs of chunks per rank); None picks 'block' when
  a climatology of the job sits behind a slab pool (consecutive chunks share its slabs), else round robin."""
  norm = []
  for name, load_chunk, metrics, aggregator in passes:
    aggs = {None: aggregator} if isinstance(aggregator, aggregation.Aggregator) else dict(aggregator)
    norm.append((name, load_chunk, metrics, aggs))
  if len({n for n, *_ in norm}) != len(norm):
    raise ValueError('pass names must be unique')
  if sharding is None:
    sharding = 'block' if climatology_cache.wanted_by([m for _, _, m, _ in norm]) else 'round_robin'
  work = distributed.shard_chunks(list(times.iter_with_chunk_offsets()), rank, world_size, sharding)
  acc = engine.Accumulation()
  # Software pipeline over chunks: nothing is waited for inside the loop except the previous chunk's kernels (to let
  # go of its inputs) after the next chunk has been enqueued, so the GPU never waits for host-side bookkeeping.
  feeders = {}
  shared = _SharedLoads()
  with engine.accumulate_results(acc):
    streams = []
    for _, load_chunk, _, _ in norm:
      if prefetch:
        # ONE feeder per loader: passes that share a loader share its chunks (and can fuse), exactly as without prefetch
        if id(load_chunk) not in feeders:
          feeders[id(load_chunk)] = ChunkFeeder(work, load_chunk, depth=prefetch)
          feeders[id(load_chunk)].source = iter(feeders[id(load_chunk)])
        streams.append(shared.tee(load_chunk, feeders[id(load_chunk)].source))
      else:
        streams.append(shared.stream(work, load_chunk))
    try:
      _consume(streams, [(n, m, a) for n, _, m, a in norm], acc)
    finally:
      for feeder in feeders.values():
        feeder.close()
  leaves, plan = distributed.reduce_accumulation(acc, group, all_reduce=all_reduce and (world_size > 1 or force_collective),
                                                 force=force_collective, comm=comm)
  if stats is not None:
    stats['collectives'] = plan.collectives
    stats['accumulator_values'] = plan.total

  # acc[pass][agg][type][stat][var][(init_off, lead_off)] -> DataArray
  trees = {n: {a: {'sum_weighted_statistics': {}, 'sum_weights': {}} for a in aggs} for n, _, _, aggs in norm}
  for (pass_name, agg_name, kind, stat_name, var_name, key), da in leaves.items():
    trees[pass_name][agg_name][kind].setdefault(stat_name, {}).setdefault(var_name, {})[key] = da
  out = {}
  for pass_name, _, _, aggs in norm:
    out[pass_name] = {}
    for agg_name in aggs:
      done = {kind: {s: {v: _concat_pieces(p) for v, p in per_var.items()}
                     for s, per_var in trees[pass_name][agg_name][kind].items()}
              for kind in ('sum_weighted_statistics', 'sum_weights')}
      out[pass_name][agg_name] = aggregation.AggregationState(done['sum_weighted_statistics'], done['sum_weights'])
  return out


def evaluate_chunks(times: tc.TimeChunks, load_chunk: LoadFn, metrics: Mapping[str, metrics_base.Metric],
                    aggregator, *, rank: int = 0, world_size: int = 1, all_reduce: bool = True, prefetch: int = 0,
                    group=None, force_collective: bool = False, comm=None, sharding: str | None = None):
  """Returns {aggregator_name: AggregationState} (key None for a single unnamed aggregator).

  `load_chunk(init_times, lead_times) -> (predictions, targets)`; chunks are sharded round-robin over ranks.
  `prefetch` > 0 loads and uploads that many chunks ahead on a feeder thread (ChunkFeeder).

  The accumulators stay in HBM for the whole loop (engine.Accumulation).  With `all_reduce` and an initialised
  torch.distributed group of more than one rank, the ranks' accumulators are combined with ONE sum all-reduce of one
  device buffer at the end (distributed.reduce_accumulation): accumulators of a reduced time dim add up, those of a
  surviving `init_time` / `lead_time` are owned by the rank that ran the chunk and are zero elsewhere, so the same
  collective also assembles the pieces the reference concatenates (beam_pipeline.py:253-319).  Every rank returns the
  complete result.  `force_collective` issues the collective even in a one-rank group (plumbing tests on one GPU).
  Several (loader, metrics, aggregator) evaluations of one job: `evaluate_passes` (still one collective).
  """
  out = evaluate_passes(times, [('', load_chunk, metrics, aggregator)], rank=rank, world_size=world_size, all_reduce=all_reduce,
                        prefetch=prefetch, group=group, force_collective=force_collective, comm=comm, sharding=sharding)
  return out['']


def resolve_out_path(out_path, agg_name):
  """`metrics.nc` + aggregator name -> `metrics_<name>.nc` (beam_pipeline.py:388-399)."""
  import os  # pylint: disable=g-import-not-at-top
  if isinstance(out_path, str):
    if agg_name is None:
      return out_path
    base, ext = os.path.splitext(out_path)
    return f'{base}_{agg_name}{ext}'
  return out_path[agg_name]
