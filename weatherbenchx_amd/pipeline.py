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
from weatherbenchx_amd import distributed
from weatherbenchx_amd import engine
from weatherbenchx_amd import time_chunks as tc
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as metrics_base

LoadFn = Callable[[np.ndarray, object], tuple[Mapping, Mapping]]


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


def _consume(chunks, metrics, aggregators, commit):
  """Launches every chunk; the accumulators of chunk k are combined after chunk k+1 has been enqueued."""
  previous = []
  for offsets, predictions, targets in chunks:
    entries = []
    for stat_name, stats in metrics_base.generate_unique_statistics_for_all_metrics(metrics, predictions, targets):
      for var_name, stat in stats.items():
        for agg_name, agg in aggregators.items():
          state = agg.aggregate_stat_var(stat)
          if state is None:
            continue
          dims = state.sum_weighted_statistics.dims
          key = (offsets.init_time if 'init_time' in dims else None,
                 offsets.lead_time if 'lead_time' in dims else None)
          entries.append((state, agg_name, stat_name, var_name, key))
    commit(previous)
    previous = entries
  return previous


def evaluate_chunks(times: tc.TimeChunks, load_chunk: LoadFn, metrics: Mapping[str, metrics_base.Metric],
                    aggregator, *, rank: int = 0, world_size: int = 1, all_reduce: bool = True, prefetch: int = 0):
  """Returns {aggregator_name: AggregationState} (key None for a single unnamed aggregator).

  `load_chunk(init_times, lead_times) -> (predictions, targets)`; chunks are sharded round-robin over ranks.
  `prefetch` > 0 loads and uploads that many chunks ahead on a feeder thread (ChunkFeeder).
  """
  aggregators = {None: aggregator} if isinstance(aggregator, aggregation.Aggregator) else dict(aggregator)
  # acc[agg][type][stat][var][(init_off, lead_off)] -> DataArray
  acc = {name: {'sum_weighted_statistics': {}, 'sum_weights': {}} for name in aggregators}
  work = distributed.shard_chunks(list(times.iter_with_chunk_offsets()), rank, world_size)

  def commit(entries):
    # CombiningSum of one chunk's accumulators (beam_pipeline.py:509-510); waits for that chunk's read-back only
    for state, agg_name, stat_name, var_name, key in entries:
      state.wait()
      for kind, da in (('sum_weighted_statistics', state.sum_weighted_statistics), ('sum_weights', state.sum_weights)):
        slot = acc[agg_name][kind].setdefault(stat_name, {}).setdefault(str(var_name), {})
        slot[key] = da if key not in slot else aggregation.combining_sum([slot[key], da])

  # Software pipeline over chunks: the sums of chunk k are read back asynchronously and combined only after the
  # kernels of chunk k+1 have been enqueued, so the GPU never waits for the host-side bookkeeping.
  with engine.deferred_results():
    if prefetch:
      chunks = ChunkFeeder(work, load_chunk, depth=prefetch)
    else:
      chunks = ((offsets, *load_chunk(init_chunk, lead_chunk)) for offsets, (init_chunk, lead_chunk) in work)
    try:
      previous = _consume(chunks, metrics, aggregators, commit)
    finally:
      if prefetch:
        chunks.close()
    commit(previous)

  out = {}
  for agg_name in aggregators:
    trees = {}
    for kind in ('sum_weighted_statistics', 'sum_weights'):
      trees[kind] = {s: {v: _concat_pieces(p) for v, p in per_var.items()} for s, per_var in acc[agg_name][kind].items()}
    state = aggregation.AggregationState(trees['sum_weighted_statistics'], trees['sum_weights'])
    if all_reduce and world_size > 1:
      reduced_all_time = all(('init_time' not in da.dims and 'lead_time' not in da.dims)
                             for _, da in distributed._leaves(state.sum_weighted_statistics))  # pylint: disable=protected-access
      if not reduced_all_time:
        raise NotImplementedError('all-reduce with surviving init_time/lead_time needs an all-gather of disjoint '
                                  'offsets; run with all_reduce=False and concatenate on the host')
      state = distributed.all_reduce_state(state)
    out[agg_name] = state
  return out


def resolve_out_path(out_path, agg_name):
  """`metrics.nc` + aggregator name -> `metrics_<name>.nc` (beam_pipeline.py:388-399)."""
  import os  # pylint: disable=g-import-not-at-top
  if isinstance(out_path, str):
    if agg_name is None:
      return out_path
    base, ext = os.path.splitext(out_path)
    return f'{base}_{agg_name}{ext}'
  return out_path[agg_name]
