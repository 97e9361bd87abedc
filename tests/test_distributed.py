"""Multi-rank path on CPU: two gloo processes shard the time chunks round-robin, aggregate locally and
all-reduce ONE packed accumulator buffer; the result must equal the single-process evaluation.
(The per-chunk device math is the NumPy plan interpreter here; on the GPU box the same code runs over RCCL.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ensemble_case(tp):
  """Probabilistic workload: accumulators with a category dim next to scalar-per-lead ones in the packed buffer."""
  import mock_data
  from weatherbenchx_amd import aggregation, time_chunks, weighting
  from weatherbenchx_amd.metrics import probabilistic
  predictions = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-04T00', lead_start_days=0,
                                               lead_stop_days=1, random=True, seed=11, ensemble_size=4)
  targets = mock_data.mock_target_data(time_start='2020-01-01T00', time_stop='2020-01-06T00', random=True, seed=12)
  init_times = predictions['geopotential']['time'].values
  lead_times = predictions['geopotential']['prediction_timedelta'].values
  times = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=1, lead_time_chunk_size=1)
  metrics = {'crps': probabilistic.CRPSEnsemble(ensemble_dim='realization', use_sort=True),
             'ssr': probabilistic.UnbiasedSpreadSkillRatio(ensemble_dim='realization'),
             'rank': probabilistic.RankHistogram(ensemble_dim='realization')}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'lead_time', 'latitude', 'longitude'],
                               weigh_by=[weighting.GridAreaWeighting()])
  return times, tp._loader(predictions, targets), metrics, agg


def _ensemble_worker(rank, world_size, port, out_dir):
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
  import torch.distributed as dist
  import fake_device
  import test_pipeline as tp
  from weatherbenchx_amd import pipeline

  class MP:  # minimal monkeypatch
    def setattr(self, obj, name, value):
      setattr(obj, name, value)
  fake_device.install(MP())
  dist.init_process_group('gloo', rank=rank, world_size=world_size)
  try:
    times, load, metrics, agg = _ensemble_case(tp)
    state = pipeline.evaluate_chunks(times, load, metrics, agg, rank=rank, world_size=world_size, prefetch=1)[None]
    vals = state.metric_values(metrics)
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), **{k: v.values for k, v in vals.items()})
  finally:
    dist.destroy_process_group()


def test_two_rank_allreduce_of_an_ensemble_workload(tmp_path, monkeypatch):
  """CRPS, spread/skill and rank histograms over 2 gloo ranks (chunk feeder on, launches alternating over two contexts
  on each rank): every rank ends up with the single-process numbers."""
  import socket
  import torch.multiprocessing as mp
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  mp.spawn(_ensemble_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  import fake_device
  import test_pipeline as tp
  from weatherbenchx_amd import pipeline
  fake_device.install(monkeypatch)
  times, load, metrics, agg = _ensemble_case(tp)
  want = pipeline.evaluate_chunks(times, load, metrics, agg)[None].metric_values(metrics)
  for rank in (0, 1):
    got = np.load(os.path.join(tmp_path, f'rank{rank}.npz'))
    assert set(got.files) == set(want)
    for k in want:
      np.testing.assert_allclose(got[k], want[k].values, rtol=1e-12, err_msg=k)


def _worker(rank, world_size, port, out_dir):
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
  import torch.distributed as dist
  import fake_device
  import test_pipeline as tp
  from weatherbenchx_amd import aggregation, pipeline, time_chunks
  from weatherbenchx_amd.metrics import deterministic

  class MP:  # minimal monkeypatch
    def setattr(self, obj, name, value):
      setattr(obj, name, value)
  fake_device.install(MP())
  dist.init_process_group('gloo', rank=rank, world_size=world_size)
  try:
    predictions, targets = tp._datasets()
    init_times = predictions['geopotential']['time'].values
    lead_times = predictions['geopotential']['prediction_timedelta'].values
    times = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=1, lead_time_chunk_size=1)
    metrics = {'rmse': deterministic.RMSE(), 'bias': deterministic.Bias()}
    agg = aggregation.Aggregator(reduce_dims=['init_time', 'lead_time', 'latitude', 'longitude'])
    state = pipeline.evaluate_chunks(times, tp._loader(predictions, targets), metrics, agg, rank=rank,
                                     world_size=world_size)[None]
    vals = state.metric_values(metrics)
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), **{k: v.values for k, v in vals.items()})
  finally:
    dist.destroy_process_group()


def test_two_rank_allreduce_matches_single_process(tmp_path, monkeypatch):
  import torch.multiprocessing as mp
  import socket
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  # single-process reference
  import fake_device
  import test_pipeline as tp
  from weatherbenchx_amd import aggregation, pipeline, time_chunks
  from weatherbenchx_amd.metrics import deterministic
  fake_device.install(monkeypatch)
  predictions, targets = tp._datasets()
  init_times = predictions['geopotential']['time'].values
  lead_times = predictions['geopotential']['prediction_timedelta'].values
  times = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=1, lead_time_chunk_size=1)
  metrics = {'rmse': deterministic.RMSE(), 'bias': deterministic.Bias()}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'lead_time', 'latitude', 'longitude'])
  want = pipeline.evaluate_chunks(times, tp._loader(predictions, targets), metrics, agg)[None].metric_values(metrics)
  for rank in (0, 1):
    got = np.load(os.path.join(tmp_path, f'rank{rank}.npz'))
    assert set(got.files) == set(want)
    for k in want:
      np.testing.assert_allclose(got[k], want[k].values, rtol=1e-12)


def test_pack_unpack_round_trip_and_sharding():
  sys.path.insert(0, ROOT)
  from weatherbenchx_amd import distributed
  from weatherbenchx_amd import xarray_lite as xr
  from weatherbenchx_amd.aggregation import AggregationState
  st = AggregationState({'s': {'a': xr.DataArray(np.arange(6.0).reshape(2, 3), dims=['x', 'y']),
                               'b': xr.DataArray(np.float64(7.0))}},
                        {'s': {'a': xr.DataArray(np.ones((2, 3)), dims=['x', 'y']), 'b': xr.DataArray(np.float64(2.0))}})
  flat, layout, items = distributed.pack_state(st)
  assert flat.shape == (14,) and len(layout) == 4
  back = distributed.unpack_state(flat * 2, items)
  np.testing.assert_allclose(back.sum_weighted_statistics['s']['a'].values, 2 * np.arange(6.0).reshape(2, 3))
  np.testing.assert_allclose(back.sum_weights['s']['b'].values, 4.0)
  assert distributed.shard_chunks(list(range(7)), 1, 3) == [1, 4]
  assert distributed.all_reduce_state(st) is st  # no process group: identity


def _ragged_worker(rank, world_size, port, out_dir):
  sys.path.insert(0, ROOT)
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
  import torch.distributed as dist
  from weatherbenchx_amd import distributed
  from weatherbenchx_amd import xarray_lite as xr
  from weatherbenchx_amd.aggregation import AggregationState
  dist.init_process_group('gloo', rank=rank, world_size=world_size)
  try:
    n = 3 + rank  # rank 1 packs one value more
    st = AggregationState({'s': {'v': xr.DataArray(np.ones(n), dims=['x'])}},
                          {'s': {'v': xr.DataArray(np.ones(n), dims=['x'])}})
    try:
      distributed.all_reduce_state(st)
      outcome = 'no error'
    except (ValueError, RuntimeError) as e:  # gloo itself may reject mismatched sizes first
      outcome = type(e).__name__
    open(os.path.join(out_dir, f'ragged{rank}.txt'), 'w').write(outcome)
  finally:
    dist.destroy_process_group()


def test_ragged_shards_fail_loudly(tmp_path):
  import socket
  import torch.multiprocessing as mp
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  mp.spawn(_ragged_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  outcomes = [open(os.path.join(tmp_path, f'ragged{r}.txt')).read() for r in (0, 1)]
  assert all(o != 'no error' for o in outcomes), outcomes
