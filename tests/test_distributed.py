"""Multi-rank path on CPU: two gloo processes shard the time chunks round-robin, accumulate locally and combine with
ONE sum all-reduce of one buffer that holds the union of every rank's accumulator slots; whatever survives of
(init_time, lead_time) -- nothing, one of them, both -- every rank must end up with the single-process result
(the reference's CombinePerKey + ConcatPerStatisticPerVariable, beam_pipeline.py:253-319, 509-510).
(The per-chunk device math is the NumPy plan interpreter here; on the GPU box the same code runs over RCCL.)"""
import os
import pickle
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init(rank, world_size, out_dir):
  """gloo group over a file store in the test's tmp dir (no TCP port to collide on when tests run in parallel)."""
  for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
      sys.path.insert(0, p)
  import torch.distributed as dist
  import fake_device

  class MP:  # minimal monkeypatch
    def setattr(self, obj, name, value):
      setattr(obj, name, value)
  fake_device.install(MP())
  dist.init_process_group('gloo', init_method='file://' + os.path.join(out_dir, 'rendezvous'), rank=rank,
                          world_size=world_size)
  return dist


def _spawn(fn, world_size, tmp_path, *args):
  import torch.multiprocessing as mp
  mp.spawn(fn, args=(world_size, str(tmp_path)) + args, nprocs=world_size, join=True)


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


def _ensemble_worker(rank, world_size, out_dir):
  dist = _init(rank, world_size, out_dir)
  import test_pipeline as tp
  from weatherbenchx_amd import pipeline
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
  _spawn(_ensemble_worker, 2, tmp_path)
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


REDUCE_SETS = {'all_time': ['init_time', 'lead_time', 'latitude', 'longitude'],
               'lead_kept': ['init_time', 'latitude', 'longitude'],       # the benchmark's default: RMSE per lead time
               'both_kept': ['latitude', 'longitude'],                    # every chunk owns its own offsets
               'maps_per_lead': ['init_time'],                            # latitude / longitude survive too
               'init_kept': ['lead_time', 'latitude', 'longitude']}


def _det_case(tp, reduce_dims, chunk=(1, 1)):
  from weatherbenchx_amd import aggregation, binning, time_chunks, weighting
  from weatherbenchx_amd.metrics import deterministic
  predictions, targets = tp._datasets()
  init_times = predictions['geopotential']['time'].values
  lead_times = predictions['geopotential']['prediction_timedelta'].values
  times = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=chunk[0], lead_time_chunk_size=chunk[1])
  metrics = {'rmse': deterministic.RMSE(), 'bias': deterministic.Bias(),
             'wind': deterministic.WindVectorRMSE('geopotential', 'geopotential', 'wind')}  # a linear combination
  aggs = {'plain': aggregation.Aggregator(reduce_dims=reduce_dims)}
  if 'latitude' in reduce_dims:
    aggs['regions'] = aggregation.Aggregator(
        reduce_dims=reduce_dims, weigh_by=[weighting.GridAreaWeighting()], masked=True,
        bin_by=[binning.Regions({'global': ((-90, 90), (0, 360)), 'nh': ((20, 90), (0, 360))})])
  return times, tp._loader(predictions, targets), metrics, aggs


def _dump(states, metrics, path):
  out = {}
  for name, st in states.items():
    out[name] = {'metrics': {k: (v.dims, np.asarray(v.values), {c: np.asarray(v[c].values) for c in v.dims if c in v.coords})
                             for k, v in st.metric_values(metrics).items()},
                 'sws': {(s, v): (da.dims, np.asarray(da.values)) for s, per in st.sum_weighted_statistics.items()
                         for v, da in per.items()},
                 'sw': {(s, v): (da.dims, np.asarray(da.values)) for s, per in st.sum_weights.items() for v, da in per.items()}}
  with open(path, 'wb') as f:
    pickle.dump(out, f)


def _det_worker(rank, world_size, out_dir, which, chunk):
  dist = _init(rank, world_size, out_dir)
  import test_pipeline as tp
  from weatherbenchx_amd import pipeline
  try:
    times, load, metrics, aggs = _det_case(tp, REDUCE_SETS[which], chunk)
    states = pipeline.evaluate_chunks(times, load, metrics, aggs, rank=rank, world_size=world_size)
    _dump(states, metrics, os.path.join(out_dir, f'rank{rank}.pkl'))
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('which,chunk,world', [('all_time', (1, 1), 2), ('lead_kept', (1, 1), 2), ('both_kept', (1, 1), 2),
                                               ('maps_per_lead', (1, 1), 2), ('init_kept', (1, 2), 2),
                                               ('lead_kept', (1, 2), 3), ('both_kept', (2, 1), 3)])
def test_ranks_equal_single_process_whatever_survives(tmp_path, monkeypatch, which, chunk, world):
  """reduce_dims in {all, [init,lat,lon], [lat,lon], [init], ...} x chunkings x 2-3 ranks (3 ranks over 2 chunks: one
  rank has nothing to contribute and still returns the complete result)."""
  _spawn(_det_worker, world, tmp_path, which, chunk)
  import fake_device
  import test_pipeline as tp
  from weatherbenchx_amd import pipeline
  fake_device.install(monkeypatch)
  times, load, metrics, aggs = _det_case(tp, REDUCE_SETS[which], chunk)
  states = pipeline.evaluate_chunks(times, load, metrics, aggs)
  _dump(states, metrics, os.path.join(tmp_path, 'want.pkl'))
  want = pickle.load(open(os.path.join(tmp_path, 'want.pkl'), 'rb'))
  assert any(len(v['metrics']) for v in want.values())
  for rank in range(world):
    got = pickle.load(open(os.path.join(tmp_path, f'rank{rank}.pkl'), 'rb'))
    assert set(got) == set(want)
    for name in want:
      for part in ('sws', 'sw'):
        assert set(got[name][part]) == set(want[name][part])
        for k, (dims, vals) in want[name][part].items():
          assert got[name][part][k][0] == dims, (name, part, k)
          np.testing.assert_allclose(got[name][part][k][1], vals, rtol=1e-12, atol=1e-300, err_msg=f'{name} {part} {k}')
      assert set(got[name]['metrics']) == set(want[name]['metrics'])
      for k, (dims, vals, coords) in want[name]['metrics'].items():
        gd, gv, gc = got[name]['metrics'][k]
        assert gd == dims
        np.testing.assert_allclose(gv, vals, rtol=1e-12, equal_nan=True, err_msg=f'{name} {k}')
        for c in coords:
          assert np.array_equal(gc[c], coords[c]), (name, k, c)  # time coordinates are concatenated in offset order


def test_host_states_and_sharding():
  """all_reduce_state on finished (host) states is the identity without a process group; sharding is round-robin."""
  sys.path.insert(0, ROOT)
  from weatherbenchx_amd import distributed, engine
  from weatherbenchx_amd import xarray_lite as xr
  from weatherbenchx_amd.aggregation import AggregationState
  st = AggregationState({'s': {'a': xr.DataArray(np.arange(6.0).reshape(2, 3), dims=['x', 'y']),
                               'b': xr.DataArray(np.float64(7.0))}},
                        {'s': {'a': xr.DataArray(np.ones((2, 3)), dims=['x', 'y']), 'b': xr.DataArray(np.float64(2.0))}})
  assert distributed.shard_chunks(list(range(7)), 1, 3) == [1, 4]
  assert distributed.all_reduce_state(st) is st  # no process group: identity
  back, plan = distributed.resolve_state(st, engine.Accumulation())  # host leaves through the packed buffer
  assert plan.total == 14 and plan.collectives == 0
  np.testing.assert_allclose(back.sum_weighted_statistics['s']['a'].values, np.arange(6.0).reshape(2, 3))
  np.testing.assert_allclose(back.sum_weights['s']['b'].values, 2.0)
  assert back.sum_weighted_statistics['s']['b'].dims == ()


def _ragged_worker(rank, world_size, out_dir):
  dist = _init(rank, world_size, out_dir)
  from weatherbenchx_amd import distributed
  from weatherbenchx_amd import xarray_lite as xr
  from weatherbenchx_amd.aggregation import AggregationState
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
  _spawn(_ragged_worker, 2, tmp_path)
  outcomes = [open(os.path.join(tmp_path, f'ragged{r}.txt')).read() for r in (0, 1)]
  assert all(o == 'ValueError' for o in outcomes), outcomes  # the layout exchange catches it before the payload collective


def _step_worker(rank, world_size, out_dir):
  """A bench-like loop: every step accumulates one launch set and all-reduces it; the layout is exchanged once."""
  dist = _init(rank, world_size, out_dir)
  from weatherbenchx_amd import aggregation, distributed, engine, weighting
  from weatherbenchx_amd import xarray_lite as xr
  from weatherbenchx_amd.metrics import base as metrics_base
  from weatherbenchx_amd.metrics import deterministic
  try:
    lat, lon = np.linspace(-80, 80, 9), np.arange(12) * 30.0
    metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE()}
    agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
    plan, sums = None, []
    for step in range(3):
      rng = np.random.default_rng(100 * step + rank)
      p = {'v': xr.DataArray(rng.normal(size=(2, 9, 12)).astype(np.float32), dims=('lead_time', 'latitude', 'longitude'),
                             coords={'latitude': lat, 'longitude': lon})}
      t = {'v': xr.DataArray(rng.normal(size=(2, 9, 12)).astype(np.float32), dims=('lead_time', 'latitude', 'longitude'),
                             coords={'latitude': lat, 'longitude': lon})}
      acc = engine.Accumulation()
      with engine.accumulate_results(acc):
        state = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, p, t))
      state, plan = distributed.resolve_state(state, acc, plan=plan)
      sums.append(state.metric_values(metrics)['rmse.v'].values)
    assert plan.collectives == 3  # one sum all-reduce per step, the layout exchange only before the first
    np.save(os.path.join(out_dir, f'steps{rank}.npy'), np.stack(sums))
  finally:
    dist.destroy_process_group()


def test_per_step_allreduce_reuses_the_layout(tmp_path):
  _spawn(_step_worker, 2, tmp_path)
  a, b = (np.load(os.path.join(tmp_path, f'steps{r}.npy')) for r in (0, 1))
  np.testing.assert_array_equal(a, b)
  lat = np.linspace(-80, 80, 9)
  from oracle import wbx_oracle as O
  w = O.grid_area_weights(lat)
  for step in range(3):
    num = den = 0.0
    for rank in (0, 1):
      rng = np.random.default_rng(100 * step + rank)
      p = rng.normal(size=(2, 9, 12)).astype(np.float32).astype(np.float64)
      t = rng.normal(size=(2, 9, 12)).astype(np.float32).astype(np.float64)
      num = num + ((p - t) ** 2 * w[None, :, None]).sum(axis=(1, 2))
      den = den + (np.ones_like(p) * w[None, :, None]).sum(axis=(1, 2))
    np.testing.assert_allclose(a[step], np.sqrt(num / den), rtol=1e-9)


def _passes_case(tp):
  """Two evaluations of one job with different loaders, metrics and aggregators: deterministic per lead time, and an ensemble
  suite reduced over everything (bench.py's configs[4] in miniature)."""
  _, load_e, metrics_e, agg_e = _ensemble_case(tp)
  times_d, load_d, metrics_d, aggs_d = _det_case(tp, REDUCE_SETS['lead_kept'])
  # one chunking for the job: the deterministic case's times (a subset of the ensemble data's: both loaders serve them)
  return times_d, [('deterministic', load_d, metrics_d, aggs_d), ('ensemble', load_e, metrics_e, agg_e)]


def _passes_worker(rank, world_size, out_dir):
  dist = _init(rank, world_size, out_dir)
  import test_pipeline as tp
  from weatherbenchx_amd import pipeline
  try:
    times, passes = _passes_case(tp)
    stats = {}
    out = pipeline.evaluate_passes(times, passes, rank=rank, world_size=world_size, stats=stats)
    assert stats['collectives'] == 1, stats  # ONE sum all-reduce for every pass, aggregator, statistic and variable of the job
    res = {}
    for name, _, metrics, _ in passes:
      for agg_name, st in out[name].items():
        res.update({f'{name}/{agg_name}/{k}': v.values for k, v in st.metric_values(metrics).items()})
    np.savez(os.path.join(out_dir, f'passes{rank}.npz'), **res)
  finally:
    dist.destroy_process_group()


def test_several_passes_share_one_collective(tmp_path, monkeypatch):
  """pipeline.evaluate_passes: the accumulators of every pass live in one Accumulation and cross the ranks in ONE
  all-reduce (the reference's single CombinePerKey over all keys, beam_pipeline.py:509-510); the result of every pass
  equals evaluating that pass alone in one process."""
  _spawn(_passes_worker, 2, tmp_path)
  import fake_device
  import test_pipeline as tp
  from weatherbenchx_amd import pipeline
  fake_device.install(monkeypatch)
  times, passes = _passes_case(tp)
  want = {}
  for name, load, metrics, aggs in passes:
    states = pipeline.evaluate_chunks(times, load, metrics, aggs)
    for agg_name, st in states.items():
      want.update({f'{name}/{agg_name}/{k}': v.values for k, v in st.metric_values(metrics).items()})
  assert len(want) > 6
  for rank in (0, 1):
    got = np.load(os.path.join(tmp_path, f'passes{rank}.npz'))
    assert set(got.files) == set(want)
    for k in want:
      np.testing.assert_allclose(got[k], want[k], rtol=1e-12, equal_nan=True, err_msg=k)


def _stale_worker(rank, world_size, out_dir):
  """Step 2 changes the latitude labels on rank 1 ONLY (same shapes, so the slot layout is unchanged): under the cached plan
  rank 1's arrays would come back with the first step's coordinates and -- had rank 1 re-planned alone -- rank 0 would sit in
  the payload all-reduce while rank 1 sits in the layout exchange.  The stale flag rides in the payload buffer: both re-plan."""
  dist = _init(rank, world_size, out_dir)
  from weatherbenchx_amd import aggregation, distributed, engine
  from weatherbenchx_amd import xarray_lite as xr
  from weatherbenchx_amd.metrics import base as metrics_base
  from weatherbenchx_amd.metrics import deterministic
  try:
    lon = np.arange(12) * 30.0
    metrics = {'mse': deterministic.MSE()}
    agg = aggregation.Aggregator(reduce_dims=['longitude'])
    plan, rows = None, []
    for step in range(3):
      lat = np.linspace(-80, 80, 9) + (1.0 if (step == 1 and rank == 1) else 0.0)
      rng = np.random.default_rng(7 * step + rank)
      mk = lambda: xr.DataArray(rng.normal(size=(9, 12)).astype(np.float32), dims=('latitude', 'longitude'),
                                coords={'latitude': lat, 'longitude': lon})
      acc = engine.Accumulation()
      with engine.accumulate_results(acc):
        state = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, {'v': mk()}, {'v': mk()}))
      state, plan = distributed.resolve_state(state, acc, plan=plan)
      out = state.metric_values(metrics)['mse.v']
      rows.append((out['latitude'].values.copy(), out.values.copy(), plan.collectives))
    with open(os.path.join(out_dir, f'stale{rank}.pkl'), 'wb') as f:
      pickle.dump(rows, f)
  finally:
    dist.destroy_process_group()


def test_changed_labels_under_a_cached_plan_replan_on_every_rank(tmp_path):
  _spawn(_stale_worker, 2, tmp_path)
  r0, r1 = (pickle.load(open(os.path.join(tmp_path, f'stale{r}.pkl'), 'rb')) for r in (0, 1))
  base = np.linspace(-80, 80, 9)
  for step in range(3):
    np.testing.assert_array_equal(r0[step][1], r1[step][1])  # the same sums on both ranks, nobody hung
    assert np.all(np.isfinite(r0[step][1]))
  np.testing.assert_array_equal(r0[0][0], base)
  # step 1: a fresh plan on BOTH ranks (collectives: 1 for step 0, then the flagged round + the re-planned round)
  assert r0[1][2] == r1[1][2] == 3
  # the plan of step 1 holds rank 0's labels first (leaf frames: first rank that has the leaf); what matters is that it is
  # the CURRENT step's frame, and that step 2 (labels back to the base grid) re-plans again instead of replaying step 1
  np.testing.assert_array_equal(r0[2][0], base)
  np.testing.assert_array_equal(r1[2][0], base)


# ---- configs[4] at world size 8: 366 one-init chunks -> 46 / 45 per rank, init_time reduced AND preserved -----------------------
C5_NINIT, C5_NLEAD, C5_NLEV, C5_NLAT, C5_NLON, C5_M = 366, 4, 2, 6, 16, 3


def _config5_accumulator_values(nlead, nlev, nlon, ninit=0):
  """Values of the job's ONE collective (bench.py config5_leg): per (lead, level) sums + weights of the six deterministic
  statistics, per (lead, level, wavenumber) of the two spectra, per lead of the five ensemble statistics; `ninit` > 0: plus the
  slots of the aggregator that preserves init_time -- per (init, lead, level) the six lanes of the deterministic launch it shares
  with the first evaluation (a launch's result buffer is accumulated whole) and their count lane."""
  base = 6 * 2 * nlead * nlev + 2 * 2 * nlead * nlev * (nlon // 2 + 1) + 5 * 2 * nlead + 7 * ninit * nlead * nlev
  # (r5) consecutive chunks deal their ensemble launches to the two launch streams the other way round, and every stream adds
  # into slots of its own (engine.Accumulation.next_chunk): the five ensemble lanes exist once more
  return base + 5 * nlead


def _config5_case():
  """bench.py's configs[4] leg in miniature: three evaluations over ONE chunking of 366 inits (1 init x all leads per chunk), plus
  a fourth aggregator that keeps init_time (its slots exist on one rank each and are concatenated by the same sum)."""
  from weatherbenchx_amd import aggregation, spectra, time_chunks, weighting
  from weatherbenchx_amd import xarray_lite as xr
  from weatherbenchx_amd.metrics import deterministic, probabilistic, wrappers
  lat, lon, level = np.linspace(-75, 75, C5_NLAT), np.arange(C5_NLON) * (360.0 / C5_NLON), np.arange(C5_NLEV)
  lead_time = (np.arange(C5_NLEAD) * 6).astype('timedelta64[h]').astype('timedelta64[ns]')
  init_times = np.datetime64('2020-01-01T00', 'ns') + np.arange(C5_NINIT) * np.timedelta64(24, 'h')
  index_of = {int(t.astype('int64')): i for i, t in enumerate(init_times)}
  ndoy = 369
  rng = np.random.default_rng(1)
  clim = xr.Dataset({'z': xr.DataArray((rng.normal(size=(ndoy, 4, C5_NLEV, C5_NLAT, C5_NLON)) * 3 + 280).astype(np.float32),
                                       dims=('dayofyear', 'hour', 'level', 'latitude', 'longitude'),
                                       coords={'dayofyear': np.arange(1, ndoy + 1), 'hour': np.array([0, 6, 12, 18]), 'level': level,
                                               'latitude': lat, 'longitude': lon})})

  def fields(i):  # chunk i holds the same numbers whichever rank runs it
    g = np.random.default_rng(1000 + i)
    zp = (g.normal(size=(1, C5_NLEAD, C5_NLEV, C5_NLAT, C5_NLON)) + 280).astype(np.float32)
    zt = (g.normal(size=(1, C5_NLEAD, C5_NLEV, C5_NLAT, C5_NLON)) + 280).astype(np.float32)
    tt = (g.normal(size=(1, C5_NLEAD, C5_NLAT, C5_NLON)) + 280).astype(np.float32)
    ep = (tt[:, :, None] + g.normal(size=(1, C5_NLEAD, C5_M, C5_NLAT, C5_NLON))).astype(np.float32)
    return zp, zt, ep, tt

  def load_det(inits, leads):
    i = index_of[int(inits[0].astype('int64'))]
    zp, zt, _, _ = fields(i)
    cs = {'init_time': inits, 'lead_time': lead_time, 'level': level, 'latitude': lat, 'longitude': lon}
    dims = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
    return {'z': xr.DataArray(zp, dims=dims, coords=cs)}, {'z': xr.DataArray(zt, dims=dims, coords=cs)}

  def load_ens(inits, leads):
    i = index_of[int(inits[0].astype('int64'))]
    _, _, ep, tt = fields(i)
    cs = {'init_time': inits, 'lead_time': lead_time, 'latitude': lat, 'longitude': lon}
    return ({'t2m': xr.DataArray(ep, dims=('init_time', 'lead_time', 'number', 'latitude', 'longitude'), coords=cs)},
            {'t2m': xr.DataArray(tt, dims=('init_time', 'lead_time', 'latitude', 'longitude'), coords=cs)})

  det = {'rmse': deterministic.RMSE(), 'mse': deterministic.MSE(), 'mae': deterministic.MAE(), 'bias': deterministic.Bias(),
         'acc': deterministic.ACC(clim), 'prediction_activity': deterministic.PredictionActivity(clim)}
  spec = {'spectrum_p': spectra.ZonalPowerSpectrum('predictions'), 'spectrum_t': spectra.ZonalPowerSpectrum('targets')}
  ens = {'crps': probabilistic.CRPSEnsemble(use_sort=True), 'unbiased_spread_skill': probabilistic.UnbiasedSpreadSkillRatio(),
         'unbiased_mean_rmse': probabilistic.UnbiasedEnsembleMeanRMSE(),
         'mean_rmse': wrappers.WrappedMetric(deterministic.RMSE(), [wrappers.EnsembleMean(which='predictions')])}
  per_init = {'rmse': deterministic.RMSE()}
  area = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  zonal = aggregation.Aggregator(reduce_dims=['init_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
  keep_init = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  passes = [('deterministic', load_det, det, area), ('spectra', load_det, spec, zonal), ('ensemble', load_ens, ens, area),
            ('per_init', load_det, per_init, keep_init)]
  times = time_chunks.TimeChunks(init_times, lead_time, init_time_chunk_size=1)
  return times, passes


def _config5_worker(rank, world_size, out_dir):
  dist = _init(rank, world_size, out_dir)
  from weatherbenchx_amd import distributed, pipeline
  try:
    times, passes = _config5_case()
    mine = distributed.shard_chunks(list(times.iter_with_chunk_offsets()), rank, world_size)
    stats = {}
    out = pipeline.evaluate_passes(times, passes, rank=rank, world_size=world_size, stats=stats)
    res = {}
    for name, _, metrics, _ in passes:
      res.update({f'{name}/{k}': np.asarray(v.values) for k, v in out[name][None].metric_values(metrics).items()})
    res['_init_time_of_per_init'] = out['per_init'][None].metric_values(passes[3][2])['rmse.z']['init_time'].values.astype('int64')
    np.savez(os.path.join(out_dir, f'c5_{rank}.npz'), _chunks=len(mine), _collectives=stats['collectives'],
             _values=stats['accumulator_values'], **res)
  finally:
    dist.destroy_process_group()


def test_config5_sharding_over_eight_ranks(tmp_path, monkeypatch):
  """the configs[4] partitioning at world size 8 with a ragged split -- 366 chunks -> 46 on six ranks, 45 on two
  (time_chunks.py:177-202: chunk i -> rank i mod 8) -- init_time both reduced (sums add up: CombinePerKey(CombiningSum()),
  beam_pipeline.py:509-510) and preserved (every slot lives on one rank; the same sum concatenates them): ONE collective per job
  on every rank, a buffer of exactly the job's accumulator values, and every rank holds the single-process result."""
  world = 8
  _spawn(_config5_worker, world, tmp_path)
  import fake_device
  from weatherbenchx_amd import pipeline
  fake_device.install(monkeypatch)
  times, passes = _config5_case()
  want = {}
  for name, load, metrics, agg in passes:
    st = pipeline.evaluate_chunks(times, load, metrics, agg)[None]
    vals = st.metric_values(metrics)
    want.update({f'{name}/{k}': np.asarray(v.values) for k, v in vals.items()})
    if name == 'per_init':
      assert vals['rmse.z'].shape[:1] == (C5_NINIT,) and vals['rmse.z'].dims[0] == 'init_time'
      want_init = vals['rmse.z']['init_time'].values.astype('int64')
  expect_values = _config5_accumulator_values(C5_NLEAD, C5_NLEV, C5_NLON, C5_NINIT)
  counts = []
  for rank in range(world):
    got = np.load(os.path.join(tmp_path, f'c5_{rank}.npz'))
    counts.append(int(got['_chunks']))
    assert int(got['_collectives']) == 1, (rank, got['_collectives'])
    assert int(got['_values']) == expect_values, (rank, int(got['_values']), expect_values)
    np.testing.assert_array_equal(got['_init_time_of_per_init'], want_init)  # concatenated in label order on every rank
    assert set(want) <= set(got.files)
    for k in want:
      np.testing.assert_allclose(got[k], want[k], rtol=1e-11, atol=1e-300, equal_nan=True, err_msg=f'rank {rank} {k}')
  assert sorted(counts, reverse=True) == [46] * 6 + [45] * 2 and sum(counts) == C5_NINIT
  # the same formula at BASELINE.json's configs[4] size: 20 leads x 37 levels x 721 wavenumbers -> the 2.14 M doubles (17 MB)
  # bench.py's config5 leg reports as `accumulator_values`
  assert _config5_accumulator_values(20, 37, 1440) == 2143240 + 100


def _slab_case():
  """ACC / activity / RMSE against a 60-day x 4-hour climatology behind an 11-slab pool: 12 daily inits x 5 six-hourly leads."""
  import test_climatology_cache as tcc
  from weatherbenchx_amd import climatology_cache, time_chunks
  from weatherbenchx_amd import xarray_lite as xr
  full, da = tcc._climatology(ndoy=60)
  clim = climatology_cache.cached(xr.Dataset({'z': da}), slots=11)
  inits, lead, p, t, load = tcc._job(12, 5, start='2020-01-03T00')
  times = time_chunks.TimeChunks(inits, lead, init_time_chunk_size=1)
  return tcc, clim, full, inits, lead, p, t, load, times


def _slab_worker(rank, world_size, out_dir, keep_init):
  dist = _init(rank, world_size, out_dir)
  from weatherbenchx_amd import aggregation, climatology_cache, distributed, pipeline, weighting
  try:
    tcc, clim, full, inits, lead, p, t, load, times = _slab_case()
    metrics = tcc._metrics(clim)
    agg = tcc._area() if not keep_init else aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
    stats = {}
    out = pipeline.evaluate_passes(times, [('', load, metrics, agg)], rank=rank, world_size=world_size, stats=stats)
    state = out[''][None]
    cache = climatology_cache.cache_for(clim['z'])
    mine = distributed.shard_chunks(list(times.iter_with_chunk_offsets()), rank, world_size, 'block')
    vals = state.metric_values(metrics)
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), uploads=cache.stats['uploads'], nmine=len(mine), collectives=stats['collectives'],
             first=int(mine[0][0].init_time) if mine else -1, **{k: np.asarray(v.transpose(*[d for d in ('init_time', 'lead_time', 'level') if d in v.dims]).values) for k, v in vals.items()})
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('world,keep_init', [(2, False), (3, True)])
def test_a_job_with_a_slab_pool_is_sharded_in_runs_of_chunks(tmp_path, world, keep_init):
  """A climatology behind a slab pool: `evaluate_passes` deals CONTIGUOUS runs of chunks to the ranks by itself (consecutive
  inits share 4 of their 5 slabs, inits `world` days apart share none), every rank uploads only its run's slabs, ONE
  collective, every rank ends with the float64 oracle's numbers -- with init_time reduced and with init_time kept."""
  _spawn(_slab_worker, world, tmp_path, keep_init)
  tcc, clim, full, inits, lead, p, t, load, times = _slab_case()
  n = len(inits)
  for rank in range(world):
    got = np.load(os.path.join(tmp_path, f'rank{rank}.npz'))
    lo, hi = rank * n // world, (rank + 1) * n // world
    assert int(got['nmine']) == hi - lo and int(got['first']) == lo and int(got['collectives']) == 1
    assert int(got['uploads']) == (hi - lo) * 4 + 1  # its run's days x 4 hours + 00 of the day after: no slab twice, none of the others'
    if not keep_init:
      want = tcc._oracle_acc(p, t, full, inits, lead)
      for name, key in (('acc', 'acc.z'), ('activity', 'activity.z'), ('rmse', 'rmse.z')):
        np.testing.assert_allclose(got[key], want[name], rtol=1e-6, err_msg=name)
    else:
      for i in range(n):
        want = tcc._oracle_acc(p[i:i + 1], t[i:i + 1], full, inits[i:i + 1], lead)
        for name, key in (('acc', 'acc.z'), ('activity', 'activity.z'), ('rmse', 'rmse.z')):
          np.testing.assert_allclose(got[key][i], want[name], rtol=1e-6, err_msg=f'{name} init {i}')
