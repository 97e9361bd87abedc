"""Chunked == unchunked evaluation, restating weatherbenchX/beam_pipeline_test.py:82-284 (1x1 chunks vs one
chunk for five reduce_dims sets, two named aggregators), weatherbenchX/time_chunks_test.py:20-57 and
weatherbenchX/xarray_tree_test.py against the Beam-free chunk loop."""
import numpy as np
import pytest

import mock_data
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import pipeline
from weatherbenchx_amd import time_chunks
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd import xarray_tree
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.metrics import deterministic


def _datasets():
  predictions = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-03T00',
                                               lead_start_days=0, lead_stop_days=1, random=True, seed=0)
  targets = mock_data.mock_target_data(time_start='2020-01-01T00', time_stop='2020-01-05T00', random=True, seed=1)
  return predictions, targets


def _loader(predictions, targets):
  """What PredictionsFromXarray / TargetsFromXarray hand to the chunk loop (xarray_loaders.py:185-256):
  predictions[init_time, lead_time, ...]; targets selected at valid_time = init + lead with the same dims."""
  pred = predictions.rename({'time': 'init_time', 'prediction_timedelta': 'lead_time'})

  def load(init_times, lead_times):
    p = {k: v.sel(init_time=init_times, lead_time=lead_times).transpose('init_time', 'lead_time', ...)
         for k, v in pred.items()}
    vt = init_times[:, None] + lead_times[None, :]
    t = {}
    for k, v in targets.items():
      sel = v.sel(time=xr.DataArray(vt, dims=('init_time', 'lead_time')))
      t[k] = sel.drop_vars('time').assign_coords(init_time=init_times, lead_time=lead_times)
    return p, t
  return load


@pytest.mark.parametrize('reduce_dims', [['init_time', 'latitude', 'longitude'], ['init_time'], ['lead_time'],
                                         ['latitude', 'longitude'], []])
def test_chunked_equals_single_chunk(backend, reduce_dims):
  predictions, targets = _datasets()
  init_times = predictions['geopotential']['time'].values
  lead_times = predictions['geopotential']['prediction_timedelta'].values
  times = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=1, lead_time_chunk_size=1)
  assert len(times) == 4
  load = _loader(predictions, targets)
  metrics = {'rmse': deterministic.RMSE(), 'mse': deterministic.MSE()}
  agg = aggregation.Aggregator(reduce_dims=reduce_dims)
  p, t = load(init_times, lead_times)
  direct = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, p, t))
  direct_metrics = direct.metric_values(metrics)
  state = pipeline.evaluate_chunks(times, load, metrics, agg)[None]
  chunked_metrics = state.metric_values(metrics)
  assert set(direct_metrics) == set(chunked_metrics)
  for k in direct_metrics:
    xr.assert_allclose(direct_metrics[k], chunked_metrics[k], atol=1e-5, check_dim_order=False)
  xarray_tree.map_structure(lambda a, b: xr.assert_allclose(a, b, atol=1e-5, check_dim_order=False),
                            (direct.sum_weighted_statistics, direct.sum_weights),
                            (state.sum_weighted_statistics, state.sum_weights))


@pytest.mark.parametrize('reduce_dims', [['init_time', 'latitude'], ['init_time', 'lead_time', 'latitude'], ['latitude']])
def test_chunked_spectra_equal_single_chunk(backend, reduce_dims):
  """Zonal spectra through the chunk loop: the power sums are accumulated in HBM, the (data-independent) sums of row
  weights are one cached array met once per chunk -- both must add up to the single-chunk state."""
  from weatherbenchx_amd import spectra, weighting  # pylint: disable=g-import-not-at-top
  predictions, targets = _datasets()
  keep = lambda ds: xr.Dataset({'2m_temperature': ds['2m_temperature'].astype(np.float32)})
  predictions, targets = keep(predictions), keep(targets)
  init_times = predictions['2m_temperature']['time'].values
  lead_times = predictions['2m_temperature']['prediction_timedelta'].values
  times = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=1, lead_time_chunk_size=1)
  load = _loader(predictions, targets)
  metrics = {'spectrum_p': spectra.ZonalPowerSpectrum('predictions'), 'spectrum_t': spectra.ZonalPowerSpectrum('targets')}
  agg = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=[weighting.GridAreaWeighting()])
  p, t = load(init_times, lead_times)
  direct = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, p, t))
  for _ in range(2):  # (the second loop meets the cached weight sums of the first)
    state = pipeline.evaluate_chunks(times, load, metrics, agg)[None]
    xarray_tree.map_structure(lambda a, b: xr.assert_allclose(a, b, rtol=1e-9, atol=1e-9, check_dim_order=False),
                              (direct.sum_weighted_statistics, direct.sum_weights),
                              (state.sum_weighted_statistics, state.sum_weights))
    want, got = direct.metric_values(metrics), state.metric_values(metrics)
    for k in want:
      xr.assert_allclose(want[k], got[k], rtol=1e-9, check_dim_order=False)


def test_host_side_constants_are_counted_per_frame():
  """`engine.Accumulation.capture` counts how often it met one cached read-only array under a path instead of adding it up
  chunk by chunk -- but only for the same frame: the same numbers under other coordinates are another term (outer join), a
  writable array is summed as before."""
  from weatherbenchx_amd import engine  # pylint: disable=g-import-not-at-top
  const = np.array(np.arange(6, dtype=np.float64).reshape(2, 3))  # (owns its memory, like the cached weight sums)
  const.flags.writeable = False
  mk = lambda lead: xr.DataArray(const, dims=('lead_time', 'k'), coords={'lead_time': np.asarray(lead), 'k': np.arange(3)})
  acc = engine.Accumulation()
  for _ in range(5):
    acc.capture('p', mk([0, 6]))
  acc.capture('p', mk([12, 18]))  # same array, other labels
  acc.capture('p', xr.DataArray(np.ones((2, 3)), dims=('lead_time', 'k'), coords={'lead_time': np.asarray([0, 6]), 'k': np.arange(3)}))
  got = acc.host['p']
  assert list(got['lead_time'].values) == [0, 6, 12, 18]
  np.testing.assert_array_equal(got.sel(lead_time=[0, 6]).values, 5 * const + 1)
  np.testing.assert_array_equal(got.sel(lead_time=[12, 18]).values, const)
  assert len(acc._host_const['p']) == 2 and acc._host_const['p'][0][1] == 5.0  # pylint: disable=protected-access


def test_multiple_named_aggregators(backend):
  predictions, targets = _datasets()
  init_times = predictions['geopotential']['time'].values
  lead_times = predictions['geopotential']['prediction_timedelta'].values
  times = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=1, lead_time_chunk_size=1)
  load = _loader(predictions, targets)
  metrics = {'rmse': deterministic.RMSE()}
  aggs = {'init_time,latitude,longitude': aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude']),
          'init_time': aggregation.Aggregator(reduce_dims=['init_time'])}
  states = pipeline.evaluate_chunks(times, load, metrics, aggs)
  p, t = load(init_times, lead_times)
  stats = metrics_base.compute_unique_statistics_for_all_metrics(metrics, p, t)
  for name, agg in aggs.items():
    want = agg.aggregate_statistics(stats).metric_values(metrics)
    got = states[name].metric_values(metrics)
    for k in want:
      xr.assert_allclose(want[k], got[k], atol=1e-5, check_dim_order=False)
  assert pipeline.resolve_out_path('/tmp/metrics.nc', 'init_time') == '/tmp/metrics_init_time.nc'
  assert pipeline.resolve_out_path('/tmp/metrics.nc', None) == '/tmp/metrics.nc'


def test_deferred_results_equal_synchronous_results(backend):
  """engine.deferred_results(): aggregate_statistics returns before the sums have arrived; the AggregationState
  waits on first use, states combine (CombiningSum, beam_pipeline.py:509-510) and pickle like synchronous ones."""
  import pickle
  from weatherbenchx_amd import binning, engine, weighting
  predictions, targets = _datasets()
  init_times = predictions['geopotential']['time'].values
  lead_times = predictions['geopotential']['prediction_timedelta'].values
  load = _loader(predictions, targets)
  # WindVectorRMSE is a LinearCombination of two fused reductions: its terms are combined on the host, so the
  # aggregator has to resolve them itself
  metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE(),
             'wind': deterministic.WindVectorRMSE('geopotential', 'geopotential', 'wind')}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'],
                               weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions({'global': ((-90, 90), (0, 360)), 'nh': ((20, 90), (0, 360))})])
  chunks = [load(init_times[i:i + 1], lead_times) for i in range(len(init_times))]
  sync_states = [agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, p, t))
                 for p, t in chunks]
  want = aggregation.AggregationState.sum(sync_states).metric_values(metrics)
  fresh = [load(init_times[i:i + 1], lead_times) for i in range(len(init_times))]  # new objects: nothing cached on them
  with engine.deferred_results():
    assert engine.deferred_active() is not None
    states = [agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, p, t))
              for p, t in fresh]  # every chunk is launched before any result is looked at
    clone = pickle.loads(pickle.dumps(states[0]))  # pickling waits
    total = aggregation.AggregationState.sum(states)
    got = total.metric_values(metrics)
  assert engine.deferred_active() is None
  assert not engine._stream_ring  # deterministic reductions stay on the default context's stream
  for k in want:
    xr.assert_allclose(got[k], want[k], rtol=1e-12, atol=0, check_dim_order=False)
  xarray_tree.map_structure(lambda a, b: xr.assert_allclose(a, b, rtol=1e-12, atol=0),
                            clone.sum_weighted_statistics, sync_states[0].sum_weighted_statistics)
  # single-variable entry points carry their own fence
  with engine.deferred_results():
    stats = metrics_base.compute_unique_statistics_for_all_metrics({'mse': deterministic.MSE()}, *chunks[0])
    one = agg.aggregate_stat_var(stats['SquaredError']['geopotential'])
    arr = agg.aggregation_fn(stats['SquaredError']['geopotential'])  # a bare DataArray: resolved before returning
    np.testing.assert_allclose(arr.values, one.wait().sum_weighted_statistics.values, rtol=1e-12)


def test_chunk_feeder_prefetch_equals_serial_loading(backend):
  """prefetch=1: a feeder thread loads and stages chunk k+1 (own context / copy stream) while chunk k is aggregated;
  results are identical, loader exceptions surface in the caller."""
  predictions, targets = _datasets()
  init_times = predictions['geopotential']['time'].values
  lead_times = predictions['geopotential']['prediction_timedelta'].values
  times = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=1, lead_time_chunk_size=1)
  load = _loader(predictions, targets)
  metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE()}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'])
  serial = pipeline.evaluate_chunks(times, load, metrics, agg)[None].metric_values(metrics)
  calls = []

  def counting_load(i, l):
    calls.append(__import__('threading').current_thread().name)
    return load(i, l)
  fed = pipeline.evaluate_chunks(times, counting_load, metrics, agg, prefetch=1)[None].metric_values(metrics)
  assert len(calls) == len(times) and set(calls) == {'wbx-chunk-feeder'}
  for k in serial:
    xr.assert_allclose(fed[k], serial[k], rtol=1e-12, atol=0)

  def failing_load(i, l):
    if len(calls) >= len(times) + 2:
      raise OSError('disk on fire')
    calls.append('x')
    return load(i, l)
  with pytest.raises(OSError, match='disk on fire'):
    pipeline.evaluate_chunks(times, failing_load, metrics, agg, prefetch=2)


def test_time_chunks_lengths_and_offsets():
  init_times = np.arange('2020-01-01T00', '2020-01-02T00', np.timedelta64(6, 'h'), dtype='datetime64[ns]')
  lead_times = np.arange(0, 18, 6, dtype='timedelta64[h]')
  assert len(list(time_chunks.TimeChunks(init_times, lead_times))) == 1
  chunks = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=2, lead_time_chunk_size=2)
  assert len(list(chunks)) == 4 and len(chunks) == 4
  offs = [(o.init_time, o.lead_time) for o, _ in chunks.iter_with_chunk_offsets()]
  assert offs == [(0, 0), (0, 2), (2, 0), (2, 2)]
  assert chunks[1][1].tolist() == lead_times[2:].astype('timedelta64[ns]').tolist()
  sl = slice(np.timedelta64(0, 'h'), np.timedelta64(6, 'h'))
  assert len(list(time_chunks.TimeChunks(init_times, sl, init_time_chunk_size=2))) == 2
  with pytest.raises(ValueError):
    time_chunks.TimeChunks(init_times, sl, lead_time_chunk_size=2)
  with pytest.raises(ValueError):
    time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=-1)
  with pytest.raises(IndexError):
    chunks[4]  # pylint: disable=pointless-statement


def test_xarray_tree_map_structure():
  a = xr.DataArray(np.arange(3.0), dims=['x'], coords={'x': [0, 1, 2]})
  ds = xr.Dataset({'a': a, 'b': a + 1})
  doubled = xarray_tree.map_structure(lambda v: v * 2, ds)
  assert isinstance(doubled, xr.Dataset) and doubled['b'].values.tolist() == [2, 4, 6]
  dropped = xarray_tree.map_structure(lambda v: None if v.name == 'a' else v, ds)
  assert list(dropped) == ['b']
  nested = xarray_tree.map_structure(lambda x, y: x + y, {'k': [a, a]}, {'k': [a, a]})
  assert nested['k'][1].values.tolist() == [0, 2, 4]
  as_dict = xarray_tree.map_structure(lambda v: 3, ds)
  assert as_dict == {'a': 3, 'b': 3}
  with pytest.raises(TypeError):
    xarray_tree.map_structure(3, ds)
  with pytest.raises(ValueError):
    xarray_tree.map_structure(lambda x: x)


def test_deferred_ensemble_reductions_alternate_streams(backend):
  """Under deferred_results() ensemble / indicator reductions are dealt to two contexts (HIP streams) in turn, so that
  one kernel's tail overlaps the next launch; the numbers and the fence semantics are those of the one-stream path."""
  from weatherbenchx_amd import engine, weighting
  from weatherbenchx_amd.metrics import probabilistic
  rng = np.random.default_rng(21)
  lat, lon = np.linspace(-87.5, 87.5, 36), np.arange(72) * 5.0
  coords = {'latitude': lat, 'longitude': lon}

  def chunk(seed):
    r = np.random.default_rng(seed)
    tv = r.normal(size=(3, 36, 72)).astype(np.float32)
    pv = (tv[:, None] + r.normal(size=(3, 5, 36, 72))).astype(np.float32)
    names = ('a', 'b', 'c')
    p = {n: xr.DataArray(pv + i, dims=('lead_time', 'number', 'latitude', 'longitude'), coords=coords) for i, n in enumerate(names)}
    t = {n: xr.DataArray(tv + i, dims=('lead_time', 'latitude', 'longitude'), coords=coords) for i, n in enumerate(names)}
    return p, t
  del rng
  metrics = {'crps': probabilistic.CRPSEnsemble(use_sort=True), 'rank': probabilistic.RankHistogram()}
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  launch = lambda p, t: agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, p, t))
  want = [launch(*chunk(s)).metric_values(metrics) for s in (1, 2, 3)]
  with engine.deferred_results():
    states = [launch(*chunk(s)) for s in (1, 2, 3)]  # 3 chunks x 3 variables x 2 statistic families, nothing read yet
    assert len(engine._stream_ring) == 2 and engine._stream_ring[0] is not engine._stream_ring[1]
    got = [s.metric_values(metrics) for s in states]
  for g, w in zip(got, want):
    for k in w:
      xr.assert_allclose(g[k], w[k], rtol=1e-12, atol=1e-15)


def test_ensemble_chunk_loop_with_feeder_and_two_launch_streams(backend):
  """evaluate_chunks on an ensemble workload with prefetch=1: the feeder's copy stream, the deferred read-back and the
  two alternating launch streams work together; chunked == one chunk == serial loading."""
  from weatherbenchx_amd import engine, weighting
  from weatherbenchx_amd.metrics import probabilistic
  predictions = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-04T00', lead_start_days=0,
                                               lead_stop_days=1, random=True, seed=7, ensemble_size=5)
  targets = mock_data.mock_target_data(time_start='2020-01-01T00', time_stop='2020-01-06T00', random=True, seed=8)
  init_times = predictions['geopotential']['time'].values
  lead_times = predictions['geopotential']['prediction_timedelta'].values
  load = _loader(predictions, targets)
  metrics = {'crps': probabilistic.CRPSEnsemble(ensemble_dim='realization', use_sort=True),
             'ssr': probabilistic.UnbiasedSpreadSkillRatio(ensemble_dim='realization'),
             'rank': probabilistic.RankHistogram(ensemble_dim='realization')}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  whole = time_chunks.TimeChunks(init_times, lead_times)
  pieces = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=1, lead_time_chunk_size=1)
  want = pipeline.evaluate_chunks(whole, load, metrics, agg)[None].metric_values(metrics)
  serial = pipeline.evaluate_chunks(pieces, load, metrics, agg)[None].metric_values(metrics)
  fed = pipeline.evaluate_chunks(pieces, load, metrics, agg, prefetch=1)[None].metric_values(metrics)
  assert len(engine._stream_ring) == 2  # the chunk loop ran under deferred_results(): ensemble launches alternated
  for k in want:
    xr.assert_allclose(serial[k], want[k], rtol=1e-9, atol=1e-12, check_dim_order=False)
    xr.assert_allclose(fed[k], serial[k], rtol=1e-12, atol=0, check_dim_order=False)


def test_slab_detection_for_the_latitude_fastest_fused_sweep():
  """engine._fused_slab_rows: a plan whose x is the strided longitude qualifies for wbx_det_spectrum_slabs only when its keys
  are whole slabs of ADJACENT rows in every input and the climatology slice does not change inside a slab (host logic, no
  device)."""
  from weatherbenchx_amd import engine, planner
  nlead, nlev, nlon, nlat = 2, 3, 1440, 11
  dims = ('init_time', 'lead_time', 'level', 'longitude', 'latitude')
  sizes = dict(zip(dims, (1, nlead, nlev, nlon, nlat)))

  def layout(order, shape):
    strides, acc = {}, 1
    for d, n in zip(reversed(order), reversed(shape)):
      strides[d] = acc
      acc *= n
    return planner.InputLayout(strides=strides, itemsize=4, base_alignment=256)
  lay = layout(dims, [sizes[d] for d in dims])
  clay = layout(('slot', 'level', 'longitude', 'latitude'), [4, nlev, nlon, nlat])
  table = (np.array([[3, 1]]) * clay.strides['slot']).astype(np.int64)
  gather = planner.GatherSpec(dims=('init_time', 'lead_time'), table=table)
  entry = {'row_dims': ('init_time', 'lead_time', 'level', 'latitude'), 'row_shape': (1, nlead, nlev, nlat), 'dev': {}}
  plan = planner.build_s1_plan(dims, sizes, [lay, lay, clay, None], ['init_time', 'latitude', 'longitude'], wdep_dims=['latitude'],
                               gather=gather, force_x_dim='longitude', allow_vec4=False)
  assert plan.nx == nlon and plan.xstride[0] == nlat and plan.key_dims == ('lead_time', 'level', 'latitude')
  assert engine._fused_slab_rows(plan, entry, 3) == nlat
  # targets stored longitude-fastest: their rows of a slab are 1440 elements apart -> no slabs
  tlay = layout(('init_time', 'lead_time', 'level', 'latitude', 'longitude'), [1, nlead, nlev, nlat, nlon])
  plan2 = planner.build_s1_plan(dims, sizes, [lay, tlay, clay, None], ['init_time', 'latitude', 'longitude'], wdep_dims=['latitude'],
                                gather=gather, force_x_dim='longitude', allow_vec4=False)
  entry2 = dict(entry, dev={})
  assert engine._fused_slab_rows(plan2, entry2, 3) is None
  # a reduction that keeps longitude rows apart from the spectra's rows (level summed inside stage 1): not the fields' rows
  plan3 = planner.build_s1_plan(dims, sizes, [lay, lay, clay, None], ['init_time', 'level', 'latitude', 'longitude'], wdep_dims=['latitude'],
                                gather=gather, force_x_dim='longitude', allow_vec4=False)
  assert engine._fused_slab_rows(plan3, dict(entry, dev={}), 3) is None


def test_accumulator_slot_handed_to_a_launch(monkeypatch):
  """engine.Accumulation.slot_pointer / accumulate(in_place=): from the second chunk on a launch that understands
  WBX_BINNED_ACCUMULATE gets the address of its result's slot and adds into it itself; the bookkeeping (ordinal, layout views,
  `multi`) is what the scratch + wbx_acc_add path does, no add is enqueued, and an address that is not the slot's is refused."""
  from weatherbenchx_amd import engine

  class Buf:
    def __init__(self, ptr, nbytes):
      self.ptr, self.nbytes = ptr, nbytes

  class Ctx:
    def __init__(self):
      self.next = 0x7f0000000000

    def alloc(self, nbytes):
      b = Buf(self.next, nbytes)
      self.next += (nbytes + 255) // 256 * 256
      return b
  adds = []
  monkeypatch.setattr(engine, '_acc_add', lambda ctx, buf, off, src, n, first: adds.append((buf.ptr + 8 * off, src, n, first)))
  ctx, acc = Ctx(), engine.Accumulation()
  acc.set_label('job')
  assert acc.slot_pointer(ctx, 12) is None                      # first chunk: the slot does not exist yet
  v0 = acc.accumulate(ctx, 0x1000, (3, 4))
  v1 = acc.accumulate(ctx, 0x2000, (5,))
  assert adds == [(acc.blocks[0].dev.ptr, 0x1000, 12, True), (acc.blocks[0].dev.ptr + 96, 0x2000, 5, True)] and not acc.multi
  acc.set_label('job')                                          # the next chunk under the same label
  p0 = acc.slot_pointer(ctx, 12)
  assert p0 == acc.blocks[0].dev.ptr and acc.slot_pointer(ctx, 11) is None and acc.slot_pointer(Ctx(), 12) is None
  w0 = acc.accumulate(ctx, None, (3, 4), in_place=p0)
  assert len(adds) == 2 and acc.multi                           # nothing enqueued: the kernel has added already
  assert w0.shape == (3, 4) and acc.locate(w0) == acc.locate(v0)
  p1 = acc.slot_pointer(ctx, 5)
  assert p1 == acc.blocks[0].dev.ptr + 96
  with pytest.raises(RuntimeError, match='not this result'):
    acc.accumulate(ctx, None, (5,), in_place=p1 + 8)
