"""One evaluation that strings the round-5 additions together on the host-side plan interpreter (tests/fake_device.py; this file was
written after the round's GPU minutes were spent, so it does not ask for the device): in-memory loaders with an interpolation and a
NaN mask, chunked over init / lead times, contingency / FSS / energy-score / REV metrics next to RMSE, two named aggregators (area
mean; latitude bands), then a paired t-test on the per-init accumulators -- against the same numbers from whole-array NumPy."""
import numpy as np
import pytest

from oracle import wbx_oracle as O
from tests import mock_data
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import binning
from weatherbenchx_amd import engine
from weatherbenchx_amd import interpolations
from weatherbenchx_amd import loaders
from weatherbenchx_amd import pipeline
from weatherbenchx_amd import time_chunks
from weatherbenchx_amd import weighting
from weatherbenchx_amd.data_loaders import xarray_loaders
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.metrics import categorical
from weatherbenchx_amd.metrics import deterministic
from weatherbenchx_amd.metrics import spatial
from weatherbenchx_amd.metrics import wrappers
from weatherbenchx_amd.statistical_inference import t_test


@pytest.fixture
def emulated(monkeypatch):
  import fake_device  # pylint: disable=g-import-not-at-top
  engine.clear_caches()
  fake_device.install(monkeypatch)
  yield
  engine.clear_caches()


def test_loaders_interpolation_scores_and_inference_in_one_evaluation(emulated):
  del emulated
  target = mock_data.mock_target_data(time_start='2020-01-01T00', time_stop='2020-01-20T00', variables_3d=[], random=True, seed=1,
                                      spatial_resolution_in_degrees=10.0)
  coarse = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-13T00', variables_3d=[], random=True, seed=2,
                                          lead_stop_days=2, spatial_resolution_in_degrees=20.0)
  target['2m_temperature'].values[5, 3, 7] = np.nan
  lt = xarray_loaders.TargetsFromXarray(ds=target, add_nan_mask=True)
  regrid = interpolations.InterpolateToFixedCoords('linear', {'latitude': target['2m_temperature']['latitude'].values,
                                                             'longitude': target['2m_temperature']['longitude'].values}, wrap_longitude=True)
  lp = xarray_loaders.PredictionsFromXarray(ds=coarse, interpolation=regrid)
  init_times = np.arange('2020-01-01T00', '2020-01-13T00', np.timedelta64(24, 'h'), dtype='datetime64[ns]')
  lead_times = np.arange(3, dtype='timedelta64[D]').astype('timedelta64[ns]')
  binarize = wrappers.ContinuousToBinary('both', [0.5], 'threshold')
  metrics = {'rmse': deterministic.RMSE(), 'csi': wrappers.WrappedMetric(categorical.CSI(), [binarize]),
             'fss': wrappers.WrappedMetric(spatial.FSS(3, wrap_longitude=True), [binarize])}
  aggregators = {'area': aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()], masked=True),
                 'bands': aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], bin_by=[binning.LatitudeBins(60)],
                                                 weigh_by=[weighting.GridAreaWeighting()], masked=True)}
  tc = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=5, lead_time_chunk_size=2)
  states = pipeline.evaluate_chunks(tc, loaders.load_chunk_fn(lp, lt), metrics, aggregators, prefetch=0)
  # ---- the same from whole arrays ---------------------------------------------------------------------------------------------
  t_all = target['2m_temperature'].values                                  # [time, lat, lon]
  p_src = coarse['2m_temperature']                                         # [lead, time, lat20, lon20]
  p_all = np.asarray(regrid.interpolate_data_array(p_src).transpose('time', 'prediction_timedelta', 'latitude', 'longitude').values)
  t_sel = t_all[np.arange(12)[:, None] + np.arange(3)[None, :]]            # [init, lead, lat, lon]
  valid = ~np.isnan(t_sel)
  lat = target['2m_temperature']['latitude'].values
  w = np.broadcast_to(O.grid_area_weights(lat)[None, None, :, None], t_sel.shape)
  wmean = lambda x, axes: (np.where(valid, x, 0) * w).sum(axis=axes) / (valid * w).sum(axis=axes)
  area = states['area'].metric_values(metrics)
  np.testing.assert_allclose(np.asarray(area['rmse.2m_temperature'].transpose('init_time', 'lead_time').values),
                             np.sqrt(wmean((p_all - t_sel) ** 2, (2, 3))), rtol=2e-6)
  pb, tb = (p_all > 0.5), (t_sel > 0.5)
  tp, fp, fn = (wmean((a & b).astype(float), (2, 3)) for a, b in ((pb, tb), (pb, ~tb), (~pb, tb)))
  got_csi = area['csi.2m_temperature'].isel(threshold=0).transpose('init_time', 'lead_time')
  np.testing.assert_allclose(np.asarray(got_csi.values), tp / (tp + fp + fn), rtol=2e-6)
  bands = states['bands'].metric_values(metrics)
  assert 'latitude_bins' in bands['rmse.2m_temperature'].dims
  for k, lo in enumerate((-90, -30, 30)):
    inside = ((lat >= lo) & (lat <= lo + 60))[None, None, :, None] & valid
    ww = inside * w
    want = np.sqrt((np.where(inside, (p_all - t_sel) ** 2, 0) * w).sum(axis=(0, 2, 3)) / ww.sum(axis=(0, 2, 3)))
    np.testing.assert_allclose(np.asarray(bands['rmse.2m_temperature'].isel(latitude_bins=k).values), want, rtol=2e-6)
  fss = area['fss.2m_temperature']
  assert {'init_time', 'lead_time'} <= set(fss.dims) and np.isfinite(np.asarray(fss.values)).any()
  # ---- inference on the per-init accumulators ----------------------------
  rmse_only = {'rmse': deterministic.RMSE()}
  inference = t_test.GeerAR2Corrected(metrics=rmse_only, aggregated_statistics=states['area'], experimental_unit_dim='init_time')
  point = inference.point_estimates()['rmse']['2m_temperature']
  np.testing.assert_allclose(np.asarray(point.values), np.sqrt(_pooled(p_all, t_sel, valid, w)), rtol=2e-6)
  lower, upper = inference.confidence_intervals(0.05)
  assert (np.asarray(lower['rmse']['2m_temperature'].values) < np.asarray(point.values)).all()
  assert (np.asarray(point.values) < np.asarray(upper['rmse']['2m_temperature'].values)).all()


def _pooled(p, t, valid, w):
  """mean over init times of the per-init accumulators: sum over inits of sum(w se) / sum over inits of sum(w), per lead time."""
  se = np.where(valid, (p - t) ** 2, 0) * w
  return se.sum(axis=(0, 2, 3)) / (valid * w).sum(axis=(0, 2, 3))


def test_define_pipeline_writes_what_the_chunk_loop_computes(emulated, tmp_path):
  """beam_pipeline_test.py:82-284 in spirit: chunked == one chunk, files per named aggregator, the targets handed to the predictions
  loader as the interpolation reference."""
  del emulated
  from weatherbenchx_amd import beam_pipeline  # pylint: disable=g-import-not-at-top
  from weatherbenchx_amd import io as wio  # pylint: disable=g-import-not-at-top
  target = mock_data.mock_target_data(time_start='2020-01-01T00', time_stop='2020-01-08T00', variables_3d=[], random=True, seed=3,
                                      spatial_resolution_in_degrees=10.0)
  coarse = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-05T00', variables_3d=[], random=True, seed=4,
                                          lead_stop_days=1, spatial_resolution_in_degrees=30.0)
  calls = []
  lt = xarray_loaders.TargetsFromXarray(ds=target)
  lp = xarray_loaders.PredictionsFromXarray(ds=coarse, interpolation=interpolations.InterpolateToReferenceCoords(
      'linear', dims=['latitude', 'longitude'], wrap_longitude=True))
  init_times = np.arange('2020-01-01T00', '2020-01-05T00', np.timedelta64(24, 'h'), dtype='datetime64[ns]')
  lead_times = np.arange(2, dtype='timedelta64[D]').astype('timedelta64[ns]')
  metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE()}
  aggregators = {'global': aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()]),
                 'per_init': aggregation.Aggregator(reduce_dims=['latitude', 'longitude'])}
  out, state_out = str(tmp_path / 'metrics.nc'), {'global': str(tmp_path / 'g.nc'), 'per_init': str(tmp_path / 'p.nc')}
  chunked = beam_pipeline.define_pipeline(None, time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=1, lead_time_chunk_size=1),
                                          lp, lt, metrics, aggregators, out_path=out, aggregation_state_out_path=state_out,
                                          setup_fn=lambda: calls.append('setup'))
  assert calls == ['setup']
  whole = beam_pipeline.define_pipeline(None, time_chunks.TimeChunks(init_times, lead_times), lp, lt, metrics, aggregators,
                                        out_path={'global': str(tmp_path / 'w_g.nc'), 'per_init': str(tmp_path / 'w_p.nc')})
  for name in aggregators:
    a, b = chunked[name].metric_values(metrics), whole[name].metric_values(metrics)
    for key in a:
      np.testing.assert_allclose(np.asarray(a[key].transpose(*b[key].dims).values), np.asarray(b[key].values), atol=1e-5)   # beam_pipeline_test.py:150-155
    written = wio.open_dataset(str(tmp_path / f'metrics_{name}.nc'))
    for key in a:
      np.testing.assert_allclose(np.asarray(written[key].transpose(*a[key].dims).values), np.asarray(a[key].values), rtol=1e-12)
    back = wio.read_aggregation_state(state_out[name])
    for key in a:
      np.testing.assert_allclose(np.asarray(back.metric_values(metrics)[key].transpose(*a[key].dims).values), np.asarray(a[key].values), rtol=1e-12)
  assert chunked['per_init'].metric_values(metrics)['rmse.2m_temperature'].sizes['init_time'] == 4
  # the predictions really were regridded to the targets' 10 degree grid (else the statistics could not have been formed)
  t_chunk = lt.load_chunk(init_times[:1], lead_times[:1])
  assert lp.load_chunk(init_times[:1], lead_times[:1], t_chunk)['2m_temperature'].sizes['latitude'] == 19
  with pytest.raises(ValueError, match='At least one of'):
    beam_pipeline.define_pipeline(None, time_chunks.TimeChunks(init_times, lead_times), lp, lt, metrics, aggregators)
  with pytest.raises(ValueError, match="don't match aggregator names"):
    beam_pipeline.define_pipeline(None, time_chunks.TimeChunks(init_times, lead_times), lp, lt, metrics, aggregators, out_path={'x': 'y'})


def test_baseline_forecasts_ensembles_from_climatology_and_inference(emulated, tmp_path):
  """The reference's baseline loaders as forecasts: the probabilistic climatology as a 5-member ensemble (CRPS against the pairwise
  definition), persistence against the climatology in a paired HAC test per lead time, ACC with a bootstrap interval from the
  per-init accumulators read back from a file, a forecast served with operational latency."""
  del emulated
  from weatherbenchx_amd import beam_pipeline  # pylint: disable=g-import-not-at-top
  from weatherbenchx_amd import io as wio  # pylint: disable=g-import-not-at-top
  from weatherbenchx_amd import xarray_lite as xr  # pylint: disable=g-import-not-at-top
  from weatherbenchx_amd.data_loaders import latency_wrappers  # pylint: disable=g-import-not-at-top
  from weatherbenchx_amd.metrics import probabilistic  # pylint: disable=g-import-not-at-top
  from weatherbenchx_amd.statistical_inference import bootstrap  # pylint: disable=g-import-not-at-top
  target = mock_data.mock_target_data(time_start='2015-01-01T00', time_stop='2021-01-01T00', variables_3d=[], random=True, seed=1,
                                      spatial_resolution_in_degrees=30.0)
  init_times = np.arange('2020-06-01T00', '2020-06-21T00', np.timedelta64(24, 'h'), dtype='datetime64[ns]')
  lead_times = np.arange(1, 4, dtype='timedelta64[D]').astype('timedelta64[ns]')
  tc = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=7)
  lt = xarray_loaders.TargetsFromXarray(ds=target)
  lat = target['2m_temperature']['latitude'].values
  area = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  per_init = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  # every year 2015..2019 a member
  pc = xarray_loaders.ProbabilisticClimatologyFromXarray(ds=target, start_year=2015, end_year=2019)
  metrics = {'crps': probabilistic.CRPSEnsemble(), 'ssr': probabilistic.UnbiasedSpreadSkillRatio()}
  values = beam_pipeline.define_pipeline(None, tc, pc, lt, metrics, area, out_path=str(tmp_path / 'm.nc'))[None].metric_values(metrics)
  pv = np.asarray(pc.load_chunk(init_times, lead_times)['2m_temperature'].transpose('number', 'init_time', 'lead_time', 'latitude', 'longitude').values)
  tv = np.asarray(lt.load_chunk(init_times, lead_times)['2m_temperature'].transpose('init_time', 'lead_time', 'latitude', 'longitude').values)
  w = O.grid_area_weights(lat)[None, None, :, None]
  pointwise = np.abs(pv - tv[None]).mean(0) - 0.5 * np.abs(pv[:, None] - pv[None, :]).sum((0, 1)) / (5 * 4)
  np.testing.assert_allclose(np.asarray(values['crps.2m_temperature'].values), (pointwise * w).sum((0, 2, 3)) / (np.ones_like(pointwise) * w).sum((0, 2, 3)),
                             rtol=1e-5)
  # persistence against a day-of-year climatology: per-init accumulators, written, read back, tested
  first_year = target['2m_temperature'].isel(time=slice(0, 366))
  clim = xr.Dataset({'2m_temperature': xr.DataArray(np.asarray(first_year.values), dims=('dayofyear', 'latitude', 'longitude'),
                                                    coords={'dayofyear': np.arange(1, 367), 'latitude': lat, 'longitude': first_year['longitude'].values})})
  m2 = {'rmse': deterministic.RMSE(), 'acc': deterministic.ACC(clim)}
  persistence = xarray_loaders.PersistenceFromXarray(ds=target)
  climatology = xarray_loaders.ClimatologyFromXarray(ds=clim, climatology_time_coords=['dayofyear'])
  s_pers = beam_pipeline.define_pipeline(None, tc, persistence, lt, m2, per_init, aggregation_state_out_path=str(tmp_path / 'p.nc'))[None]
  s_clim = beam_pipeline.define_pipeline(None, tc, climatology, lt, {'rmse': deterministic.RMSE()}, per_init,
                                         aggregation_state_out_path=str(tmp_path / 'c.nc'))[None]
  paired = t_test.LazarusHACEWC.for_baseline_comparison(metrics={'rmse': deterministic.RMSE()}, aggregated_statistics=s_pers,
                                                        baseline_aggregated_statistics=s_clim, experimental_unit_dim='init_time')
  diff = np.asarray(paired.point_estimates()['rmse']['2m_temperature'].values)
  rm = lambda s: np.asarray(s.sum_along_dims(['init_time']).metric_values({'rmse': deterministic.RMSE()})['rmse.2m_temperature'].values)
  np.testing.assert_allclose(diff, rm(s_pers) - rm(s_clim), rtol=1e-6, atol=1e-9)
  p = np.asarray(paired.p_values()['rmse']['2m_temperature'].values)
  assert p.shape == (3,) and ((0 <= p) & (p <= 1)).all()
  back = wio.read_aggregation_state(str(tmp_path / 'p.nc'))
  boot = bootstrap.StationaryBootstrap(metrics=m2, aggregated_statistics=back, experimental_unit_dim='init_time', n_replicates=60,
                                       rng=np.random.default_rng(0))
  lower, upper = boot.confidence_intervals(0.1)
  acc = np.asarray(boot.point_estimates()['acc']['2m_temperature'].values)
  np.testing.assert_allclose(acc, np.asarray(s_pers.sum_along_dims(['init_time']).metric_values(m2)['acc.2m_temperature'].values), rtol=1e-9)
  assert (np.asarray(lower['acc']['2m_temperature'].values) < acc).all() and (acc < np.asarray(upper['acc']['2m_temperature'].values)).all()
  # a forecast that is 30 h late: every query is answered from the run issued before it
  forecasts = mock_data.mock_prediction_data(time_start='2020-05-25T00', time_stop='2020-06-25T00', variables_3d=[], random=True, seed=5,
                                             lead_stop_days=6, spatial_resolution_in_degrees=30.0)
  plain = xarray_loaders.PredictionsFromXarray(ds=forecasts)
  late = latency_wrappers.XarrayConstantLatencyWrapper(plain, latency=np.timedelta64(30, 'h'))
  rmse = {'rmse': deterministic.RMSE()}
  got = beam_pipeline.define_pipeline(None, tc, late, lt, rmse, area, out_path=str(tmp_path / 'l.nc'))[None].metric_values(rmse)
  shifted = np.asarray(plain.load_chunk(init_times - np.timedelta64(48, 'h'), lead_times + np.timedelta64(48, 'h'))['2m_temperature'].transpose(
      'init_time', 'lead_time', 'latitude', 'longitude').values)         # 30 h late on a daily cycle: the run of two days earlier
  want = np.sqrt((((shifted - tv) ** 2) * w).sum((0, 2, 3)) / (np.ones_like(tv) * w).sum((0, 2, 3)))
  np.testing.assert_allclose(np.asarray(got['rmse.2m_temperature'].values), want, rtol=1e-5)


def test_station_chunks_binned_by_a_coordinate_accumulate_on_the_host(emulated, tmp_path):
  """Station targets from Parquet, gridded forecasts interpolated pointwise in (init_time, lead_time, latitude, longitude) with the
  altitude adjustment, statistics over `index` binned by the stations' lead_time: the bin labels depend on the chunk, so the chunk
  states are added with an outer join (accumulate='host'); the device accumulators refuse such results instead of adding them
  position by position."""
  del emulated
  import os  # pylint: disable=g-import-not-at-top
  from weatherbenchx_amd import beam_pipeline  # pylint: disable=g-import-not-at-top
  from weatherbenchx_amd import xarray_lite as xr  # pylint: disable=g-import-not-at-top
  from weatherbenchx_amd.data_loaders import sparse_parquet  # pylint: disable=g-import-not-at-top
  metar = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'metar-timeNominal-by-month')
  lt = sparse_parquet.METARFromParquet(path=metar, variables=['2m_temperature'], partitioned_by='month', split_variables=True, dropna=True,
                                       time_dim='timeNominal')
  lat, lon = np.linspace(-90, 90, 37), np.arange(0, 360, 5.0)
  init = np.arange('2020-01-02T00', '2020-01-05T00', np.timedelta64(12, 'h'), dtype='datetime64[ns]')
  lead = (np.arange(0, 5) * 6).astype('timedelta64[h]').astype('timedelta64[ns]')
  rng = np.random.default_rng(0)
  values = 280 + rng.normal(size=(init.size, lead.size, 37, 72))
  forecasts = xr.Dataset({'2m_temperature': xr.DataArray(values, dims=('time', 'prediction_timedelta', 'latitude', 'longitude'),
                                                         coords={'time': init, 'prediction_timedelta': lead, 'latitude': lat, 'longitude': lon})})
  orography = xr.DataArray(rng.uniform(0, 800, size=(37, 72)), dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  to_stations = interpolations.GridToSparseWithAltitudeAdjustment('linear', grid_elevation=orography, wrap_longitude=True)
  lp = xarray_loaders.PredictionsFromXarray(ds=forecasts, interpolation=to_stations)
  metrics = {'rmse': deterministic.RMSE(), 'bias': deterministic.Bias()}
  agg = aggregation.Aggregator(reduce_dims=['index'], bin_by=[binning.ByExactCoord('lead_time')])
  tc = time_chunks.TimeChunks(init, lead[1:], init_time_chunk_size=2, lead_time_chunk_size=2)
  state = beam_pipeline.define_pipeline(None, tc, lp, lt, metrics, agg, out_path=str(tmp_path / 'm.nc'), accumulate='host')[None]
  got = state.metric_values(metrics)
  np.testing.assert_array_equal(got['rmse.2m_temperature'].coords['lead_time'].values, lead[1:])      # the union of the chunks' labels
  # one pass over everything, written out
  t_all = lt.load_chunk(init, lead[1:])
  p_all = lp.load_chunk(init, lead[1:], t_all)['2m_temperature']
  obs = t_all['2m_temperature']
  err = np.asarray(p_all.values, dtype=np.float64) - np.asarray(obs.values, dtype=np.float64)
  lts = obs.coords['lead_time'].values
  for k, one in enumerate(lead[1:]):
    sel = lts == one
    np.testing.assert_allclose(float(np.asarray(got['rmse.2m_temperature'].values)[k]), np.sqrt((err[sel] ** 2).mean()), rtol=1e-5)
    np.testing.assert_allclose(float(np.asarray(got['bias.2m_temperature'].values)[k]), err[sel].mean(), rtol=1e-4, atol=1e-6)
  # the interpolated forecast at a station = bilinear value of its (init, lead) field + the lapse-rate adjustment
  i = 5
  field = values[list(init).index(obs.coords['init_time'].values[i]), list(lead).index(obs.coords['lead_time'].values[i])]
  grid = xr.DataArray(field, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon}, name='2m_temperature')
  one_station = obs.isel(index=[i])
  expect = to_stations.interpolate_data_array(grid, one_station)
  np.testing.assert_allclose(float(np.asarray(p_all.values)[i]), float(np.asarray(expect.values)[0]), rtol=1e-9)
  with pytest.raises(ValueError, match="accumulate='host'"):
    beam_pipeline.define_pipeline(None, tc, lp, lt, metrics, agg, out_path=str(tmp_path / 'd.nc'))
  with pytest.raises(ValueError, match="'device' or 'host'"):
    beam_pipeline.define_pipeline(None, tc, lp, lt, metrics, agg, out_path=str(tmp_path / 'd.nc'), accumulate='beam')


def test_device_accumulators_refuse_results_whose_labels_change_between_chunks(emulated):
  """Two chunks of station data binned by station name: two bins each time, but other stations.  Added position by position the sums
  would land under the first chunk's labels; the accumulation raises instead, and the host route joins the labels."""
  del emulated
  from weatherbenchx_amd import xarray_lite as xr  # pylint: disable=g-import-not-at-top
  init_times = np.array(['2020-01-01T00', '2020-01-02T00'], dtype='datetime64[ns]')
  names = {init_times[0]: np.array(['A', 'A', 'B']), init_times[1]: np.array(['C', 'D', 'D'])}

  def load(init, lead):
    del lead
    n = names[init[0]]
    coords = {'index': np.arange(3), 'stationName': (('index',), n)}
    return ({'t': xr.DataArray(np.array([1.0, 2.0, 3.0]), dims=('index',), coords=coords)},
            {'t': xr.DataArray(np.zeros(3), dims=('index',), coords=coords)})

  metrics = {'mae': deterministic.MAE()}
  agg = aggregation.Aggregator(reduce_dims=['index'], bin_by=[binning.ByExactCoord('stationName')])
  tc = time_chunks.TimeChunks(init_times, np.array([0], dtype='timedelta64[h]'), init_time_chunk_size=1)
  with pytest.raises(ValueError, match='labels of dimension .stationName. changed between chunks'):
    pipeline.evaluate_chunks(tc, load, metrics, agg)
  total = aggregation.AggregationState.zero()
  for it, ld in tc:
    p, t = load(it, ld)
    total = total + agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, p, t))
  out = total.metric_values(metrics)['mae.t']
  assert out.coords['stationName'].values.tolist() == ['A', 'B', 'C', 'D']
  np.testing.assert_allclose(np.asarray(out.values), [1.5, 3.0, 1.0, 2.5])


def test_time_unit_bins_whose_labels_depend_on_the_chunk(emulated, tmp_path):
  """Gridded data can produce chunk-dependent frames too: `ByTimeUnit('hour', 'init_time')` over chunks of two 6-hourly inits sees the
  hours [0, 6] in one chunk and [12, 18] in the next.  Chunks that each hold all four hours accumulate on the device and equal the
  single-chunk result; the half-day chunks are refused there and are right on the host route."""
  del emulated
  from weatherbenchx_amd import beam_pipeline  # pylint: disable=g-import-not-at-top
  target = mock_data.mock_target_data(time_start='2020-01-01T00', time_stop='2020-01-08T00', variables_3d=[], random=True, seed=1,
                                      spatial_resolution_in_degrees=30.0, time_resolution_hours=6)
  forecast = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-05T00', variables_3d=[], random=True, seed=2,
                                            lead_stop_days=1, spatial_resolution_in_degrees=30.0, time_resolution_hours=6)
  lt, lp = xarray_loaders.TargetsFromXarray(ds=target), xarray_loaders.PredictionsFromXarray(ds=forecast)
  init = np.arange('2020-01-01T00', '2020-01-05T00', np.timedelta64(6, 'h'), dtype='datetime64[ns]')
  lead = np.arange(2, dtype='timedelta64[D]').astype('timedelta64[ns]')
  metrics = {'rmse': deterministic.RMSE()}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.ByTimeUnit('hour', 'init_time')])
  run = lambda size, **kw: beam_pipeline.define_pipeline(None, time_chunks.TimeChunks(init, lead, init_time_chunk_size=size), lp, lt, metrics, agg,
                                                         out_path=str(tmp_path / 'm.nc'), **kw)[None].metric_values(metrics)['rmse.2m_temperature']
  whole = run(None)
  assert whole.coords['init_time_hour'].values.tolist() == [0, 6, 12, 18]
  days = run(4)
  np.testing.assert_allclose(np.asarray(days.transpose(*whole.dims).values), np.asarray(whole.values), rtol=1e-6)
  with pytest.raises(ValueError, match="labels of dimension 'init_time_hour' changed between chunks"):
    run(2)
  halves = run(2, accumulate='host')
  np.testing.assert_allclose(np.asarray(halves.transpose(*whole.dims).values), np.asarray(whole.values), rtol=1e-6)


def test_loaders_metrics_and_interpolations_pickle():
  """The reference ships its loaders, metrics and aggregators to Beam workers by pickle (ownership); the objects added in
  round 5 survive the same trip with the standard pickle (no closures inside)."""
  import os  # pylint: disable=g-import-not-at-top
  import pickle  # pylint: disable=g-import-not-at-top
  from weatherbenchx_amd.data_loaders import latency_wrappers, sparse_parquet  # pylint: disable=g-import-not-at-top
  from weatherbenchx_amd.metrics import probabilistic  # pylint: disable=g-import-not-at-top
  t = mock_data.mock_target_data(time_start='2020-01-01T00', time_stop='2020-01-05T00', variables_3d=[], random=True, seed=1,
                                 spatial_resolution_in_degrees=30.0)
  f = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-05T00', variables_3d=[], lead_stop_days=2,
                                     spatial_resolution_in_degrees=30.0)
  metar = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'metar-timeNominal-by-month')
  objects = [
      xarray_loaders.TargetsFromXarray(ds=t, interpolation=interpolations.InterpolateToFixedCoords('linear', {'latitude': np.arange(-80, 81, 20.0)})),
      latency_wrappers.XarrayConstantLatencyWrapper(xarray_loaders.PredictionsFromXarray(ds=f), latency=np.timedelta64(6, 'h')),
      sparse_parquet.METARFromParquet(path=metar, variables=['2m_temperature'], time_dim='timeNominal'),
      wrappers.WrappedMetric(categorical.CSI(), [wrappers.ContinuousToBinary('both', [0.5], 'threshold')]),
      spatial.FSS([1, 3], True), probabilistic.EnergyScore('longitude'), probabilistic.RelativeEconomicValue(ensemble_size=5),
      probabilistic.EnsembleRankedProbabilityScore([0.1, 0.5], [0.1, 0.5], 'bin', 's'), categorical.SEEPS(['x'], t),
      categorical.Opportunism('number', t, True, True, None), categorical.Reliability(),
      aggregation.Aggregator(reduce_dims=['index'], bin_by=[binning.ByExactCoord('lead_time'), binning.BySets({'a': ['x']}, 'stationName', 'sub')],
                             weigh_by=[weighting.StationDensityWeighting()]),
      interpolations.GridToSparseWithAltitudeAdjustment('linear', grid_elevation=t['2m_temperature'].isel(time=0, drop=True)),
  ]
  for obj in objects:
    back = pickle.loads(pickle.dumps(obj))
    assert type(back) is type(obj)
  loader = pickle.loads(pickle.dumps(objects[2]))
  chunk = loader.load_chunk(np.array(['2020-01-02T00'], dtype='datetime64[ns]'), np.array([6], dtype='timedelta64[h]'))
  assert chunk['2m_temperature'].size > 5
  assert set(pickle.loads(pickle.dumps(objects[9])).statistics) == {'Confident', 'Covered'}


def test_archive_conventions_descending_latitude_lat_lon_names_level_selection_and_crop(emulated, tmp_path):
  """Datasets as the archives have them: `lat` / `lon` / `time` / `prediction_timedelta` names, latitude from north to south, a
  level picked with `sel_kwargs`, both sides cropped to a box (which sorts the axes ascending) -- through `define_pipeline` in chunks
  of five init times, against whole-array NumPy with the regional area weights."""
  del emulated
  from weatherbenchx_amd import beam_pipeline  # pylint: disable=g-import-not-at-top
  from weatherbenchx_amd import xarray_lite as xr  # pylint: disable=g-import-not-at-top
  rng = np.random.default_rng(0)
  lat, lon, lev = np.linspace(90, -90, 19), np.arange(0, 360, 20.0), np.array([500, 850])
  times = np.datetime64('2020-01-01', 'ns') + np.arange(20) * np.timedelta64(12, 'h')
  leads = (np.arange(3) * 12).astype('timedelta64[h]').astype('timedelta64[ns]')
  truth, forecast = rng.normal(size=(20, 2, 19, 18)), rng.normal(size=(12, 3, 2, 19, 18))
  tds = xr.Dataset({'z': xr.DataArray(truth, dims=('time', 'level', 'lat', 'lon'), coords={'time': times, 'level': lev, 'lat': lat, 'lon': lon})})
  pds = xr.Dataset({'z': xr.DataArray(forecast, dims=('time', 'prediction_timedelta', 'level', 'lat', 'lon'),
                                      coords={'time': times[:12], 'prediction_timedelta': leads, 'level': lev, 'lat': lat, 'lon': lon})})
  box = interpolations.CropToBox(-60, 60, 0, 180)
  lt = xarray_loaders.TargetsFromXarray(ds=tds, sel_kwargs={'level': [850]}, interpolation=box)
  lp = xarray_loaders.PredictionsFromXarray(ds=pds, sel_kwargs={'level': [850]}, interpolation=box)
  metrics = {'rmse': deterministic.RMSE(), 'bias': deterministic.Bias()}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  state = beam_pipeline.define_pipeline(None, time_chunks.TimeChunks(times[:12], leads, init_time_chunk_size=5), lp, lt, metrics, agg,
                                        out_path=str(tmp_path / 'm.nc'))[None]
  got = state.metric_values(metrics)
  assert got['rmse.z'].dims == ('lead_time', 'level') and got['rmse.z'].coords['level'].values.tolist() == [850]
  south_to_north = lat[::-1]
  keep_lat, keep_lon = (south_to_north >= -60) & (south_to_north <= 60), (lon >= 0) & (lon <= 180)
  t_box = truth[:, 1][:, ::-1][:, keep_lat][:, :, keep_lon]
  p_box = forecast[:, :, 1][:, :, ::-1][:, :, keep_lat][:, :, :, keep_lon]
  w = O.grid_area_weights(south_to_north[keep_lat])[None, :, None]
  for k in range(3):
    err = p_box[:, k] - t_box[np.arange(12) + k]
    np.testing.assert_allclose(float(np.asarray(got['rmse.z'].values)[k, 0]), np.sqrt((err ** 2 * w).sum() / (np.ones_like(err) * w).sum()), rtol=1e-6)
    np.testing.assert_allclose(float(np.asarray(got['bias.z'].values)[k, 0]), (err * w).sum() / (np.ones_like(err) * w).sum(), rtol=1e-5, atol=1e-9)
