"""GPU parity tests at the sizes bench.py times ("full-size parity for what the bench times"), plus the
device-resident plumbing of round 2: masks built and consumed in HBM, page-locked feeder uploads, a GPU-produced
AggregationState through file -> read -> statistical-inference consumer.  Tolerance: rtol 1e-6 (north_star)."""
import os
import sys

import numpy as np
import pytest

from oracle import wbx_oracle as O
from weatherbenchx_amd import _hip
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import binning
from weatherbenchx_amd import data as wdata
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.metrics import deterministic
from weatherbenchx_amd.metrics import probabilistic

pytestmark = pytest.mark.gpu
RTOL = 1e-6
NLAT, NLON = 721, 1440
LAT = np.linspace(-90, 90, NLAT)
LON = np.linspace(0, 360, NLON, endpoint=False)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def ctx():
  assert _hip.is_available(), 'gpu tests need libwbx_hip.so and a HIP device'
  return _hip.default_context(0)


def _randn(shape, seed, offset=0.0, scale=1.0):
  import torch
  g = torch.Generator(device='cuda')
  g.manual_seed(seed)
  return torch.randn(shape, generator=g, device='cuda', dtype=torch.float32) * scale + offset


def _sp(layout):
  return ('latitude', 'longitude') if layout == 'lon_fastest' else ('longitude', 'latitude')


def _sp_shape(layout):
  return (NLAT, NLON) if layout == 'lon_fastest' else (NLON, NLAT)


def _to_latlon(a, layout):
  """numpy [..., sp] -> [..., lat, lon]"""
  return a if layout == 'lon_fastest' else np.swapaxes(a, -1, -2)


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
def test_headline_kernel_shape_det6_with_gather_full_grid(ctx, layout):
  """What bench.py's main line launches: DET6 + the (dayofyear, hour) climatology gather at 721 x 1440 with 40 inits
  and a 44-slot (not 366) table, reduce (init_time, latitude, longitude) with area weights -- against float64 torch
  on the same device data for every (lead, level), and against the NumPy oracle on one (lead, level) slice.
  Reference: metrics/base.py:382-403 (alignment), deterministic.py:91-123,222-259, aggregation.py:339-366."""
  import torch
  ni, nl, nlev, ndoy = 40, 3, 2, 44
  sp, sps = _sp(layout), _sp_shape(layout)
  init_time = np.datetime64('2020-01-01T00', 'ns') + np.arange(ni) * np.timedelta64(24, 'h')
  lead_time = (np.arange(nl) * 18).astype('timedelta64[h]').astype('timedelta64[ns]')  # 0, 18, 36 h: hours 0 / 18 / 12
  coords = {'init_time': init_time, 'lead_time': lead_time, 'level': np.array([500, 850]), 'latitude': LAT, 'longitude': LON}
  dims = ('init_time', 'lead_time', 'level') + sp
  cdims = ('dayofyear', 'hour', 'level') + sp
  clim_t = _randn((ndoy, 4, nlev) + sps, 1, 280.0, 10.0)
  p_t = _randn((ni, nl, nlev) + sps, 2, 280.0, 3.0)
  t_t = _randn((ni, nl, nlev) + sps, 3, 280.0, 3.0)
  clim = xr.Dataset({'z': xr.DataArray(clim_t, dims=cdims, coords={
      'dayofyear': np.arange(1, ndoy + 1), 'hour': np.array([0, 6, 12, 18]), 'level': coords['level'],
      'latitude': LAT, 'longitude': LON})})
  metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE(), 'bias': deterministic.Bias(),
             'acc': deterministic.ACC(clim), 'activity': deterministic.PredictionActivity(clim)}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  got = aggregation.compute_metric_values_for_single_chunk(
      metrics, agg, {'z': xr.DataArray(p_t, dims=dims, coords=coords)}, {'z': xr.DataArray(t_t, dims=dims, coords=coords)})

  # float64 torch on the same device data
  vt = init_time[:, None] + lead_time[None, :]
  doy = (vt.astype('datetime64[D]') - vt.astype('datetime64[Y]').astype('datetime64[D]')).astype(int)  # 0-based
  hour = ((vt - vt.astype('datetime64[D]')).astype('timedelta64[h]').astype(int) // 6)
  c = clim_t[torch.as_tensor(doy, device='cuda'), torch.as_tensor(hour, device='cuda')].double()  # [ni, nl, nlev, sp]
  p, t = p_t.double(), t_t.double()
  w = torch.as_tensor(O.grid_area_weights(LAT), device='cuda')
  wsp = w[:, None] if layout == 'lon_fastest' else w[None, :]
  den = float(w.sum()) * ni * NLON

  def mean(x):
    return ((x * wsp).sum(dim=(0, 3, 4)) / den).cpu().numpy()
  e = p - t
  want = {'rmse': np.sqrt(mean(e * e)), 'mae': mean(e.abs()), 'bias': mean(e),
          'acc': mean((p - c) * (t - c)) / np.sqrt(mean((p - c) ** 2) * mean((t - c) ** 2)),
          'activity': np.sqrt(mean((p - c) ** 2))}
  for k, v in want.items():
    np.testing.assert_allclose(got[f'{k}.z'].transpose('lead_time', 'level').values, v, rtol=RTOL, atol=1e-9, err_msg=k)
  # the NumPy oracle on one (lead, level) slice, with its own climatology alignment
  l, k = 1, 1
  ph = _to_latlon(p_t[:, l, k].cpu().numpy(), layout)
  th = _to_latlon(t_t[:, l, k].cpu().numpy(), layout)
  ch_all = _to_latlon(clim_t[:, :, k].cpu().numpy(), layout)
  ch, _ = O.align_climatology(ch_all, ('dayofyear', 'hour', 'latitude', 'longitude'), vt[:, l], ('init_time',))
  odims = ('init_time', 'latitude', 'longitude')
  wo = (O.grid_area_weights(LAT), ('latitude',))

  def omean(a):
    sws, sw, _ = O.aggregate(a, odims, list(odims), weights=[wo])
    return sws / sw
  oracle_acc = O.acc(omean(O.anomaly_covariance(ph, th, ch)), omean(O.squared_prediction_anomaly(ph, ch)),
                     omean(O.squared_target_anomaly(th, ch)))
  sel = dict(lead_time=lead_time[l], level=coords['level'][k])
  np.testing.assert_allclose(got['acc.z'].sel(**sel).values, oracle_acc, rtol=RTOL)
  np.testing.assert_allclose(got['rmse.z'].sel(**sel).values, O.rmse(omean(O.squared_error(ph, th))), rtol=RTOL)


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
def test_public_chunk_34_bins_masked_with_acc_against_the_oracle(ctx, layout):
  """The public benchmark's chunk shape through its production configuration (run_benchmark_evaluation.py:97-131,
  369-382): 1 init x 12 leads x 13 levels, 17 regions x land/sea = 34 bins, masked=True with NaN targets whose mask is
  built on the device, GridAreaWeighting, RMSE / bias / ACC -- against the oracle on one (lead, level) slice, and the
  'global' bin against float64 torch for every (lead, level)."""
  import torch
  sys.path.insert(0, os.path.join(ROOT, 'tools'))
  from wb_regions import REGIONS
  nl, nlev, ndoy = 12, 13, 8
  sp, sps = _sp(layout), _sp_shape(layout)
  coords = {'init_time': np.array(['2020-01-01T00'], dtype='datetime64[ns]'),
            'lead_time': (np.arange(nl) * 12).astype('timedelta64[h]').astype('timedelta64[ns]'),
            'level': np.arange(nlev), 'latitude': LAT, 'longitude': LON}
  dims = ('init_time', 'lead_time', 'level') + sp
  cdims = ('dayofyear', 'hour', 'level') + sp
  p_t = _randn((1, nl, nlev) + sps, 11, 280.0, 2.0)
  t_t = _randn((1, nl, nlev) + sps, 12, 280.0, 2.0)
  clim_t = _randn((ndoy, 4, nlev) + sps, 13, 280.0, 10.0)
  hole = torch.rand(sps, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5)) < 0.2  # "sea ice": NaN targets
  t_t[..., hole] = float('nan')
  clim = xr.Dataset({'z': xr.DataArray(clim_t, dims=cdims, coords={
      'dayofyear': np.arange(1, ndoy + 1), 'hour': np.array([0, 6, 12, 18]), 'level': coords['level'],
      'latitude': LAT, 'longitude': LON})})
  rng = np.random.default_rng(7)
  land = rng.random((NLAT, NLON)) > 0.7
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': LAT, 'longitude': LON})
  metrics = {'rmse': deterministic.RMSE(), 'bias': deterministic.Bias(), 'acc': deterministic.ACC(clim)}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)], masked=True)
  targets = wdata.add_nan_mask_to_data({'z': xr.DataArray(t_t, dims=dims, coords=coords)})
  assert targets['z']._coords['mask'][1].is_cuda  # built in HBM, stays there
  got = aggregation.compute_metric_values_for_single_chunk(metrics, agg, {'z': xr.DataArray(p_t, dims=dims, coords=coords)}, targets)
  assert got['rmse.z'].sizes['region'] == 34
  # oracle on one (lead, level) slice
  l, k = 5, 7
  ph = _to_latlon(p_t[0, l, k].cpu().numpy(), layout)[None]
  th = _to_latlon(t_t[0, l, k].cpu().numpy(), layout)[None]
  vt = coords['init_time'][:1] + coords['lead_time'][l]
  ch, _ = O.align_climatology(_to_latlon(clim_t[:, :, k].cpu().numpy(), layout), ('dayofyear', 'hour', 'latitude', 'longitude'),
                              vt, ('init_time',))
  names, masks = O.region_masks(LAT, LON, REGIONS, land_sea_mask=land)
  odims = ('init_time', 'latitude', 'longitude')
  kw = dict(weights=[(O.grid_area_weights(LAT), ('latitude',))], bin_masks=[('region', masks, ('region', 'latitude', 'longitude'))],
            mask=~np.isnan(th), mask_dims=odims)

  def omean(a):
    with np.errstate(invalid='ignore'):
      sws, sw, od = O.aggregate(a, odims, list(odims), **kw)
    return sws / sw
  sel = dict(lead_time=coords['lead_time'][l], level=k)
  assert list(got['acc.z']['region'].values) == names
  with np.errstate(invalid='ignore'):
    want_acc = O.acc(omean(O.anomaly_covariance(ph, th, ch)), omean(O.squared_prediction_anomaly(ph, ch)),
                     omean(O.squared_target_anomaly(th, ch)))
    want_rmse = O.rmse(omean(O.squared_error(ph, th)))
    want_bias = omean(O.error(ph, th))
  np.testing.assert_allclose(got['acc.z'].sel(**sel).values, want_acc, rtol=RTOL)
  np.testing.assert_allclose(got['rmse.z'].sel(**sel).values, want_rmse, rtol=RTOL)
  np.testing.assert_allclose(got['bias.z'].sel(**sel).values, want_bias, rtol=RTOL, atol=1e-9)
  # 'global' for every (lead, level) against float64 torch
  w = torch.as_tensor(O.grid_area_weights(LAT), device='cuda')
  wsp = w[:, None] if layout == 'lon_fastest' else w[None, :]
  valid = ~torch.isnan(t_t[0])
  e = torch.where(valid, p_t[0].double() - t_t[0].double(), torch.zeros((), device='cuda', dtype=torch.float64))
  num = (e * e * wsp).sum(dim=(2, 3))
  den = (valid.double() * wsp).sum(dim=(2, 3))
  np.testing.assert_allclose(got['rmse.z'].sel(region='global').transpose('lead_time', 'level').values,
                             torch.sqrt(num / den).cpu().numpy(), rtol=RTOL)


def test_masks_of_device_payloads_never_visit_the_host(ctx, monkeypatch):
  """add_nan_mask_to_data on payloads in HBM builds the mask there (wbx_notnan_mask) and the masked kernels read it in
  place: no upload anywhere near a byte per point happens (data_loaders/base.py:25-56, aggregation.py:339-352)."""
  import torch
  shape = (4, 181, 360)
  lat, lon = np.linspace(-90, 90, 181), np.arange(360.0)
  dims = ('lead_time', 'latitude', 'longitude')
  coords = {'latitude': lat, 'longitude': lon}
  tv = _randn(shape, 21, 280.0)
  tv[:, 20:50, 100:200] = float('nan')
  pv = _randn(shape, 22, 280.0)
  ens = _randn((4, 5, 181, 360), 23) + tv[:, None]
  uploads = []
  real = _hip.Context.upload

  def counting(self, arr):
    uploads.append(int(np.asarray(arr).nbytes))
    return real(self, arr)
  monkeypatch.setattr(_hip.Context, 'upload', counting)
  targets = wdata.add_nan_mask_to_data({'v': xr.DataArray(tv, dims=dims, coords=coords)})
  mask = targets['v']._coords['mask'][1]
  assert mask.is_cuda and mask.dtype == torch.bool and tuple(mask.stride()) == tuple(tv.stride())
  assert torch.equal(mask, ~torch.isnan(tv))
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()], masked=True)
  det = aggregation.compute_metric_values_for_single_chunk({'rmse': deterministic.RMSE()}, agg,
                                                           {'v': xr.DataArray(pv, dims=dims, coords=coords)}, targets)
  crps = aggregation.compute_metric_values_for_single_chunk(
      {'crps': probabilistic.CRPSEnsemble(use_sort=True)}, agg,
      {'v': xr.DataArray(ens, dims=('lead_time', 'number', 'latitude', 'longitude'), coords=coords)}, targets)
  assert max(uploads, default=0) < tv.numel() // 4, uploads  # tables and weights only
  th, ph, eh = tv.cpu().numpy(), pv.cpu().numpy(), ens.cpu().numpy()
  valid = ~np.isnan(th)
  w = (O.grid_area_weights(lat), ('latitude',))
  with np.errstate(invalid='ignore'):
    sws, sw, _ = O.aggregate(O.squared_error(ph, th), dims, ['latitude', 'longitude'], weights=[w], mask=valid, mask_dims=dims)
  np.testing.assert_allclose(det['rmse.v'].values, np.sqrt(sws / sw), rtol=RTOL)
  edims = ('lead_time', 'number', 'latitude', 'longitude')
  with np.errstate(invalid='ignore'):
    sk = O.aggregate(O.crps_skill(eh, edims, th, dims, 'number')[0], dims, ['latitude', 'longitude'], weights=[w], mask=valid, mask_dims=dims)
    sp = O.aggregate(O.crps_spread(eh, edims, 'number', use_sort=True)[0], dims, ['latitude', 'longitude'], weights=[w])
  np.testing.assert_allclose(crps['crps.v'].values, O.crps(sk[0] / sk[1], sp[0] / sp[1]), rtol=RTOL)
  # a non-dense view falls back to the elementwise route and still lives on the device
  view = {'v': xr.DataArray(tv[:, ::2], dims=dims, coords={'latitude': lat[::2], 'longitude': lon})}
  m2 = wdata.add_nan_mask_to_data(view)['v']._coords['mask'][1]
  assert m2.is_cuda and torch.equal(m2, ~torch.isnan(tv[:, ::2]))


def test_feeder_with_page_locked_chunks_against_the_oracle(ctx):
  """The chunk loop fed from page-locked arrays (pipeline.pinned_empty): asynchronous DMA on the feeder's copy stream,
  the launch stream waits on the copy's event -- chunked + prefetched == the oracle on the whole data set
  (LoadPredictionsAndTargets ahead of the aggregation DoFn, beam_pipeline.py:69-116)."""
  from weatherbenchx_amd import pipeline, time_chunks
  rng = np.random.default_rng(31)
  ni, nl, nlat, nlon = 6, 4, 181, 360
  lat, lon = np.linspace(-90, 90, nlat), np.arange(nlon) * 1.0
  init_times = np.datetime64('2020-01-01T00', 'ns') + np.arange(ni) * np.timedelta64(12, 'h')
  lead_times = (np.arange(nl) * 6).astype('timedelta64[h]').astype('timedelta64[ns]')
  pv = (rng.normal(size=(ni, nl, nlat, nlon)) + 280).astype(np.float32)
  tv = (rng.normal(size=(ni, nl, nlat, nlon)) + 280).astype(np.float32)
  dims = ('init_time', 'lead_time', 'latitude', 'longitude')
  pinned_seen = []

  def load(inits, leads):
    ii = np.searchsorted(init_times, inits)
    li = np.searchsorted(lead_times, leads)
    out = []
    for src in (pv, tv):
      buf = pipeline.pinned_empty((len(ii), len(li), nlat, nlon), np.float32)  # a loader decoding straight into it
      np.copyto(buf, src[np.ix_(ii, li)])
      pinned_seen.append(_hip.is_pinned(buf))
      out.append({'v': xr.DataArray(buf, dims=dims, coords={'init_time': inits, 'lead_time': leads, 'latitude': lat,
                                                            'longitude': lon})})
    return out[0], out[1]
  metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE()}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  times = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=1, lead_time_chunk_size=2)
  got = pipeline.evaluate_chunks(times, load, metrics, agg, prefetch=2)[None].metric_values(metrics)
  assert pinned_seen and all(pinned_seen)
  w = (O.grid_area_weights(lat), ('latitude',))
  sws, sw, od = O.aggregate(O.squared_error(pv, tv), dims, ['init_time', 'latitude', 'longitude'], weights=[w])
  np.testing.assert_allclose(got['rmse.v'].values, np.sqrt(sws / sw), rtol=RTOL)
  sws, sw, od = O.aggregate(O.absolute_error(pv, tv), dims, ['init_time', 'latitude', 'longitude'], weights=[w])
  np.testing.assert_allclose(got['mae.v'].values, sws / sw, rtol=RTOL)


def test_gpu_state_through_file_to_the_bootstrap_consumer(ctx, tmp_path):
  """a GPU-produced AggregationState with init_time preserved is written, read back, and consumed the
  way statistical_inference does (sum_along_dims + mean_statistics for point estimates, `dot` with a replicate x
  init_time count matrix for resampled values: statistical_inference/base.py:42-76, bootstrap.py:140-160) -- equal to
  the same operations on the oracle's accumulators."""
  from weatherbenchx_amd import io as wio
  rng = np.random.default_rng(41)
  ni, nl, nlat, nlon = 5, 3, 91, 180
  lat, lon = np.linspace(-90, 90, nlat), np.arange(nlon) * 2.0
  init_times = np.datetime64('2020-01-01T00', 'ns') + np.arange(ni) * np.timedelta64(24, 'h')
  lead_times = (np.arange(nl) * 24).astype('timedelta64[h]').astype('timedelta64[ns]')
  dims = ('init_time', 'lead_time', 'latitude', 'longitude')
  coords = {'init_time': init_times, 'lead_time': lead_times, 'latitude': lat, 'longitude': lon}
  pv = _randn((ni, nl, nlat, nlon), 51, 280.0)
  tv = _randn((ni, nl, nlat, nlon), 52, 280.0)
  metrics = {'rmse': deterministic.RMSE(), 'bias': deterministic.Bias()}
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])  # init_time survives
  stats = metrics_base.compute_unique_statistics_for_all_metrics(
      metrics, {'z': xr.DataArray(pv, dims=dims, coords=coords)}, {'z': xr.DataArray(tv, dims=dims, coords=coords)})
  state = agg.aggregate_statistics(stats)
  path = os.path.join(tmp_path, 'aggregation_state.nc')
  wio.write_aggregation_state(state, path)
  back = wio.read_aggregation_state(path)
  assert back.sum_weighted_statistics['SquaredError']['z'].dims == ('init_time', 'lead_time')
  np.testing.assert_array_equal(back.sum_weighted_statistics['SquaredError']['z']['init_time'].values, init_times)
  # the oracle's accumulators
  ph, th = pv.cpu().numpy(), tv.cpu().numpy()
  w = (O.grid_area_weights(lat), ('latitude',))
  se_s, se_w, _ = O.aggregate(O.squared_error(ph, th), dims, ['latitude', 'longitude'], weights=[w])
  er_s, er_w, _ = O.aggregate(O.error(ph, th), dims, ['latitude', 'longitude'], weights=[w])
  np.testing.assert_allclose(back.sum_weighted_statistics['SquaredError']['z'].values, se_s, rtol=RTOL)
  # point estimates: reduce the experimental-unit dim, then the metrics
  point = back.sum_along_dims(['init_time']).metric_values(metrics)
  np.testing.assert_allclose(point['rmse.z'].values, np.sqrt(se_s.sum(0) / se_w.sum(0)), rtol=RTOL)
  np.testing.assert_allclose(point['bias.z'].values, er_s.sum(0) / er_w.sum(0), rtol=RTOL, atol=1e-9)
  # bootstrap replicates: one matrix product with the multinomial counts
  counts = rng.multinomial(ni, np.full(ni, 1 / ni), size=16)
  resampled = back.dot(xr.DataArray(counts, dims=['replicate', 'init_time']), dim='init_time').metric_values(metrics)
  want = np.sqrt((counts.astype(np.float64) @ se_s) / (counts.astype(np.float64) @ se_w))
  np.testing.assert_allclose(resampled['rmse.z'].transpose('replicate', 'lead_time').values, want, rtol=RTOL)


def _run_bench(*flags, timeout=600):
  import json
  import subprocess
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
  res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(flags), capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
  assert res.returncode == 0, res.stderr[-3000:]
  lines = [l for l in res.stdout.splitlines() if l.strip()]
  # stdout is ONE JSON line and nothing else (library banners -- RCCL's version block, gloo's rank messages -- go to stderr)
  assert len(lines) == 1 and lines[0].startswith('{'), res.stdout[-2000:]
  line = json.loads(lines[0])
  if 'full' not in line:  # (a "skipped" line)
    return line
  # (r5) the line is the compact one the driver parses (< 8 KB: headline + roofline + cpu_baseline + one record per leg); every
  # leg's full result sits in the file it names, beside bench.py
  assert len(lines[0]) < 8192 and ({'roofline', 'config'} <= set(line) or 'note' in line)  # ('note': --legs without the main leg)
  full = json.load(open(os.path.join(ROOT, line['full'])))
  assert full['value'] == line['value'] and full.get('ms_per_step') == line.get('ms_per_step')
  return full


def test_bench_contract_small(ctx):
  """bench.py keeps the driver's contract (one JSON line with value / roofline / config.workload, every leg present) --
  at debugging sizes, so that a broken leg shows up here and not in the round-end run."""
  out = _run_bench('--small', '--steps', '2', '--warmup', '1', '--cpu-workers', '2')
  for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'configs1', 'ensemble',
              'public_chunk', 'spectrum', 'lat_fastest', 'config5'):
    assert key in out, key
  import json
  baseline = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
  assert out['metric'] == baseline['metric']  # the driver line quotes BASELINE.json's metric string itself
  assert out['n_gpus'] == 1 and out['steps'] == 2 and out['value'] > 0
  # the main line is the north_star field: one 51-member forecast per GPU, the ensemble kernel's roofline beside it
  assert 'north_star' in out['config']['workload'] and '51 member' in out['config']['workload']
  assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(out['roofline']) and 'ens_pipe_kernel' in out['roofline']['kernel']
  assert out['roofline']['launches_per_step'] == 1  # all four metrics of a variable out of ONE ensemble launch
  cpu = out['cpu_baseline']
  assert {'value', 'unit', 'cores', 'kind', 'sample'} <= set(cpu) and cpu['cores'] == 1
  for suite in ('ensemble', 'deterministic'):  # the main line's workload and configs1, one core and all cores each
    assert {'value', 'unit', 'cores', 'kind', 'sample', 'all_cores'} <= set(cpu[suite]), suite
    assert cpu[suite]['all_cores']['cores'] == 2 and cpu[suite]['all_cores']['value'] > 0
  assert cpu['value'] == cpu['ensemble']['value']
  # the reference-default CRPSEnsemble() (use_sort=False) is ONE ensemble launch per variable; the pair-form kernel is
  # timed on its own events
  assert out['ensemble']['default_crps_ensemble']['ensemble_launches_per_variable'] == 1
  assert 'PAIRWISE' in out['ensemble']['pairwise_form']['kernel']
  np.testing.assert_allclose(out['ensemble']['pairwise_form']['crps'], out['ensemble']['default_crps_ensemble']['crps'], rtol=1e-6)
  assert {'main', 'configs1', 'spectrum'} <= set(out['lat_fastest'])
  assert 'lat_fastest' in out['lat_fastest']['main']['workload']
  assert out['config5']['scaling'] == 'strong' and out['config5']['chunks'] == 6
  assert abs(out['check']['crps_mean'] - 0.5642) < 0.01 and abs(out['configs1']['check']['rmse_mean'] - 2 ** 0.5) < 0.01


def test_bench_two_ranks_from_a_bare_shell(ctx):
  """`python bench.py --gpus 2` from a bare shell launches itself under torch.distributed.run.  On a 1-GPU box the RCCL
  run is skipped with a clear JSON line (rc 0); with --backend gloo the two ranks share the device and the whole N > 1
  path -- per-step device accumulators + one all-reduce, config5 sharded i mod 2 with one reduce per pass -- runs for
  real (the collective goes through the host: a plumbing check, not a measurement)."""
  import torch
  if torch.cuda.device_count() < 2:
    skipped = _run_bench('--gpus', '2', '--small', '--steps', '2', '--warmup', '1')
    assert skipped.get('skipped') is True and skipped['n_gpus'] == 2 and 'devices' in skipped['reason']
  out = _run_bench('--gpus', '2', '--backend', 'gloo', '--small', '--steps', '2', '--warmup', '1', '--legs', 'main,config5')
  assert out['n_gpus'] == 2 and out['config']['collectives_per_step'] == 1.0 and out['config']['backend'] == 'gloo'
  assert abs(out['check']['crps_mean'] - 0.5642) < 0.01
  one = _run_bench('--small', '--steps', '2', '--warmup', '1', '--legs', 'config5')
  # the sharded, all-reduced result of the streamed suite equals the one-rank result
  assert out['config5']['n_gpus'] == 2 and out['config5']['check']['shape_rmse_z'] == one['config5']['check']['shape_rmse_z']
  np.testing.assert_allclose(out['config5']['check']['rmse_z_mean'], one['config5']['check']['rmse_z_mean'], rtol=1e-12)
  np.testing.assert_allclose(out['config5']['check']['crps_t2m_mean'], one['config5']['check']['crps_t2m_mean'], rtol=1e-12)
  # (the spectra's sums of row weights never touch the device: every rank counts how often it met the cached array)
  np.testing.assert_allclose(out['config5']['check']['spectrum_p_z_mean'], one['config5']['check']['spectrum_p_z_mean'], rtol=1e-9)
  assert out['config5']['check']['spectrum_p_z_mean'] > 0
