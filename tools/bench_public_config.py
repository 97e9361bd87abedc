"""Public-benchmark style chunk (public_benchmark/run_benchmark_evaluation.py:301-382): one chunk of 1 init x 12 leads
x 13 levels at 0.25 deg, rmse/mse/bias/acc/activity, GridAreaWeighting, Regions(17) x land-sea (34 bins), masked=True;
latitude-fastest like the real zarr chunks.  Prints ms per chunk and the algorithmic GB/s (12 B/point)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from weatherbenchx_amd import _hip, aggregation, binning, engine, weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as mb, deterministic
from wb_regions import REGIONS  # noqa: E402

layout = sys.argv[1] if len(sys.argv) > 1 else 'lat_fastest'
ni = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nl, nlev, nlat, nlon = 12, 13, 721, 1440
lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
sp = ('longitude', 'latitude') if layout == 'lat_fastest' else ('latitude', 'longitude')
dims = ('init_time', 'lead_time', 'level') + sp
coords = {'init_time': np.datetime64('2020-01-01T00', 'ns') + np.arange(ni) * np.timedelta64(12, 'h'),
          'lead_time': (np.arange(nl) * 12).astype('timedelta64[h]').astype('timedelta64[ns]'),
          'level': np.arange(nlev), 'latitude': lat, 'longitude': lon}
shape = tuple(len(coords[d]) for d in dims)
p_t, t_t = torch.randn(shape, device='cuda') + 280, torch.randn(shape, device='cuda') + 280
clim_t = torch.randn((10, 4) + shape[2:], device='cuda') + 280
clim = xr.Dataset({'z': xr.DataArray(clim_t, dims=('dayofyear', 'hour') + dims[2:], coords={
    'dayofyear': np.arange(1, 11), 'hour': np.array([0, 6, 12, 18]), **{d: coords[d] for d in dims[2:]}})})
# land-sea mask: spatially coherent "continents" by default (like the real one); 'random' = worst case for the
# wave-level bin skipping of wbx_det_binned
if len(sys.argv) > 3 and sys.argv[3] == 'random':
  land = np.random.default_rng(0).random((nlat, nlon)) > 0.7
else:
  land = (np.sin(np.deg2rad(lon) * 3)[None, :] * np.cos(np.deg2rad(lat) * 2.5)[:, None]) > 0.35
lsm = xr.DataArray(land, dims=('latitude', 'longitude'),
                   coords={'latitude': lat, 'longitude': lon})
metrics = {'rmse': deterministic.RMSE(), 'mse': deterministic.MSE(), 'bias': deterministic.Bias(),
           'acc': deterministic.ACC(clim), 'prediction_activity': deterministic.PredictionActivity(clim)}
ctx = _hip.default_context(0)
if os.environ.get('NREGIONS'):  # diagnostic: fewer regions (the bit path is forced)
  REGIONS = dict(list(REGIONS.items())[:int(os.environ['NREGIONS'])])
  engine.BITS_MIN_BINS = 1
pts = int(np.prod(shape))
for name, mode, agg in (
    ('no bins', 'auto', aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'],
                                               weigh_by=[weighting.GridAreaWeighting()], masked=True)),
    ('34 bins two-stage', 'never', aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'],
                                                          weigh_by=[weighting.GridAreaWeighting()],
                                                          bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)], masked=True)),
    ('34 bins fused', 'always', aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'],
                                                       weigh_by=[weighting.GridAreaWeighting()],
                                                       bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)], masked=True))):
  if os.environ.get('ONLY_FUSED') and mode != 'always':
    continue
  engine.BINNED_MODE = mode
  def step():
    pp = {'z': xr.DataArray(p_t, dims=dims, coords=coords)}
    tt = {'z': xr.DataArray(t_t, dims=dims, coords=coords)}
    return agg.aggregate_statistics(mb.compute_unique_statistics_for_all_metrics(metrics, pp, tt)).metric_values(metrics)
  for _ in range(3):
    out = step()
  t0 = time.perf_counter()
  n = 10
  for _ in range(n):
    out = step()
  ms = (time.perf_counter() - t0) / n * 1e3
  # the same chunks through the software pipeline of pipeline.evaluate_chunks (read-back deferred one chunk)
  t0 = time.perf_counter()
  with engine.deferred_results():
    prev = None
    for _ in range(n * 3):
      pp = {'z': xr.DataArray(p_t, dims=dims, coords=coords)}
      tt = {'z': xr.DataArray(t_t, dims=dims, coords=coords)}
      cur = agg.aggregate_statistics(mb.compute_unique_statistics_for_all_metrics(metrics, pp, tt))
      if prev is not None:
        out = prev.metric_values(metrics)
      prev = cur
    out = prev.metric_values(metrics)
  ms_pipe = (time.perf_counter() - t0) / (n * 3) * 1e3
  engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], 5
  step()
  k_ms = [e['ms'] for e in engine.S1_EVENT_LOG]
  engine.S1_EVENT_LOG = None
  print(f'{layout} ni={ni} {name:18s}: {ms:7.2f} ms/chunk sync, {ms_pipe:6.2f} pipelined ({pts * 12 / ms_pipe / 1e6:7.1f} GB/s algorithmic, {pts * 12 / 1e9:.2f} GB)  '
        f'stage-1/fused kernels {sum(k_ms):.2f} ms   acc[0]={np.asarray(out["acc.z"].values).reshape(-1)[0]:.4f}')
