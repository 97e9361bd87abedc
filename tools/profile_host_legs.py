"""cProfile of the host-side (Python) part of one step of the public-benchmark chunk and of the spectrum leg (steps pipelined
like bench.py runs them: deferred results, one step in flight)."""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from weatherbenchx_amd import _hip, aggregation, binning, engine, spectra, weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as mb, deterministic
from wb_regions import REGIONS

which = sys.argv[1] if len(sys.argv) > 1 else 'public'
nlat, nlon = 721, 1440
lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
ctx = _hip.default_context(0)
if which == 'public':
  nl, nlev = 12, 13
  dims = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
  coords = {'init_time': np.array(['2020-01-01T00'], dtype='datetime64[ns]'),
            'lead_time': (np.arange(nl) * 12).astype('timedelta64[h]').astype('timedelta64[ns]'),
            'level': np.arange(nlev), 'latitude': lat, 'longitude': lon}
  shape = tuple(len(coords[d]) for d in dims)
  p_t, t_t = torch.randn(shape, device='cuda') + 280, torch.randn(shape, device='cuda') + 280
  clim = xr.Dataset({'z': xr.DataArray(torch.randn((10, 4) + shape[2:], device='cuda') + 280, dims=('dayofyear', 'hour') + dims[2:],
                                       coords={'dayofyear': np.arange(1, 11), 'hour': np.array([0, 6, 12, 18]),
                                               **{d: coords[d] for d in dims[2:]}})})
  land = (np.sin(np.deg2rad(lon) * 3)[None, :] * np.cos(np.deg2rad(lat) * 2.5)[:, None]) > 0.35
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  metrics = {'acc': deterministic.ACC(clim), 'rmse': deterministic.RMSE(), 'mse': deterministic.MSE(), 'bias': deterministic.Bias(),
             'activity': deterministic.PredictionActivity(clim)}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)], masked=True)
else:
  nt, nlev = 8, 37
  dims = ('lead_time', 'level', 'latitude', 'longitude')
  coords = {'lead_time': (np.arange(nt) * 6).astype('timedelta64[h]').astype('timedelta64[ns]'), 'level': np.arange(nlev),
            'latitude': lat, 'longitude': lon}
  shape = tuple(len(coords[d]) for d in dims)
  p_t, t_t = torch.randn(shape, device='cuda') + 280, torch.randn(shape, device='cuda') + 280
  metrics = {'spec_p': spectra.ZonalPowerSpectrum('predictions'), 'spec_t': spectra.ZonalPowerSpectrum('targets')}
  agg = aggregation.Aggregator(reduce_dims=['lead_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])


def launch():
  pp = {'z': xr.DataArray(p_t, dims=dims, coords=coords)}
  tt = {'z': xr.DataArray(t_t, dims=dims, coords=coords)}
  return agg.aggregate_statistics(mb.compute_unique_statistics_for_all_metrics(metrics, pp, tt))


def run(n):
  with engine.deferred_results():
    pending = None
    for _ in range(n):
      cur = launch()
      if pending is not None:
        pending.metric_values(metrics)
      pending = cur
    pending.metric_values(metrics)


run(5)
ctx.synchronize()
t0 = time.perf_counter()
run(200)
ctx.synchronize()
print(f'{which}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per step (wall)')
pr = cProfile.Profile()
pr.enable()
run(200)
pr.disable()
ctx.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)
st.sort_stats('tottime').print_stats(18)
