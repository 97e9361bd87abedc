"""cProfile of the host-side (Python) part of one bench step, to find per-step overhead outside the kernels."""
import cProfile
import pstats
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from weatherbenchx_amd import aggregation, weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as mb, deterministic

ni, nl, nlev, nlat, nlon = 8, 10, 5, 721, 1440
lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
coords = {'init_time': np.datetime64('2020-01-01T00', 'ns') + np.arange(ni) * np.timedelta64(24, 'h'),
          'lead_time': (np.arange(nl) * 6).astype('timedelta64[h]').astype('timedelta64[ns]'),
          'level': np.arange(nlev), 'latitude': lat, 'longitude': lon}
dims = tuple(coords)
shape = tuple(len(coords[d]) for d in dims)
dev = torch.device('cuda')
p_t, t_t = torch.randn(shape, device=dev) + 280, torch.randn(shape, device=dev) + 280
clim_t = torch.randn((ni + 4, 4) + shape[2:], device=dev) + 280
clim = xr.Dataset({'z': xr.DataArray(clim_t, dims=('dayofyear', 'hour') + dims[2:], coords={
    'dayofyear': np.arange(1, ni + 5), 'hour': np.array([0, 6, 12, 18]), **{d: coords[d] for d in dims[2:]}})})
metrics = {'rmse': deterministic.RMSE(), 'mse': deterministic.MSE(), 'mae': deterministic.MAE(),
           'bias': deterministic.Bias(), 'acc': deterministic.ACC(clim),
           'prediction_activity': deterministic.PredictionActivity(clim)}
agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])


def step():
  pp = {'z': xr.DataArray(p_t, dims=dims, coords=coords)}
  tt = {'z': xr.DataArray(t_t, dims=dims, coords=coords)}
  stats = mb.compute_unique_statistics_for_all_metrics(metrics, pp, tt)
  return agg.aggregate_statistics(stats).metric_values(metrics)


for _ in range(3):
  step()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
  step()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(40)
