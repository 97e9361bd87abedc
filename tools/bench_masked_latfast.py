"""Latitude-fastest chunk with a (lon, lat) validity mask on the targets (SST-style NaNs, masked=True): the x-kept
generic kernel with mask loads vs the same chunk without a mask (LDS plane mode)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from weatherbenchx_amd import aggregation, engine, weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as mb, deterministic

ni, nl, nlev, nlat, nlon = 8, 10, 5, 721, 1440
lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
dims = ('init_time', 'lead_time', 'level', 'longitude', 'latitude')
coords = {'init_time': np.datetime64('2020-01-01T00', 'ns') + np.arange(ni) * np.timedelta64(24, 'h'),
          'lead_time': (np.arange(nl) * 6).astype('timedelta64[h]').astype('timedelta64[ns]'),
          'level': np.arange(nlev), 'latitude': lat, 'longitude': lon}
shape = tuple(len(coords[d]) for d in dims)
p_t, t_t = torch.randn(shape, device='cuda') + 280, torch.randn(shape, device='cuda') + 280
mask = np.random.default_rng(0).random((nlon, nlat)) > 0.3
metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE(), 'bias': deterministic.Bias()}
agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                             masked=True)
nbytes = int(np.prod(shape)) * 8
for name, with_mask in (('no mask', False), ('mask(lon,lat)', True)):
  def step():
    pp = xr.DataArray(p_t, dims=dims, coords=coords)
    tt = xr.DataArray(t_t, dims=dims, coords=coords)
    if with_mask:
      tt.coords['mask'] = xr.DataArray(mask, dims=('longitude', 'latitude'))
    return agg.aggregate_statistics(mb.compute_unique_statistics_for_all_metrics(metrics, {'z': pp}, {'z': tt}))
  for _ in range(3):
    step().metric_values(metrics)
  engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], 10
  out = step().metric_values(metrics)
  log = engine.S1_EVENT_LOG
  engine.S1_EVENT_LOG = None
  ms = sum(e['ms'] for e in log)
  print(f'{name:14s}: stage-1 {ms:6.3f} ms = {nbytes / ms / 1e6:7.1f} GB/s ({nbytes / ms / 1e6 / 80:4.1f} % of 8 TB/s)  '
        f'x_kept={log[0].get("x_kept")} plane_rows={log[0].get("plane_rows")}  rmse={float(np.asarray(out["rmse.z"].values).reshape(-1)[0]):.4f}')
