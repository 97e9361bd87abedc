"""configs[3] side measurement: zonal spectra of predictions and targets (batched rocFFT along longitude + |F|^2
reduction, area-weighted mean over latitude and time) plus the deterministic suite, 37 levels, device-resident."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from weatherbenchx_amd import _hip, aggregation, spectra, weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as mb, deterministic

nt, nlev, nlat, nlon = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), 37, 721, 1440
layout = sys.argv[2] if len(sys.argv) > 2 else 'lon_fastest'
lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
sp = ('latitude', 'longitude') if layout == 'lon_fastest' else ('longitude', 'latitude')
dims = ('lead_time', 'level') + sp
coords = {'lead_time': (np.arange(nt) * 6).astype('timedelta64[h]').astype('timedelta64[ns]'), 'level': np.arange(nlev),
          'latitude': lat, 'longitude': lon}
shape = tuple(len(coords[d]) for d in dims)
p_t, t_t = torch.randn(shape, device='cuda') + 280, torch.randn(shape, device='cuda') + 280
torch.cuda.synchronize()
ctx = _hip.default_context(0)
spec_metrics = {'spec_p': spectra.ZonalPowerSpectrum('predictions'), 'spec_t': spectra.ZonalPowerSpectrum('targets')}
det_metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE(), 'bias': deterministic.Bias()}
agg_s = aggregation.Aggregator(reduce_dims=['lead_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
agg_d = aggregation.Aggregator(reduce_dims=['lead_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])


def step(metrics, agg):
  pp = {'z': xr.DataArray(p_t, dims=dims, coords=coords)}
  tt = {'z': xr.DataArray(t_t, dims=dims, coords=coords)}
  return agg.aggregate_statistics(mb.compute_unique_statistics_for_all_metrics(metrics, pp, tt)).metric_values(metrics)


pts = int(np.prod(shape))
for name, metrics, agg, nbytes in (('spectra(p)+spectra(t)', spec_metrics, agg_s, pts * 8),
                                   ('rmse+mae+bias', det_metrics, agg_d, pts * 8)):
  for _ in range(4):
    out = step(metrics, agg)
  ctx.synchronize()
  t0 = time.perf_counter()
  n = 20
  for _ in range(n):
    out = step(metrics, agg)
  ctx.synchronize()
  ms = (time.perf_counter() - t0) / n * 1e3
  k = list(out)[0]
  print(f'{layout} T={nt} x {nlev} levels: {name:24s} {ms:8.2f} ms/step  {nbytes / ms / 1e6:8.1f} GB/s algorithmic '
        f'({pts * len(metrics) / ms / 1e6:.1f} M evals/ms)  {k} shape={out[k].shape}')
spec = out if 'spec_p.z' in out else step(spec_metrics, agg_s)
s = np.asarray(spec['spec_p.z'].values)
print('Parseval check (N(280,1) field): sum_k S_k =', float(s[0].sum()), ' expected ~', 280.0 ** 2 + 1.0)
