"""Times the zonal-spectrum launch alone (HIP-side wall clock over back-to-back steps) for one field repeated and for two
fields alternating; usage: python tools/kbench_spectrum.py [nt] [layout]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from weatherbenchx_amd import _hip, aggregation, spectra, weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as mb

nt, nlev, nlat, nlon = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), 37, 721, 1440
layout = sys.argv[2] if len(sys.argv) > 2 else 'lon_fastest'
lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
sp = ('latitude', 'longitude') if layout == 'lon_fastest' else ('longitude', 'latitude')
dims = ('lead_time', 'level') + sp
coords = {'lead_time': (np.arange(nt) * 6).astype('timedelta64[h]').astype('timedelta64[ns]'), 'level': np.arange(nlev),
          'latitude': lat, 'longitude': lon}
shape = tuple(len(coords[d]) for d in dims)
fields = [torch.randn(shape, device='cuda') + 280 for _ in range(2)]
torch.cuda.synchronize()
ctx = _hip.default_context(0)
agg = aggregation.Aggregator(reduce_dims=['lead_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
one = {'spec_p': spectra.ZonalPowerSpectrum('predictions')}
two = {'spec_p': spectra.ZonalPowerSpectrum('predictions'), 'spec_t': spectra.ZonalPowerSpectrum('targets')}


def step(metrics, a, b):
  pp = {'z': xr.DataArray(a, dims=dims, coords=coords)}
  tt = {'z': xr.DataArray(b, dims=dims, coords=coords)}
  return agg.aggregate_statistics(mb.compute_unique_statistics_for_all_metrics(metrics, pp, tt))


pts = int(np.prod(shape))
for name, metrics, nf in (('one field, repeated', one, 1), ('two fields, alternating', two, 2)):
  for _ in range(4):
    out = step(metrics, fields[0], fields[1])
  ctx.synchronize()
  t0 = time.perf_counter()
  n = 40
  for _ in range(n):
    out = step(metrics, fields[0], fields[1])
  ctx.synchronize()
  ms = (time.perf_counter() - t0) / n * 1e3
  print(f'{layout} {name:26s} {ms:7.3f} ms/step = {ms / nf:7.3f} ms per field, {pts * 4 * nf / ms / 1e6:8.1f} GB/s')
