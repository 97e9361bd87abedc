"""Indicator statistics on an ensemble chunk (51 members x 8 leads at 0.25 deg): RankHistogram (52 ranks) and
EnsembleErrorExceedance (4 thresholds), area-weighted; and ErrorExceedance on a deterministic chunk."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from weatherbenchx_amd import aggregation, weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as mb, deterministic, probabilistic

m, nl, nlat, nlon = 51, 8, 721, 1440
LATFAST = len(sys.argv) > 1 and sys.argv[1] == 'lat_fastest'
SP = ('longitude', 'latitude') if LATFAST else ('latitude', 'longitude')
lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
coords = {'lead_time': (np.arange(nl) * 12).astype('timedelta64[h]').astype('timedelta64[ns]'), 'latitude': lat,
          'longitude': lon}
sshape = (nlon, nlat) if LATFAST else (nlat, nlon)
t_t = torch.randn((nl,) + sshape, device='cuda') + 280
p_t = t_t[:, None] + torch.randn((nl, m) + sshape, device='cuda')
agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
nbytes = nl * nlat * nlon * (m + 1) * 4
for name, metrics, pd in (('RankHistogram', {'rank': probabilistic.RankHistogram()}, True),
                          ('EnsembleErrorExceedance x4', {'exc': probabilistic.EnsembleErrorExceedance([0.5, 1, 2, 3])}, True),
                          ('ErrorExceedance x4 (member 0)', {'exc': deterministic.ErrorExceedance([0.5, 1, 2, 3])}, False)):
  def step():
    pp = xr.DataArray(p_t if pd else p_t[:, 0], dims=(('lead_time', 'number') + SP) if pd else (('lead_time',) + SP), coords=coords)
    tt = xr.DataArray(t_t, dims=('lead_time',) + SP, coords=coords)
    return agg.aggregate_statistics(mb.compute_unique_statistics_for_all_metrics(metrics, {'v': pp}, {'v': tt})).metric_values(metrics)
  for _ in range(2):
    out = step()
  n = 10
  t0 = time.perf_counter()
  for _ in range(n):
    out = step()
  ms = (time.perf_counter() - t0) / n * 1e3
  nb = nbytes if pd else nl * nlat * nlon * 8
  k = next(iter(out))
  print(f'{name:32s}: {ms:7.2f} ms/chunk ({nb / ms / 1e6:7.1f} GB/s algorithmic)  {k} -> {np.asarray(out[k].values).reshape(-1)[:3]}')
