"""IFS-ENS style chunk (lead, number, longitude, latitude) with GridAreaWeighting through the API: latitude weights
folded into stage 1 (x summed) vs kept for stage 2 (x-kept kernel)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from weatherbenchx_amd import aggregation, engine, weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as mb, probabilistic

m, nl, nlat, nlon = 51, 8, 721, 1440
lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
coords = {'lead_time': (np.arange(nl) * 12).astype('timedelta64[h]').astype('timedelta64[ns]'), 'latitude': lat, 'longitude': lon}
LONFAST = len(sys.argv) > 1 and sys.argv[1] == 'lon_fastest'  # the same chunk stored (..., latitude, longitude), for reference
sp = ('latitude', 'longitude') if LONFAST else ('longitude', 'latitude')
shp = (nlat, nlon) if LONFAST else (nlon, nlat)
t_t = torch.randn((nl,) + shp, device='cuda') + 280
p_t = t_t[:, None] + torch.randn((nl, m) + shp, device='cuda')
metrics = {'crps': probabilistic.CRPSEnsemble(use_sort=True), 'ssr': probabilistic.UnbiasedSpreadSkillRatio()}


class _LonWeights(weighting.Weighting):
  """Diagnostic: weights on longitude, so the flat fold also runs on longitude-fastest data."""

  def weights(self, statistic):
    return xr.DataArray(1.0 + 0.001 * np.arange(nlon), dims=('longitude',), coords={'longitude': lon})


wts = _LonWeights() if (len(sys.argv) > 2 and sys.argv[2] == 'lonw') else weighting.GridAreaWeighting()
agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[wts])
nbytes = nl * nlat * nlon * (m + 1) * 4
for fold in (True, False, True, False):
  engine.FOLD_X_WEIGHTS = fold
  engine.clear_caches()

  def step():
    pp = {'v': xr.DataArray(p_t, dims=('lead_time', 'number') + sp, coords=coords)}
    tt = {'v': xr.DataArray(t_t, dims=('lead_time',) + sp, coords=coords)}
    return agg.aggregate_statistics(mb.compute_unique_statistics_for_all_metrics(metrics, pp, tt)).metric_values(metrics)
  for _ in range(3):
    out = step()
  engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], 10
  out = step()
  ms = sum(e['ms'] for e in engine.S1_EVENT_LOG)
  engine.S1_EVENT_LOG = None
  print(f'fold={fold}: stage-1 {ms:6.3f} ms = {nbytes / ms / 1e6:7.1f} GB/s ({nbytes / ms / 1e6 / 80:4.1f} %)  crps={float(np.asarray(out["crps.v"].values).reshape(-1)[0]):.5f}')
