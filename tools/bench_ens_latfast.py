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
t_t = torch.randn((nl, nlon, nlat), device='cuda') + 280
p_t = t_t[:, None] + torch.randn((nl, m, nlon, nlat), device='cuda')
metrics = {'crps': probabilistic.CRPSEnsemble(use_sort=True), 'ssr': probabilistic.UnbiasedSpreadSkillRatio()}
agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
nbytes = nl * nlat * nlon * (m + 1) * 4
for fold in (True, False):
  engine.FOLD_X_WEIGHTS = fold
  engine.clear_caches()

  def step():
    pp = {'v': xr.DataArray(p_t, dims=('lead_time', 'number', 'longitude', 'latitude'), coords=coords)}
    tt = {'v': xr.DataArray(t_t, dims=('lead_time', 'longitude', 'latitude'), coords=coords)}
    return agg.aggregate_statistics(mb.compute_unique_statistics_for_all_metrics(metrics, pp, tt)).metric_values(metrics)
  for _ in range(3):
    out = step()
  engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], 10
  out = step()
  ms = sum(e['ms'] for e in engine.S1_EVENT_LOG)
  engine.S1_EVENT_LOG = None
  print(f'fold={fold}: stage-1 {ms:6.3f} ms = {nbytes / ms / 1e6:7.1f} GB/s ({nbytes / ms / 1e6 / 80:4.1f} %)  crps={float(np.asarray(out["crps.v"].values).reshape(-1)[0]):.5f}')
