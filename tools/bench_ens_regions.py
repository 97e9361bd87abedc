"""Probabilistic public-benchmark style chunk: CRPS + spread/skill of a 51-member ensemble, 8 lead times at 0.25 deg,
GridAreaWeighting, Regions(17) x land-sea (34 bins).  ms per chunk with and without bins, synchronous and pipelined."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from weatherbenchx_amd import aggregation, binning, engine, weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as mb, probabilistic
from wb_regions import REGIONS

m, nl, nlat, nlon = 51, 8, 721, 1440
LATFAST = len(sys.argv) > 1 and sys.argv[1] == 'lat_fastest'
SP = ('longitude', 'latitude') if LATFAST else ('latitude', 'longitude')
lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
coords = {'lead_time': (np.arange(nl) * 12).astype('timedelta64[h]').astype('timedelta64[ns]'), 'latitude': lat,
          'longitude': lon}
sshape = (nlon, nlat) if LATFAST else (nlat, nlon)
t_t = torch.randn((nl,) + sshape, device='cuda') + 280
p_t = t_t[:, None] + torch.randn((nl, m) + sshape, device='cuda')
land = (np.sin(np.deg2rad(lon) * 3)[None, :] * np.cos(np.deg2rad(lat) * 2.5)[:, None]) > 0.35
lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
metrics = {'crps': probabilistic.CRPSEnsemble(use_sort=True), 'ssr': probabilistic.UnbiasedSpreadSkillRatio()}
nbytes = nl * nlat * nlon * (m + 1) * 4
for name, agg in (('no bins', aggregation.Aggregator(reduce_dims=['latitude', 'longitude'],
                                                     weigh_by=[weighting.GridAreaWeighting()])),
                  ('34 bins', aggregation.Aggregator(reduce_dims=['latitude', 'longitude'],
                                                     weigh_by=[weighting.GridAreaWeighting()],
                                                     bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)]))):
  def launch():
    pp = {'v': xr.DataArray(p_t, dims=('lead_time', 'number') + SP, coords=coords)}
    tt = {'v': xr.DataArray(t_t, dims=('lead_time',) + SP, coords=coords)}
    return agg.aggregate_statistics(mb.compute_unique_statistics_for_all_metrics(metrics, pp, tt))
  for _ in range(3):
    out = launch().metric_values(metrics)
  n = 20
  t0 = time.perf_counter()
  for _ in range(n):
    out = launch().metric_values(metrics)
  ms = (time.perf_counter() - t0) / n * 1e3
  with engine.deferred_results():  # warm the pipelined route too (second launch stream: plans, scratch, pinned blocks)
    for _ in range(3):
      out = launch().metric_values(metrics)
  t0 = time.perf_counter()
  with engine.deferred_results():
    prev = None
    for _ in range(n):
      cur = launch()
      if prev is not None:
        out = prev.metric_values(metrics)
      prev = cur
    out = prev.metric_values(metrics)
  ms_pipe = (time.perf_counter() - t0) / n * 1e3
  print(f'ens M=51 x {nl} leads {name:8s}: {ms:6.2f} ms/chunk sync, {ms_pipe:6.2f} pipelined '
        f'({nbytes / ms_pipe / 1e6:7.1f} GB/s algorithmic, {nbytes / 1e9:.2f} GB)  crps[0]={float(np.asarray(out["crps.v"].values).reshape(-1)[0]):.4f}')
