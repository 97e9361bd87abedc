"""The public benchmark's probabilistic chunk on wbx_ens_binned: CRPS + spread/skill + ensemble-mean RMSE of a 51-member
ensemble, NL lead times at 0.25 deg, GridAreaWeighting, Regions(17) x land-sea (34 bins), masked=True [with a (lat, lon)
validity mask].  Kernel time per launch (HIP event marks around each launch of the pipelined loop), ms per chunk, fraction of
the HBM peak on the algorithmic bytes (M + 1) * 4 per point.
usage: bench_ens_binned.py [lon_fastest|lat_fastest|ifs] [mask|nanmask] [skipna] [nl=8] [m=51]
  nanmask: NaN targets (a polar hole that grows with the lead time) + add_nan_mask_to_data: the per-point mask with a lead_time
           stride (data_loaders/base.py:25-56); skipna: Aggregator(skipna=True) on NaN targets (no mask coordinate unless asked)"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from weatherbenchx_amd import aggregation, binning, engine, weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as mb, deterministic, probabilistic, wrappers
from wb_regions import REGIONS

args = sys.argv[1:]
layout = next((a for a in args if a in ('lon_fastest', 'lat_fastest', 'ifs')), 'lon_fastest')
with_mask = 'mask' in args
nan_mask, skipna = 'nanmask' in args, 'skipna' in args
no_bins = 'nobins' in args  # the un-binned pipelined kernel on the same members, for reference
nl = int(next((a[3:] for a in args if a.startswith('nl=')), 8))
m = int(next((a[2:] for a in args if a.startswith('m=')), 51))
nlat, nlon = 721, 1440
lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
sp = ('latitude', 'longitude') if layout == 'lon_fastest' else ('longitude', 'latitude')
sshape = tuple({'latitude': nlat, 'longitude': nlon}[d] for d in sp)
coords = {'lead_time': (np.arange(nl) * 12).astype('timedelta64[h]').astype('timedelta64[ns]'), 'latitude': lat, 'longitude': lon,
          'init_time': np.array(['2020-01-01T00'], dtype='datetime64[ns]')}
if layout == 'ifs':
  pdims, tdims = ('init_time', 'number', 'lead_time') + sp, ('init_time', 'lead_time') + sp
  t_t = torch.randn((1, nl) + sshape, device='cuda') + 280
  p_t = t_t[:, None] + torch.randn((1, m, nl) + sshape, device='cuda')
  reduce_dims = ['init_time', 'latitude', 'longitude']
else:
  pdims, tdims = ('lead_time', 'number') + sp, ('lead_time',) + sp
  t_t = torch.randn((nl,) + sshape, device='cuda') + 280
  p_t = t_t[:, None] + torch.randn((nl, m) + sshape, device='cuda')
  reduce_dims = ['latitude', 'longitude']
land = (np.sin(np.deg2rad(lon) * 3)[None, :] * np.cos(np.deg2rad(lat) * 2.5)[:, None]
        + 0.3 * np.sin(np.deg2rad(lon) * 17)[None, :] * np.sin(np.deg2rad(lat) * 13)[:, None]) > 0.35
lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
valid = ~((np.abs(lat)[:, None] > 80) & (np.cos(np.deg2rad(lon) * 5)[None, :] > 0.2))
mask_da = None
if with_mask:
  mv = valid if sp == ('latitude', 'longitude') else np.ascontiguousarray(valid.T)
  mask_da = xr.DataArray(torch.as_tensor(mv, device='cuda'), dims=sp, coords={'latitude': lat, 'longitude': lon})
if nan_mask or skipna:
  from weatherbenchx_amd import data as wdata
  holes = np.stack([(np.abs(lat)[:, None] > 80 - 2 * l) & (np.cos(np.deg2rad(lon) * (5 + l))[None, :] > 0.2) for l in range(nl)])
  hv = holes if sp == ('latitude', 'longitude') else np.ascontiguousarray(np.swapaxes(holes, 1, 2))
  (t_t[0] if layout == 'ifs' else t_t)[torch.as_tensor(hv, device='cuda')] = float('nan')
  if nan_mask:
    mask_da = wdata.add_nan_mask_to_data({'v': xr.DataArray(t_t, dims=tdims, coords={k: v for k, v in coords.items() if k in tdims})})['v'].coords['mask']
metrics = {'crps': probabilistic.CRPSEnsemble(use_sort=True), 'ssr': probabilistic.UnbiasedSpreadSkillRatio(),
           'unbiased_mean_rmse': probabilistic.UnbiasedEnsembleMeanRMSE(),
           'mean_rmse': wrappers.WrappedMetric(deterministic.RMSE(), [wrappers.EnsembleMean(which='predictions')])}
agg = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=[weighting.GridAreaWeighting()],
                             bin_by=None if no_bins else [binning.Regions(REGIONS, land_sea_mask=lsm)], masked=True,
                             skipna=skipna)
nbytes = nl * nlat * nlon * (m + 1) * 4


def launch():
  pp = {'v': xr.DataArray(p_t, dims=pdims, coords={k: v for k, v in coords.items() if k in pdims})}
  t = xr.DataArray(t_t, dims=tdims, coords={k: v for k, v in coords.items() if k in tdims})
  if mask_da is not None:
    t = t.assign_coords(mask=mask_da)
  return agg.aggregate_statistics(mb.compute_unique_statistics_for_all_metrics(metrics, pp, {'v': t}))


with engine.deferred_results():
  for _ in range(3):
    out = launch().metric_values(metrics)
n = 20
engine.S1_EVENT_LOG, engine.S1_EVENT_MARKS = [], True
t0 = time.perf_counter()
with engine.deferred_results():
  prev = None
  for _ in range(n):
    cur = launch()
    if prev is not None:
      out = prev.metric_values(metrics)
    prev = cur
  out = prev.metric_values(metrics)
ms_chunk = (time.perf_counter() - t0) / n * 1e3
log = engine.resolve_event_marks(engine.S1_EVENT_LOG)
engine.S1_EVENT_LOG, engine.S1_EVENT_MARKS = None, False
kinds = sorted({e['kind'] for e in log})
per_kind = {k: float(np.mean([e['ms'] for e in log if e['kind'] == k])) for k in kinds}
launches = {k: sum(e['kind'] == k for e in log) / n for k in kinds}
kernel_ms = sum(per_kind[k] * launches[k] for k in kinds)
print(json.dumps({'layout': layout, 'bins': 0 if no_bins else 34, 'lib': os.path.basename(os.environ.get('WBX_LIBRARY_PATH', 'libwbx_hip.so')), 'mask': 'nan' if nan_mask else with_mask, 'skipna': skipna, 'M': m, 'leads': nl, 'GB': round(nbytes / 1e9, 3),
                  'ms_per_chunk': round(ms_chunk, 4), 'launches_per_chunk': launches, 'ms_per_launch': {k: round(v, 4) for k, v in per_kind.items()},
                  'kernel_ms_per_chunk': round(kernel_ms, 4),
                  'frac_of_hbm_peak_per_launch': {k: round(nbytes / (v * 1e-3) / 8e12, 4) for k, v in per_kind.items()},
                  'rows': os.environ.get('WBX_ENS_ATOMS_ROWS', 'default'), 'ens_binned': os.environ.get('WBX_ENS_BINNED', '1'),
                  'crps_global': float(np.asarray(out['crps.v'].values).reshape(-1)[0])}))
