"""Times wbx_det_binned on the public-benchmark chunk (HIP events, 10 launches per pair) for both layouts and two
land-sea masks (smooth continents / random points); prints one JSON line.  The kernel variant comes from the
environment: WBX_BINNED_ATOMS=0|1, WBX_ATOMS_NT=0|1, WBX_LIBRARY_PATH (A/B builds), WBX_KBENCH_LAYOUTS=lat_fastest.  usage: python tools/kbench_binned_ab.py [tag]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from weatherbenchx_amd import aggregation, binning, engine, weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as mb, deterministic
from wb_regions import REGIONS

nl, nlev, nlat, nlon = 12, 13, 721, 1440
lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
out = {'tag': sys.argv[1] if len(sys.argv) > 1 else '', 'atoms': os.environ.get('WBX_BINNED_ATOMS', '1'),
       'nt': os.environ.get('WBX_ATOMS_NT', 'auto')}
engine.BINNED_MODE = 'always'
for layout in os.environ.get('WBX_KBENCH_LAYOUTS', 'lon_fastest,lat_fastest').split(','):
  sp = ('longitude', 'latitude') if layout == 'lat_fastest' else ('latitude', 'longitude')
  dims = ('init_time', 'lead_time', 'level') + sp
  coords = {'init_time': np.array(['2020-01-01T00'], dtype='datetime64[ns]'),
            'lead_time': (np.arange(nl) * 12).astype('timedelta64[h]').astype('timedelta64[ns]'),
            'level': np.arange(nlev), 'latitude': lat, 'longitude': lon}
  shape = tuple(len(coords[d]) for d in dims)
  g = torch.Generator(device='cuda').manual_seed(1)
  p_t = torch.randn(shape, device='cuda', generator=g) + 280
  t_t = torch.randn(shape, device='cuda', generator=g) + 280
  clim = xr.Dataset({'z': xr.DataArray(torch.randn((10, 4) + shape[2:], device='cuda', generator=g) + 280,
                                       dims=('dayofyear', 'hour') + dims[2:],
                                       coords={'dayofyear': np.arange(1, 11), 'hour': np.array([0, 6, 12, 18]),
                                               **{d: coords[d] for d in dims[2:]}})})
  for mask_kind in ('smooth', 'random'):
    if mask_kind == 'smooth':
      land = (np.sin(np.deg2rad(lon) * 3)[None, :] * np.cos(np.deg2rad(lat) * 2.5)[:, None]) > 0.35
    else:
      land = np.random.default_rng(3).random((nlat, nlon)) > 0.7
    lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
    metrics = {'acc': deterministic.ACC(clim), 'rmse': deterministic.RMSE()}
    agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                                 bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)], masked=True)

    def run():
      pp = {'z': xr.DataArray(p_t, dims=dims, coords=coords)}
      tt = {'z': xr.DataArray(t_t, dims=dims, coords=coords)}
      return agg.aggregate_statistics(mb.compute_unique_statistics_for_all_metrics(metrics, pp, tt)).metric_values(metrics)
    for _ in range(2):
      res = run()
    engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], 10
    for _ in range(3):
      res = run()
    ms = [e['ms'] for e in engine.S1_EVENT_LOG if e['kind'] == 'det_binned']
    engine.S1_EVENT_LOG = None
    points = int(np.prod(shape))
    out[f'{layout}/{mask_kind}'] = {'ms': round(float(np.mean(ms)), 4),
                                    'frac_hbm': round(points * 12 / (float(np.mean(ms)) * 1e-3) / 8e12, 4),
                                    'acc_sum': float(np.nansum(res['acc.z'].values)), 'rmse_sum': float(np.nansum(res['rmse.z'].values))}
print(json.dumps(out))
