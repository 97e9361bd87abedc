"""One public-benchmark chunk through the fused binned kernel only (for rocprofv3 --pmc passes): a few launches, no
two-stage comparison.  usage: python tools/kbench_binned.py [lat_fastest|lon_fastest] [steps] [skipna] [mask]
(skipna: Aggregator(skipna=True) on targets with 1 % NaN -- the flavours with one count lane per statistic; mask: a `mask`
coordinate on the targets)"""
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

layout = sys.argv[1] if len(sys.argv) > 1 else 'lat_fastest'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
skipna, with_mask = 'skipna' in sys.argv[3:], 'mask' in sys.argv[3:]
nl, nlev, nlat, nlon = 12, 13, 721, 1440
lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
sp = ('longitude', 'latitude') if layout == 'lat_fastest' else ('latitude', 'longitude')
dims = ('init_time', 'lead_time', 'level') + sp
coords = {'init_time': np.array(['2020-01-01T00'], dtype='datetime64[ns]'),
          'lead_time': (np.arange(nl) * 12).astype('timedelta64[h]').astype('timedelta64[ns]'),
          'level': np.arange(nlev), 'latitude': lat, 'longitude': lon}
shape = tuple(len(coords[d]) for d in dims)
p_t, t_t = torch.randn(shape, device='cuda') + 280, torch.randn(shape, device='cuda') + 280
clim = xr.Dataset({'z': xr.DataArray(torch.randn((10, 4) + shape[2:], device='cuda') + 280,
                                     dims=('dayofyear', 'hour') + dims[2:],
                                     coords={'dayofyear': np.arange(1, 11), 'hour': np.array([0, 6, 12, 18]),
                                             **{d: coords[d] for d in dims[2:]}})})
land = (np.sin(np.deg2rad(lon) * 3)[None, :] * np.cos(np.deg2rad(lat) * 2.5)[:, None]) > 0.35
lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
metrics = {'acc': deterministic.ACC(clim), 'rmse': deterministic.RMSE()}
agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                             bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)], masked=True, skipna=skipna)
if skipna:
  t_t[torch.rand(shape, device='cuda') < 0.01] = float('nan')
tcoords = dict(coords)
if with_mask:
  tcoords['mask'] = (sp, np.asarray(np.random.default_rng(0).random((len(coords[sp[0]]), len(coords[sp[1]]))) > 0.1))
engine.BINNED_MODE = 'always'
engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], 5
for _ in range(steps):
  pp = {'z': xr.DataArray(p_t, dims=dims, coords=coords)}
  tt = {'z': xr.DataArray(t_t, dims=dims, coords=tcoords)}
  out = agg.aggregate_statistics(mb.compute_unique_statistics_for_all_metrics(metrics, pp, tt)).metric_values(metrics)
log = [e for e in engine.S1_EVENT_LOG if 'ms' in e]
engine.S1_EVENT_LOG = None
print('acc[0] =', float(np.asarray(out['acc.z'].values).reshape(-1)[0]))
for kind in sorted({e['kind'] for e in log}):
  ms = [e['ms'] for e in log if e['kind'] == kind]
  print(f'{kind:16s} {np.median(ms):.4f} ms per launch (median of {len(ms)})  ->  {np.prod(shape) * 12 / np.median(ms) / 1e9:.3f} TB/s on 12 B/point')
