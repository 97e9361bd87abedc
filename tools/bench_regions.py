"""Times the public-benchmark style aggregation (17 regions x {all, land} = 34 bins + area weights, masked) through
the drop-in API on device-resident data, against the same pass without bins."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from weatherbenchx_amd import _hip, aggregation, binning, weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as mb, deterministic

from wb_regions import REGIONS  # noqa: E402
ni, nl, nlev = (int(sys.argv[1]) if len(sys.argv) > 1 else 40), 4, 5
nlat, nlon = 721, 1440
lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
coords = {'init_time': np.datetime64('2020-01-01T00', 'ns') + np.arange(ni) * np.timedelta64(24, 'h'),
          'lead_time': (np.arange(nl) * 6).astype('timedelta64[h]').astype('timedelta64[ns]'),
          'level': np.arange(nlev), 'latitude': lat, 'longitude': lon}
dims = tuple(coords)
shape = tuple(len(coords[d]) for d in dims)
p_t = torch.randn(shape, device='cuda') + 280
t_t = torch.randn(shape, device='cuda') + 280
rng = np.random.default_rng(0)
lsm = xr.DataArray(rng.random((nlat, nlon)) > 0.7, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE(), 'bias': deterministic.Bias()}
ctx = _hip.default_context(0)


def run(agg, n=5):
  def step():
    pp = {'z': xr.DataArray(p_t, dims=dims, coords=coords)}
    tt = {'z': xr.DataArray(t_t, dims=dims, coords=coords)}
    return agg.aggregate_statistics(mb.compute_unique_statistics_for_all_metrics(metrics, pp, tt)).metric_values(metrics)
  t0 = time.perf_counter()
  out = step()
  first = time.perf_counter() - t0
  step()
  t0 = time.perf_counter()
  for _ in range(n):
    out = step()
  return first * 1e3, (time.perf_counter() - t0) / n * 1e3, out


pts = int(np.prod(shape))
for name, agg in (
    ('area weights only', aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'],
                                                 weigh_by=[weighting.GridAreaWeighting()])),
    ('17 regions', aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'],
                                          weigh_by=[weighting.GridAreaWeighting()], bin_by=[binning.Regions(REGIONS)])),
    ('17 regions x land (34 bins), masked', aggregation.Aggregator(
        reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
        bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)], masked=True)),
):
  first, ms, out = run(agg)
  print(f'{name:40s} first call {first:9.1f} ms   steady {ms:8.2f} ms/step   {pts * 8 / ms / 1e6:8.1f} GB/s algorithmic '
        f'(inits={ni})  rmse[0,0]={np.asarray(out["rmse.z"].values).reshape(-1)[0]:.5f}')
