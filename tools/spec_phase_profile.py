"""Phase profile of the 1440-point spectrum kernel: runs configs[3]'s field through the s_memtime-stamped instantiation
(WBX_SPECTRUM_PROF) and prints the average shader-clock cycles wave 0 of each block spends per row pair and phase."""
import os
# the knock-out / phase-stamped kernels live in the diagnostic build only (make -C weatherbenchx_amd/csrc diag)
_diag = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'weatherbenchx_amd', 'libwbx_hip_diag.so')
if os.path.exists(_diag):
  os.environ.setdefault('WBX_LIBRARY_PATH', _diag)
import sys
import tempfile
path = os.path.join(tempfile.gettempdir(), 'wbx_spec_prof.txt')
if os.path.exists(path):
  os.remove(path)
os.environ['WBX_SPECTRUM_PROF'] = path
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from weatherbenchx_amd import aggregation, spectra, weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as mb

nt, nlev, nlat, nlon = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), 37, 721, 1440
lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
dims = ('lead_time', 'level', 'latitude', 'longitude')
coords = {'lead_time': (np.arange(nt) * 6).astype('timedelta64[h]').astype('timedelta64[ns]'), 'level': np.arange(nlev),
          'latitude': lat, 'longitude': lon}
p_t = torch.randn((nt, nlev, nlat, nlon), device='cuda') + 280
metrics = {'spec_p': spectra.ZonalPowerSpectrum('predictions')}
agg = aggregation.Aggregator(reduce_dims=['lead_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
for _ in range(4):
  pp = {'z': xr.DataArray(p_t, dims=dims, coords=coords)}
  out = agg.aggregate_statistics(mb.compute_unique_statistics_for_all_metrics(metrics, pp, pp)).metric_values(metrics)
rows = np.loadtxt(path, dtype=np.float64)[1:]
names = ['pairs', 'wait rows', 'pass 1', 'transpose 1', 'pass 2', 'transpose 2', 'pass 3', 'mirror + unpack']
tot = rows.sum(0)
per_pair = tot[1:8] / tot[0]
nblocks = 256
last = rows[-1]
print(f'wave lifetime {last[10] / nblocks / 100:.1f} us on average; first start -> last start {(last[13] - last[11]) / 100:.1f} us, '
      f'first start -> last end {(last[12] - last[11]) / 100:.1f} us (last launch)')
print('wave lifetime by team (us):', ' '.join(f'{v / nblocks / 100:.0f}' for v in last[14:26]))
print(f'prologue {tot[8] / len(rows) / nblocks:9.0f} cycles, closing flush {tot[9] / len(rows) / nblocks:9.0f} cycles per profiled wave (assuming {nblocks} blocks)')
for n, v in zip(names[1:], per_pair):
  print(f'{n:18s} {v:9.0f} cycles per pair ({100 * v / per_pair.sum():5.1f} %)')
print(f'{"total":18s} {per_pair.sum():9.0f} cycles per pair, {tot[0] / len(rows):.0f} pairs per profiled wave')
