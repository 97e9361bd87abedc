"""wbx_zonal_spectrum through the C ABI alone, HIP events on the launch stream: N back-to-back launches per event pair, with
and without idle gaps between the pairs (is the kernel slower in a sustained loop than in isolation?)."""
import os
# the knock-out / phase-stamped kernels live in the diagnostic build only (make -C weatherbenchx_amd/csrc diag)
_diag = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'weatherbenchx_amd', 'libwbx_hip_diag.so')
if os.path.exists(_diag):
  os.environ.setdefault('WBX_LIBRARY_PATH', _diag)
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from weatherbenchx_amd import _hip

nt, nlev, nlat, nlon = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), 37, 721, 1440
layout = sys.argv[2] if len(sys.argv) > 2 else 'lon_fastest'
ctx = _hip.default_context(0)
lib = ctx.lib
nrows = nt * nlev * nlat
fields = [torch.randn((nrows, nlon), device='cuda') + 280 for _ in range(2)]
group = torch.arange(nrows, device='cuda', dtype=torch.int32) // nlat % nlev
group = group.to(torch.int32).contiguous()
scale = torch.full((nrows,), 1.0 / nrows, device='cuda', dtype=torch.float64)
power = torch.zeros((nlev, nlon // 2 + 1), device='cuda', dtype=torch.float64)
torch.cuda.synchronize()


import ctypes
slab_off = (np.arange(nt * nlev, dtype=np.int64) * nlat * nlon)
if len(sys.argv) > 3 and sys.argv[3] == 'sorted':  # the slabs of one group (level) next to each other, as spectra.py lists them
  order = np.argsort(np.arange(nt * nlev) % nlev, kind='stable')
  slab_off = np.ascontiguousarray(slab_off[order])
  group = torch.from_numpy(np.repeat((np.arange(nt * nlev) % nlev)[order], nlat).astype(np.int32)).cuda()


def launch(f):
  if layout == 'lat_fastest':  # [slab][lon][lat]: rows adjacent, longitude strided by nlat
    _hip.check(lib.wbx_zonal_spectrum_slabs(ctx.handle, f.data_ptr(), nlat, 1, nlat, nt * nlev, slab_off.ctypes.data_as(ctypes.c_void_p),
                                            nlon, group.data_ptr(), scale.data_ptr(), nlev, 0, power.data_ptr()), 'spectrum')
    return
  _hip.check(lib.wbx_zonal_spectrum(ctx.handle, f.data_ptr(), 1, nlon, nrows, nlon, group.data_ptr(), scale.data_ptr(), nlev, 0,
                                    power.data_ptr()), 'spectrum')


for f in fields:
  launch(f)
ctx.synchronize()
gb = nrows * nlon * 4 / 1e9
for reps, gap in ((1, 0.0), (1, 0.003), (2, 0.0), (10, 0.0), (50, 0.0), (200, 0.0), (1, 0.003)):
  ms = []
  for it in range(8 if reps < 50 else 3):
    if gap:
      ctx.synchronize()
      time.sleep(gap)
    ctx.timer_start()
    for i in range(reps):
      launch(fields[i & 1])
    ms.append(ctx.timer_stop() / reps)
  print(f'{reps:4d} launches per event pair, gap {gap * 1e3:.0f} ms: {np.mean(ms):.4f} ms per launch (min {np.min(ms):.4f}) = {gb / np.mean(ms):.2f} TB/s')
