"""Hashes of what the spectrum entry points return for seeded inputs -- to compare two builds of the library bit for bit
(WBX_LIBRARY_PATH selects one), and the source of tests/golden/spectra_bits.json (`--write`; tests/test_gpu_round6.py compares): the plain spectrum of a field, the fused det + spectra sweep (folded and not), accumulate mode,
a group table whose groups span many teams and one whose records overflow a block's list.
usage: python tools/spectra_bits.py [--write]"""
import ctypes as C
import hashlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import numpy as np
from weatherbenchx_amd import _hip, engine, planner
from weatherbenchx_amd import xarray_lite as xr

ctx = _hip.default_context(0)
lib = ctx.lib
nlead, nlev, nlat, nlon = 3, 5, 181, 1440
dims = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
shape = (1, nlead, nlev, nlat, nlon)
gen = np.random.default_rng(1234)  # (host-side generator: the inputs do not depend on the torch build)
arrs = [xr.DataArray((gen.standard_normal(shape) * (3 if i < 2 else 10) + 280).astype(np.float32), dims=dims) for i in range(3)]
devs = [engine._to_device(ctx, a, _hip.F32) for a in arrs]
lays = [d.layout for d in devs] + [None]
plan = planner.build_s1_plan(dims, dict(zip(dims, shape)), lays, ['init_time', 'latitude', 'longitude'], wdep_dims=['latitude'], allow_vec4=True)
dplan = engine._device_plan(ctx, plan)
nrows, nk = plan.nkey, nlon // 2 + 1
ptr = lambda d: C.c_void_p(d.ptr)
w = np.cos(np.deg2rad(np.linspace(-90, 90, nlat))) + 0.01
rng = np.random.default_rng(7)
results = {}


def sha(*bufs):
  h = hashlib.sha256()
  for b in bufs:
    h.update(np.ascontiguousarray(b).tobytes())
  return h.hexdigest()[:16]


for name, group in (('per (lead, level)', np.repeat(np.arange(nlead * nlev, dtype=np.int32), nlat)),
                    ('one group', np.zeros(nrows, dtype=np.int32)),
                    ('random, 7 groups', rng.integers(0, 7, nrows).astype(np.int32)),
                    ('a new group every row of three', (np.arange(nrows) // 3 % 40).astype(np.int32))):
  ngroup = int(group.max()) + 1
  scale = np.tile(w, nlead * nlev)
  g_dev, s_dev = ctx.upload(group), ctx.upload(scale)
  out = [ctx.alloc(ngroup * nk * 8) for _ in range(3)]
  part = ctx.alloc(nrows * 6 * 8)
  for acc in (0, 1, 1):
    _hip.check(lib.wbx_zonal_spectrum(ctx.handle, ptr(devs[0]), 1, nlon, nrows, nlon, ptr(g_dev), ptr(s_dev), ngroup, acc, ptr(out[0])), 'spectrum')
  _hip.check(lib.wbx_det_spectrum(ctx.handle, C.byref(dplan.struct), _hip.DET6, _hip.F32, ptr(devs[0]), ptr(devs[1]), ptr(devs[2]),
                                  ptr(g_dev), ptr(s_dev), ngroup, ptr(part), ptr(out[1]), ptr(out[2])), 'det_spectrum')
  ctx.synchronize()
  got = [ctx.download(o.ptr, (ngroup, nk)).copy() for o in out]
  results[name] = {'spectrum_x3': sha(got[0]), 'fused_p': sha(got[1]), 'fused_t': sha(got[2]), 'partial': sha(ctx.download(part.ptr, (nrows, 6)).copy())}
  print(f'{name:32s} ' + '  '.join(f'{k} {v}' for k, v in results[name].items()))
  if name == 'per (lead, level)':
    det = ctx.alloc(ngroup * 6 * 8)
    ds = ctx.upload(scale)
    _hip.check(lib.wbx_det_spectrum_folded(ctx.handle, C.byref(dplan.struct), _hip.DET6, _hip.F32, ptr(devs[0]), ptr(devs[1]), ptr(devs[2]),
                                           ptr(g_dev), ptr(s_dev), ptr(ds), ngroup, ptr(det), ptr(out[1]), ptr(out[2])), 'folded')
    ctx.synchronize()
    results['folded'] = {'det': sha(ctx.download(det.ptr, (ngroup, 6)).copy()), 'p': sha(ctx.download(out[1].ptr, (ngroup, nk)).copy())}
    print(f'{"  folded":32s} ' + '  '.join(f'{k} {v}' for k, v in results['folded'].items()))

golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'spectra_bits.json')
if '--write' in sys.argv:
  json.dump(results, open(golden, 'w'), indent=1, sort_keys=True)
  print('wrote', golden)
elif os.path.exists(golden):
  want = json.load(open(golden))
  print('golden:', 'equal' if want == results else 'DIFFERENT ' + json.dumps({k: v for k, v in results.items() if want.get(k) != v}))
