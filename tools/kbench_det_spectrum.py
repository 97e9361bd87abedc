"""wbx_det_spectrum (spectra of p and t + the DET6 lanes in one sweep) against wbx_det_partial + two wbx_zonal_spectrum launches
on a configs[4] z chunk f32[1 init, 20 lead, 37 level, 721, 1440]: same outputs, HIP-event timing of both routes.
usage: python tools/kbench_det_spectrum.py [nlead]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from weatherbenchx_amd import _hip, engine, planner
from weatherbenchx_amd import xarray_lite as xr

ctx = _hip.default_context(0)
lib = ctx.lib
nlead = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nlev, nlat, nlon = 37, 721, 1440
dims = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
shape = (1, nlead, nlev, nlat, nlon)
arrs = [xr.DataArray(torch.randn(shape, device='cuda') * (3 if i < 2 else 10) + 280, dims=dims) for i in range(3)]
torch.cuda.synchronize()
devs = [engine._to_device(ctx, a, _hip.F32) for a in arrs] + [None]
lays = [d.layout if d else None for d in devs]
sizes = dict(zip(dims, shape))
plan = planner.build_s1_plan(dims, sizes, lays, ['init_time', 'latitude', 'longitude'], wdep_dims=['latitude'], allow_vec4=True)
assert plan.nchunk == 1 and plan.ndepth == 1 and plan.nx == nlon and not plan.x_kept, (plan.nchunk, plan.ndepth)
dplan = engine._device_plan(ctx, plan)
nrows = plan.nkey
assert nrows == nlead * nlev * nlat
# spectra rows: mean over latitude per (lead, level), cos-lat weights
lat = np.linspace(-90, 90, nlat)
w = np.cos(np.deg2rad(lat)); w /= w.sum()
group = np.repeat(np.arange(nlead * nlev, dtype=np.int32), nlat)
scale = np.tile(w, nlead * nlev)
g_dev, s_dev = ctx.upload(group), ctx.upload(scale)
ngroup, nk = nlead * nlev, nlon // 2 + 1
part_a, part_b = ctx.alloc(nrows * 6 * 8), ctx.alloc(nrows * 6 * 8)
pw = [ctx.alloc(ngroup * nk * 8) for _ in range(4)]
ptr = lambda d: C.c_void_p(d.ptr)


def separate():
  _hip.check(lib.wbx_det_partial(ctx.handle, C.byref(dplan.struct), _hip.DET6, _hip.F32, ptr(devs[0]), ptr(devs[1]), ptr(devs[2]),
                                 None, ptr(part_a)), 'wbx_det_partial')
  for i in (0, 1):
    _hip.check(lib.wbx_zonal_spectrum(ctx.handle, ptr(devs[i]), 1, nlon, nrows, nlon, ptr(g_dev), ptr(s_dev), ngroup, 0, ptr(pw[i])),
               'wbx_zonal_spectrum')


def fused():
  _hip.check(lib.wbx_det_spectrum(ctx.handle, C.byref(dplan.struct), _hip.DET6, _hip.F32, ptr(devs[0]), ptr(devs[1]), ptr(devs[2]),
                                  ptr(g_dev), ptr(s_dev), ngroup, ptr(part_b), ptr(pw[2]), ptr(pw[3])), 'wbx_det_spectrum')


separate()
fused()
ctx.synchronize()
a = ctx.download(part_a.ptr, (nrows, 6)).copy()
b = ctx.download(part_b.ptr, (nrows, 6)).copy()
print('det partial: max rel diff', float(np.max(np.abs(a - b) / (np.abs(a) + 1e-30))), 'sample', a[5], b[5])
for i, name in ((0, 'p'), (1, 't')):
  sa = ctx.download(pw[i].ptr, (ngroup, nk)).copy()
  sb = ctx.download(pw[2 + i].ptr, (ngroup, nk)).copy()
  print(f'spectrum {name}: max rel diff', float(np.max(np.abs(sa - sb) / (np.abs(sa) + 1e-30))), 'sum_k', sa[0].sum(), sb[0].sum())
det_out = ctx.alloc(ngroup * 6 * 8)


def folded():
  _hip.check(lib.wbx_det_spectrum_folded(ctx.handle, C.byref(dplan.struct), _hip.DET6, _hip.F32, ptr(devs[0]), ptr(devs[1]), ptr(devs[2]),
                                         ptr(g_dev), ptr(s_dev), ptr(s_dev), ngroup, ptr(det_out), ptr(pw[2]), ptr(pw[3])),
             'wbx_det_spectrum_folded')


def folded3():
  _hip.check(lib.wbx_det_spectrum_folded(ctx.handle, C.byref(dplan.struct), _hip.DET3, _hip.F32, ptr(devs[0]), ptr(devs[1]), None,
                                         ptr(g_dev), ptr(s_dev), ptr(s_dev), ngroup, ptr(det_out), ptr(pw[2]), ptr(pw[3])),
             'wbx_det_spectrum_folded')


folded()
ctx.synchronize()
want = (a.reshape(ngroup, nlat, 6) * w[None, :, None]).sum(axis=1)
got = ctx.download(det_out.ptr, (ngroup, 6)).copy()
print('folded det sums: max rel diff to the contraction of the partial', float(np.max(np.abs(got - want) / (np.abs(want) + 1e-300))))
for i, name in ((0, 'p'), (1, 't')):
  sa = ctx.download(pw[i].ptr, (ngroup, nk)).copy()
  sb = ctx.download(pw[2 + i].ptr, (ngroup, nk)).copy()
  print(f'folded spectrum {name}: max rel diff', float(np.max(np.abs(sa - sb) / (np.abs(sa) + 1e-30))))
folded3()
ctx.synchronize()
got3 = ctx.download(det_out.ptr, (ngroup, 3)).copy()
print('folded DET3 sums: max rel diff', float(np.max(np.abs(got3 - want[:, :3]) / (np.abs(want[:, :3]) + 1e-300))))


def fused3():
  _hip.check(lib.wbx_det_spectrum(ctx.handle, C.byref(dplan.struct), _hip.DET3, _hip.F32, ptr(devs[0]), ptr(devs[1]), None,
                                  ptr(g_dev), ptr(s_dev), ngroup, ptr(part_b), ptr(pw[2]), ptr(pw[3])), 'wbx_det_spectrum')


def spectra_only():
  for i in (0, 1):
    _hip.check(lib.wbx_zonal_spectrum(ctx.handle, ptr(devs[i]), 1, nlon, nrows, nlon, ptr(g_dev), ptr(s_dev), ngroup, 0, ptr(pw[i])),
               'wbx_zonal_spectrum')


gb = nrows * nlon * 4 / 1e9
for name, fn, nbytes in (('separate (det + 2 spectra)', separate, 12), ('fused', fused, 12), ('folded (stage 2 inside)', folded, 12),
                         ('fused DET3 (no c)', fused3, 8), ('folded DET3', folded3, 8),
                         ('two spectra alone', spectra_only, 8)):
  ms = []
  for it in range(5):
    ctx.timer_start()
    for _ in range(5):
      fn()
    ms.append(ctx.timer_stop() / 5)
  m = float(np.median(ms))
  print(f'{name:28s} {m:.4f} ms per chunk  ({gb * nbytes / 4 / m:.2f} TB/s on {nbytes} B/point)')
