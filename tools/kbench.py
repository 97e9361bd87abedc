"""In-process A/B timing of stage-1 kernel variants (same box, same clocks): HIP events, 20 launches per variant.

usage: python tools/kbench.py [ens|det]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from weatherbenchx_amd import _hip, engine, planner
from weatherbenchx_amd import xarray_lite as xr

ctx = _hip.default_context(0)
NLAT, NLON = int(os.environ.get('KB_NLAT', 721)), 1440


def time_s1(kind, plan, devs, nl, reps=20, **kw):
  dplan = engine._device_plan(ctx, plan)
  engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], reps
  engine._run_s1(ctx, kind, dplan, plan, devs, _hip.F32, nl, **kw)  # warm
  engine.S1_EVENT_LOG.clear()
  for _ in range(3):
    engine._run_s1(ctx, kind, dplan, plan, devs, _hip.F32, nl, **kw)
  ms = float(np.median([e['ms'] for e in engine.S1_EVENT_LOG]))
  engine.S1_EVENT_LOG = None
  return ms


def ens():
  m, ns = 51, 8
  t = xr.DataArray(torch.randn(ns, NLAT, NLON, device='cuda') + 280, dims=('lead_time', 'latitude', 'longitude'))
  p = xr.DataArray(t.data[:, None] + torch.randn(ns, m, NLAT, NLON, device='cuda'),
                   dims=('lead_time', 'number', 'latitude', 'longitude'))
  torch.cuda.synchronize()
  devs = [engine._to_device(ctx, p, _hip.F32), engine._to_device(ctx, t, _hip.F32), None, None]
  lays = [d.layout if d else None for d in devs]
  sizes = {'lead_time': ns, 'latitude': NLAT, 'longitude': NLON}
  nbytes = ns * NLAT * NLON * (m + 1) * 4
  for bt in (64, 128, 256):
    plan = planner.build_s1_plan(('lead_time', 'latitude', 'longitude'), sizes, lays, ['latitude', 'longitude'],
                                 wdep_dims=['latitude'], allow_vec4=False, flags=_hip.FLAG_FAIR)
    plan.block_threads = bt
    for name, algo in (('sort', 0), ('pairwise', 1), ('pairwise through LDS tiles', 98), ('loadonly', 99)):
      if algo == 98 and bt != 64:
        continue  # (the LDS-tiled diagnostic is built for one-wave blocks)
      ms = time_s1('ens', plan, devs, 5, ens=(m, devs[0].layout.stride('number'), algo))
      print(f'ens M=51 block={bt:3d} nkey={plan.nkey} nchunk={plan.nchunk} {name:9s} {ms:7.4f} ms  '
            f'{nbytes / ms / 1e6:7.1f} GB/s  {nbytes / ms / 1e6 / 80:5.1f}% of 8 TB/s')


  # the LDS-tiled diagnostic forms the same 1275 terms: its partial sums must agree with the register pair form's
  plan.block_threads = 64
  dplan = engine._device_plan(ctx, plan)
  got = {}
  for algo in (1, 98):
    part = engine._run_s1(ctx, 'ens', dplan, plan, devs, _hip.F32, 5, ens=(m, devs[0].layout.stride('number'), algo))
    got[algo] = ctx.download(part.ptr, (plan.nkey * plan.nchunk, 5)).copy()
  err = np.max(np.abs(got[98] - got[1]) / np.maximum(np.abs(got[1]), 1e-30))
  print(f'LDS-tiled pair form against the register pair form: max relative difference of the partial sums {err:.2e}')
  assert err < 1e-6, err


def det():
  ni, nl, nz = int(os.environ.get('KB_NI', 16)), 10, 5
  shape = (ni, nl, nz, NLAT, NLON)
  dims = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
  arrs = [xr.DataArray(torch.randn(shape, device='cuda') + 280, dims=dims) for _ in range(3)]
  torch.cuda.synchronize()
  devs = [engine._to_device(ctx, a, _hip.F32) for a in arrs] + [None]
  lays = [d.layout if d else None for d in devs]
  sizes = dict(zip(dims, shape))
  nbytes = int(np.prod(shape)) * 12
  for bt in (64, 128, 256):
    plan = planner.build_s1_plan(dims, sizes, lays, ['init_time', 'latitude', 'longitude'], wdep_dims=['latitude'])
    plan.block_threads = bt
    ms = time_s1('det', plan, devs, 6, func=_hip.DET6)
    print(f'det DET6 vec={plan.vec} block={bt:3d} nkey={plan.nkey} {ms:7.4f} ms {nbytes / ms / 1e6:7.1f} GB/s '
          f'{nbytes / ms / 1e6 / 80:5.1f}%')
  # latitude-fastest (x kept): chunk geometry sweep
  dims2 = ('init_time', 'lead_time', 'level', 'longitude', 'latitude')
  shape2 = (ni, nl, nz, NLON, NLAT)
  arrs2 = [xr.DataArray(a.data.reshape(shape2), dims=dims2) for a in arrs]
  devs2 = [engine._to_device(ctx, a, _hip.F32) for a in arrs2] + [None]
  lays2 = [d.layout if d else None for d in devs2]
  sizes2 = dict(zip(dims2, shape2))
  for tb in (1024, 4096, 16384, 65536):
    for bt in (128, 256):
      plan = planner.build_s1_plan(dims2, sizes2, lays2, ['init_time', 'latitude', 'longitude'], wdep_dims=['latitude'],
                                   target_blocks=tb)
      plan.block_threads = bt
      ms = time_s1('det', plan, devs2, 6, func=_hip.DET6)
      print(f'det DET6 lat-fastest XK vec={plan.vec} block={bt} target={tb} nchunk={plan.nchunk} '
            f'dchunk={plan.depth_chunk} {ms:7.4f} ms {nbytes / ms / 1e6:7.1f} GB/s {nbytes / ms / 1e6 / 80:5.1f}%')


def det_masked():
  """Deterministic family with the mask / skipna count lanes (public-benchmark default masked=True with NaN targets)."""
  ni, nl, nz = 16, 10, 5
  shape = (ni, nl, nz, NLAT, NLON)
  dims = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
  arrs = [xr.DataArray(torch.randn(shape, device='cuda') + 280, dims=dims) for _ in range(3)]
  mask = xr.DataArray(np.random.default_rng(0).random((NLAT, NLON)) > 0.3, dims=('latitude', 'longitude'))
  torch.cuda.synchronize()
  devs = [engine._to_device(ctx, a, _hip.F32) for a in arrs] + [engine._mask_to_device(ctx, mask)]
  lays = [d.layout if d else None for d in devs]
  sizes = dict(zip(dims, shape))
  nbytes = int(np.prod(shape)) * 12
  for name, flags, use in (('plain', 0, devs[:3] + [None]), ('masked', _hip.FLAG_MASKED, devs),
                           ('skipna', _hip.FLAG_SKIPNA, devs[:3] + [None])):
    plan = planner.build_s1_plan(dims, sizes, [d.layout if d else None for d in use], ['init_time', 'latitude', 'longitude'],
                                 wdep_dims=['latitude'], flags=flags)
    for vec in sorted({plan.vec, 1}, reverse=True):
      plan.vec = vec
      ms = time_s1("det", plan, use, {0: 6, _hip.FLAG_MASKED: 7}.get(flags, 12), func=_hip.DET6)
      print(f'det DET6 {name:7s} vec={plan.vec} x_kept={plan.x_kept} {ms:7.4f} ms {nbytes / ms / 1e6:7.1f} GB/s '
            f'{nbytes / ms / 1e6 / 80:5.1f}%')


def ens_latfast():
  """IFS-ENS style chunk (init, number, lead, longitude, latitude): latitude fastest, members slow."""
  m, nl = 51, 8
  t = xr.DataArray(torch.randn(nl, NLON, NLAT, device='cuda') + 280, dims=('lead_time', 'longitude', 'latitude'))
  p = xr.DataArray(t.data[None] + torch.randn(m, nl, NLON, NLAT, device='cuda'),
                   dims=('number', 'lead_time', 'longitude', 'latitude'))
  torch.cuda.synchronize()
  devs = [engine._to_device(ctx, p, _hip.F32), engine._to_device(ctx, t, _hip.F32), None, None]
  lays = [d.layout if d else None for d in devs]
  sizes = {'lead_time': nl, 'latitude': NLAT, 'longitude': NLON}
  nbytes = nl * NLAT * NLON * (m + 1) * 4
  for tb in (1024, 4096, 16384):
    plan = planner.build_s1_plan(('lead_time', 'longitude', 'latitude'), sizes, lays, ['latitude', 'longitude'],
                                 wdep_dims=['latitude'], allow_vec4=False, flags=_hip.FLAG_FAIR, target_blocks=tb)
    for name, algo in (('sort', 0), ('loadonly', 99)):
      ms = time_s1('ens', plan, devs, 5, ens=(m, devs[0].layout.stride('number'), algo))
      print(f'ens M=51 lat-fastest x_kept={plan.x_kept} block={plan.block_threads} nkey={plan.nkey} nchunk={plan.nchunk} '
            f'{name:9s} {ms:7.4f} ms  {nbytes / ms / 1e6:7.1f} GB/s  {nbytes / ms / 1e6 / 80:5.1f}% of 8 TB/s')


if __name__ == '__main__':
  which = sys.argv[1] if len(sys.argv) > 1 else 'ens'
  {'ens': ens, 'det': det, 'ens_latfast': ens_latfast, 'det_masked': det_masked}[which]()
