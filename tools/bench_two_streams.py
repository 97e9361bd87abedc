"""Six back-to-back ensemble stage-1 launches (one per variable of a configs[2] step) on ONE stream vs alternating over
TWO: does the tail of one launch overlap the head of the next?"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weatherbenchx_amd import _hip, engine, planner
from weatherbenchx_amd import xarray_lite as xr

NLAT, NLON, m, ns, nvar = 721, 1440, 51, 8, 6
c0 = _hip.default_context(0)
c1 = engine.new_context()
fields = []
for v in range(nvar):
  t = xr.DataArray(torch.randn(ns, NLAT, NLON, device='cuda') + 280, dims=('lead_time', 'latitude', 'longitude'))
  p = xr.DataArray(t.data[:, None] + torch.randn(ns, m, NLAT, NLON, device='cuda'),
                   dims=('lead_time', 'number', 'latitude', 'longitude'))
  fields.append((p, t))
torch.cuda.synchronize()
sizes = {'lead_time': ns, 'latitude': NLAT, 'longitude': NLON}


def prep(ctx):
  out = []
  for p, t in fields:
    devs = [engine._to_device(ctx, p, _hip.F32), engine._to_device(ctx, t, _hip.F32), None, None]
    lays = [d.layout if d else None for d in devs]
    plan = planner.build_s1_plan(('lead_time', 'latitude', 'longitude'), sizes, lays, ['latitude', 'longitude'],
                                 wdep_dims=['latitude'], allow_vec4=False, flags=_hip.FLAG_FAIR)
    out.append((plan, engine._device_plan(ctx, plan), devs))
  return out


sets = {id(c0): prep(c0), id(c1): prep(c1)}
for name, ring in (('one stream ', [c0]), ('two streams', [c0, c1])):
  def sweep():
    for v in range(nvar):
      ctx = ring[v % len(ring)]
      plan, dplan, devs = sets[id(ctx)][v]
      # each variable needs its own partial buffer when launches may overlap
      engine._run_s1(ctx, 'ens', dplan, plan, devs, _hip.F32, 5, ens=(m, devs[0].layout.stride('number'), 0))
  for _ in range(3):
    sweep()
  c0.synchronize(); c1.synchronize()
  t0 = time.perf_counter()
  n = 20
  for _ in range(n):
    sweep()
  c0.synchronize(); c1.synchronize()
  ms = (time.perf_counter() - t0) / n * 1e3
  print(f'{name}: {ms:6.3f} ms per {nvar}-variable sweep = {ms / nvar:6.4f} ms per launch')
