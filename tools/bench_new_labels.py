"""Streamed deterministic suite (ACC against a (dayofyear, hour) climatology) where EVERY chunk carries its own time labels,
as a real evaluation does: [1 init x 20 lead (24 h apart) x 37 level x 721 x 1440] chunks against a resident climatology of
40 days.  Prints the wall time per chunk with all-new labels and with repeated labels (the same init again and again); the
two agree when the per-chunk plan work does not depend on the labels (engine._planned swaps the climatology gather table
into the cached plan).  Usage (GPU box): python tools/bench_new_labels.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from weatherbenchx_amd import _hip, aggregation, engine, pipeline, time_chunks, weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import deterministic

nlat, nlon, nlead, nlev, ndoy, ninit = 721, 1440, 20, 37, 40, 20
lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
ctx = _hip.default_context(0)
dev = torch.device('cuda', 0)
lead = (np.arange(nlead) * 24).astype('timedelta64[h]').astype('timedelta64[ns]')
inits = np.datetime64('2020-01-01T00', 'ns') + np.arange(ninit) * np.timedelta64(24, 'h')
level = np.arange(nlev)
dims = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
pool = [(torch.randn((1, nlead, nlev, nlat, nlon), device=dev) + 280, torch.randn((1, nlead, nlev, nlat, nlon), device=dev) + 280)
        for _ in range(2)]
clim = xr.Dataset({'z': xr.DataArray(torch.randn((ndoy, 1, nlev, nlat, nlon), device=dev) + 280,
                                     dims=('dayofyear', 'hour', 'level', 'latitude', 'longitude'),
                                     coords={'dayofyear': np.arange(1, ndoy + 1), 'hour': np.array([0]), 'level': level,
                                             'latitude': lat, 'longitude': lon})})
torch.cuda.synchronize()
count = [0]


def load(init_chunk, lead_chunk):
  p, t = pool[count[0] % 2]
  count[0] += 1
  cs = {'init_time': init_chunk, 'lead_time': lead, 'level': level, 'latitude': lat, 'longitude': lon}
  return {'z': xr.DataArray(p, dims=dims, coords=cs)}, {'z': xr.DataArray(t, dims=dims, coords=cs)}


metrics = {'rmse': deterministic.RMSE(), 'acc': deterministic.ACC(clim), 'activity': deterministic.PredictionActivity(clim)}
agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])


def run(times, n):
  chunks = time_chunks.TimeChunks(times, lead, init_time_chunk_size=1)
  t0 = time.perf_counter()
  out = pipeline.evaluate_chunks(chunks, load, metrics, agg)[None].metric_values(metrics)
  ctx.synchronize()
  return (time.perf_counter() - t0) / n * 1e3, out


for swap in (True, False):
  engine.SWAP_GATHER_TABLES = swap
  engine._fast_plan_cache.clear()  # pylint: disable=protected-access
  run(inits[:2], 2)  # builds the plan
  new_ms, out = run(inits, ninit)  # 18 of the 20 label sets have never been seen
  again_ms, _ = run(inits, ninit)  # every label set is cached now
  print(f'gather table swapped into the cached plan: {swap}; {ninit} chunks, all-new time labels: {new_ms:.3f} ms per chunk; '
        f'the same labels again: {again_ms:.3f} ms per chunk; acc mean {float(np.asarray(out["acc.z"].values).mean()):.4f}')
