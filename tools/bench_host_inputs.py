"""PCIe-inclusive rate: the public-benchmark chunk (1 init x 12 leads x 13 levels x 721 x 1440, DET6, area weights) with
predictions / targets handed over as HOST numpy arrays (what the reference's zarr loaders produce), fresh every chunk."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from weatherbenchx_amd import _hip, aggregation, engine, weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as mb, deterministic

nl, nlev, nlat, nlon = 12, 13, 721, 1440
lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
dims = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
coords = {'init_time': np.array(['2020-01-01T00'], dtype='datetime64[ns]'),
          'lead_time': (np.arange(nl) * 12).astype('timedelta64[h]').astype('timedelta64[ns]'),
          'level': np.arange(nlev), 'latitude': lat, 'longitude': lon}
shape = tuple(len(coords[d]) for d in dims)
rng = np.random.default_rng(0)
bufs = [(rng.standard_normal(shape, dtype=np.float32) + 280, rng.standard_normal(shape, dtype=np.float32) + 280)
        for _ in range(3)]
metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE(), 'bias': deterministic.Bias()}
agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
ctx = _hip.default_context(0)
nbytes = 2 * int(np.prod(shape)) * 4


def chunk(i):
  p, t = bufs[i % 3]
  pp = {'z': xr.DataArray(p, dims=dims, coords=coords)}
  tt = {'z': xr.DataArray(t, dims=dims, coords=coords)}
  return agg.aggregate_statistics(mb.compute_unique_statistics_for_all_metrics(metrics, pp, tt))


for i in range(2):
  out = chunk(i).metric_values(metrics)
n = 6
t0 = time.perf_counter()
for i in range(n):
  out = chunk(i).metric_values(metrics)
ms = (time.perf_counter() - t0) / n * 1e3
print(f'host numpy inputs, synchronous: {ms:7.1f} ms/chunk = {nbytes / ms / 1e6:6.1f} GB/s over PCIe ({nbytes / 1e9:.2f} GB per chunk)')
t0 = time.perf_counter()
with engine.deferred_results():
  prev = None
  for i in range(n):
    cur = chunk(i)
    if prev is not None:
      out = prev.metric_values(metrics)
    prev = cur
  out = prev.metric_values(metrics)
ms = (time.perf_counter() - t0) / n * 1e3
print(f'host numpy inputs, deferred   : {ms:7.1f} ms/chunk = {nbytes / ms / 1e6:6.1f} GB/s over PCIe   rmse={float(np.asarray(out["rmse.z"].values).reshape(-1)[0]):.4f}')
# raw copy rates for reference
a = bufs[0][0]
d = ctx.alloc(a.nbytes)
import ctypes as C
for _ in range(2):
  t0 = time.perf_counter()
  _hip.check(ctx.lib.wbx_memcpy_h2d(ctx.handle, C.c_void_p(d.ptr), a.ctypes.data_as(C.c_void_p), a.nbytes), 'h2d')
  ctx.synchronize()
  dt = time.perf_counter() - t0
print(f'raw wbx_memcpy_h2d from pageable memory: {a.nbytes / dt / 1e9:6.1f} GB/s')

# ---- the same chunks through pipeline.evaluate_chunks, with and without the chunk feeder ------------------
from weatherbenchx_amd import pipeline, time_chunks
nchunk = 8
inits = np.datetime64('2020-01-01T00', 'ns') + np.arange(nchunk) * np.timedelta64(12, 'h')
leads = coords['lead_time']
times = time_chunks.TimeChunks(inits, leads, init_time_chunk_size=1)


def load(init_chunk, lead_chunk):
  i = int((init_chunk[0] - inits[0]) / np.timedelta64(12, 'h'))
  p, t = bufs[i % 3]
  cc = dict(coords, init_time=init_chunk)
  # zero-cost loader (arrays already decoded in host memory): what is left is upload + kernels + bookkeeping
  return ({'z': xr.DataArray(p, dims=dims, coords=cc)}, {'z': xr.DataArray(t, dims=dims, coords=cc)})


for prefetch in (0, 1):
  pipeline.evaluate_chunks(times, load, metrics, agg, prefetch=prefetch)
  t0 = time.perf_counter()
  state = pipeline.evaluate_chunks(times, load, metrics, agg, prefetch=prefetch)[None]
  ms = (time.perf_counter() - t0) / nchunk * 1e3
  r = float(np.asarray(state.metric_values(metrics)['rmse.z'].values).reshape(-1)[0])
  print(f'evaluate_chunks(prefetch={prefetch}): {ms:7.1f} ms/chunk = {nbytes / ms / 1e6:6.1f} GB/s over PCIe   rmse={r:.4f}')
