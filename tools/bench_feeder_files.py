"""A file-backed evaluation end to end: chunks of a forecast archive on local disk -> page-locked memory -> asynchronous H2D
-> kernels, through weatherbenchx_amd.loaders + pipeline.evaluate_chunks(prefetch=...).  Reports what each leg costs and how
much of it overlaps:
  read    loaders' gather from the memory-mapped files into page-locked buffers (page cache after the first pass)
  h2d     bytes uploaded / wall time of the job (a lower bound of the PCIe rate while kernels run)
  kernel  HIP-event time of the stage-1 launches per chunk
  job     wall time per chunk with prefetch = 0 (serial: read, upload, launch) and prefetch = 2 (feeder thread + copy stream)
Shape: the deterministic side of configs[4], scaled to fit a box's disk: f32[NI init, 4 lead, 13 level, 1440, 721] forecasts
and the matching analyses on a time axis (latitude fastest, like the public archives); RMSE + MAE + bias, GridAreaWeighting.
usage: bench_feeder_files.py [dir=/tmp/wbx_files] [ni=12] [npy|nc]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from weatherbenchx_amd import aggregation, engine, loaders, pipeline, time_chunks, weighting
from weatherbenchx_amd.metrics import deterministic

args = sys.argv[1:]
root = next((a[4:] for a in args if a.startswith('dir=')), '/tmp/wbx_files')
ni = int(next((a[3:] for a in args if a.startswith('ni=')), 12))
fmt = 'nc' if 'nc' in args else 'npy'
nlead, nlev, nlon, nlat = 4, 13, 1440, 721
os.makedirs(root, exist_ok=True)
init_times = np.datetime64('2020-01-01T00', 'ns') + np.arange(ni) * np.timedelta64(12, 'h')
lead_times = (np.arange(nlead) * 12).astype('timedelta64[h]').astype('timedelta64[ns]')
times = np.datetime64('2020-01-01T00', 'ns') + np.arange(ni + nlead) * np.timedelta64(12, 'h')
dims = ('level', 'longitude', 'latitude')
coords = {'level': np.arange(nlev), 'longitude': np.arange(nlon) * 0.25, 'latitude': np.linspace(-90, 90, nlat)}
pp, tp = os.path.join(root, f'p.{fmt}'), os.path.join(root, f't.{fmt}')
rng = np.random.default_rng(0)
t0 = time.perf_counter()
if fmt == 'npy':
  if not os.path.exists(pp):
    p = np.lib.format.open_memmap(pp, mode='w+', dtype=np.float32, shape=(ni, nlead, nlev, nlon, nlat))
    for i in range(ni):
      p[i] = rng.standard_normal((nlead, nlev, nlon, nlat), dtype=np.float32) + 280
    p.flush()
    t = np.lib.format.open_memmap(tp, mode='w+', dtype=np.float32, shape=(times.size, nlev, nlon, nlat))
    for i in range(times.size):
      t[i] = rng.standard_normal((nlev, nlon, nlat), dtype=np.float32) + 280
    t.flush()
    del p, t
  src_p, src_t = {'z': pp}, {'z': tp}
else:
  from scipy.io import netcdf_file
  if not os.path.exists(pp):
    for path, shape, names in ((pp, (ni, nlead, nlev, nlon, nlat), ('init_time', 'lead_time') + dims), (tp, (times.size, nlev, nlon, nlat), ('time',) + dims)):
      f = netcdf_file(path, 'w', version=2)
      for n, s in zip(names, shape):
        f.createDimension(n, s)
      v = f.createVariable('z', np.float32, names)
      for i in range(shape[0]):
        v[i] = rng.standard_normal(shape[1:], dtype=np.float32) + 280
      f.close()
  src_p, src_t = {'z': (pp, 'z')}, {'z': (tp, 'z')}
write_s = time.perf_counter() - t0
chunk_bytes = 2 * nlead * nlev * nlon * nlat * 4
metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE(), 'bias': deterministic.Bias()}
agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
tc = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=1)


def job(prefetch, log_kernels=False):
  lp = loaders.PredictionsFromFiles(src_p, init_times, lead_times, dims, coords)
  lt = loaders.TargetsFromFiles(src_t, times, dims, coords)
  if log_kernels:
    engine.S1_EVENT_LOG, engine.S1_EVENT_MARKS = [], True
  t0 = time.perf_counter()
  out = pipeline.evaluate_chunks(tc, loaders.load_chunk_fn(lp, lt), metrics, agg, prefetch=prefetch)[None].metric_values(metrics)
  wall = time.perf_counter() - t0
  kernel_ms = None
  if log_kernels:
    log = engine.resolve_event_marks(engine.S1_EVENT_LOG)
    engine.S1_EVENT_LOG, engine.S1_EVENT_MARKS = None, False
    kernel_ms = float(np.sum([e['ms'] for e in log])) / ni
  read_s = lp.timings['seconds'] + lt.timings['seconds']
  return out, wall, read_s, kernel_ms


job(0)  # first pass: page cache, plans, pools
out0, wall0, read0, _ = job(0)
out2, wall2, read2, kernel_ms = job(2, log_kernels=True)
for k in metrics:
  assert np.array_equal(np.asarray(out0[f'{k}.z'].values), np.asarray(out2[f'{k}.z'].values)), k
total = chunk_bytes * ni
print(json.dumps({'format': fmt, 'chunks': ni, 'chunk': f'2 x f32[1,{nlead},{nlev},{nlon},{nlat}]', 'chunk_GB': round(chunk_bytes / 1e9, 3),
                  'files_written_s': round(write_s, 1),
                  'read_GBps': round(total / read2 / 1e9, 2), 'read_ms_per_chunk': round(read2 / ni * 1e3, 2),
                  'kernel_ms_per_chunk': round(kernel_ms, 3),
                  'serial_ms_per_chunk': round(wall0 / ni * 1e3, 2), 'prefetch2_ms_per_chunk': round(wall2 / ni * 1e3, 2),
                  'end_to_end_GBps_prefetch2': round(total / wall2 / 1e9, 2), 'end_to_end_GBps_serial': round(total / wall0 / 1e9, 2),
                  'overlap': 'prefetch=2: the loader thread reads chunk k+1 / k+2 and its copy stream uploads them while the launch stream '
                             'runs chunk k; the job is bound by max(read, upload, kernel) instead of their sum',
                  'rmse_level0': float(np.asarray(out2['rmse.z'].values).reshape(-1)[0])}))
