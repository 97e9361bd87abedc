"""Host cost of the transposing loader copy (wbx_host_transpose) against the plain copy of the same bytes, by thread count.
One chunk-sized block of 37-level fields [n, 1440, 721] -> [n, 721, 1440], page-locked destination."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weatherbenchx_amd import _hip  # noqa: E402

lib = _hip.load_library()
n, rows, cols = 148, 1440, 721  # 0.61 GB
src = np.random.default_rng(0).standard_normal((n, rows, cols), dtype=np.float32)
try:
  dst = _hip.default_context().pinned_empty((n, cols, rows), np.float32)
  where = 'page-locked'
except Exception:  # pylint: disable=broad-except
  dst = np.empty((n, cols, rows), np.float32)
  where = 'pageable'
flat = dst.reshape(n, rows, cols)
print(f'{src.nbytes / 1e9:.2f} GB, destination {where}, host cpus {os.cpu_count()}')
for threads in (1, 2, 4, 8, 16, 32):
  if threads > (os.cpu_count() or 1):
    break
  pool = ThreadPoolExecutor(threads)
  cuts = np.linspace(0, n, 4 * threads + 1).astype(int)
  parts = list(zip(cuts[:-1], cuts[1:]))

  def tr(ab):
    lib.wbx_host_transpose(dst[ab[0]:ab[1]].ctypes.data, src[ab[0]:ab[1]].ctypes.data, int(ab[1] - ab[0]), rows, cols, 4)

  def cp(ab):
    np.copyto(flat[ab[0]:ab[1]], src[ab[0]:ab[1]])
  res = {}
  for name, fn in (('copy', cp), ('transpose', tr)):
    best = 1e9
    for _ in range(4):
      t0 = time.perf_counter()
      list(pool.map(fn, parts))
      best = min(best, time.perf_counter() - t0)
    res[name] = src.nbytes / best / 1e9
  print(f'threads {threads:3d}: copy {res["copy"]:6.1f} GB/s   transpose {res["transpose"]:6.1f} GB/s   ratio {res["copy"] / res["transpose"]:.2f}')
  pool.shutdown()
assert np.array_equal(dst[3], src[3].T)
