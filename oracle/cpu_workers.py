"""TEST / MEASUREMENT INFRASTRUCTURE (not product code): worker processes for bench.py's multi-core `cpu_baseline`.

BASELINE.md section 5(b): `os.cpu_count()` worker processes over time chunks, the stand-in for Beam's
`DirectRunner --direct_num_workers` (docs/source/beam_dataflow.md:5-17).  Every worker generates its own seeded
(init, lead) slices of the configs[1] workload -- nothing large is pickled -- waits on a barrier, and runs the oracle's
"reference structure" NumPy path (oracle/wbx_oracle.py: one pass per statistic with full-size float32 temporaries + two
einsums, aggregation.py:337-366) on them.  Only the compute between the barrier and the last worker's finish is timed.
"""
import os
import time


def _worker(rank, nslices, nlev, nlat, nlon, barrier, out):
  os.environ.setdefault('OMP_NUM_THREADS', '1')
  os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
  os.environ.setdefault('MKL_NUM_THREADS', '1')
  import numpy as np  # pylint: disable=g-import-not-at-top
  from oracle import wbx_oracle as O  # pylint: disable=g-import-not-at-top
  rng = np.random.default_rng(1000 + rank)
  shape = (1, nslices, nlev, nlat, nlon)
  c = (rng.standard_normal(shape, dtype=np.float32) * 10 + 280)
  p = c + rng.standard_normal(shape, dtype=np.float32)
  t = c + rng.standard_normal(shape, dtype=np.float32)
  w = O.grid_area_weights(np.linspace(-90, 90, nlat))
  try:
    barrier.wait(timeout=600)
  except Exception:  # pylint: disable=broad-except
    out.put((rank, None, None, 0))
    return
  t0 = time.time()
  res = O.reference_structure_deterministic(p, t, c, w)
  t1 = time.time()
  out.put((rank, t0, t1, int(p.size), float(np.asarray(res['SquaredError']).mean())))


def run(nworkers: int, nslices: int, nlev: int, nlat: int, nlon: int):
  """-> {'seconds': wall between the barrier and the last finish, 'points': total points, 'workers': n}."""
  import multiprocessing as mp  # pylint: disable=g-import-not-at-top
  ctx = mp.get_context('spawn')  # the parent holds a HIP context: never fork it
  barrier = ctx.Barrier(nworkers)
  out = ctx.Queue()
  procs = [ctx.Process(target=_worker, args=(r, nslices, nlev, nlat, nlon, barrier, out), daemon=True)
           for r in range(nworkers)]
  for p in procs:
    p.start()
  results = [out.get(timeout=900) for _ in procs]
  for p in procs:
    p.join(timeout=60)
  if any(r[1] is None for r in results):
    raise RuntimeError('a CPU baseline worker failed to reach the start barrier')
  start = min(r[1] for r in results)
  end = max(r[2] for r in results)
  return {'seconds': end - start, 'points': sum(r[3] for r in results), 'workers': nworkers,
          'check_mse': sum(r[4] for r in results) / len(results)}
