"""TEST / MEASUREMENT INFRASTRUCTURE (not product code): worker processes for bench.py's multi-core `cpu_baseline`.

BASELINE.md section 5(b): `os.cpu_count()` worker processes over time chunks, the stand-in for Beam's
`DirectRunner --direct_num_workers` (docs/source/beam_dataflow.md:5-17).  Every worker generates its own seeded
(init, lead) slices of the configs[1] workload, or 51-member fields of the north_star workload (kind='ensemble') -- nothing
large is pickled -- waits on a barrier, and runs the oracle's
"reference structure" NumPy path (oracle/wbx_oracle.py: one pass per statistic with full-size float32 temporaries + two
einsums, aggregation.py:337-366) on them.  Only the compute between the barrier and the last worker's finish is timed.
"""
import os
import time


def _worker_ens(rank, nslices, m, nlat, nlon, barrier, out):
  """`nslices` fields of float32[m members, nlat, nlon] + targets: the public-benchmark ensemble suite per field
  (oracle.reference_structure_ensemble: CRPS rank form, variance, unbiased MSE, ensemble-mean SE + two einsums each)."""
  os.environ.setdefault('OMP_NUM_THREADS', '1')
  os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
  os.environ.setdefault('MKL_NUM_THREADS', '1')
  import numpy as np  # pylint: disable=g-import-not-at-top
  from oracle import wbx_oracle as O  # pylint: disable=g-import-not-at-top
  rng = np.random.default_rng(2000 + rank)
  t = rng.standard_normal((nslices, nlat, nlon), dtype=np.float32) + 280
  p = t[:, None] + rng.standard_normal((nslices, m, nlat, nlon), dtype=np.float32)
  t = t + rng.standard_normal((nslices, nlat, nlon), dtype=np.float32)
  w = O.grid_area_weights(np.linspace(-90, 90, nlat))
  try:
    barrier.wait(timeout=600)
  except Exception:  # pylint: disable=broad-except
    out.put((rank, None, None, 0))
    return
  t0 = time.time()
  check = 0.0
  for k in range(nslices):
    res = O.reference_structure_ensemble(p[k], t[k], w)
    check += float(res['CRPSSkill'] - 0.5 * res['CRPSSpread'])
  t1 = time.time()
  out.put((rank, t0, t1, int(t.size), check / nslices))


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


def run(nworkers: int, nslices: int, nlev: int, nlat: int, nlon: int, kind: str = 'deterministic'):
  """-> {'seconds': wall between the barrier and the last finish, 'points': total points, 'workers': n}.
  kind='ensemble': `nlev` is the ensemble size and a worker's slice is one float32[nlev members, nlat, nlon] field."""
  import multiprocessing as mp  # pylint: disable=g-import-not-at-top
  ctx = mp.get_context('spawn')  # the parent holds a HIP context: never fork it
  barrier = ctx.Barrier(nworkers)
  out = ctx.Queue()
  target = _worker_ens if kind == 'ensemble' else _worker
  procs = [ctx.Process(target=target, args=(r, nslices, nlev, nlat, nlon, barrier, out), daemon=True)
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
          'check_mse' if kind != 'ensemble' else 'check_crps': sum(r[4] for r in results) / len(results)}
