"""Where a configs[4] chunk's wall time goes: per pass (deterministic / spectra / ensemble) the wall time per chunk, the part
of it the host spends blocked on the previous chunk's fence (`AggregationState.wait` in `pipeline._consume`) and the rest
(host work: labeled arrays, statistics, plan look-ups, launches).  A pass whose host work per chunk exceeds its kernels'
time is host-bound: its waits are ~0.  Usage (GPU box): python tools/config5_host_split.py [inits]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ninit = int(sys.argv[1]) if len(sys.argv) > 1 else 120
sys.argv = ['bench.py', '--legs', 'config5', '--no-cpu', '--config5-inits', str(ninit)]
import bench  # noqa: E402
from weatherbenchx_amd import aggregation, pipeline  # noqa: E402

waited = [0.0, 0]
calls = []
stamps = []
_wait = aggregation.AggregationState.wait


def wait(self):
  t0 = time.perf_counter()
  out = _wait(self)
  t1 = time.perf_counter()
  waited[0] += t1 - t0
  waited[1] += 1
  stamps.append(t1)
  return out


_evaluate = pipeline.evaluate_chunks


def evaluate_chunks(times, *a, **k):
  waited[0], waited[1] = 0.0, 0
  del stamps[:]
  t0 = time.perf_counter()
  out = _evaluate(times, *a, **k)
  t1 = time.perf_counter()
  n = len(list(times.iter_with_chunk_offsets()))
  per = max(1, waited[1] // max(n, 1))  # waits per chunk: the last one of a chunk marks the end of its kernels
  ends = [stamps[i] - t0 for i in range(per - 1, len(stamps), per)]
  calls.append((n, t1 - t0, waited[0], waited[1], ends, t1 - t0))
  return out


aggregation.AggregationState.wait = wait
pipeline.evaluate_chunks = evaluate_chunks
args = bench.parse()
env = bench.Env(args)
res = bench.config5_leg(env)
print('ms per chunk by pass (bench):', res['ms_per_chunk_by_pass_rank0'])
for name, (n, wall, w, nw, ends, total) in zip(['deterministic', 'spectra', 'ensemble'], calls[-3:]):
  print(f'{name:14s} {n} chunks: wall {wall / n * 1e3:.3f} ms/chunk, blocked on fences {w / n * 1e3:.3f} ms/chunk '
        f'({nw / n:.1f} waits), host work {(wall - w) / n * 1e3:.3f} ms/chunk')
  d = [(b - a) * 1e3 for a, b in zip(ends[:-1], ends[1:])]
  mid = d[len(d) // 4:]
  print(f'    first chunk done at {ends[0] * 1e3:.2f} ms; next intervals {[round(x, 2) for x in d[:12]]}; '
        f'mean of the last three quarters {sum(mid) / len(mid):.3f} ms; last chunk done at {ends[-1] * 1e3:.1f} ms, '
        f'evaluate_chunks returned at {total * 1e3:.1f} ms')
