#!/usr/bin/env python3
"""Host-side profile of bench.py's public-chunk loop: cProfile around the timed loop only (the kernels run asynchronously, so
what is listed is what the Python side of a chunk costs).  usage: python tools/profile_host_chunk.py [lon_fastest|lat_fastest] [--small]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = (['bench.py'] + (['--small'] if '--small' in sys.argv else []) +  # --small: tiny grids = the Python cost alone
            ['--legs', os.environ.get('WBX_PROFILE_LEG', 'public_chunk'), '--no-cpu', '--no-config5', '--steps', '300', '--warmup', '20',
             '--layout', 'lat_fastest' if 'lat_fastest' in sys.argv else 'lon_fastest'] + os.environ.get('WBX_PROFILE_ARGS', '').split())
import bench  # noqa: E402

prof = cProfile.Profile()
orig = bench.pipelined
state = {'t': None}


def pipelined(launch, finish, n):
  if n < 100:
    return orig(launch, finish, n)
  t0 = time.perf_counter()
  out = orig(launch, finish, n)  # un-profiled: the native host time per chunk (an upper bound: it includes waiting on fences)
  state['native'] = (time.perf_counter() - t0) / n * 1e3
  prof.enable()
  out = orig(launch, finish, n)
  prof.disable()
  return out


bench.pipelined = pipelined
bench.main()
sys.stdout.flush()
print('\nwall per chunk without the profiler: %.4f ms' % state.get('native', float('nan')), file=sys.stderr)
st = pstats.Stats(prof, stream=sys.stderr)
st.sort_stats('cumulative').print_stats(45)
st.sort_stats('tottime').print_stats(40)
