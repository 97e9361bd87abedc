#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace run (rocpd SQLite): start offset, duration and the gap to the previous kernel,
for the last N kernels -- what the GPU does between the dominant launches of a pipelined loop.
usage: python tools/timeline.py <results.db> [N]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
rows = con.execute('select name, start, end, duration' + (', stream_id' if 'stream_id' in cols else ', 0') +
                   ' from kernels order by start').fetchall()
rows = rows[-n:]
t0 = rows[0][1]
prev_end = None
for name, s, e, d, stream in rows:
  gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
  short = name.replace('void ', '').replace('wbx::', '')[:70]
  print(f'{(s - t0) / 1e3:10.1f} us  +{d / 1e3:8.1f} us  gap {gap:8.1f} us  stream {stream}  {short}')
  prev_end = max(prev_end, e) if prev_end is not None else e
