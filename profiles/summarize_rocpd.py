#!/usr/bin/env python3
"""Turns rocprofv3's rocpd SQLite output (ROCm 7.2 default) into the compact text summaries committed here.

usage: summarize_rocpd.py <results.db> [more.db ...] > profiles/rNN_<what>.txt
For every kernel: calls, total / average / min / max / MEDIAN duration (us) -- the median is what bench.py's HIP-event figure
must agree with (the average carries the first launches after an idle gap: clocks, cold L2); for PMC runs: per-kernel counter averages.
FETCH_SIZE / WRITE_SIZE are in KiB as reported; on gfx950 FETCH_SIZE under-reports wide coalesced reads by
exactly 2x (/opt/skills/guides/MI355X_MICROARCH.md, HBM section) -- the `hbm_read_bytes_corrected` column is
FETCH_SIZE * 1024 * 2.
"""
import sqlite3
import sys


def short(name, n=110):
  name = name.replace('void ', '')
  return name if len(name) <= n else name[:n - 3] + '...'


def main():
  for path in sys.argv[1:]:
    con = sqlite3.connect(path)
    cur = con.cursor()
    print(f'== {path}')
    per = {}
    for name, dur in cur.execute('select name, duration from kernels'):
      per.setdefault(name, []).append(dur)
    rows = sorted(((n, len(d), sum(d), sum(d) / len(d), min(d), max(d), sorted(d)[len(d) // 2]) for n, d in per.items()),
                  key=lambda r: -r[2])
    total = sum(r[2] for r in rows) or 1
    print(f'{"kernel":112s} {"calls":>6s} {"total_us":>12s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s} {"pct":>6s} {"median_us":>10s}')
    for name, calls, tot, avg, mn, mx, med in rows[:25]:
      print(f'{short(name):112s} {calls:6d} {tot / 1e3:12.1f} {avg / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} '
            f'{100 * tot / total:6.2f} {med / 1e3:10.2f}')
    try:
      pmc = cur.execute('select kernel_name, counter_name, count(*), avg(value) from counters_collection '
                        'group by kernel_name, counter_name order by kernel_name').fetchall()
    except sqlite3.Error:
      pmc = []
    if pmc:
      print(f'\n{"kernel":112s} {"counter":>14s} {"n":>5s} {"avg":>16s} {"bytes (x1024, FETCH x2)":>26s}')
      for k, c, n, v in pmc:
        if 'wbx::' not in k:
          continue
        b = v * 1024 * (2 if c == 'FETCH_SIZE' else 1) if c in ('FETCH_SIZE', 'WRITE_SIZE') else float('nan')
        print(f'{short(k):112s} {c:>14s} {n:5d} {v:16.1f} {b:26.0f}')
    print()


if __name__ == '__main__':
  main()
