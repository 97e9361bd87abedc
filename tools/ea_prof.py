"""Phase stamps of ens_atoms_kernel (make ab-eaprof; WBX_EA_PROF_DUMP=<file>): where a launch's time goes.
stamps per patch: 0 start, 1 sweep done, 2 flushed, 3 arrived (level 1), 4 level 1 done, 5 arrived (level 2), 6 level 2 done, 7 cell written
(wall_clock64: 100 MHz)."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.int64)
ncell, nrs, nxt, ns = raw[:4]
t = raw[4:].reshape(-1, ns).astype(np.float64)
ok = t[:, 0] > 0
t0 = t[ok, 0].min()
us = (t - t0) / 100.0
us[t == 0] = np.nan
print(f'patches {ok.sum()}  cells {ncell}  nrs {nrs}  nxt {nxt}')
print(f'kernel span (first start -> last stamp): {np.nanmax(us):8.1f} us')
print(f'last wave START                  : {np.nanmax(us[:, 0]):8.1f}')
print(f'last sweep end                   : {np.nanmax(us[:, 1]):8.1f}')
print(f'last flush end                   : {np.nanmax(us[:, 2]):8.1f}')
print(f'last level-1 arrival             : {np.nanmax(us[:, 3]):8.1f}')
print(f'last level-1 done                : {np.nanmax(us[:, 4]):8.1f}')
print(f'last level-2 arrival             : {np.nanmax(us[:, 5]):8.1f}')
print(f'last level-2 done                : {np.nanmax(us[:, 6]):8.1f}')
print(f'last cell written                : {np.nanmax(us[:, 7]):8.1f}')
d = lambda a, b: us[:, b] - us[:, a]
for name, a, b in (('prologue (decode, table pointers, pre-zero)', 0, 8), ('row lookups + first issue', 8, 9), ('first row in flight', 9, 10),
                   ('first row -> second row landed', 10, 11), ('sweep', 0, 1), ('flush', 1, 2), ('arrive1 (store ack + atomic)', 2, 3), ('level 1', 3, 4), ('arrive2', 4, 5), ('level 2', 5, 6),
                   ('arrive3 + level 3', 6, 7)):
  x = d(a, b)
  x = x[~np.isnan(x)]
  if b >= ns:
    continue
  print(f'{name:44s} n {x.size:6d}  median {np.median(x):8.2f}  mean {x.mean():8.2f}  p95 {np.percentile(x, 95):8.2f}  max {x.max():8.2f} us')
# waves in flight over time
st, en = us[ok, 0], np.nanmax(us[ok], axis=1)
for q in np.linspace(0, np.nanmax(us), 12):
  print(f't = {q:7.1f} us: {int(((st <= q) & (en > q)).sum()):5d} waves in flight')
