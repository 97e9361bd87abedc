"""Executes the ctypes binding printed in INTEGRATION.md (Level 2) verbatim against libwbx_hip.so and checks it
against the oracle -- so the documented reference-side stub is known to work, not just to read well."""
import os
import re

import numpy as np
import pytest

from oracle import wbx_oracle as O
from weatherbenchx_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
  text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
  blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
  block = next(b for b in blocks if '_wbx_hip.py' in b)
  return block.replace("C.CDLL('libwbx_hip.so')", f"C.CDLL({_hip.lib_path()!r})")


def test_stub_is_present_and_names_real_symbols():
  src = _stub_source()
  for sym in re.findall(r'_lib\.(wbx_[a-z0-9_]+)', src):
    assert sym in _hip.EXPORTED_SYMBOLS, sym


@pytest.mark.gpu
def test_stub_runs_and_matches_oracle():
  ns = {}
  exec(compile(_stub_source(), 'INTEGRATION.md', 'exec'), ns)  # pylint: disable=exec-used
  rng = np.random.default_rng(0)
  p = rng.normal(size=(3, 2, 2, 32, 64)).astype(np.float32)
  t = rng.normal(size=(3, 2, 2, 32, 64)).astype(np.float32)
  lat = np.linspace(-87.1875, 87.1875, 32)
  w = O.grid_area_weights(lat)
  e, ae, se = ns['weighted_sums_lon_fastest'](p, t, w)
  dims = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
  for got, fn in ((e, O.error), (ae, O.absolute_error), (se, O.squared_error)):
    want, _, _ = O.aggregate(fn(p, t), dims, ['init_time', 'latitude', 'longitude'], weights=[(w, ('latitude',))])
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)


def _replay_source():
  text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
  blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
  return next(b for b in blocks if 'def replay_chunks' in b)


def test_chunk_record_stub_names_real_symbols_and_function_ids():
  src = _replay_source()
  for sym in re.findall(r'_lib\.(wbx_[a-z0-9_]+)', src):
    assert sym in _hip.EXPORTED_SYMBOLS, sym
  m = re.search(r'(WBX_FN_\w+), (WBX_FN_\w+), (WBX_FN_\w+) = (\d+), (\d+), (\d+)', src)
  ids = dict(zip(m.groups()[:3], map(int, m.groups()[3:])))
  assert ids == {'WBX_FN_DET_PARTIAL': _hip.FN_IDS['wbx_det_partial'], 'WBX_FN_CONTRACT': _hip.FN_IDS['wbx_contract'],
                 'WBX_FN_ACC_ADD': _hip.FN_IDS['wbx_acc_add']}
  header = open(os.path.join(ROOT, 'include', 'wbx.h')).read()
  for name, val in ids.items():
    assert re.search(rf'{name} = {val}\b', header), name


@pytest.mark.gpu
def test_chunk_record_stub_replays_chunks():
  """The chunk-record binding of INTEGRATION.md, verbatim: three chunks replayed from one call list add up to what the level-2
  stub computes chunk by chunk."""
  import ctypes as C
  ns = {}
  exec(compile(_stub_source(), 'INTEGRATION.md', 'exec'), ns)  # pylint: disable=exec-used
  exec(compile(_replay_source(), 'INTEGRATION.md', 'exec'), ns)  # pylint: disable=exec-used
  lib, ctx, dev, check = ns['_lib'], ns['ctx'], ns['_dev'], ns['_check']
  rng = np.random.default_rng(1)
  ni, nl, nz, ny, nx = 2, 2, 3, 16, 64
  lat = np.linspace(-84.375, 84.375, ny)
  w = np.ascontiguousarray(O.grid_area_weights(lat), np.float64)
  chunks = [(rng.normal(size=(ni, nl, nz, ny, nx)).astype(np.float32), rng.normal(size=(ni, nl, nz, ny, nx)).astype(np.float32)) for _ in range(3)]
  want = sum(np.stack(ns['weighted_sums_lon_fastest'](p, t, w), axis=-1) for p, t in chunks)
  key = (np.arange(nl)[:, None, None] * nz * ny * nx + np.arange(nz)[None, :, None] * ny * nx + np.arange(ny)[None, None, :] * nx).reshape(-1).astype(np.int64)
  depth = (np.arange(ni) * nl * nz * ny * nx).astype(np.int64)
  plan = ns['S1Plan'](nkey=key.size, ndepth=ni, nx=nx, x_kept=0, nchunk=1, depth_chunk=ni, block_threads=256, vec=4)
  dk, dd = dev(key), dev(depth)
  for i in (0, 1):
    plan.xstride[i], plan.key_off[i], plan.depth_off[i] = 1, dk.value, dd.value
  s2 = ns['S2Plan'](nA=nl * nz, nBk=1, nBr=ny, nchunk=1, nlane=3, nj=1, nbin=1, sum_j=0)
  n = nl * nz * 3
  partial, out, acc = C.c_void_p(), C.c_void_p(), C.c_void_p()
  check(lib.wbx_malloc(ctx, key.size * 3 * 8, C.byref(partial)))
  check(lib.wbx_malloc(ctx, n * 8, C.byref(out)))
  check(lib.wbx_malloc(ctx, n * 8, C.byref(acc)))
  check(lib.wbx_memset(ctx, acc, 0, n * 8))
  ns['replay_chunks'](plan, s2, partial, dev(w), out, acc, n, [(dev(p), dev(t)) for p, t in chunks])
  host = np.empty((nl, nz, 3))
  check(lib.wbx_memcpy_d2h(ctx, host.ctypes.data_as(C.c_void_p), acc, host.nbytes))
  np.testing.assert_allclose(host, want, rtol=1e-12, atol=1e-12)
