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
