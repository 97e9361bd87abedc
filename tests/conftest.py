"""pytest configuration: `gpu` marker + the two execution backends of the host-side tests.

`backend` fixture:
  * 'emulated' -- tests/fake_device.py interprets the stage-1/stage-2 plans with NumPy by monkeypatching the
    launch functions of weatherbenchx_amd.engine.  It exists so the planner / labeled-array / Aggregator logic
    is exercised on a GPU-less box; it is test infrastructure, never part of the product path.
  * 'hip'      -- the real libwbx_hip.so on cuda:0 (marked gpu).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
  if p not in sys.path:
    sys.path.insert(0, p)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(params=['emulated', pytest.param('hip', marks=pytest.mark.gpu)])
def backend(request, monkeypatch):
  from weatherbenchx_amd import engine
  engine.clear_caches()
  if request.param == 'emulated':
    import fake_device
    fake_device.install(monkeypatch)
  else:
    from weatherbenchx_amd import _hip
    if not _hip.is_available():
      pytest.fail('gpu test selected but libwbx_hip.so / a HIP device is not available')
  yield request.param
  engine.clear_caches()
