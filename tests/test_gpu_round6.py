"""Round 6 on the device: stage 2 of the deterministic lanes folded into the fused det + spectra sweep
(wbx_det_spectrum_folded), through the raw C ABI and through the chunk loop (records on), against wbx_det_spectrum + wbx_contract
and the float64 oracle; the host transposition entry point; the clock probe."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import wbx_oracle as O  # noqa: E402

from weatherbenchx_amd import _hip  # noqa: E402
from weatherbenchx_amd import aggregation  # noqa: E402
from weatherbenchx_amd import engine  # noqa: E402
from weatherbenchx_amd import pipeline  # noqa: E402
from weatherbenchx_amd import planner  # noqa: E402
from weatherbenchx_amd import replay  # noqa: E402
from weatherbenchx_amd import spectra  # noqa: E402
from weatherbenchx_amd import time_chunks  # noqa: E402
from weatherbenchx_amd import weighting  # noqa: E402
from weatherbenchx_amd import xarray_lite as xr  # noqa: E402
from weatherbenchx_amd.metrics import deterministic  # noqa: E402

pytestmark = pytest.mark.gpu


def _torch():
  import torch
  if not _hip.is_available():
    pytest.fail('gpu test selected but libwbx_hip.so / a HIP device is not available')
  return torch


@pytest.mark.parametrize('func', ['DET6', 'DET3'])
def test_folded_entry_point_equals_partial_times_weights(func):
  """wbx_det_spectrum_folded(det_scale = w[lat]) == sum over latitude of w[lat] x wbx_det_spectrum's per-row partial, per
  (lead, level) group and lane, to fp64 rounding of another summation order; both spectra bit-identical to wbx_det_spectrum's (the
  same records); a team that walks several groups, a group that spans several teams; twenty runs bit-identical."""
  torch = _torch()
  ctx = _hip.default_context(0)
  nlead, nlev, nlat, nlon = 3, 2, 181, 1440
  dims = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
  shape = (1, nlead, nlev, nlat, nlon)
  g = torch.Generator(device='cuda')
  g.manual_seed(3)
  arrs = [xr.DataArray(torch.randn(shape, generator=g, device='cuda') * (3 if i < 2 else 10) + 280, dims=dims) for i in range(3)]
  torch.cuda.synchronize()
  devs = [engine._to_device(ctx, a, _hip.F32) for a in arrs]
  lays = [d.layout for d in devs] + [None]
  if func == 'DET3':
    lays[2] = None
  plan = planner.build_s1_plan(dims, dict(zip(dims, shape)), lays, ['init_time', 'latitude', 'longitude'], wdep_dims=['latitude'])
  assert plan.nkey == nlead * nlev * nlat and plan.ndepth == 1 and plan.nchunk == 1
  dplan = engine._device_plan(ctx, plan)
  code = getattr(_hip, func)
  nl = _hip.DET_LANES[code]
  nrows, ngroup, nk = plan.nkey, nlead * nlev, nlon // 2 + 1
  w = O.grid_area_weights(np.linspace(-90, 90, nlat))
  group = np.repeat(np.arange(ngroup, dtype=np.int32), nlat)
  scale = np.tile(w / nlat, ngroup)
  dscale = np.tile(w, ngroup)
  g_dev, s_dev, d_dev = ctx.upload(group), ctx.upload(scale), ctx.upload(dscale)
  part = ctx.alloc(nrows * nl * 8)
  det = ctx.alloc(ngroup * nl * 8)
  pw = [ctx.alloc(ngroup * nk * 8) for _ in range(4)]
  ptr = lambda d: C.c_void_p(d.ptr) if d is not None else None
  cdev = devs[2] if func == 'DET6' else None
  _hip.check(ctx.lib.wbx_det_spectrum(ctx.handle, C.byref(dplan.struct), code, _hip.F32, ptr(devs[0]), ptr(devs[1]), ptr(cdev),
                                      ptr(g_dev), ptr(s_dev), ngroup, ptr(part), ptr(pw[0]), ptr(pw[1])), 'wbx_det_spectrum')
  runs = []
  for _ in range(20):
    _hip.check(ctx.lib.wbx_det_spectrum_folded(ctx.handle, C.byref(dplan.struct), code, _hip.F32, ptr(devs[0]), ptr(devs[1]), ptr(cdev),
                                               ptr(g_dev), ptr(s_dev), ptr(d_dev), ngroup, ptr(det), ptr(pw[2]), ptr(pw[3])),
               'wbx_det_spectrum_folded')
    ctx.synchronize()
    runs.append((ctx.download(det.ptr, (ngroup, nl)).copy(), ctx.download(pw[2].ptr, (ngroup, nk)).copy(),
                 ctx.download(pw[3].ptr, (ngroup, nk)).copy()))
  for r in runs[1:]:
    for a, b in zip(runs[0], r):
      assert np.array_equal(a, b)
  rows = ctx.download(part.ptr, (nrows, nl)).copy()
  want = (rows.reshape(ngroup, nlat, nl) * w[None, :, None]).sum(axis=1)
  scale_of = np.abs(rows.reshape(ngroup, nlat, nl) * w[None, :, None]).sum(axis=1)
  assert np.all(np.abs(runs[0][0] - want) <= 1e-13 * scale_of)
  np.testing.assert_array_equal(runs[0][1], ctx.download(pw[0].ptr, (ngroup, nk)))
  np.testing.assert_array_equal(runs[0][2], ctx.download(pw[1].ptr, (ngroup, nk)))
  # ... and against the oracle on the fields themselves
  p64, t64, c64 = (np.asarray(a.data.cpu().numpy(), np.float64)[0] for a in arrs)
  lanes = [O.error(p64, t64), O.absolute_error(p64, t64), O.squared_error(p64, t64)]
  if func == 'DET6':
    lanes += [O.squared_prediction_anomaly(p64, c64), O.squared_target_anomaly(t64, c64), O.anomaly_covariance(p64, t64, c64)]
  for l, lane in enumerate(lanes):
    ref = (lane.sum(axis=-1) * w[None, None, :]).sum(axis=-1).reshape(-1)
    tol = 1e-9 * (np.abs(lane).sum(axis=-1) * w[None, None, :]).sum(axis=-1).reshape(-1)
    assert np.all(np.abs(runs[0][0][:, l] - ref) <= tol), l
  # argument checks of the new entry point
  assert ctx.lib.wbx_det_spectrum_folded(ctx.handle, C.byref(dplan.struct), code, _hip.F32, ptr(devs[0]), ptr(devs[1]), ptr(cdev),
                                         ptr(g_dev), ptr(s_dev), None, ngroup, ptr(det), ptr(pw[2]), ptr(pw[3])) != 0
  assert b'NULL' in ctx.lib.wbx_last_error()


def _job(torch, n, nlat=61, nlev=2, nlead=2, seed=8, permuted=False):
  nlon = 1440
  g = torch.Generator(device='cuda')
  g.manual_seed(seed)
  lat, lon = np.linspace(-75, 75, nlat), np.linspace(0, 360, nlon, endpoint=False)
  lead = (np.arange(nlead) * 6).astype('timedelta64[h]').astype('timedelta64[ns]')
  inits = np.datetime64('2020-01-01T00', 'ns') + np.arange(n) * np.timedelta64(24, 'h')
  level = np.arange(nlev)
  zd = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
  fields = [tuple(torch.randn((1, nlead, nlev, nlat, nlon), generator=g, device='cuda') * 3 + 280 for _ in range(2)) for _ in range(n)]
  clim_t = torch.randn((n + 3, 4, nlev, nlat, nlon), generator=g, device='cuda') * 10 + 280
  if permuted:
    # the SAME values in another storage order: level outermost in the fields, level before (dayofyear, hour) in the climatology
    # -- the DataArrays are strided views, the plan's key-offset tables are no longer monotonic in the key
    fields = [tuple(f.permute(0, 2, 1, 3, 4).contiguous().permute(0, 2, 1, 3, 4) for f in pt) for pt in fields]
    clim_t = clim_t.permute(2, 0, 1, 3, 4).contiguous().permute(1, 2, 0, 3, 4)
    assert not fields[0][0].is_contiguous() and not clim_t.is_contiguous()
  clim = xr.Dataset({'z': xr.DataArray(clim_t, dims=('dayofyear', 'hour', 'level', 'latitude', 'longitude'),
                                       coords={'dayofyear': np.arange(1, n + 4), 'hour': np.array([0, 6, 12, 18]), 'level': level,
                                               'latitude': lat, 'longitude': lon})})
  index = {int(t.astype('int64')): i for i, t in enumerate(inits)}

  def load(ic, lc):
    i = index[int(ic[0].astype('int64'))]
    cs = {'init_time': ic, 'lead_time': lead, 'level': level, 'latitude': lat, 'longitude': lon}
    return {'z': xr.DataArray(fields[i][0], dims=zd, coords=cs)}, {'z': xr.DataArray(fields[i][1], dims=zd, coords=cs)}
  det = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE(), 'bias': deterministic.Bias(), 'acc': deterministic.ACC(clim),
         'activity': deterministic.PredictionActivity(clim)}
  spec = {'sp': spectra.ZonalPowerSpectrum('predictions'), 'st': spectra.ZonalPowerSpectrum('targets')}
  area = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  zonal = aggregation.Aggregator(reduce_dims=['init_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
  passes = [('det', load, det, area), ('spec', load, spec, zonal)]
  times = time_chunks.TimeChunks(inits, lead, init_time_chunk_size=1)
  return passes, times, det, spec, fields, clim_t, lat, inits, lead


@pytest.mark.parametrize('records', [True, False])
def test_chunk_loop_folds_stage_two_into_the_fused_sweep(monkeypatch, records):
  """configs[4] in miniature: the deterministic suite + ACC (area mean per (lead, level)) and the zonal spectra of p and t share
  a loader: ONE launch per chunk and variable serves all of it, and with GridAreaWeighting that launch also does the
  deterministic lanes' stage 2 (no partial, no wbx_contract).  Folded == unfolded to fp64 rounding, both equal the oracle; with
  chunk records on the folded call is replayed."""
  torch = _torch()
  n = 8
  out = {}
  for fold in (True, False):
    engine.clear_caches()
    monkeypatch.setattr(engine, 'FOLD_DET_SPECTRA', fold)
    monkeypatch.setattr(replay, 'ENABLED', records)
    replay.reset_stats()
    passes, times, det, spec, fields, clim_t, lat, inits, lead = _job(torch, n)
    log = []
    monkeypatch.setattr(engine, 'S1_EVENT_LOG', log if not records else None)
    st = pipeline.evaluate_passes(times, passes)
    monkeypatch.setattr(engine, 'S1_EVENT_LOG', None)
    out[fold] = (st['det'][None], st['spec'][None], st['det'][None].metric_values(det), st['spec'][None].metric_values(spec))
    if not records:
      fused = [e for e in log if e.get('kind') == 'det_spectrum']
      assert len(fused) == n and all(bool(e['folded']) == fold for e in fused), [(e.get('kind'), e.get('folded')) for e in log][:6]
    else:
      assert replay.STATS['replayed'] >= n - 4, dict(replay.STATS)
  for part in (0, 1):
    for kind in ('sum_weighted_statistics', 'sum_weights'):
      ta, tb = getattr(out[True][part], kind), getattr(out[False][part], kind)
      for stat in ta:
        a, b = np.asarray(ta[stat]['z'].values), np.asarray(tb[stat]['z'].values)
        if part == 1:
          np.testing.assert_array_equal(a, b, err_msg=f'{kind} {stat}')  # (the spectra come out of the same records)
        else:
          np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-9, err_msg=f'{kind} {stat}')
  # the oracle: area-weighted means over (init, lat, lon) per (lead, level)
  p64 = np.stack([np.asarray(f[0].cpu().numpy(), np.float64)[0] for f in fields])
  t64 = np.stack([np.asarray(f[1].cpu().numpy(), np.float64)[0] for f in fields])
  valid = inits[:, None] + lead[None, :]
  cfull = np.asarray(clim_t.cpu().numpy(), np.float64)
  c64, _ = O.align_climatology(cfull, ('dayofyear', 'hour', 'level', 'latitude', 'longitude'), valid, ('init_time', 'lead_time'))
  zdims = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
  w = (O.grid_area_weights(lat), ('latitude',))

  def mean(stat):
    sws, sw, od = O.aggregate(stat, zdims, ['init_time', 'latitude', 'longitude'], weights=[w])
    return sws / sw
  want = {'rmse.z': O.rmse(mean(O.squared_error(p64, t64))), 'mae.z': mean(O.absolute_error(p64, t64)), 'bias.z': mean(O.error(p64, t64)),
          'acc.z': O.acc(mean(O.anomaly_covariance(p64, t64, c64)), mean(O.squared_prediction_anomaly(p64, c64)),
                         mean(O.squared_target_anomaly(t64, c64))),
          'activity.z': np.sqrt(mean(O.squared_prediction_anomaly(p64, c64)))}
  for fold in (True, False):
    for k, v in want.items():
      got = np.asarray(out[fold][2][k].transpose('lead_time', 'level').values)
      np.testing.assert_allclose(got, v, rtol=1e-6, atol=1e-9 if k == 'bias.z' else 0, err_msg=f'{k} fold={fold}')
  engine.clear_caches()


@pytest.mark.parametrize('fold', [True, False])
def test_fused_sweep_on_strided_storage_equals_contiguous(monkeypatch, fold):
  """The fused sweep resolves a row's three base pointers from the plan's offset / gather tables with scalar loads, one row
  ahead (r6).  The same job on fields stored level-outermost and a climatology stored level-first (strided views: key offsets that
  jump back and forth from row to row) must give the sums of the contiguous job bit for bit -- same rows, same order."""
  torch = _torch()
  n = 3
  out = {}
  for permuted in (False, True):
    engine.clear_caches()
    monkeypatch.setattr(engine, 'FOLD_DET_SPECTRA', fold)
    monkeypatch.setattr(replay, 'ENABLED', False)
    passes, times, det, spec, *_ = _job(torch, n, nlat=37, nlev=3, nlead=3, seed=21, permuted=permuted)
    log = []
    monkeypatch.setattr(engine, 'S1_EVENT_LOG', log)
    st = pipeline.evaluate_passes(times, passes)
    monkeypatch.setattr(engine, 'S1_EVENT_LOG', None)
    assert len([e for e in log if e.get('kind') == 'det_spectrum']) == n, [e.get('kind') for e in log]
    out[permuted] = (st['det'][None], st['spec'][None])
  for part in (0, 1):
    for kind in ('sum_weighted_statistics', 'sum_weights'):
      ta, tb = getattr(out[True][part], kind), getattr(out[False][part], kind)
      for stat in ta:
        np.testing.assert_array_equal(np.asarray(ta[stat]['z'].values), np.asarray(tb[stat]['z'].values), err_msg=f'{kind} {stat}')
  engine.clear_caches()


def test_spectrum_entry_points_return_the_pinned_bits():
  """A REGRESSION pin, not a parity vector: sha256 of what wbx_zonal_spectrum (x3, accumulate), wbx_det_spectrum and
  wbx_det_spectrum_folded return for seeded host-generated inputs and four group tables, as the library of round 6 returned them
  (tests/golden/spectra_bits.json, written by `python tools/spectra_bits.py --write` on an MI355X).  The sums are ordered and the
  kernels deterministic, so a build whose hashes differ has changed its ARITHMETIC (round 6: grouping the transform's LDS reads
  changed how the compiler paired multiplies and adds into FMAs -- the last fp32 bit of the fused sweep's spectra)."""
  import subprocess
  _torch()
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  proc = subprocess.run([sys.executable, os.path.join(root, 'tools', 'spectra_bits.py')], capture_output=True, text=True, timeout=300)
  assert proc.returncode == 0, proc.stderr[-2000:]
  assert 'golden: equal' in proc.stdout, proc.stdout[-2000:]


def test_clock_probe_and_host_transpose_on_the_box():
  _torch()
  ctx = _hip.default_context(0)
  one, busy = ctx.clock_probe(1), ctx.clock_probe(2048)
  assert 500.0 < busy <= one * 1.05 < 4000.0, (one, busy)
  src = np.random.default_rng(0).standard_normal((3, 1440, 721)).astype(np.float32)
  dst = ctx.pinned_empty((3, 721, 1440), np.float32)
  _hip.check(ctx.lib.wbx_host_transpose(dst.ctypes.data, src.ctypes.data, 3, 1440, 721, 4), 'wbx_host_transpose')
  np.testing.assert_array_equal(dst, np.swapaxes(src, 1, 2))
