"""GPU-only parity tests that go straight through the C ABI (libwbx_hip.so) -- edge cases, error behaviour and
size-independent properties at BASELINE.json's full grid (721 x 1440), where the oracle would be too slow
to brute-force everything.  Tolerance: rtol 1e-6 (north_star); fp64 accumulation actually gives ~1e-12."""
import ctypes as C

import numpy as np
import pytest

from oracle import wbx_oracle as O
from weatherbenchx_amd import _hip
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import engine
from weatherbenchx_amd import planner
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.metrics import deterministic
from weatherbenchx_amd.metrics import probabilistic

pytestmark = pytest.mark.gpu
RTOL = 1e-6


@pytest.fixture(scope='module')
def ctx():
  assert _hip.is_available(), 'gpu tests need libwbx_hip.so and a HIP device'
  return _hip.default_context(0)


def _reduce(inputs, dims, sizes, reduce_dims, func, **kw):
  return engine.reduce_statistics('det', inputs, dims, sizes, reduce_dims, None, (), func=func, **kw)


def test_device_is_gfx950(ctx):
  assert 'gfx950' in ctx.device_name()


@pytest.mark.parametrize('nx', [1, 3, 4, 63, 64, 65, 255, 257, 1024, 1444])
def test_ragged_row_lengths_both_modes(ctx, nx):
  rng = np.random.default_rng(nx)
  p = xr.DataArray(rng.normal(size=(5, 7, nx)).astype(np.float32), dims=('a', 'b', 'x'))
  t = xr.DataArray(rng.normal(size=(5, 7, nx)).astype(np.float32), dims=('a', 'b', 'x'))
  sizes = {'a': 5, 'b': 7, 'x': nx}
  e = p.values.astype(np.float64) - t.values
  # x summed (XR kernel; vec4 when nx % 4 == 0)
  vals, cnt, od = _reduce([p, t], p.dims, sizes, ['b', 'x'], _hip.DET3)
  assert od == ('a',)
  np.testing.assert_allclose(vals[2], (e * e).sum(axis=(1, 2)), rtol=1e-12)
  np.testing.assert_allclose(cnt[0], 7 * nx)
  # x kept (XK kernel)
  vals, cnt, od = _reduce([p, t], p.dims, sizes, ['a', 'b'], _hip.DET3)
  assert od == ('x',)
  np.testing.assert_allclose(vals[1], np.abs(e).sum(axis=(0, 1)), rtol=1e-12)
  np.testing.assert_allclose(vals[0], e.sum(axis=(0, 1)), rtol=1e-9, atol=1e-12)


def test_empty_reductions(ctx):
  p = xr.DataArray(np.zeros((3, 0, 8), np.float32), dims=('a', 'b', 'x'))
  vals, cnt, od = _reduce([p, p], p.dims, {'a': 3, 'b': 0, 'x': 8}, ['b', 'x'], _hip.DET3)
  np.testing.assert_array_equal(vals[2], np.zeros(3))
  np.testing.assert_array_equal(cnt[2], np.zeros(3))
  q = xr.DataArray(np.zeros((0, 4), np.float32), dims=('a', 'x'))
  vals, _, od = _reduce([q, q], q.dims, {'a': 0, 'x': 4}, ['x'], _hip.DET3)
  assert vals[0].shape == (0,)


def test_vec4_and_scalar_paths_agree(ctx):
  rng = np.random.default_rng(0)
  base = rng.normal(size=(6, 9, 130)).astype(np.float32)
  p = xr.DataArray(base[:, :, :128], dims=('a', 'b', 'x'))       # sliced view -> contiguous upload, nx % 4 == 0
  t = xr.DataArray(base[:, :, 1:129].copy(), dims=('a', 'b', 'x'))
  sizes = {'a': 6, 'b': 9, 'x': 128}
  lays = [engine._to_device(ctx, v, _hip.F32).layout for v in (p, t)]
  plan4 = planner.build_s1_plan(p.dims, sizes, lays, ['b', 'x'])
  plan1 = planner.build_s1_plan(p.dims, sizes, lays, ['b', 'x'], allow_vec4=False)
  assert plan4.vec == 4 and plan1.vec == 1
  outs = []
  for plan in (plan4, plan1):
    devs = [engine._to_device(ctx, v, _hip.F32) for v in (p, t)] + [None, None]
    buf = engine._run_s1(ctx, 'det', engine._device_plan(ctx, plan), plan, devs, _hip.F32, 3, func=_hip.DET3)
    outs.append(ctx.download(buf.ptr, plan.partial_shape(3)))
  np.testing.assert_allclose(outs[0].sum(axis=3), outs[1].sum(axis=3), rtol=1e-13)


def test_bad_arguments_return_error_codes(ctx):
  lib = ctx.lib
  plan = _hip.S1PlanStruct()
  plan.nkey, plan.ndepth, plan.nx, plan.nchunk, plan.depth_chunk, plan.block_threads, plan.vec = 1, 1, 4, 1, 1, 100, 1
  rc = lib.wbx_det_partial(ctx.handle, C.byref(plan), _hip.DET3, _hip.F32, None, None, None, None, None)
  assert rc == -1 and b'block_threads' in lib.wbx_last_error()
  plan.block_threads = 64
  rc = lib.wbx_det_partial(ctx.handle, C.byref(plan), 7, _hip.F32, C.c_void_p(8), C.c_void_p(8), None, None, C.c_void_p(8))
  assert rc == -1 and b'unknown deterministic family' in lib.wbx_last_error()
  rc = lib.wbx_ens_partial(ctx.handle, C.byref(plan), _hip.F32, 0, 1, 0, C.c_void_p(8), C.c_void_p(8), None, C.c_void_p(8))
  assert rc == -1 and b'ensemble size' in lib.wbx_last_error()
  with pytest.raises(_hip.WbxError, match='lane'):
    _hip.check(lib.wbx_det_map(ctx.handle, C.byref(plan), _hip.DET3, _hip.F32, 5, C.c_void_p(8), C.c_void_p(8), None,
                               C.c_void_p(8)), 'wbx_det_map')
  h = C.c_void_p(0)
  assert lib.wbx_ctx_create(99, None, C.byref(h)) == -1


def test_nan_poisons_every_bin_but_skipna_counts_it_out(ctx):
  from weatherbenchx_amd import binning
  lat, lon = np.linspace(-85, 85, 18), np.arange(36) * 10.0
  pv = np.ones((18, 36), np.float32)
  pv[15, 3] = np.nan  # northern hemisphere only
  coords = {'latitude': lat, 'longitude': lon}
  p = {'v': xr.DataArray(pv, dims=('latitude', 'longitude'), coords=coords)}
  t = {'v': xr.DataArray(np.zeros((18, 36), np.float32), dims=('latitude', 'longitude'), coords=coords)}
  regions = binning.Regions({'north': ((0, 90), (0, 360)), 'south': ((-90, 0), (0, 360))})
  m = {'mse': deterministic.MSE()}
  res = aggregation.compute_metric_values_for_single_chunk(
      m, aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], bin_by=[regions]), p, t)
  assert np.isnan(res['mse.v'].values).all()  # NaN * 0 = NaN: aggregation.py:272-277
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], bin_by=[regions], skipna=True)
  state = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(m, p, t))
  np.testing.assert_allclose(state.sum_weights['SquaredError']['v'].values, [9 * 36 - 1, 9 * 36])
  np.testing.assert_allclose(state.metric_values(m)['mse.v'].values, [1.0, 1.0])


# ---- full-grid properties (721 x 1440, BASELINE.json sizes) ------------------------------------------------
NLAT, NLON = 721, 1440
LAT = np.linspace(-90, 90, NLAT)
LON = np.linspace(0, 360, NLON, endpoint=False)


def _torch_field(shape, seed, offset=280.0):
  import torch
  g = torch.Generator(device='cuda')
  g.manual_seed(seed)
  return torch.randn(shape, generator=g, device='cuda', dtype=torch.float32) + offset


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
def test_full_grid_constant_error_is_exact_and_layouts_agree(ctx, layout):
  import torch
  sp = ('latitude', 'longitude') if layout == 'lon_fastest' else ('longitude', 'latitude')
  dims = ('init_time', 'lead_time', 'level') + sp
  n = {'init_time': 3, 'lead_time': 2, 'level': 2, 'latitude': NLAT, 'longitude': NLON}
  shape = tuple(n[d] for d in dims)
  tt = _torch_field(shape, 1)
  pt = tt + 1.5
  coords = {'latitude': LAT, 'longitude': LON}
  p = {'z': xr.DataArray(pt, dims=dims, coords=coords)}
  t = {'z': xr.DataArray(tt, dims=dims, coords=coords)}
  metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE(), 'bias': deterministic.Bias()}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  res = aggregation.compute_metric_values_for_single_chunk(metrics, agg, p, t)
  # fp32(t + 1.5) - t is 1.5 up to fp32 rounding of the sum: |e - 1.5| <= ulp(281.5)/2 ~ 1.5e-5
  for k in ('rmse.z', 'mae.z', 'bias.z'):
    assert res[k].shape == (2, 2)
    np.testing.assert_allclose(res[k].values, 1.5, rtol=2e-5)
  # exact cross-check against torch fp64 on the same device data (independent code path)
  e = (pt.double() - tt.double())
  w = torch.as_tensor(O.grid_area_weights(LAT), device='cuda')
  wshape = [1] * 5
  wshape[dims.index('latitude')] = NLAT
  red = tuple(dims.index(d) for d in ('init_time', 'latitude', 'longitude'))
  want = ((e * e) * w.reshape(wshape)).sum(dim=red) / (w.sum() * NLON * 3)
  np.testing.assert_allclose(res['rmse.z'].values, np.sqrt(want.cpu().numpy()), rtol=1e-10)


def test_full_grid_chunking_is_invisible(ctx):
  """Splitting depth in many chunks or few must not change the sums (different launch geometry, same bytes)."""
  dims = ('init_time', 'latitude', 'longitude')
  pt, tt = _torch_field((8, NLAT, NLON), 2), _torch_field((8, NLAT, NLON), 3)
  p, t = xr.DataArray(pt, dims=dims), xr.DataArray(tt, dims=dims)
  sizes = {'init_time': 8, 'latitude': NLAT, 'longitude': NLON}
  lays = [planner.layout_of(v.data, dims) for v in (p, t)]
  outs = []
  for target in (1, 64, 100000):
    plan = planner.build_s1_plan(dims, sizes, lays, ['init_time', 'longitude'], wdep_dims=['latitude'],
                                 target_blocks=target)
    devs = [engine._to_device(ctx, v, _hip.F32) for v in (p, t)] + [None, None]
    buf = engine._run_s1(ctx, 'det', engine._device_plan(ctx, plan), plan, devs, _hip.F32, 3, func=_hip.DET3)
    outs.append(ctx.download(buf.ptr, plan.partial_shape(3)).sum(axis=(3, 5)))
  np.testing.assert_allclose(outs[0], outs[1], rtol=1e-12)
  np.testing.assert_allclose(outs[0], outs[2], rtol=1e-12)


@pytest.mark.parametrize('use_sort', [True, False])
def test_full_grid_ensemble_closed_form(ctx, use_sort):
  """Members t + k*d (k = 0..M-1): skill = d(M-1)/2, fair spread = d(M+1)/3, var = d^2 M(M+1)/12 -- exact."""
  import torch
  m, d = 51, 0.25
  tt = _torch_field((NLAT, NLON), 4, offset=0.0).round()  # integers: t + k*d is exact in fp32
  k = torch.arange(m, device='cuda', dtype=torch.float32)
  perm = torch.randperm(m, device='cuda')
  pt = tt[None] + (k[perm] * d)[:, None, None]
  coords = {'latitude': LAT, 'longitude': LON}
  p = {'v': xr.DataArray(pt, dims=('number', 'latitude', 'longitude'), coords=coords)}
  t = {'v': xr.DataArray(tt, dims=('latitude', 'longitude'), coords=coords)}
  metrics = {'skill': probabilistic.CRPSSkill(), 'spread': probabilistic.CRPSSpread(use_sort=use_sort),
             'var': probabilistic.EnsembleVariance(), 'crps': probabilistic.CRPSEnsemble(use_sort=use_sort)}
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  res = aggregation.compute_metric_values_for_single_chunk(metrics, agg, p, t)
  np.testing.assert_allclose(res['skill.v'].values, d * (m - 1) / 2, rtol=1e-12)
  np.testing.assert_allclose(res['spread.v'].values, d * (m + 1) / 3, rtol=1e-6 if not use_sort else 1e-12)
  np.testing.assert_allclose(res['var.v'].values, d * d * m * (m + 1) / 12, rtol=1e-10)
  np.testing.assert_allclose(res['crps.v'].values, d * (m - 1) / 2 - 0.5 * d * (m + 1) / 3, rtol=1e-6)


def test_lds_tiled_pair_form_diagnostic_agrees_with_the_register_pair_form(ctx):
  """The north_star's LDS-tiled pairwise CRPS spread exists as a diagnostic instantiation (algo 98, tools/kbench.py times it
  next to the register-tiled pair form `use_sort=False` runs): same 1275 terms per point, other association of the fp32 row
  sums -> the partial sums agree to fp32 round-off of the rows, the other lanes exactly."""
  import torch
  m, ns = 51, 2
  g = torch.Generator(device='cuda')
  g.manual_seed(11)
  tt = torch.randn((ns, NLAT, NLON), generator=g, device='cuda') * 3.7 + 1.3
  pt = tt[:, None] + torch.randn((ns, m, NLAT, NLON), generator=g, device='cuda') * 2.9
  torch.cuda.synchronize()  # (engine._run_s1 is called directly: nothing orders torch's stream in front of the launch stream)
  p = xr.DataArray(pt, dims=('lead_time', 'number', 'latitude', 'longitude'))
  t = xr.DataArray(tt, dims=('lead_time', 'latitude', 'longitude'))
  devs = [engine._to_device(ctx, p, _hip.F32), engine._to_device(ctx, t, _hip.F32), None, None]
  lays = [d.layout if d else None for d in devs]
  sizes = {'lead_time': ns, 'latitude': NLAT, 'longitude': NLON}
  plan = planner.build_s1_plan(('lead_time', 'latitude', 'longitude'), sizes, lays, ['latitude', 'longitude'],
                               wdep_dims=['latitude'], allow_vec4=False, flags=_hip.FLAG_FAIR)
  plan.block_threads = 64
  dplan = engine._device_plan(ctx, plan)
  got = {}
  for algo in (_hip.ENS_PAIRWISE, 98):
    part = engine._run_s1(ctx, 'ens', dplan, plan, devs, _hip.F32, 5, ens=(m, devs[0].layout.stride('number'), algo))
    got[algo] = ctx.download(part.ptr, (plan.nkey * plan.nchunk, 5)).copy()
  a, b = got[_hip.ENS_PAIRWISE], got[98]
  assert np.isfinite(a).all() and a[:, 1].min() > 0
  np.testing.assert_allclose(b[:, 1], a[:, 1], rtol=5e-7)
  np.testing.assert_array_equal(b[:, [0, 2, 3, 4]], a[:, [0, 2, 3, 4]])
  plan.block_threads = 256  # the tile is sized for one-wave blocks: anything else is refused, not mis-run
  with pytest.raises(_hip.WbxError):
    engine._run_s1(ctx, 'ens', engine._device_plan(ctx, plan), plan, devs, _hip.F32, 5,
                   ens=(m, devs[0].layout.stride('number'), 98))


def test_oracle_parity_on_a_full_grid_slice(ctx):
  """One 721x1440 field pair + 5 members: every fused lane against the float64 oracle."""
  rng = np.random.default_rng(7)
  tv = (rng.normal(size=(NLAT, NLON)) + 280).astype(np.float32)
  pv = (tv[None] + rng.normal(size=(5, NLAT, NLON))).astype(np.float32)
  coords = {'latitude': LAT, 'longitude': LON}
  p = {'v': xr.DataArray(pv, dims=('number', 'latitude', 'longitude'), coords=coords)}
  t = {'v': xr.DataArray(tv, dims=('latitude', 'longitude'), coords=coords)}
  metrics = {'crps': probabilistic.CRPSEnsemble(use_sort=True), 'ssr': probabilistic.UnbiasedSpreadSkillRatio()}
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  res = aggregation.compute_metric_values_for_single_chunk(metrics, agg, p, t)
  pd, td = ('number', 'latitude', 'longitude'), ('latitude', 'longitude')
  w = (O.grid_area_weights(LAT), ('latitude',))

  def mean(a):
    sws, sw, _ = O.aggregate(a, td, list(td), weights=[w])
    return sws / sw
  want = O.crps(mean(O.crps_skill(pv, pd, tv, td, 'number')[0]), mean(O.crps_spread(pv, pd, 'number', use_sort=True)[0]))
  np.testing.assert_allclose(res['crps.v'].values, want, rtol=RTOL)
  want = O.unbiased_spread_skill_ratio(mean(O.ensemble_variance(pv, pd, 'number')[0]),
                                       mean(O.unbiased_ensemble_mean_squared_error(pv, pd, tv, td, 'number')[0]))
  np.testing.assert_allclose(res['ssr.v'].values, want, rtol=RTOL)
  # deterministic suite on member 0
  p0 = {'v': xr.DataArray(pv[0], dims=td, coords=coords)}
  r = aggregation.compute_metric_values_for_single_chunk({'rmse': deterministic.RMSE(), 'bias': deterministic.Bias()},
                                                         agg, p0, t)
  np.testing.assert_allclose(r['rmse.v'].values, np.sqrt(mean(O.squared_error(pv[0], tv))), rtol=RTOL)
  np.testing.assert_allclose(r['bias.v'].values, mean(O.error(pv[0], tv)), rtol=RTOL, atol=1e-9)


# ---- device-resident torch inputs with non-trivial strides (views are consumed in place, never copied) ------------
def _np(t):
  return t.detach().cpu().numpy()


@pytest.mark.parametrize('case', ['permuted', 'strided_slices', 'broadcast_target', 'member_fast_view'])
def test_torch_views_are_consumed_in_place(ctx, case):
  import torch
  g = torch.Generator(device='cuda')
  g.manual_seed(5)
  lat, lon = np.linspace(-80, 80, 17), np.arange(40) * 9.0
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  w = (O.grid_area_weights(lat), ('latitude',))
  if case == 'member_fast_view':
    base = torch.randn((3, 17, 40, 9), generator=g, device='cuda') + 280          # members fastest in memory
    pt = base.permute(3, 0, 1, 2)[1:8]                                             # -> (number=7, time, lat, lon) view
    tt = torch.randn((3, 17, 40), generator=g, device='cuda') + 280
    assert not pt.is_contiguous()
    p = {'v': xr.DataArray(pt, dims=('number', 'time', 'latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})}
    t = {'v': xr.DataArray(tt, dims=('time', 'latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})}
    res = aggregation.compute_metric_values_for_single_chunk({'crps': probabilistic.CRPSEnsemble(use_sort=True)}, agg, p, t)
    pd, td = ('number', 'time', 'latitude', 'longitude'), ('time', 'latitude', 'longitude')
    skill = O.crps_skill(_np(pt), pd, _np(tt), td, 'number')[0]
    spread = O.crps_spread(_np(pt), pd, 'number', use_sort=True)[0]
    a = O.aggregate(skill, td, ['latitude', 'longitude'], weights=[w])
    b = O.aggregate(spread, td, ['latitude', 'longitude'], weights=[w])
    np.testing.assert_allclose(res['crps.v'].values, O.crps(a[0] / a[1], b[0] / b[1]), rtol=RTOL)
    return
  if case == 'permuted':
    base_p = torch.randn((40, 3, 17), generator=g, device='cuda') + 280            # (lon, time, lat) in memory
    base_t = torch.randn((17, 40, 3), generator=g, device='cuda') + 280            # (lat, lon, time) in memory
    pt, tt = base_p.permute(1, 2, 0), base_t.permute(2, 0, 1)                      # both viewed as (time, lat, lon)
  elif case == 'strided_slices':
    pt = (torch.randn((6, 34, 83), generator=g, device='cuda') + 280)[::2, ::2, 3::2]
    tt = (torch.randn((3, 17, 120), generator=g, device='cuda') + 280)[:, :, ::3]
  else:  # target without a time dim, broadcast along it
    pt = torch.randn((3, 17, 40), generator=g, device='cuda') + 280
    tt = torch.randn((17, 40), generator=g, device='cuda') + 280
  assert pt.shape == (3, 17, 40)
  tdims = ('time', 'latitude', 'longitude') if tt.dim() == 3 else ('latitude', 'longitude')
  coords = {'latitude': lat, 'longitude': lon}
  p = {'v': xr.DataArray(pt, dims=('time', 'latitude', 'longitude'), coords=coords)}
  t = {'v': xr.DataArray(tt, dims=tdims, coords=coords)}
  res = aggregation.compute_metric_values_for_single_chunk({'rmse': deterministic.RMSE(), 'bias': deterministic.Bias()},
                                                           agg, p, t)
  dims = ('time', 'latitude', 'longitude')
  te = O.expand_to(_np(tt), tdims, dims)
  for name, fn, post in (('rmse', O.squared_error, np.sqrt), ('bias', O.error, lambda v: v)):
    sws, sw, _ = O.aggregate(fn(_np(pt), te), dims, ['latitude', 'longitude'], weights=[w])
    np.testing.assert_allclose(res[f'{name}.v'].values, post(sws / sw), rtol=RTOL, atol=1e-9)
  # and the per-point statistic through the map kernel
  se = deterministic.SquaredError().compute(p, t)['v']
  np.testing.assert_allclose(se.values, O.squared_error(_np(pt), te), rtol=1e-12)


@pytest.mark.parametrize('nlane,nchunk,nj,nbin', [(1, 1, 64, 5), (3, 1, 200, 14), (5, 1, 130, 34), (6, 3, 97, 34),
                                                  (7, 1, 64, 64), (10, 2, 70, 20), (12, 1, 257, 9), (2, 1, 1440, 40)])
def test_stage2_patch_kernel_equals_dense_contraction(ctx, monkeypatch, nlane, nchunk, nj, nbin):
  """wbx_contract_bits over full-map partials takes the patch kernel (wbx_s2_patch.hip); it must equal the dense
  contraction  out[a,b,l,bin] = sum_{r,c,j} partial[a,b,r,c,l,j] * wt[b,r,j] * member[b,r,j,bin]  including the
  NaN-poisons-every-bin rule and more bins in a patch than accumulator slots."""
  monkeypatch.setenv('WBX_S2_PATCH_MIN', '0')
  rng = np.random.default_rng(nlane * 1000 + nj)
  nA, nBk, nBr = 3, 2, 37
  partial = rng.normal(size=(nA, nBk, nBr, nchunk, nlane, nj))
  partial[1, 0, 5, 0, 0, 3] = np.nan  # lane 0 of cell (1, 0) is poisoned for every bin
  wt = rng.random((nBk, nBr, nj)) + 0.5
  member = rng.random((nBk, nBr, nj, nbin)) < 0.3
  member[..., 0] = True
  member[:, :, :, nbin - 1] = False  # an empty bin
  bits = np.zeros((nBk, nBr, nj), np.uint64)
  for b in range(nbin):
    bits |= member[..., b].astype(np.uint64) << np.uint64(b)
  with np.errstate(invalid='ignore'):
    want = np.einsum('abrclj,brjn->abln', partial, wt[..., None] * member)
  plan = _hip.S2PlanStruct(nA, nBk, nBr, nchunk, nlane, nj, nbin, 1)
  bufs = [ctx.upload(partial), ctx.upload(wt), ctx.upload(bits)]
  out = ctx.alloc(want.size * 8)
  _hip.check(ctx.lib.wbx_contract_bits(ctx.handle, C.byref(plan), C.c_void_p(bufs[0].ptr), C.c_void_p(bufs[1].ptr),
                                       C.c_void_p(bufs[2].ptr), C.c_void_p(out.ptr)), 'wbx_contract_bits')
  got = ctx.download(out.ptr, want.shape, np.float64)
  assert np.isnan(got[1, 0, 0]).all() and np.isnan(want[1, 0, 0]).all()
  np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
  assert (got[..., nbin - 1][~np.isnan(got[..., nbin - 1])] == 0).all()
  # the block-per-(cell, lane) kernel gives the same numbers
  monkeypatch.setenv('WBX_S2_PATCH_MIN', str(1 << 40))
  _hip.check(ctx.lib.wbx_contract_bits(ctx.handle, C.byref(plan), C.c_void_p(bufs[0].ptr), C.c_void_p(bufs[1].ptr),
                                       C.c_void_p(bufs[2].ptr), C.c_void_p(out.ptr)), 'wbx_contract_bits')
  np.testing.assert_allclose(ctx.download(out.ptr, want.shape, np.float64), got, rtol=1e-12, atol=1e-12)


def test_ensemble_crps_per_region_uses_the_patch_contraction(ctx, monkeypatch):
  """CRPS per region x land/sea (the public benchmark's probabilistic configuration) on a 1-init chunk: the ensemble
  kernel writes a full-map partial and wbx_contract_bits bins it; result == oracle."""
  from weatherbenchx_amd import binning
  monkeypatch.setenv('WBX_S2_PATCH_MIN', '0')
  rng = np.random.default_rng(3)
  nlat, nlon, m = 48, 128, 8
  lat, lon = np.linspace(-87, 87, nlat), np.linspace(0, 360, nlon, endpoint=False)
  tv = rng.normal(size=(2, nlat, nlon)).astype(np.float32)
  pv = (tv[:, None] + rng.normal(size=(2, m, nlat, nlon))).astype(np.float32)
  coords = {'lead_time': np.arange(2) * np.timedelta64(6, 'h'), 'latitude': lat, 'longitude': lon}
  t = xr.DataArray(tv, dims=('lead_time', 'latitude', 'longitude'), coords=coords)
  p = xr.DataArray(pv, dims=('lead_time', 'number', 'latitude', 'longitude'), coords=coords)
  regions = {'global': ((-90, 90), (0, 360)), 'tropics': ((-20, 20), (0, 360)), 'nh': ((20, 90), (0, 360)),
             'sh': ((-90, -20), (0, 360)), 'europe': ((35, 75), (-12.5, 42.5)), 'namerica': ((25, 60), (240, 285))}
  land = rng.random((nlat, nlon)) > 0.6
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(regions, land_sea_mask=lsm)])
  got = aggregation.compute_metric_values_for_single_chunk({'crps': probabilistic.CRPSEnsemble(use_sort=True)}, agg,
                                                           {'v': p}, {'v': t})['crps.v']
  pd, td = ('lead_time', 'number', 'latitude', 'longitude'), ('lead_time', 'latitude', 'longitude')
  skill = O.crps_skill(pv, pd, tv, td, 'number')[0]
  spread = O.crps_spread(pv, pd, 'number', use_sort=True)[0]
  w = (O.grid_area_weights(lat), ('latitude',))
  names, masks = O.region_masks(lat, lon, regions, land_sea_mask=land)
  bm = [('region', masks, ('region', 'latitude', 'longitude'))]
  a = O.aggregate(skill, td, ['latitude', 'longitude'], weights=[w], bin_masks=bm)
  b = O.aggregate(spread, td, ['latitude', 'longitude'], weights=[w], bin_masks=bm)
  assert list(got['region'].values) == names
  np.testing.assert_allclose(got.transpose(*a[2]).values, O.crps(a[0] / a[1], b[0] / b[1]), rtol=RTOL)


@pytest.mark.parametrize('nx', [64, 130, 1440])
@pytest.mark.parametrize('skipna', [False, True])
def test_masked_reduction_with_depth_independent_mask(ctx, nx, skipna):
  """(lat, lon) validity mask under a reduction over (init_time, longitude): the key's mask row is staged in LDS
  (s1_xr_kernel<.., MROW>), with dword mask loads when nx % 4 == 0; masked-out points contribute exactly 0 even when
  they hold NaN / inf, and the count lane is the number of valid points (aggregation.py:339-357)."""
  rng = np.random.default_rng(nx)
  shape = (5, 2, 6, nx)
  dims = ('init_time', 'lead_time', 'latitude', 'longitude')
  pv, tv = rng.normal(size=shape).astype(np.float32), rng.normal(size=shape).astype(np.float32)
  valid = rng.random((6, nx)) > 0.3
  tv[:, :, ~valid] = np.nan
  pv[0, 0, ~valid] = np.inf
  if skipna:
    tv[1, 1, 2, 5] = np.nan  # an additional NaN under a valid mask point: skipna counts it out, mask alone would poison
  p, t = xr.DataArray(pv, dims=dims), xr.DataArray(tv, dims=dims)
  mask = xr.DataArray(valid, dims=('latitude', 'longitude'))
  vals, cnt, od = engine.reduce_statistics('det', [p, t], dims, dict(zip(dims, shape)), ['init_time', 'longitude'], None,
                                           (), func=_hip.DET3, mask=mask, skipna=skipna)
  assert od == ('lead_time', 'latitude')
  e = pv.astype(np.float64) - tv
  ok = np.broadcast_to(valid, shape) & (~np.isnan(e) if skipna else True)
  want = np.where(ok, e * e, 0.0).sum(axis=(0, 3))
  np.testing.assert_allclose(vals[2], want, rtol=1e-12)
  np.testing.assert_allclose(vals[1], np.where(ok, np.abs(e), 0.0).sum(axis=(0, 3)), rtol=1e-12)
  np.testing.assert_allclose(cnt[2], ok.sum(axis=(0, 3)))


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
@pytest.mark.parametrize('mode', ['plain', 'masked', 'skipna'])
def test_region_binned_paths_on_a_one_degree_grid(ctx, monkeypatch, layout, mode):
  """17 regions x land/sea (34 bins) on a 181 x 360 grid, several x tiles / row splits / sweeps per patch: the fused
  kernel (wbx_det_binned), the two-stage path with the stage-2 patch kernel and the oracle must agree for RMSE, ACC
  and bias, including NaN targets under masked / skipna and a NaN that poisons every bin without them."""
  import sys
  import os
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
  from wb_regions import REGIONS
  from weatherbenchx_amd import binning
  rng = np.random.default_rng(23)
  nlat, nlon = 181, 360
  lat, lon = np.linspace(-90, 90, nlat), np.arange(nlon) * 1.0
  sp = ('latitude', 'longitude') if layout == 'lon_fastest' else ('longitude', 'latitude')
  dims = ('init_time', 'lead_time', 'level') + sp
  sizes = {'init_time': 2, 'lead_time': 3, 'level': 2, 'latitude': nlat, 'longitude': nlon}
  coords = {'init_time': np.datetime64('2020-01-01T00', 'ns') + np.arange(2) * np.timedelta64(12, 'h'),
            'lead_time': np.arange(3) * np.timedelta64(6, 'h'), 'level': [500, 850], 'latitude': lat, 'longitude': lon}
  shape = tuple(sizes[d] for d in dims)
  pv = (rng.normal(size=shape) + 280).astype(np.float32)
  tv = (rng.normal(size=shape) + 280).astype(np.float32)
  if mode == 'plain':
    tv[(1, 2, 1) + ((90, 10) if layout == 'lon_fastest' else (10, 90))] = np.nan  # poisons every bin of that cell
  else:
    tv[rng.random(shape) < 0.05] = np.nan
  p = xr.DataArray(pv, dims=dims, coords=coords)
  t = xr.DataArray(tv, dims=dims, coords=coords)
  if mode == 'masked':
    t.coords['mask'] = ~np.isnan(t)
  cdims = ('dayofyear', 'hour', 'level') + sp
  cv = (rng.normal(size=(3, 4, 2) + tuple(sizes[d] for d in sp)) + 280).astype(np.float32)
  clim = xr.DataArray(cv, dims=cdims, coords={'dayofyear': [1, 2, 3], 'hour': [0, 6, 12, 18], 'level': [500, 850],
                                              'latitude': lat, 'longitude': lon})
  land = rng.random((nlat, nlon)) > 0.6
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  metrics = {'rmse': deterministic.RMSE(), 'acc': deterministic.ACC({'z': clim}), 'bias': deterministic.Bias()}
  monkeypatch.setenv('WBX_S2_PATCH_MIN', '0')
  results = {}
  for binned in ('always', 'never'):
    monkeypatch.setattr(engine, 'BINNED_MODE', binned)
    engine.clear_caches()
    agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'],
                                 weigh_by=[weighting.GridAreaWeighting()],
                                 bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)],
                                 masked=(mode == 'masked'), skipna=(mode == 'skipna'))
    results[binned] = aggregation.compute_metric_values_for_single_chunk(metrics, agg, {'z': p}, {'z': t})
  for k in results['never']:
    np.testing.assert_allclose(results['always'][k].values, results['never'][k].values, rtol=1e-9, atol=1e-12,
                               equal_nan=True)
  # oracle for the squared error
  w = (O.grid_area_weights(lat), ('latitude',))
  names, masks = O.region_masks(lat, lon, REGIONS, land_sea_mask=land)
  kw = {}
  if mode == 'masked':
    kw = dict(mask=~np.isnan(tv), mask_dims=dims)
  if mode == 'skipna':
    kw = dict(skipna=True)
  sws, sw, od = O.aggregate(O.squared_error(pv, tv), dims, ['init_time', 'latitude', 'longitude'], weights=[w],
                            bin_masks=[('region', masks, ('region', 'latitude', 'longitude'))], **kw)
  with np.errstate(all='ignore'):
    want = np.sqrt(sws / sw)
  got = results['always']['rmse.z']
  assert list(got['region'].values) == names and len(names) == 34
  np.testing.assert_allclose(got.transpose(*od).values, want, rtol=RTOL, equal_nan=True)
  if mode == 'plain':
    assert np.isnan(got.sel(lead_time=coords['lead_time'][2], level=850).values).all()
    assert np.isfinite(got.sel(lead_time=coords['lead_time'][0], level=850).values).all()


def test_rccl_all_reduce_of_the_device_accumulators_single_rank(ctx):
  """The RCCL leg of distributed.reduce_accumulation (backend 'nccl' = RCCL on ROCm) on the one GPU a test box has: the
  chunks' sums are added into device accumulator slots (wbx_acc_add), the slots are laid out in one device buffer, that
  buffer is all-reduced IN PLACE through its device pointer (no host hop: torch sees wbx memory through the CUDA array
  interface) and read back once -- the identity for a one-rank group, and exactly ONE collective per reduction.
  (World sizes 2-3 run on gloo in tests/test_distributed.py; N > 1 on RCCL is the driver's scaling run.)"""
  import socket
  import torch
  import torch.distributed as dist
  from weatherbenchx_amd import distributed, pipeline, time_chunks
  if dist.is_initialized():
    pytest.skip('a process group already exists in this process')
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  torch.cuda.set_device(0)
  dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1,
                          device_id=torch.device('cuda', 0))
  try:
    rng = np.random.default_rng(0)
    lat = np.linspace(-80, 80, 16)
    p = xr.DataArray(rng.normal(size=(3, 16, 32)).astype(np.float32), dims=('lead_time', 'latitude', 'longitude'),
                     coords={'latitude': lat})
    t = xr.DataArray(rng.normal(size=(3, 16, 32)).astype(np.float32), dims=('lead_time', 'latitude', 'longitude'),
                     coords={'latitude': lat})
    metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE()}
    agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
    want = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, {'v': p}, {'v': t}))
    # the device pointer really is what torch reduces: same address, no copy
    buf = ctx.upload(np.arange(8.0))
    view = distributed.device_tensor(buf.ptr, 8, ctx.device_id)
    assert view.data_ptr() == buf.ptr and view.dtype == torch.float64
    dist.all_reduce(view)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(ctx.download(buf.ptr, (8,)), np.arange(8.0))
    del view
    plan = None
    for step in range(3):  # a per-step all-reduce loop: layout exchanged once, one collective per step
      acc = engine.Accumulation()
      with engine.accumulate_results(acc):
        fresh = lambda a: {'v': xr.DataArray(a.data, dims=a.dims, coords={'latitude': lat})}
        state = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, fresh(p), fresh(t)))
      reduced, plan = distributed.resolve_state(state, acc, plan=plan, force=True)
      assert plan.collectives == step + 1
      for k, v in want.metric_values(metrics).items():
        np.testing.assert_array_equal(reduced.metric_values(metrics)[k].values, v.values)
    # the chunk loop: accumulate 2 + 2 chunks on the device, one collective at the end
    predictions, targets = None, None
    import mock_data
    predictions = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-03T00', lead_start_days=0,
                                                 lead_stop_days=1, random=True, seed=0)
    targets = mock_data.mock_target_data(time_start='2020-01-01T00', time_stop='2020-01-05T00', random=True, seed=1)
    import test_pipeline as tp
    init_times = predictions['geopotential']['time'].values
    lead_times = predictions['geopotential']['prediction_timedelta'].values
    load = tp._loader(predictions, targets)
    pieces = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=1, lead_time_chunk_size=1)
    whole = time_chunks.TimeChunks(init_times, lead_times)
    agg2 = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
    a = pipeline.evaluate_chunks(pieces, load, metrics, agg2, force_collective=True)[None].metric_values(metrics)
    b = pipeline.evaluate_chunks(whole, load, metrics, agg2)[None].metric_values(metrics)
    for k in b:
      xr.assert_allclose(a[k], b[k], rtol=1e-9, atol=1e-12, check_dim_order=False)
  finally:
    dist.destroy_process_group()


def test_full_grid_region_bins_size_independent_properties(ctx, monkeypatch):
  """721 x 1440, 17 regions x land/sea (the public benchmark's binning): (i) the 'global' bin equals the unbinned
  aggregation, accumulator by accumulator; (ii) every '<region>_land' accumulator is <= its '<region>' one;
  (iii) the fused kernel and the two-stage path agree; (iv) doubling the predictions' error quadruples every bin's
  squared-error sum (linearity of the accumulators in the statistic)."""
  import os
  import sys
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
  from wb_regions import REGIONS
  from weatherbenchx_amd import binning
  rng = np.random.default_rng(5)
  nlat, nlon = 721, 1440
  lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
  dims = ('lead_time', 'longitude', 'latitude')  # latitude-fastest, like the real archives
  coords = {'lead_time': np.arange(2) * np.timedelta64(6, 'h'), 'latitude': lat, 'longitude': lon}
  tv = rng.normal(size=(2, nlon, nlat)).astype(np.float32)
  ev = rng.normal(size=(2, nlon, nlat)).astype(np.float32)
  t = xr.DataArray(tv, dims=dims, coords=coords)
  p1 = xr.DataArray(tv + ev, dims=dims, coords=coords)
  p2 = xr.DataArray(tv + 2 * ev, dims=dims, coords=coords)
  land = rng.random((nlat, nlon)) > 0.7
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  kw = dict(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])

  def state(p, **extra):
    agg = aggregation.Aggregator(**kw, **extra)
    stats = metrics_base.compute_unique_statistics_for_all_metrics({'mse': deterministic.MSE()}, {'v': p}, {'v': t})
    s = agg.aggregate_statistics(stats)
    return s.sum_weighted_statistics['SquaredError']['v'], s.sum_weights['SquaredError']['v']
  plain_s, plain_w = state(p1)
  bins = dict(bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)])
  monkeypatch.setattr(engine, 'BINNED_MODE', 'always')
  fused_s, fused_w = state(p1, **bins)
  fused2_s, _ = state(p2, **bins)
  monkeypatch.setattr(engine, 'BINNED_MODE', 'never')
  engine.clear_caches()
  two_s, two_w = state(p1, **bins)
  names = list(fused_s['region'].values)
  assert len(names) == 34
  np.testing.assert_allclose(fused_s.sel(region='global').values, plain_s.values, rtol=1e-12)
  np.testing.assert_allclose(fused_w.sel(region='global').values, plain_w.values, rtol=1e-12)
  for r in REGIONS:
    assert (fused_s.sel(region=f'{r}_land').values <= fused_s.sel(region=r).values * (1 + 1e-12)).all()
    assert (fused_w.sel(region=f'{r}_land').values <= fused_w.sel(region=r).values * (1 + 1e-12)).all()
  np.testing.assert_allclose(fused_s.transpose(*two_s.dims).values, two_s.values, rtol=1e-11)
  np.testing.assert_allclose(fused_w.transpose(*two_w.dims).values, two_w.values, rtol=1e-12)
  np.testing.assert_allclose(fused2_s.values, 4.0 * fused_s.values, rtol=1e-5)  # fp32 rounding of t + 2e vs t + e


@pytest.mark.parametrize('mode', ['plain', 'masked', 'skipna'])
def test_full_grid_latitude_fastest_ensemble_closed_form(ctx, mode):
  """IFS-ENS style chunk (lead, number, longitude, latitude): the area weights are folded into the flat
  one-point-per-lane sweep (s1_xf1_kernel), also under a validity mask / skipna.  Members t + k*d give the same closed
  forms at every point, whatever the weights and whichever points are masked out or NaN."""
  import torch
  m, d, nl = 51, 0.25, 2
  tt = _torch_field((nl, NLON, NLAT), 6, offset=0.0).round()
  k = torch.arange(m, device='cuda', dtype=torch.float32)[torch.randperm(m, device='cuda')]
  pt = tt[:, None] + (k * d)[None, :, None, None]
  dims, edims = ('lead_time', 'longitude', 'latitude'), ('lead_time', 'number', 'longitude', 'latitude')
  coords = {'latitude': LAT, 'longitude': LON, 'lead_time': (np.arange(nl) * 12).astype('timedelta64[h]').astype('timedelta64[ns]')}
  hole = np.zeros((nl, NLON, NLAT), dtype=bool)
  hole[:, 100:400, 50:300] = True
  if mode == 'skipna':
    tt[torch.from_numpy(hole).cuda()] = float('nan')
  t = xr.DataArray(tt, dims=dims, coords=coords)
  if mode == 'masked':
    t.coords['mask'] = xr.DataArray(~hole, dims=dims, coords=coords)
  p = {'v': xr.DataArray(pt, dims=edims, coords=coords)}
  metrics = {'skill': probabilistic.CRPSSkill(), 'spread': probabilistic.CRPSSpread(use_sort=True),
             'var': probabilistic.EnsembleVariance()}
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               masked=(mode == 'masked'), skipna=(mode == 'skipna'))
  engine.S1_EVENT_LOG = []
  try:
    res = aggregation.compute_metric_values_for_single_chunk(metrics, agg, p, {'v': t})
    assert engine.S1_EVENT_LOG and all(e.get('flat') for e in engine.S1_EVENT_LOG if e.get('kind') == 'ens')
  finally:
    engine.S1_EVENT_LOG = None
  np.testing.assert_allclose(res['skill.v'].values, d * (m - 1) / 2, rtol=1e-12)
  np.testing.assert_allclose(res['spread.v'].values, d * (m + 1) / 3, rtol=1e-12)
  np.testing.assert_allclose(res['var.v'].values, d * d * m * (m + 1) / 12, rtol=1e-10)


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
def test_full_grid_rank_histogram_and_exceedance_closed_form(ctx, layout):
  """51 members t + (k - 20.5) d: exactly 21 members lie below the target at every point, so the rank histogram is the
  unit vector e_21; |p_k - t| = |k - 20.5| d exceeds a threshold for a known number of members (the kernel instantiated
  for M = 51)."""
  import torch
  m, d = 51, 0.5
  sp = ('latitude', 'longitude') if layout == 'lon_fastest' else ('longitude', 'latitude')
  shp = (NLAT, NLON) if layout == 'lon_fastest' else (NLON, NLAT)
  tt = _torch_field(shp, 8, offset=0.0).round()
  k = torch.arange(m, device='cuda', dtype=torch.float32)[torch.randperm(m, device='cuda')]
  pt = tt[None] + ((k - 20.5) * d)[:, None, None]
  coords = {'latitude': LAT, 'longitude': LON}
  p = {'v': xr.DataArray(pt, dims=('number',) + sp, coords=coords)}
  t = {'v': xr.DataArray(tt, dims=sp, coords=coords)}
  thr = [0.2, 2.6, 7.0, 100.0]
  metrics = {'rank': probabilistic.RankHistogram(), 'exc': probabilistic.EnsembleErrorExceedance(thresholds=thr)}
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  res = aggregation.compute_metric_values_for_single_chunk(metrics, agg, p, t)
  want = np.zeros(m + 1)
  want[21] = 1.0
  np.testing.assert_allclose(np.asarray(res['rank.v'].values).reshape(-1), want, atol=1e-12)
  frac = [np.mean(np.abs(np.arange(m) - 20.5) * d > x) for x in thr]
  np.testing.assert_allclose(np.asarray(res['exc.v'].values).reshape(-1), frac, rtol=1e-12)


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
def test_full_grid_spectrum_parseval(ctx, layout):
  """sum_k S_k (Nyquist counted once) == mean over longitude of f^2 for every row (odd row count 3 x 721: the last team ends on a single
  row), through the fused kernel (lon-fastest) and the transposing slab path (lat-fastest)."""
  from weatherbenchx_amd import spectra
  sp = ('latitude', 'longitude') if layout == 'lon_fastest' else ('longitude', 'latitude')
  shp = (3,) + ((NLAT, NLON) if layout == 'lon_fastest' else (NLON, NLAT))
  f = _torch_field(shp, 9, offset=1.0)
  da = xr.DataArray(f, dims=('lead_time',) + sp, coords={'latitude': LAT, 'longitude': LON,
                                                           'lead_time': (np.arange(3) * 6).astype('timedelta64[h]').astype('timedelta64[ns]')})
  stat = spectra.ZonalPowerSpectrum().compute({'v': da}, {'v': da})['v']
  st = aggregation.Aggregator(reduce_dims=['lead_time']).aggregate_stat_var(stat)
  got = st.sum_weighted_statistics.transpose('latitude', 'zonal_wavenumber').values  # sum over lead of S_k per latitude
  lon_ax = 1 + sp.index('longitude')
  ms = (f.double() ** 2).mean(dim=lon_ax).sum(dim=0).cpu().numpy()  # [latitude]
  # the Nyquist bin is doubled like every k > 0 (WeatherBench-2 convention): half of it is not part of Parseval's sum
  np.testing.assert_allclose(got.sum(axis=1) - 0.5 * got[:, -1], ms, rtol=2e-6)


def test_full_grid_dense_stage2_split_equals_folded_route(ctx, monkeypatch):
  """Latitude kept through stage 1 with few outputs (8 leads x 5 ensemble lanes over a 100+ MB partial): the dense
  contraction deals the (Br, chunk) rows of each output to many blocks and adds the pieces (s2_reduce_kernel with
  nsplit > 1).  The folded route (weights inside stage 1, no stage 2 to speak of) must give the same numbers."""
  m, nl = 4, 8
  tt = _torch_field((nl, NLON, NLAT), 12)
  pt = tt[:, None] + _torch_field((nl, m, NLON, NLAT), 13, offset=0.0)
  coords = {'latitude': LAT, 'longitude': LON, 'lead_time': (np.arange(nl) * 12).astype('timedelta64[h]').astype('timedelta64[ns]')}
  p = {'v': xr.DataArray(pt, dims=('lead_time', 'number', 'longitude', 'latitude'), coords=coords)}
  t = {'v': xr.DataArray(tt, dims=('lead_time', 'longitude', 'latitude'), coords=coords)}
  metrics = {'crps': probabilistic.CRPSEnsemble(use_sort=True), 'ssr': probabilistic.UnbiasedSpreadSkillRatio()}
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  res = {}
  for fold in (True, False):
    monkeypatch.setattr(engine, 'FOLD_X_WEIGHTS', fold)
    engine.clear_caches()
    res[fold] = aggregation.compute_metric_values_for_single_chunk(metrics, agg, p, t)
  for k in res[True]:
    np.testing.assert_allclose(res[False][k].values, res[True][k].values, rtol=1e-11)


@pytest.mark.parametrize('entry', ['rows', 'slabs'])
@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
def test_zonal_spectrum_entry_points_raw_1440(ctx, layout, entry):
  """wbx_zonal_spectrum / wbx_zonal_spectrum_slabs called with raw pointers (include/wbx.h) on 1440-point rows: 3 slabs of
  37 rows (odd: a lone last row per slab / run) in either layout, two groups per slab with unequal row scales, against
  numpy.fft in float64; `accumulate = 1` adds a second pass onto the first."""
  import torch
  nslab, rps, nlon = 3, 37, 1440
  rng = np.random.default_rng(5)
  if entry == 'rows':
    nslab = 1
  vals = (rng.normal(size=(nslab, rps, nlon)) + 2.0).astype(np.float32)  # [slab][row][lon]
  if layout == 'lon_fastest':
    dev = torch.from_numpy(vals).cuda()
    lon_stride, row_stride = 1, nlon
    offs = np.arange(nslab, dtype=np.int64) * rps * nlon
  else:
    dev = torch.from_numpy(np.ascontiguousarray(vals.transpose(0, 2, 1))).cuda()  # [slab][lon][row]
    lon_stride, row_stride = rps, 1
    offs = np.arange(nslab, dtype=np.int64) * rps * nlon
  ngroup = 2 * nslab
  group = (np.arange(nslab)[:, None] * 2 + (np.arange(rps)[None, :] >= 20)).astype(np.int32).reshape(-1)
  scale = rng.uniform(0.5, 2.0, size=nslab * rps)
  g_dev, s_dev = torch.from_numpy(group).cuda(), torch.from_numpy(scale).cuda()
  out = torch.full((ngroup, nlon // 2 + 1), 7.0, device='cuda', dtype=torch.float64)  # accumulate = 0 must overwrite it
  torch.cuda.synchronize()
  lib = ctx.lib

  def call(accumulate):
    if entry == 'rows':
      _hip.check(lib.wbx_zonal_spectrum(ctx.handle, dev.data_ptr(), lon_stride, row_stride, rps, nlon, g_dev.data_ptr(),
                                        s_dev.data_ptr(), ngroup, accumulate, out.data_ptr()), 'wbx_zonal_spectrum')
    else:
      _hip.check(lib.wbx_zonal_spectrum_slabs(ctx.handle, dev.data_ptr(), lon_stride, row_stride, rps, nslab,
                                              offs.ctypes.data_as(C.c_void_p), nlon, g_dev.data_ptr(), s_dev.data_ptr(), ngroup,
                                              accumulate, out.data_ptr()), 'wbx_zonal_spectrum_slabs')
  call(0)
  ctx.synchronize()
  per_row = O.zonal_power_spectrum(vals.reshape(-1, nlon)) * scale[:, None]
  want = np.zeros((ngroup, nlon // 2 + 1))
  np.add.at(want, group, per_row)
  got = out.cpu().numpy()
  from test_spectra import bound_1440
  assert float(np.max(np.abs(got - want) / bound_1440(want))) <= 1.0
  call(1)
  ctx.synchronize()
  np.testing.assert_allclose(out.cpu().numpy(), 2 * got, rtol=1e-12)
