"""Round-3 GPU parity tests: the pipelined ensemble kernel (LDS-DMA prefetch, fp32 chain sums) at the geometry bench.py
launches, its escape for magnitudes the fp32 sums cannot hold, the one-launch contract of the reference-default
CRPSEnsemble(), the configs[4] composite (three evaluations, device accumulators, new time labels per chunk) against the
oracle, and the C-ABI collective (wbx_comm_* / wbx_acc_allreduce) on a one-rank communicator.
Tolerance: rtol 1e-6 (north_star) unless a test says otherwise."""
import ctypes as C

import numpy as np
import pytest

from oracle import wbx_oracle as O
from weatherbenchx_amd import _hip
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import distributed
from weatherbenchx_amd import engine
from weatherbenchx_amd import lazy
from weatherbenchx_amd import pipeline
from weatherbenchx_amd import spectra
from test_spectra import bound_1440
from weatherbenchx_amd import time_chunks
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.metrics import deterministic
from weatherbenchx_amd.metrics import probabilistic
from weatherbenchx_amd.metrics import wrappers

pytestmark = pytest.mark.gpu
RTOL = 1e-6
NLAT, NLON = 721, 1440
LAT = np.linspace(-90, 90, NLAT)
LON = np.linspace(0, 360, NLON, endpoint=False)
PD, TD = ('number', 'latitude', 'longitude'), ('latitude', 'longitude')


@pytest.fixture(scope='module')
def ctx():
  assert _hip.is_available(), 'gpu tests need libwbx_hip.so and a HIP device'
  return _hip.default_context(0)


def _randn(shape, seed, offset=0.0, scale=1.0):
  import torch
  g = torch.Generator(device='cuda')
  g.manual_seed(seed)
  return torch.randn(shape, generator=g, device='cuda', dtype=torch.float32) * scale + offset


def _suite(use_sort=True):
  return {'crps': probabilistic.CRPSEnsemble(use_sort=use_sort),
          'unbiased_spread_skill': probabilistic.UnbiasedSpreadSkillRatio(),
          'unbiased_mean_rmse': probabilistic.UnbiasedEnsembleMeanRMSE(),
          'mean_rmse': wrappers.WrappedMetric(deterministic.RMSE(), [wrappers.EnsembleMean(which='predictions')])}


def _five_lanes(pv, tv):
  """The five per-point ensemble statistics of the float64 oracle, [lat, lon] each."""
  return {'CRPSSkill': O.crps_skill(pv, PD, tv, TD, 'number')[0],
          'CRPSSpread': O.crps_spread(pv, PD, 'number', fair=True, use_sort=True)[0],
          'EnsembleVariance': O.ensemble_variance(pv, PD, 'number')[0],
          'UnbiasedEnsembleMeanSquaredError': O.unbiased_ensemble_mean_squared_error(pv, PD, tv, TD, 'number')[0],
          'EnsembleMeanSquaredError': O.ensemble_mean_squared_error(pv, PD, tv, TD, 'number')[0]}


def _lane_statistics():
  return {'CRPSSkill': probabilistic.CRPSSkill(), 'CRPSSpread': probabilistic.CRPSSpread(use_sort=True),
          'EnsembleVariance': probabilistic.EnsembleVariance(),
          'UnbiasedEnsembleMeanSquaredError': probabilistic.UnbiasedEnsembleMeanSquaredError(),
          'EnsembleMeanSquaredError': wrappers.WrappedStatistic(deterministic.SquaredError(),
                                                                wrappers.EnsembleMean(which='predictions'))}


@pytest.mark.parametrize('form', ['rank', 'pair'])
def test_m51_full_grid_every_latitude_row_against_the_oracle(ctx, form, monkeypatch):
  """The exact geometry of the driver line's kernel -- ens_pipe_kernel<51> (rank form) resp. s1_xr_kernel<EnsOpF32<51, true,
  PAIRWISE>> (pair form), one wave per (latitude) row of 1440 points -- on random N(280, 1) members: every one of the 721
  per-row means of all five statistics against the float64 oracle (reduce `longitude` only, so no row can hide behind
  another).  The rank form runs the fp32 chain sums: its rows must still hold 1e-6."""
  monkeypatch.setattr(lazy, 'PAIR_FORM_KERNEL', form == 'pair')
  rng = np.random.default_rng(51)
  tv = (rng.normal(size=(NLAT, NLON)) + 280).astype(np.float32)
  pv = (tv[None] + rng.normal(size=(51, NLAT, NLON))).astype(np.float32)
  tv = (tv + rng.normal(size=(NLAT, NLON))).astype(np.float32)
  coords = {'latitude': LAT, 'longitude': LON}
  p = {'v': xr.DataArray(pv, dims=PD, coords=coords)}
  t = {'v': xr.DataArray(tv, dims=TD, coords=coords)}
  stats = _lane_statistics()
  if form == 'pair':
    stats['CRPSSpread'] = probabilistic.CRPSSpread(use_sort=False)
  agg = aggregation.Aggregator(reduce_dims=['longitude'])
  engine.S1_EVENT_LOG = []
  try:
    state = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(stats, p, t))
    means = state.mean_statistics()
    log = [e for e in engine.S1_EVENT_LOG if e['kind'] == 'ens']
  finally:
    engine.S1_EVENT_LOG = None
  assert len(log) == 1 and log[0]['algo'] == (1 if form == 'pair' else 0) and log[0]['block'] == 64 and log[0]['grid'] == NLAT
  want = _five_lanes(pv, tv)
  for name, lane in want.items():
    got = np.asarray(means[stats[name].unique_name]['v'].values)
    assert got.shape == (NLAT,)
    np.testing.assert_allclose(got, lane.mean(axis=-1), rtol=RTOL, err_msg=f'{form} {name}')
  if form == 'rank':  # how close the fp32 chain sums really are (bound: 9 x 2^-24 = 5.4e-7 per point, rows average 1440 points)
    got = np.asarray(means[stats['CRPSSpread'].unique_name]['v'].values)
    assert np.abs(got / want['CRPSSpread'].mean(axis=-1) - 1).max() < 5e-8


def test_bias_ten_thousand_times_the_spread(ctx):
  """|mean - target| = 1e4 x spread: a one-pass variance on x - t cancels eight digits.  The fp32 chain sums centre on the sorted
  median (variance, spread: bias-free) and form mean - t from an fp64 (median - t); the fp64 sums of the other kernels shift
  by the target in fp64.  Both against the oracle, every lane, for a small bias too."""
  rng = np.random.default_rng(9)
  nlat, nlon = 12, 1440
  lat, lon = np.linspace(-82.5, 82.5, nlat), LON
  coords = {'latitude': lat, 'longitude': lon}
  for bias in (1e4, -3e4, 0.5):
    tv = (rng.normal(size=(nlat, nlon)) * 3 + 250).astype(np.float32)
    pv = (tv[None] + np.float32(bias) + rng.normal(size=(51, nlat, nlon))).astype(np.float32)
    spread = pv.astype(np.float64).std(axis=0).mean()
    assert abs(bias) < 1 or abs((pv.astype(np.float64).mean(axis=0) - tv).mean()) > 5e3 * spread
    want = _five_lanes(pv, tv)
    stats = _lane_statistics()
    for kernel in ('ens_pipe_kernel (fp32 chain sums)', 's1_xr_kernel with a mask (fp64 sums)'):
      p = xr.DataArray(pv, dims=PD, coords=coords)
      t = xr.DataArray(tv, dims=TD, coords=coords)
      agg = aggregation.Aggregator(reduce_dims=['longitude'])
      if kernel.startswith('s1_xr'):  # an all-true mask coordinate: same numbers through the masked wrapper of the old skeleton
        t = t.assign_coords(mask=xr.DataArray(np.ones((nlat, nlon), bool), dims=TD, coords=coords))
        agg = aggregation.Aggregator(reduce_dims=['longitude'], masked=True)
      means = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(stats, {'v': p}, {'v': t})).mean_statistics()
      for name, lane in want.items():
        got = np.asarray(means[stats[name].unique_name]['v'].values)
        np.testing.assert_allclose(got, lane.mean(axis=-1), rtol=RTOL, err_msg=f'bias {bias} {kernel} {name}')


def test_magnitudes_the_fp32_sums_cannot_hold_take_the_fp64_escape(ctx):
  """Rows whose members span 1e-30 .. 1e25, are all denormal, or sit next to a 1e30 target: an fp32 square would overflow /
  underflow, so the wave redoes those points with the generic fp64 op (wbx_ens_impl.hpp: compute() -> finish()).  Every row
  against the oracle at the fp64 tolerance of round 2 -- and an ordinary row in the same launch stays on the fp32 sums."""
  rng = np.random.default_rng(77)
  nlat, nlon, m = 6, 1440, 51
  pv = np.empty((m, nlat, nlon), np.float32)
  tv = np.empty((nlat, nlon), np.float32)
  pv[:, 0] = rng.normal(size=(m, nlon)) * 10.0 ** rng.integers(-30, 36, size=(m, nlon))   # mixed magnitudes up to 1e36
  pv[0, 0] = 3e22                                                                            # ... range above 2^60 in EVERY point
  tv[0] = pv[3, 0]
  pv[:, 1] = (rng.integers(-4000, 4000, size=(m, nlon)) * np.float64(1.4e-45)).astype(np.float32)  # denormals: range < 2^-50
  tv[1] = 0.0
  pv[:, 2] = rng.normal(size=(m, nlon)) + 280
  tv[2] = 1e32                                                                               # |target| > 2^100
  pv[:, 3] = rng.normal(size=(m, nlon)) * 1e-18                                              # squares underflow fp32 normals
  tv[3] = pv[7, 3]
  pv[:, 4] = 5.0                                                                             # zero range: stays fast, exact
  tv[4] = 4.0
  pv[:, 5] = rng.normal(size=(m, nlon)) + 280                                                # ordinary
  tv[5] = 280.5
  coords = {'latitude': np.linspace(-75, 75, nlat), 'longitude': LON}
  stats = _lane_statistics()
  agg = aggregation.Aggregator(reduce_dims=['longitude'])
  p = {'v': xr.DataArray(pv, dims=PD, coords=coords)}
  t = {'v': xr.DataArray(tv, dims=TD, coords=coords)}
  means = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(stats, p, t)).mean_statistics()
  want = _five_lanes(pv, tv)
  with np.errstate(all='ignore'):
    for name, lane in want.items():
      got = np.asarray(means[stats[name].unique_name]['v'].values)
      ref = lane.mean(axis=-1)
      np.testing.assert_allclose(got[:5], ref[:5], rtol=1e-9, atol=1e-300, err_msg=name)  # rows 0-3: fp64 escape; row 4: exact
      np.testing.assert_allclose(got[5], ref[5], rtol=RTOL, err_msg=name)
  assert np.isfinite(np.asarray(means[stats['EnsembleVariance'].unique_name]['v'].values)).all()


def test_scaling_by_a_power_of_two_is_exact_at_the_driver_line_size(ctx):
  """A size-independent property of the fp32 chain sums at the full 37-level field (7.99 GB): multiplying members and targets by
  4 multiplies every fp32 operation's result by 4 exactly, so skill / spread scale by 4 and the squared lanes by 16 BIT FOR BIT --
  any lost update, stale staging buffer or mis-ordered LDS-DMA tile would break the equality somewhere in 38 million points."""
  import torch
  nlev = 37
  tv = _randn((nlev, NLAT, NLON), 1, 280.0)
  ens = _randn((nlev, 51, NLAT, NLON), 2)
  ens += tv[:, None]
  tv += _randn((nlev, NLAT, NLON), 3)
  coords = {'latitude': LAT, 'longitude': LON}
  stats = _lane_statistics()
  agg = aggregation.Aggregator(reduce_dims=['longitude'])

  def run():
    p = {'v': xr.DataArray(ens, dims=('level',) + PD, coords=coords)}
    t = {'v': xr.DataArray(tv, dims=('level',) + TD, coords=coords)}
    st = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(stats, p, t))
    return {k: np.asarray(st.sum_weighted_statistics[stats[k].unique_name]['v'].values).copy() for k in stats}
  a = run()
  ens *= 4.0
  tv *= 4.0
  torch.cuda.synchronize()
  b = run()
  for name, factor in (('CRPSSkill', 4.0), ('CRPSSpread', 4.0), ('EnsembleVariance', 16.0),
                       ('UnbiasedEnsembleMeanSquaredError', 16.0), ('EnsembleMeanSquaredError', 16.0)):
    assert a[name].shape == (nlev, NLAT) and np.isfinite(a[name]).all()
    np.testing.assert_array_equal(b[name], a[name] * factor, err_msg=name)
  # and the level means are what an exchangeable N(0, 1) ensemble must give
  crps = (a['CRPSSkill'] - 0.5 * a['CRPSSpread']).sum(axis=1) / (NLAT * NLON)
  np.testing.assert_allclose(crps, 0.5642, atol=2e-3)


def test_reference_default_crps_ensemble_is_one_launch_per_variable(ctx):
  """CRPSEnsemble() with the reference defaults (use_sort=False, probabilistic.py:644) + spread/skill + both RMSEs on two
  variables: ONE ensemble kernel per variable (round 2 launched the rank kernel for the skill lane and then the pair kernel for
  the spread lane: the ensemble was read twice).  An unfair spread next to the other lanes: still one launch (look-ahead)."""
  rng = np.random.default_rng(4)
  nlat, nlon = 32, 128
  coords = {'latitude': np.linspace(-87, 87, nlat), 'longitude': np.arange(nlon) * (360 / nlon)}
  p, t = {}, {}
  for v in ('a', 'b'):
    tv = (rng.normal(size=(nlat, nlon)) + 280).astype(np.float32)
    t[v] = xr.DataArray(tv, dims=TD, coords=coords)
    p[v] = xr.DataArray((tv[None] + rng.normal(size=(51, nlat, nlon))).astype(np.float32), dims=PD, coords=coords)
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  for metrics in (_suite(use_sort=False),
                  {'mean_rmse': _suite()['mean_rmse'], 'crps_unfair': probabilistic.CRPSEnsemble(fair=False)}):
    engine.S1_EVENT_LOG = []
    try:
      res = aggregation.compute_metric_values_for_single_chunk(metrics, agg, p, t)
      log = [e for e in engine.S1_EVENT_LOG if e['kind'] == 'ens']
    finally:
      engine.S1_EVENT_LOG = None
    assert len(log) == 2 and all(e['algo'] == _hip.ENS_SORT for e in log), log
    w = (O.grid_area_weights(coords['latitude']), ('latitude',))
    for v in ('a', 'b'):
      pv, tv = p[v].values, t[v].values
      mean = lambda a: (lambda s: s[0] / s[1])(O.aggregate(a, TD, list(TD), weights=[w]))
      fair = 'crps' in metrics
      key = 'crps' if fair else 'crps_unfair'
      want = O.crps(mean(O.crps_skill(pv, PD, tv, TD, 'number')[0]), mean(O.crps_spread(pv, PD, 'number', fair=fair, use_sort=True)[0]))
      np.testing.assert_allclose(res[f'{key}.{v}'].values, want, rtol=RTOL)


def test_configs4_composite_against_the_oracle(ctx):
  """bench.py's configs[4] leg in miniature time but at full field size: 6 inits x 20 leads x 37 levels of z (p, t, climatology
  gather with NEW time labels in every chunk) -> RMSE / ACC per (lead, level) and zonal spectra; a 51-member t2m ensemble ->
  CRPS / spread-skill per lead; all three evaluations through pipeline.evaluate_passes (interleaved chunks, ONE Accumulation
  in HBM).  Against the float64 oracle on sampled (lead, level) cells -- inputs of those cells are downloaded, the oracle
  never sees the rest."""
  import torch
  ninit, nlead, nlev, m = 6, 20, 37, 51
  lead_time = (np.arange(nlead) * 6).astype('timedelta64[h]').astype('timedelta64[ns]')
  init_times = np.datetime64('2020-03-01T00', 'ns') + np.arange(ninit) * np.timedelta64(24, 'h')
  level = np.arange(nlev)
  sp = ('latitude', 'longitude')
  ndoy = 12
  clim_t = _randn((ndoy, 4, nlev, NLAT, NLON), 5, 280.0, 10.0)
  clim = xr.Dataset({'z': xr.DataArray(clim_t, dims=('dayofyear', 'hour', 'level') + sp, coords={
      'dayofyear': np.arange(61, 61 + ndoy), 'hour': np.array([0, 6, 12, 18]), 'level': level, 'latitude': LAT, 'longitude': LON})})
  zp = [_randn((1, nlead, nlev, NLAT, NLON), 100 + i, 280.0, 3.0) for i in range(ninit)]
  zt = [_randn((1, nlead, nlev, NLAT, NLON), 200 + i, 280.0, 3.0) for i in range(ninit)]
  et = [_randn((1, nlead, NLAT, NLON), 300 + i, 280.0) for i in range(2)]      # the ensemble pool: 2 buffers, reused cyclically
  ep = [_randn((1, nlead, m, NLAT, NLON), 400 + i) + et[i][:, :, None] for i in range(2)]
  et = [et[i] + _randn((1, nlead, NLAT, NLON), 500 + i) for i in range(2)]     # (target and members exchangeable: spread/skill ~ 1)
  torch.cuda.synchronize()
  index_of = {int(t.astype('int64')): i for i, t in enumerate(init_times)}

  def coords_for(inits):
    return {'init_time': inits, 'lead_time': lead_time, 'latitude': LAT, 'longitude': LON}

  def load_det(inits, leads):
    i = index_of[int(inits[0].astype('int64'))]
    cz = dict(coords_for(inits), level=level)
    dims = ('init_time', 'lead_time', 'level') + sp
    return {'z': xr.DataArray(zp[i], dims=dims, coords=cz)}, {'z': xr.DataArray(zt[i], dims=dims, coords=cz)}

  def load_ens(inits, leads):
    i = index_of[int(inits[0].astype('int64'))] % 2
    return ({'t2m': xr.DataArray(ep[i], dims=('init_time', 'lead_time', 'number') + sp, coords=coords_for(inits))},
            {'t2m': xr.DataArray(et[i], dims=('init_time', 'lead_time') + sp, coords=coords_for(inits))})
  det = {'rmse': deterministic.RMSE(), 'acc': deterministic.ACC(clim), 'bias': deterministic.Bias()}
  spec = {'spectrum_p': spectra.ZonalPowerSpectrum('predictions'), 'spectrum_t': spectra.ZonalPowerSpectrum('targets')}
  ens = {'crps': probabilistic.CRPSEnsemble(), 'ssr': probabilistic.UnbiasedSpreadSkillRatio()}
  area = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  zonal = aggregation.Aggregator(reduce_dims=['init_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
  times = time_chunks.TimeChunks(init_times, lead_time, init_time_chunk_size=1)
  stats = {}
  engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], 1
  try:
    out = pipeline.evaluate_passes(times, [('deterministic', load_det, det, area), ('spectra', load_det, spec, zonal),
                                           ('ensemble', load_ens, ens, area)], stats=stats)
    kinds = [e['kind'] for e in engine.S1_EVENT_LOG]
  finally:
    engine.S1_EVENT_LOG = None
  assert stats['collectives'] == 0  # one rank: nothing to combine
  # the two evaluations of z share their loader: every chunk is ONE sweep over p, t, c for the deterministic lanes AND both
  # spectra (wbx_det_spectrum) -- no separate spectrum launch, no separate deterministic launch
  assert kinds.count('det_spectrum') == ninit and 'spectrum' not in kinds and 'det' not in kinds, kinds
  dvals = out['deterministic'][None].metric_values(det)
  svals = out['spectra'][None].metric_values(spec)
  evals = out['ensemble'][None].metric_values(ens)
  assert dvals['rmse.z'].shape == (nlead, nlev) and svals['spectrum_p.z'].shape[:2] == (nlead, nlev)
  w = O.grid_area_weights(LAT)
  wn = w / w.sum()
  for lead, lev in ((0, 0), (7, 20), (19, 36)):
    num = {k: 0.0 for k in ('se', 'e', 'cov', 'spa', 'sta')}
    power = power_t = 0.0
    for i in range(ninit):
      p64 = zp[i][0, lead, lev].cpu().numpy().astype(np.float64)
      t64 = zt[i][0, lead, lev].cpu().numpy().astype(np.float64)
      valid = init_times[i] + lead_time[lead]
      doy = int((valid.astype('datetime64[D]') - valid.astype('datetime64[Y]').astype('datetime64[D]')) / np.timedelta64(1, 'D')) + 1
      hour = int((valid - valid.astype('datetime64[D]')) / np.timedelta64(1, 'h'))
      c64 = clim_t[doy - 61, hour // 6, lev].cpu().numpy().astype(np.float64)
      num['se'] += ((p64 - t64) ** 2 * wn[:, None]).sum() / NLON
      num['e'] += ((p64 - t64) * wn[:, None]).sum() / NLON
      num['cov'] += ((p64 - c64) * (t64 - c64) * wn[:, None]).sum() / NLON
      num['spa'] += ((p64 - c64) ** 2 * wn[:, None]).sum() / NLON
      num['sta'] += ((t64 - c64) ** 2 * wn[:, None]).sum() / NLON
      power = power + (O.zonal_power_spectrum(p64) * wn[:, None]).sum(axis=0)
      power_t = power_t + (O.zonal_power_spectrum(t64) * wn[:, None]).sum(axis=0)
    np.testing.assert_allclose(dvals['rmse.z'].values[lead, lev], np.sqrt(num['se'] / ninit), rtol=RTOL)
    np.testing.assert_allclose(dvals['bias.z'].values[lead, lev], num['e'] / ninit, rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(dvals['acc.z'].values[lead, lev], num['cov'] / np.sqrt(num['spa'] * num['sta']), rtol=RTOL)
    for name, ref in (('spectrum_p.z', power / ninit), ('spectrum_t.z', power_t / ninit)):
      got = np.asarray(svals[name].values)[lead, lev]
      # (the spectrum's own bound: the FFT is fp32, include/wbx.h "zonal spectrum")
      assert np.all(np.abs(got - ref) <= bound_1440(ref)), name
  for lead in (3,):
    skill = spread = var = ue = 0.0
    for i in range(ninit):
      pv = ep[i % 2][0, lead].cpu().numpy()
      tv = et[i % 2][0, lead].cpu().numpy()
      lanes = _five_lanes(pv, tv)
      mean = lambda a: (a * wn[:, None]).sum() / NLON
      skill += mean(lanes['CRPSSkill'])
      spread += mean(lanes['CRPSSpread'])
      var += mean(lanes['EnsembleVariance'])
      ue += mean(lanes['UnbiasedEnsembleMeanSquaredError'])
    np.testing.assert_allclose(evals['crps.t2m'].values[lead], (skill - 0.5 * spread) / ninit, rtol=RTOL)
    assert abs(np.sqrt(var / ue) - 1.0) < 0.01
    np.testing.assert_allclose(evals['ssr.t2m'].values[lead], np.sqrt(var / ue), rtol=RTOL)


# ---- the cross-rank combine behind the C ABI --------------------------------------------------------------------------
def test_cabi_communicator_one_rank(ctx):
  """wbx_comm_unique_id / wbx_comm_create / wbx_acc_allreduce / wbx_acc_read / wbx_acc_reset with raw pointers on a one-rank
  RCCL communicator (one GPU per box here: N > 1 is the driver's scaling run): the sum over one rank is the buffer itself,
  the collective counter advances, reset zeroes."""
  lib = ctx.lib
  ident = C.create_string_buffer(_hip.COMM_ID_BYTES)
  _hip.check(lib.wbx_comm_unique_id(ident), 'wbx_comm_unique_id')
  assert any(ident.raw)
  comm = C.c_void_p()
  _hip.check(lib.wbx_comm_create(ctx.handle, ident, 1, 0, C.byref(comm)), 'wbx_comm_create')
  n = 1000
  src = np.random.default_rng(0).normal(size=n)
  buf = ctx.alloc(n * 8)
  _hip.check(lib.wbx_memcpy_h2d(ctx.handle, C.c_void_p(buf.ptr), src.ctypes.data_as(C.c_void_p), n * 8), 'h2d')
  _hip.check(lib.wbx_acc_allreduce(ctx.handle, comm, C.c_void_p(buf.ptr), n), 'wbx_acc_allreduce')
  back = np.empty(n)
  _hip.check(lib.wbx_acc_read(ctx.handle, C.c_void_p(buf.ptr), n, back.ctypes.data_as(C.c_void_p)), 'wbx_acc_read')
  np.testing.assert_array_equal(back, src)
  nr, rk, nc = C.c_int32(), C.c_int32(), C.c_int64()
  _hip.check(lib.wbx_comm_info(comm, C.byref(nr), C.byref(rk), C.byref(nc)), 'wbx_comm_info')
  assert (nr.value, rk.value, nc.value) == (1, 0, 1)
  _hip.check(lib.wbx_acc_reset(ctx.handle, C.c_void_p(buf.ptr), n), 'wbx_acc_reset')
  _hip.check(lib.wbx_acc_read(ctx.handle, C.c_void_p(buf.ptr), n, back.ctypes.data_as(C.c_void_p)), 'wbx_acc_read')
  assert not back.any()
  assert lib.wbx_comm_create(ctx.handle, ident, 2, 5, C.byref(C.c_void_p())) == -1  # rank outside the group: WBX_ERR_INVALID
  _hip.check(lib.wbx_comm_destroy(comm), 'wbx_comm_destroy')


def test_chunk_loop_combines_through_the_cabi_collective(ctx):
  """pipeline.evaluate_chunks with `comm=CabiCommunicator`: the accumulators of the loop cross the (one-rank) group through
  wbx_acc_allreduce on the library's stream -- torch.distributed is not initialised at all -- and equal the plain result."""
  rng = np.random.default_rng(12)
  nlat, nlon, ninit, nlead = 19, 36, 4, 3
  lat, lon = np.linspace(-90, 90, nlat), np.arange(nlon) * 10.0
  init_times = np.datetime64('2020-01-01T00', 'ns') + np.arange(ninit) * np.timedelta64(24, 'h')
  lead_time = (np.arange(nlead) * 12).astype('timedelta64[h]').astype('timedelta64[ns]')
  pv = rng.normal(size=(ninit, nlead, nlat, nlon)).astype(np.float32)
  tv = rng.normal(size=(ninit, nlead, nlat, nlon)).astype(np.float32)

  def load(inits, leads):
    i = [int(np.where(init_times == x)[0][0]) for x in inits]
    cs = {'init_time': inits, 'lead_time': lead_time, 'latitude': lat, 'longitude': lon}
    dims = ('init_time', 'lead_time', 'latitude', 'longitude')
    return {'v': xr.DataArray(pv[i], dims=dims, coords=cs)}, {'v': xr.DataArray(tv[i], dims=dims, coords=cs)}
  metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE()}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  times = time_chunks.TimeChunks(init_times, lead_time, init_time_chunk_size=1)
  plain = pipeline.evaluate_chunks(times, load, metrics, agg)[None].metric_values(metrics)
  comm = distributed.CabiCommunicator(engine.new_context(), distributed.CabiCommunicator.new_unique_id(), 1, 0)
  try:
    got = pipeline.evaluate_chunks(times, load, metrics, agg, force_collective=True, comm=comm)[None].metric_values(metrics)
    assert comm.collectives == 1
  finally:
    comm.close()
  for k in plain:
    np.testing.assert_array_equal(got[k].values, plain[k].values)
  w = O.grid_area_weights(lat)
  want = np.sqrt((((pv.astype(np.float64) - tv) ** 2) * w[None, None, :, None]).sum(axis=(0, 2, 3)) / (w.sum() * nlon * ninit))
  np.testing.assert_allclose(plain['rmse.v'].values, want, rtol=RTOL)


# ---- spectra + deterministic lanes in one sweep ---------------------------------------------------------------------------
@pytest.mark.parametrize('func', ['DET6', 'DET3'])
def test_det_spectrum_entry_point_against_the_oracle(ctx, func):
  """wbx_det_spectrum with raw pointers: rows of 1440 points of p, t and a climatology addressed through a gather table (other
  time labels per lead), ragged row count (not a multiple of the teams).  The partial buffer must be exactly wbx_det_partial's,
  the per-row sums the float64 formulas of the oracle, both spectra numpy.fft's within the transform's bound."""
  from weatherbenchx_amd import planner
  rng = np.random.default_rng(3)
  nlead, nlev, nlat, nlon = 3, 2, 37, 1440
  dims = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
  shape = (1, nlead, nlev, nlat, nlon)
  pv = (rng.normal(size=shape) * 3 + 280).astype(np.float32)
  tv = (rng.normal(size=shape) * 3 + 280).astype(np.float32)
  nslot = 5
  cv = (rng.normal(size=(nslot, nlev, nlat, nlon)) * 10 + 280).astype(np.float32)
  slot_of_lead = np.array([4, 0, 2])
  p, t = xr.DataArray(pv, dims=dims), xr.DataArray(tv, dims=dims)
  c = xr.DataArray(cv, dims=('slot', 'level', 'latitude', 'longitude'))
  devs = [engine._to_device(ctx, a, _hip.F32) for a in (p, t, c)] + [None]
  lays = [d.layout for d in devs[:3]] + [None]
  sizes = dict(zip(dims, shape))
  table = (slot_of_lead * devs[2].layout.stride('slot')).reshape(1, nlead).astype(np.int64)
  gather = planner.GatherSpec(dims=('init_time', 'lead_time'), table=table) if func == 'DET6' else None
  if func == 'DET3':
    lays[2] = None
  plan = planner.build_s1_plan(dims, sizes, lays, ['init_time', 'latitude', 'longitude'], wdep_dims=['latitude'], gather=gather)
  assert plan.ndepth == 1 and plan.nchunk == 1 and plan.nkey == nlead * nlev * nlat and plan.key_dims == ('lead_time', 'level', 'latitude')
  dplan = engine._device_plan(ctx, plan)
  code = getattr(_hip, func)
  nl = _hip.DET_LANES[code]
  nrows, ngroup, nk = plan.nkey, nlead * nlev, nlon // 2 + 1
  w = np.cos(np.deg2rad(np.linspace(-88, 88, nlat)))
  group = np.repeat(np.arange(ngroup, dtype=np.int32), nlat)
  scale = np.tile(w, ngroup)
  g_dev, s_dev = ctx.upload(group), ctx.upload(scale)
  part_a, part_b = ctx.alloc(nrows * nl * 8), ctx.alloc(nrows * nl * 8)
  pw_p, pw_t = ctx.alloc(ngroup * nk * 8), ctx.alloc(ngroup * nk * 8)
  ptr = lambda d: C.c_void_p(d.ptr) if d is not None else None
  cdev = devs[2] if func == 'DET6' else None
  _hip.check(ctx.lib.wbx_det_partial(ctx.handle, C.byref(dplan.struct), code, _hip.F32, ptr(devs[0]), ptr(devs[1]), ptr(cdev), None,
                                     ptr(part_a)), 'wbx_det_partial')
  _hip.check(ctx.lib.wbx_det_spectrum(ctx.handle, C.byref(dplan.struct), code, _hip.F32, ptr(devs[0]), ptr(devs[1]), ptr(cdev),
                                      ptr(g_dev), ptr(s_dev), ngroup, ptr(part_b), ptr(pw_p), ptr(pw_t)), 'wbx_det_spectrum')
  ctx.synchronize()
  a = ctx.download(part_a.ptr, (nrows, nl)).copy()
  b = ctx.download(part_b.ptr, (nrows, nl)).copy()
  np.testing.assert_allclose(b, a, rtol=1e-13, atol=1e-9)
  p64, t64 = pv.astype(np.float64)[0], tv.astype(np.float64)[0]
  c64 = cv.astype(np.float64)[slot_of_lead]
  want = [O.error(p64, t64), O.absolute_error(p64, t64), O.squared_error(p64, t64)]
  if func == 'DET6':
    want += [O.squared_prediction_anomaly(p64, c64), O.squared_target_anomaly(t64, c64), O.anomaly_covariance(p64, t64, c64)]
  for lane, wv in enumerate(want):
    np.testing.assert_allclose(b[:, lane], wv.sum(axis=-1).reshape(-1), rtol=1e-11, atol=1e-6, err_msg=f'lane {lane}')
  for buf, f64 in ((pw_p, p64), (pw_t, t64)):
    got = ctx.download(buf.ptr, (ngroup, nk)).copy()
    ref = (O.zonal_power_spectrum(f64) * w[None, None, :, None]).sum(axis=2).reshape(ngroup, nk)
    assert np.all(np.abs(got - ref) <= bound_1440(ref))
  # misuse is refused, not mis-run
  assert ctx.lib.wbx_det_spectrum(ctx.handle, C.byref(dplan.struct), _hip.PASS1, _hip.F32, ptr(devs[0]), ptr(devs[1]), ptr(cdev),
                                  ptr(g_dev), ptr(s_dev), ngroup, ptr(part_b), ptr(pw_p), ptr(pw_t)) == -1


def test_fused_and_separate_launches_agree_through_the_chunk_loop(ctx, monkeypatch):
  """Two evaluations over the same loader (RMSE / ACC per (lead, level); zonal spectra of predictions and targets): with the
  fused launch and with engine.FUSE_DET_SPECTRA off -- the deterministic values bit for bit, the spectra to the last few ulps
  (same transform, other row pairing)."""
  rng = np.random.default_rng(8)
  nlat, nlon, ninit, nlead, nlev = 25, 1440, 3, 2, 3
  lat, lon = np.linspace(-84, 84, nlat), np.arange(nlon) * 0.25
  init_times = np.datetime64('2021-06-01T00', 'ns') + np.arange(ninit) * np.timedelta64(24, 'h')
  lead_time = (np.arange(nlead) * 12).astype('timedelta64[h]').astype('timedelta64[ns]')
  level = np.array([500, 700, 850])
  pv = (rng.normal(size=(ninit, nlead, nlev, nlat, nlon)) * 2 + 270).astype(np.float32)
  tv = (rng.normal(size=(ninit, nlead, nlev, nlat, nlon)) * 2 + 270).astype(np.float32)
  clim = xr.Dataset({'z': xr.DataArray((rng.normal(size=(8, 2, nlev, nlat, nlon)) * 5 + 270).astype(np.float32),
                                       dims=('dayofyear', 'hour', 'level', 'latitude', 'longitude'),
                                       coords={'dayofyear': np.arange(152, 160), 'hour': np.array([0, 12]), 'level': level,
                                               'latitude': lat, 'longitude': lon})})

  def load(inits, leads):
    i = [int(np.where(init_times == x)[0][0]) for x in inits]
    cs = {'init_time': inits, 'lead_time': lead_time, 'level': level, 'latitude': lat, 'longitude': lon}
    dims = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
    return {'z': xr.DataArray(pv[i], dims=dims, coords=cs)}, {'z': xr.DataArray(tv[i], dims=dims, coords=cs)}
  det = {'rmse': deterministic.RMSE(), 'acc': deterministic.ACC(clim), 'mae': deterministic.MAE()}
  spec = {'sp': spectra.ZonalPowerSpectrum('predictions'), 'st': spectra.ZonalEnergySpectrum('targets')}
  spec_same = {'sp': spectra.ZonalPowerSpectrum('predictions'), 'st': spectra.ZonalPowerSpectrum('targets')}
  area = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  zonal = aggregation.Aggregator(reduce_dims=['init_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
  times = time_chunks.TimeChunks(init_times, lead_time, init_time_chunk_size=1)

  def run(spec_metrics, fuse):
    monkeypatch.setattr(engine, 'FUSE_DET_SPECTRA', fuse)
    engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], 1
    try:
      out = pipeline.evaluate_passes(times, [('det', load, det, area), ('spec', load, spec_metrics, zonal)])
      kinds = [e['kind'] for e in engine.S1_EVENT_LOG]
    finally:
      engine.S1_EVENT_LOG = None
    return out['det'][None].metric_values(det), out['spec'][None].metric_values(spec_metrics), kinds
  monkeypatch.setattr(engine, 'FOLD_DET_SPECTRA', False)  # the fused launch writes wbx_det_partial's buffer: bit for bit
  d1, s1, k1 = run(spec_same, True)
  d0, s0, k0 = run(spec_same, False)
  assert k1.count('det_spectrum') == ninit and 'spectrum' not in k1
  assert 'det_spectrum' not in k0 and k0.count('spectrum') == 2 * ninit
  for k in d0:
    np.testing.assert_array_equal(d1[k].values, d0[k].values, err_msg=k)
  for k in s0:
    np.testing.assert_allclose(s1[k].values, s0[k].values, rtol=1e-12, err_msg=k)
  # (r6) stage 2 folded into the fused launch (the default): the same sums in another order -- fp64 rounding apart
  monkeypatch.setattr(engine, 'FOLD_DET_SPECTRA', True)
  df, sf, kf = run(spec_same, True)
  assert kf.count('det_spectrum') == ninit and 'spectrum' not in kf
  for k in d0:
    np.testing.assert_allclose(df[k].values, d0[k].values, rtol=1e-13, err_msg=k)
  for k in s0:
    np.testing.assert_array_equal(sf[k].values, s1[k].values, err_msg=k)
  # spectra with different row scales (power vs energy) do not share a (group, scale) table: no fusion, same numbers
  d2, s2, k2 = run(spec, True)
  assert 'det_spectrum' not in k2
  np.testing.assert_array_equal(d2['rmse.z'].values, d0['rmse.z'].values)


# ---- the pipelined sweep over latitude-fastest planes (ens_pipe_kernel<.., FLAT>) ----------------------------------------------
@pytest.mark.parametrize('nlat,nlon,m', [(721, 96, 51), (97, 40, 51), (33, 50, 50), (181, 64, 16), (97, 40, 13), (65, 48, 60),
                                         (97, 24, 4), (129, 40, 32)])
def test_latitude_fastest_planes_through_the_pipelined_sweep(ctx, nlat, nlon, m, monkeypatch):
  """Area-weighted ensemble suite on [init, level, member, longitude, latitude] arrays (the public IFS-ENS layout): the
  latitude weights are folded into stage 1 and the planes are walked flat by one-wave blocks.  Rows of 721 / 97 / 33 / 181
  floats put every chunk boundary inside a 64-element tile (the lanes in front of it are dropped), two inits make two planes
  per key, a NaN member sits right in front of a chunk boundary and the weight index wraps inside a tile (nlat = 33 < 64);
  M = 13 and 60 run the padded buckets (16, 64), M = 4 and 32 the small exact ones.
  Against the float64 oracle per (level), and against the 256-thread flat sweep (s1_xf1_kernel) the same library runs when the
  engine does not ask for one-wave blocks."""
  rng = np.random.default_rng(nlat * 1000 + nlon)
  ninit, nlev = 2, 3
  lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
  tv = (rng.normal(size=(ninit, nlev, nlon, nlat)) + 280).astype(np.float32)
  pv = (tv[:, :, None] + rng.normal(size=(ninit, nlev, m, nlon, nlat))).astype(np.float32)
  tv = (tv + rng.normal(size=tv.shape)).astype(np.float32)
  pv[1, 2, min(7, m - 1), nlon // 2, nlat - 1] = np.nan  # the last point of a row: whatever chunk ends there ends on it
  pd, td = ('init_time', 'level', 'number', 'longitude', 'latitude'), ('init_time', 'level', 'longitude', 'latitude')
  coords = {'init_time': np.array(['2020-01-01', '2020-01-02'], dtype='datetime64[ns]'), 'level': np.arange(nlev),
            'latitude': lat, 'longitude': lon}
  import torch
  p = {'v': xr.DataArray(torch.from_numpy(pv).cuda(), dims=pd, coords=coords)}
  t = {'v': xr.DataArray(torch.from_numpy(tv).cuda(), dims=td, coords=coords)}
  metrics = _suite()
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])

  def run(one_wave):
    monkeypatch.setattr(engine, 'ENS_PIPE', one_wave)
    engine.clear_caches()
    engine.S1_EVENT_LOG = []
    try:
      fresh = lambda d: {k: xr.DataArray(v.data, dims=v.dims, coords={c: v[c].values for c in v.dims}) for k, v in d.items()}
      out = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, fresh(p), fresh(t))).metric_values(metrics)
      log = [e for e in engine.S1_EVENT_LOG if e['kind'] == 'ens']
    finally:
      engine.S1_EVENT_LOG = None
    return {k: np.asarray(out[f'{k}.v'].values) for k in metrics}, log

  got, log = run(True)
  assert len(log) == 1 and log[0]['flat'] and log[0]['block'] == 64, log
  ref, rlog = run(False)
  assert len(rlog) == 1 and rlog[0]['flat'] and rlog[0]['block'] == 256, rlog
  wt = O.grid_area_weights(lat)
  p64, t64 = pv.astype(np.float64), tv.astype(np.float64)
  lanes = {'skill': O.crps_skill(p64, pd, t64, td, 'number')[0], 'spread': O.crps_spread(p64, pd, 'number', fair=True, use_sort=True)[0],
           'var': O.ensemble_variance(p64, pd, 'number')[0],
           'uemse': O.unbiased_ensemble_mean_squared_error(p64, pd, t64, td, 'number')[0],
           'emse': O.ensemble_mean_squared_error(p64, pd, t64, td, 'number')[0]}
  mean = {k: (v * wt).sum(axis=(0, 2, 3)) / (wt.sum() * ninit * nlon) for k, v in lanes.items()}  # NaN at level 2, like xr.dot
  want = {'crps': mean['skill'] - 0.5 * mean['spread'],
          'unbiased_spread_skill': np.sqrt(mean['var'] / mean['uemse']),
          'unbiased_mean_rmse': np.sqrt(mean['uemse']), 'mean_rmse': np.sqrt(mean['emse'])}
  for k in metrics:
    assert got[k].shape == (nlev,)
    assert np.isnan(got[k][2]) and np.isnan(want[k][2]) and np.isnan(ref[k][2]), k  # the NaN member poisons its level only
    np.testing.assert_allclose(got[k][:2], want[k][:2], rtol=RTOL, err_msg=k)
    np.testing.assert_allclose(got[k][:2], ref[k][:2], rtol=RTOL, err_msg=k + ' (256-thread sweep)')

