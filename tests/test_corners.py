"""The two parameter combinations that used to be evaluated as elementwise labeled-array arithmetic (VERDICT r3 missing #5) run
on HIP kernels now -- asserted from the launch log on both backends, results against the oracle:
  * skipna_ensemble=True with ensemble-valued targets (probabilistic.py:133-145, 304-336): wbx_ens2_partial, both statistics of
    a (predictions, targets) pair in ONE launch;
  * ErrorExceedance against thresholds that vary with the statistic's dims (deterministic.py:262-295): wbx_cat_exceed_field."""
import numpy as np
import pytest

from oracle import wbx_oracle as O
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import engine
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import deterministic
from weatherbenchx_amd.metrics import probabilistic

RTOL = 1e-6


def _logged(fn):
  engine.S1_EVENT_LOG = []
  try:
    out = fn()
    log = [e['kind'] for e in engine.S1_EVENT_LOG]
  finally:
    engine.S1_EVENT_LOG = None
  return out, log


def test_skipna_ensemble_with_an_ensemble_of_targets_is_one_kernel_launch(backend):
  rng = np.random.default_rng(21)
  nlead, m, n, nlat, nlon = 2, 6, 4, 9, 16
  lat = np.linspace(-80, 80, nlat)
  pdims = ('lead_time', 'number', 'latitude', 'longitude')
  pv = rng.normal(size=(nlead, m, nlat, nlon)).astype(np.float32)
  tv = rng.normal(size=(nlead, n, nlat, nlon)).astype(np.float32)
  pv[rng.random(pv.shape) < 0.15] = np.nan
  tv[rng.random(tv.shape) < 0.15] = np.nan
  pv[0, :, 3, 5] = np.nan       # no prediction member at all -> NaN point
  tv[1, 1:, 2, 2] = np.nan      # a single target member -> its variance (ddof = 1) is NaN
  coords = {'latitude': lat, 'longitude': np.arange(nlon) * 22.5}
  p = {'v': xr.DataArray(pv, dims=pdims, coords=coords)}
  t = {'v': xr.DataArray(tv, dims=pdims, coords=coords)}
  stats = {'skill': probabilistic.CRPSSkill(skipna_ensemble=True), 'uemse': probabilistic.UnbiasedEnsembleMeanSquaredError(skipna_ensemble=True)}
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()], skipna=True)

  def run():
    return agg.aggregate_statistics({k: s.compute(p, t) for k, s in stats.items()}).mean_statistics()
  means, log = _logged(run)
  assert log == ['ens2'], log
  w = [(O.grid_area_weights(lat), ('latitude',))]
  sdims = ('lead_time', 'latitude', 'longitude')
  p64, t64 = pv.astype(np.float64), tv.astype(np.float64)
  with np.errstate(all='ignore'):
    import warnings
    with warnings.catch_warnings():
      warnings.simplefilter('ignore', RuntimeWarning)
      uemse = ((np.nanmean(p64, 1) - np.nanmean(t64, 1)) ** 2 - np.nanvar(p64, 1, ddof=1) / (~np.isnan(p64)).sum(1)
               - np.nanvar(t64, 1, ddof=1) / (~np.isnan(t64)).sum(1))  # probabilistic.py:304-336 with skipna=True on both sides
  for k, lane in (('skill', O.crps_skill(pv, pdims, tv, pdims, 'number', skipna_ensemble=True)), ('uemse', (uemse, sdims))):
    assert lane[1] == sdims
    assert np.isnan(lane[0]).any()  # the NaN points are there, and skipna=True of the aggregator counts them out
    a = O.aggregate(lane[0], sdims, ['latitude', 'longitude'], weights=w, skipna=True)
    np.testing.assert_allclose(np.asarray(means[k]['v'].values), a[0] / a[1], rtol=RTOL, err_msg=k)
  # the per-point values (Statistic.compute's contract) come from the same kernel with every dim kept
  vals, log = _logged(lambda: np.asarray(stats['skill'].compute(p, t)['v'].values))
  assert log == ['ens2'], log
  np.testing.assert_allclose(vals, O.crps_skill(pv, pdims, tv, pdims, 'number', skipna_ensemble=True)[0], rtol=1e-6, equal_nan=True)


def test_error_exceedance_against_a_threshold_field_is_a_kernel_launch(backend):
  rng = np.random.default_rng(22)
  dims = ('lead_time', 'level', 'latitude', 'longitude')
  lat = np.linspace(-80, 80, 9)
  pv = rng.normal(size=(3, 2, 9, 12)).astype(np.float32)
  tv = rng.normal(size=(3, 2, 9, 12)).astype(np.float32)
  tv[2, 1, 4, 4] = np.nan
  coords = {'level': np.array([500, 850]), 'latitude': lat}
  # thresholds per (level, latitude), stored latitude-major: consumed through their own strides
  tvals = np.abs(rng.normal(size=(9, 2, 3))) + 0.2
  tvals[4, 0, 1] = np.nan
  thr = xr.DataArray(tvals, dims=('latitude', 'level', 'thr'), coords={'level': coords['level'], 'latitude': lat, 'thr': np.arange(3)})
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()], skipna=True)
  p, t = {'v': xr.DataArray(pv, dims=dims, coords=coords)}, {'v': xr.DataArray(tv, dims=dims, coords=coords)}

  def run():
    stat = deterministic.ErrorExceedance(thr).compute(p, t)['v']
    return agg.aggregate_stat_var(stat).mean_statistics().transpose('lead_time', 'level', 'thr').values
  got, log = _logged(run)
  assert log == ['cat'], log
  ae = np.abs(pv.astype(np.float64) - tv.astype(np.float64))[..., None]
  th = np.transpose(tvals, (1, 0, 2))[None, :, :, None, :]
  with np.errstate(invalid='ignore'):
    want = np.where(np.isnan(ae) | np.isnan(th), np.nan, (ae > th).astype(np.float64))
  w = O.grid_area_weights(lat)[None, None, :, None, None]
  ok = ~np.isnan(want)
  ref = (np.where(ok, want, 0) * w).sum(axis=(2, 3)) / (ok * w).sum(axis=(2, 3))
  np.testing.assert_allclose(np.asarray(got), ref, rtol=RTOL)


def test_no_elementwise_payload_arithmetic_left_under_metrics():
  """The metric classes only build lazy statistics (launch descriptions): no `abs(p - t)` / comparisons on the inputs.  (What is
  left: PredictionPassthrough / TargetPassthrough with copy_nans_from_* = True select with `.where` -- a copy, no arithmetic.)"""
  import os
  import re
  root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'weatherbenchx_amd', 'metrics')
  for name in ('deterministic.py', 'probabilistic.py'):
    text = open(os.path.join(root, name)).read()
    code = re.sub(r'''""".*?"""''', '', text, flags=re.S)
    code = '\n'.join(line.split('#')[0] for line in code.split('\n'))
    assert not re.search(r'abs\(predictions - |abs_error|\.notnull\(\)|predictions - targets|> thresholds', code), name


@pytest.mark.parametrize('kind', ['scalar', 'level_latitude', 'two_new_dims'])
def test_error_exceedance_thresholds_that_add_no_or_several_dims(backend, kind):
  """the reference takes ANY broadcastable thresholds array (`abs_error > thresholds`, deterministic.py:283-295) -- a
  0-D threshold, per-(level, latitude) thresholds with no category dim, thresholds that add two dims.  All of them run the
  threshold-field kernel in one launch; the per-point values (Statistic.compute's contract) and the aggregated means agree
  with the labeled-array formula."""
  rng = np.random.default_rng(3)
  dims = ('lead_time', 'level', 'latitude', 'longitude')
  lat = np.linspace(-80, 80, 9)
  pv = rng.normal(size=(3, 2, 9, 12)).astype(np.float32)
  tv = rng.normal(size=(3, 2, 9, 12)).astype(np.float32)
  tv[2, 1, 4, 4] = np.nan
  coords = {'level': np.array([500, 850]), 'latitude': lat}
  if kind == 'scalar':
    thr, th, new = xr.DataArray(np.float64(0.7)), np.float64(0.7), ()
  elif kind == 'level_latitude':
    tvals = np.abs(rng.normal(size=(9, 2))) + 0.2
    tvals[4, 0] = np.nan
    thr = xr.DataArray(tvals, dims=('latitude', 'level'), coords=coords)
    th, new = tvals.T[None, :, :, None], ()
  else:
    tvals = np.abs(rng.normal(size=(2, 3, 4))) + 0.2
    tvals[1, 2, 3] = np.nan
    thr = xr.DataArray(tvals, dims=('level', 'quantile', 'season'), coords={'level': coords['level'], 'quantile': [0.1, 0.5, 0.9],
                                                                            'season': np.arange(4)})
    th, new = tvals[None, :, None, None, :, :], ('quantile', 'season')
  ae = np.abs(pv.astype(np.float64) - tv.astype(np.float64))
  ae = ae.reshape(ae.shape + (1,) * len(new))
  with np.errstate(invalid='ignore'):
    want = np.where(np.isnan(ae) | np.isnan(th), np.nan, (ae > th).astype(np.float64))
  p, t = {'v': xr.DataArray(pv, dims=dims, coords=coords)}, {'v': xr.DataArray(tv, dims=dims, coords=coords)}
  stat = deterministic.ErrorExceedance(thr).compute(p, t)['v']
  assert tuple(stat.dims) == dims + new and tuple(stat.shape) == want.shape
  for d in new:
    np.testing.assert_array_equal(stat.coords[d].values, thr.coords[d].values)
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()], skipna=True)
  got, log = _logged(lambda: agg.aggregate_stat_var(deterministic.ErrorExceedance(thr).compute(p, t)['v']).mean_statistics())
  assert log == ['cat'], log
  assert tuple(got.dims) == ('lead_time', 'level') + new
  w = O.grid_area_weights(lat).reshape((1, 1, 9, 1) + (1,) * len(new))
  ok = ~np.isnan(want)
  ref = (np.where(ok, want, 0) * w).sum(axis=(2, 3)) / (ok * w).sum(axis=(2, 3))
  np.testing.assert_allclose(np.asarray(got.values), ref, rtol=RTOL)
  vals, log = _logged(lambda: np.asarray(stat.values))
  assert log == ['cat'], log
  np.testing.assert_array_equal(vals, want)
