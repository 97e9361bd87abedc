"""The ensemble family under Regions x land/sea bins, GridAreaWeighting and masked=True -- the public benchmark's
probabilistic configuration (public_benchmark/run_benchmark_evaluation.py:341-354, 365-382) -- through the drop-in API on
both backends: ONE wbx_ens_binned launch per (variable, mask setting), every bin of every lane against the oracle.
Tolerance: rtol 1e-6 (north_star)."""
import numpy as np
import pytest

from oracle import wbx_oracle as O
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import binning
from weatherbenchx_amd import engine
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.metrics import deterministic
from weatherbenchx_amd.metrics import probabilistic
from weatherbenchx_amd.metrics import wrappers

RTOL = 1e-6
REGIONS = {'global': ((-90, 90), (0, 360)), 'tropics': ((-20, 20), (0, 360)), 'nh': ((20, 90), (0, 360)),
           'sh': ((-90, -20), (0, 360)), 'europe': ((35, 75), (-12.5, 42.5)), 'namerica': ((25, 60), (240, 285)),
           'nowhere': ((100, 101), (0, 360))}

LAYOUTS = {
    'lon_fastest': (('lead_time', 'number', 'latitude', 'longitude'), ('lead_time', 'latitude', 'longitude')),
    'lat_fastest': (('lead_time', 'number', 'longitude', 'latitude'), ('lead_time', 'longitude', 'latitude')),
    # the recorded IFS-ENS chunk (docs/source/how_to/metric_wrappers.ipynb:955-964): member-slow, latitude fastest
    'ifs': (('init_time', 'number', 'lead_time', 'longitude', 'latitude'), ('init_time', 'lead_time', 'longitude', 'latitude')),
}


def lane_statistics():
  return {'CRPSSkill': probabilistic.CRPSSkill(), 'CRPSSpread': probabilistic.CRPSSpread(use_sort=True),
          'EnsembleVariance': probabilistic.EnsembleVariance(),
          'UnbiasedEnsembleMeanSquaredError': probabilistic.UnbiasedEnsembleMeanSquaredError(),
          'EnsembleMeanSquaredError': wrappers.WrappedStatistic(deterministic.SquaredError(),
                                                                wrappers.EnsembleMean(which='predictions'))}


def oracle_lanes(pv, pd, tv, td):
  return {'CRPSSkill': O.crps_skill(pv, pd, tv, td, 'number'),
          'CRPSSpread': O.crps_spread(pv, pd, 'number', fair=True, use_sort=True),
          'EnsembleVariance': O.ensemble_variance(pv, pd, 'number'),
          'UnbiasedEnsembleMeanSquaredError': O.unbiased_ensemble_mean_squared_error(pv, pd, tv, td, 'number'),
          'EnsembleMeanSquaredError': O.ensemble_mean_squared_error(pv, pd, tv, td, 'number')}


def make_case(layout, m, nlat, nlon, nlead, seed, *, offset=280.0, mask=None, nan_at=None, ninit=2):
  """(predictions, targets, raw arrays) in `layout`; `mask`: bool[lat, lon] validity attached to the targets as the `mask`
  coordinate (data_loaders/base.py:25-56)."""
  pd, td = LAYOUTS[layout]
  rng = np.random.default_rng(seed)
  lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
  sizes = {'lead_time': nlead, 'number': m, 'latitude': nlat, 'longitude': nlon, 'init_time': ninit}
  tv = (rng.normal(size=[sizes[d] for d in td]) + offset).astype(np.float32)
  pv = (np.expand_dims(tv, pd.index('number')) + rng.normal(size=[sizes[d] for d in pd])).astype(np.float32)
  tv = (tv + rng.normal(size=tv.shape)).astype(np.float32)
  if nan_at is not None:
    pv[nan_at] = np.nan
  coords = {'latitude': lat, 'longitude': lon, 'lead_time': (np.arange(nlead) * 12).astype('timedelta64[h]').astype('timedelta64[ns]'),
            'init_time': np.datetime64('2020-01-01T00', 'ns') + np.arange(ninit) * np.timedelta64(1, 'D')}
  p = xr.DataArray(pv, dims=pd, coords={k: v for k, v in coords.items() if k in pd})
  t = xr.DataArray(tv, dims=td, coords={k: v for k, v in coords.items() if k in td})
  if mask is not None:
    sp = tuple(d for d in td if d in ('latitude', 'longitude'))
    mv = mask if sp == ('latitude', 'longitude') else np.ascontiguousarray(mask.T)
    t = t.assign_coords(mask=xr.DataArray(mv, dims=sp, coords={'latitude': lat, 'longitude': lon}))
  return p, t, pv, tv, lat, lon


def check_against_oracle(state, stats, pv, tv, layout, lat, lon, land, reduce_dims, mask=None, rtol=RTOL, regions=None):
  pd, td = LAYOUTS[layout]
  regions = REGIONS if regions is None else regions
  names, masks = O.region_masks(lat, lon, regions, land_sea_mask=land)
  bm = [('region', masks, ('region', 'latitude', 'longitude'))]
  w = (O.grid_area_weights(lat), ('latitude',))
  want = oracle_lanes(pv, pd, tv, td)
  for name, (lane, ldims) in want.items():
    member_only = name in ('CRPSSpread', 'EnsembleVariance')  # statistics of the predictions alone carry no mask
    use = None if (mask is None or member_only) else mask
    sws, sw, out_dims = O.aggregate(lane, ldims, reduce_dims, weights=[w], bin_masks=bm, mask=use,
                                    mask_dims=('latitude', 'longitude') if use is not None else None)
    key = stats[name].unique_name
    got_s, got_w = state.sum_weighted_statistics[key]['v'], state.sum_weights[key]['v']
    assert list(got_s['region'].values) == names
    gs, gw = np.asarray(got_s.transpose(*out_dims).values), np.asarray(got_w.transpose(*out_dims).values)
    np.testing.assert_allclose(gw, sw, rtol=1e-12, atol=1e-12, err_msg=f'{layout} {name} sum_weights')
    scale = np.abs(sws).max() if np.isfinite(sws).all() else 1.0
    np.testing.assert_allclose(gs, sws, rtol=rtol, atol=rtol * 1e-3 * scale, err_msg=f'{layout} {name} sum_weighted_statistics')
    if 'nowhere' in names:
      assert (gw[..., names.index('nowhere')] == 0).all()


def run(stats, agg, p, t):
  engine.S1_EVENT_LOG = []
  try:
    state = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(stats, {'v': p}, {'v': t}))
    state.wait() if hasattr(state, 'wait') else None
    log = list(engine.S1_EVENT_LOG)
  finally:
    engine.S1_EVENT_LOG = None
  return state, log


@pytest.mark.parametrize('layout', sorted(LAYOUTS))
@pytest.mark.parametrize('m', [5, 51])
def test_every_bin_of_every_lane_one_launch(backend, layout, m):
  nlat, nlon = 37, 72
  rng = np.random.default_rng(7)
  land = rng.random((nlat, nlon)) > 0.6
  p, t, pv, tv, lat, lon = make_case(layout, m, nlat, nlon, 3, seed=m)
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  reduce_dims = ['latitude', 'longitude'] + (['init_time'] if layout == 'ifs' else [])
  agg = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)], masked=True)
  stats = lane_statistics()
  state, log = run(stats, agg, p, t)
  kinds = [e['kind'] for e in log]
  assert kinds == ['ens_binned'], kinds  # ONE pass over the members, no partial, no second stage
  check_against_oracle(state, stats, pv, tv, layout, lat, lon, land, reduce_dims)


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
def test_validity_mask_on_latitude_longitude(backend, layout, monkeypatch):
  """masked=True with a (latitude, longitude) `mask` coordinate on the targets: skill / unbiased MSE / mean MSE are masked
  (masked-out points contribute 0 whatever they hold -- NaN members there must not poison anything, aggregation.py:339-352),
  spread and variance are statistics of the predictions alone and stay unmasked: BOTH sets come out of one wbx_ens_binned
  launch (masked-out points are accumulated under their atom's twin); with engine.ENS_TWIN_MASK off they are two launches, and
  either way the numbers are the oracle's.  The order of the statistics does not matter (the member-only ones first here)."""
  nlat, nlon, m = 37, 72, 8
  rng = np.random.default_rng(11)
  land = rng.random((nlat, nlon)) > 0.5
  valid = rng.random((nlat, nlon)) > 0.3
  p, t, pv, tv, lat, lon = make_case(layout, m, nlat, nlon, 2, seed=3, mask=valid)
  # a NaN target where the mask says invalid: the masked lanes ignore it
  lat_i, lon_i = np.argwhere(~valid)[5]
  sp = LAYOUTS[layout][1]
  idx = tuple(1 if d == 'lead_time' else (lat_i if d == 'latitude' else lon_i) for d in sp)
  tv[idx] = np.nan
  t = xr.DataArray(tv, dims=t.dims, coords={k: t.coords[k].values for k in t.dims}).assign_coords(mask=t.coords['mask'])
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)], masked=True)
  stats = lane_statistics()
  stats = {k: stats[k] for k in ('EnsembleVariance', 'CRPSSpread', 'CRPSSkill', 'UnbiasedEnsembleMeanSquaredError', 'EnsembleMeanSquaredError')}
  state, log = run(stats, agg, p, t)
  assert [(e['kind'], e['flags'] & 1) for e in log] == [('ens_binned', 1)], log  # ONE pass over the members
  check_against_oracle(state, stats, pv, tv, layout, lat, lon, land, ['latitude', 'longitude'], mask=valid)
  monkeypatch.setattr(engine, 'ENS_TWIN_MASK', False)
  engine.clear_caches()
  p2 = xr.DataArray(pv, dims=p.dims, coords={k: p.coords[k].values for k in p.dims if k != 'number'})
  t2 = xr.DataArray(tv, dims=t.dims, coords={k: t.coords[k].values for k in t.dims}).assign_coords(mask=t.coords['mask'])
  state2, log2 = run(stats, agg, p2, t2)
  assert sorted((e['kind'], e['flags'] & 1) for e in log2) == [('ens_binned', 0), ('ens_binned', 1)], log2
  check_against_oracle(state2, stats, pv, tv, layout, lat, lon, land, ['latitude', 'longitude'], mask=valid)


def test_nan_member_poisons_every_bin_of_its_lead_time_only(backend):
  """skipna=False: a NaN anywhere in the reduced set makes EVERY bin of that output cell NaN -- also the bins the point is
  not in (NaN * 0, aggregation.py:272-277) -- and leaves the other cells alone."""
  nlat, nlon, m = 24, 128, 16
  pd = LAYOUTS['lon_fastest'][0]
  nan_at = tuple({'lead_time': 1, 'number': 3, 'latitude': 20, 'longitude': 100}[d] for d in pd)
  p, t, pv, tv, lat, lon = make_case('lon_fastest', m, nlat, nlon, 3, seed=5, nan_at=nan_at)
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(REGIONS)])
  stats = lane_statistics()
  state, _ = run(stats, agg, p, t)
  for name, s in stats.items():
    got = np.asarray(state.sum_weighted_statistics[s.unique_name]['v'].transpose('lead_time', 'region').values)
    assert np.isnan(got[1]).all(), name
    assert np.isfinite(got[[0, 2]]).all(), name
  check_against_oracle(state, stats, pv, tv, 'lon_fastest', lat, lon, None, ['latitude', 'longitude'])


def test_routes_that_stay_two_stage(backend):
  """skipna_ensemble (per-point member counts) and float64 members are not wbx_ens_binned's: same numbers through the two-stage
  route (x-kept ensemble kernel + wbx_contract_bits)."""
  nlat, nlon, m = 19, 36, 5
  p, t, pv, tv, lat, lon = make_case('lon_fastest', m, nlat, nlon, 2, seed=2)
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(REGIONS)])
  stats = {'CRPSSkill': probabilistic.CRPSSkill(skipna_ensemble=True)}
  state, log = run(stats, agg, p, t)
  assert not any(e['kind'] == 'ens_binned' for e in log)
  pd, td = LAYOUTS['lon_fastest']
  names, masks = O.region_masks(lat, lon, REGIONS)
  lane, ldims = O.crps_skill(pv, pd, tv, td, 'number')
  sws, sw, out_dims = O.aggregate(lane, ldims, ['latitude', 'longitude'], weights=[(O.grid_area_weights(lat), ('latitude',))],
                                  bin_masks=[('region', masks, ('region', 'latitude', 'longitude'))])
  got = state.sum_weighted_statistics[stats['CRPSSkill'].unique_name]['v'].transpose(*out_dims).values
  np.testing.assert_allclose(np.asarray(got), sws, rtol=RTOL)
  p64 = xr.DataArray(pv.astype(np.float64), dims=p.dims, coords={k: p.coords[k].values for k in p.dims if k != 'number'})
  t64 = xr.DataArray(tv.astype(np.float64), dims=t.dims, coords={k: t.coords[k].values for k in t.dims})
  state, log = run({'CRPSSkill': probabilistic.CRPSSkill()}, agg, p64, t64)
  assert not any(e['kind'] == 'ens_binned' for e in log)
  got = state.sum_weighted_statistics['CRPSSkill_number']['v'].transpose(*out_dims).values
  np.testing.assert_allclose(np.asarray(got), sws, rtol=RTOL)


def _check_lanes(state, stats, pv, tv, layout, lat, lon, land, reduce_dims, *, mask=None, mask_dims=None, skipna=False,
                 masked_member_only=False, rtol=RTOL):
  """Every bin of every lane AND of its weights against the oracle.  `mask` applies to the statistics that look at the targets
  (the member-only ones carry no mask coordinate: probabilistic.py:165-273) unless `masked_member_only`."""
  pd, td = LAYOUTS[layout]
  names, masks = O.region_masks(lat, lon, REGIONS, land_sea_mask=land)
  bm = [('region', masks, ('region', 'latitude', 'longitude'))]
  w = (O.grid_area_weights(lat), ('latitude',))
  for name, (lane, ldims) in oracle_lanes(pv, pd, tv, td).items():
    member_only = name in ('CRPSSpread', 'EnsembleVariance')
    use = mask if (mask is not None and (masked_member_only or not member_only)) else None
    sws, sw, out_dims = O.aggregate(lane, ldims, reduce_dims, weights=[w], bin_masks=bm, mask=use,
                                    mask_dims=mask_dims if use is not None else None, skipna=skipna)
    key = stats[name].unique_name
    gs = np.asarray(state.sum_weighted_statistics[key]['v'].transpose(*out_dims).values)
    gw = np.asarray(state.sum_weights[key]['v'].transpose(*out_dims).values)
    np.testing.assert_allclose(gw, sw, rtol=1e-12, atol=1e-12, err_msg=f'{layout} {name} sum_weights')
    scale = np.abs(sws).max() if np.isfinite(sws).all() else 1.0
    np.testing.assert_allclose(gs, sws, rtol=rtol, atol=rtol * 1e-3 * scale, err_msg=f'{layout} {name} sum_weighted_statistics')


@pytest.mark.parametrize('layout', sorted(LAYOUTS))
def test_nan_mask_with_time_strides_is_one_launch(backend, layout):
  """What the reference's loaders build (data_loaders/base.py:25-56, add_nan_mask_to_data): targets with NaNs and
  `mask = ~isnan(targets)` over EVERY dim of the targets -- another hole per lead time -- consumed by Aggregator(masked=True)
  (aggregation.py:339-352; public_benchmark/run_benchmark_evaluation.py:379-381).  Still ONE wbx_ens_binned launch for the five
  lanes (round 4 fell back to the two-stage route for any mask with a time / level stride)."""
  from weatherbenchx_amd import data as wdata
  nlat, nlon, m = 37, 72, 9
  rng = np.random.default_rng(21)
  land = rng.random((nlat, nlon)) > 0.55
  p, t, pv, tv, lat, lon = make_case(layout, m, nlat, nlon, 3, seed=6)
  tv[rng.random(tv.shape) < 0.2] = np.nan  # (another pattern at every (init, lead))
  t = wdata.add_nan_mask_to_data({'v': xr.DataArray(tv, dims=t.dims, coords={k: t.coords[k].values for k in t.dims})})['v']
  assert tuple(t.coords['mask'].dims) == tuple(t.dims)
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  reduce_dims = ['latitude', 'longitude'] + (['init_time'] if layout == 'ifs' else [])
  agg = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)], masked=True)
  stats = lane_statistics()
  state, log = run(stats, agg, p, t)
  assert [(e['kind'], e['flags'] & 1) for e in log] == [('ens_binned', 1)], log
  _check_lanes(state, stats, pv, tv, layout, lat, lon, land, reduce_dims, mask=~np.isnan(tv), mask_dims=LAYOUTS[layout][1])


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
@pytest.mark.parametrize('masked', [False, True])
def test_skipna_aggregation_is_one_launch(backend, layout, masked):
  """Aggregator(skipna=True) (aggregation.py:343-344, 353-355): every statistic leaves ITS OWN NaN points out of its sum and its
  weights -- skill / unbiased MSE / MSE of the mean where the target or a member is NaN, spread / variance where a member is.
  One wbx_ens_binned launch (NaN-target points go to their atom's twin).  With `masked`: a (latitude, longitude) mask coordinate
  on the targets on top, both sets from the same launch."""
  nlat, nlon, m = 37, 72, 7
  rng = np.random.default_rng(33)
  land = rng.random((nlat, nlon)) > 0.5
  valid = rng.random((nlat, nlon)) > 0.25
  p, t, pv, tv, lat, lon = make_case(layout, m, nlat, nlon, 3, seed=9, mask=valid if masked else None)
  tv[rng.random(tv.shape) < 0.15] = np.nan
  pv[rng.random(pv.shape) < 0.01] = np.nan
  pd, td = LAYOUTS[layout]
  p = xr.DataArray(pv, dims=pd, coords={k: p.coords[k].values for k in pd if k != 'number'})
  t2 = xr.DataArray(tv, dims=td, coords={k: t.coords[k].values for k in td})
  t = t2.assign_coords(mask=t.coords['mask']) if masked else t2
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)], masked=masked, skipna=True)
  stats = lane_statistics()
  state, log = run(stats, agg, p, t)
  assert [e['kind'] for e in log] == ['ens_binned'], log
  _check_lanes(state, stats, pv, tv, layout, lat, lon, land, ['latitude', 'longitude'], mask=valid if masked else None,
               mask_dims=('latitude', 'longitude'), skipna=True)
  # the member-only statistics first: the launch they trigger is the same one
  engine.clear_caches()
  order = ('EnsembleVariance', 'CRPSSpread', 'CRPSSkill', 'UnbiasedEnsembleMeanSquaredError', 'EnsembleMeanSquaredError')
  p = xr.DataArray(pv, dims=pd, coords={k: p.coords[k].values for k in pd if k != 'number'})
  state, log = run({k: stats[k] for k in order}, agg, p, t)
  assert [e['kind'] for e in log] == ['ens_binned'], log
  _check_lanes(state, stats, pv, tv, layout, lat, lon, land, ['latitude', 'longitude'], mask=valid if masked else None,
               mask_dims=('latitude', 'longitude'), skipna=True)


def test_skipna_with_the_same_mask_on_predictions_and_targets(backend):
  """Predictions that carry the targets' mask coordinate: spread / variance are then statistics of the masked group itself
  (lazy.ens_statistic: no member-only companion), i.e. lanes 1, 2 of the launch WITHOUT a twin output -- mask AND valid members,
  whatever the target."""
  nlat, nlon, m = 19, 36, 5
  rng = np.random.default_rng(12)
  valid = rng.random((nlat, nlon)) > 0.3
  p, t, pv, tv, lat, lon = make_case('lon_fastest', m, nlat, nlon, 2, seed=4, mask=valid)
  tv[rng.random(tv.shape) < 0.2] = np.nan
  pv[rng.random(pv.shape) < 0.02] = np.nan
  pd, td = LAYOUTS['lon_fastest']
  mask_da = t.coords['mask']
  p = xr.DataArray(pv, dims=pd, coords={k: p.coords[k].values for k in pd if k != 'number'}).assign_coords(mask=mask_da)
  t = xr.DataArray(tv, dims=td, coords={k: t.coords[k].values for k in td}).assign_coords(mask=mask_da)
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(REGIONS)], masked=True, skipna=True)
  stats = lane_statistics()
  state, log = run(stats, agg, p, t)
  assert [e['kind'] for e in log] == ['ens_binned'], log
  _check_lanes(state, stats, pv, tv, 'lon_fastest', lat, lon, None, ['latitude', 'longitude'], mask=valid,
               mask_dims=('latitude', 'longitude'), skipna=True, masked_member_only=True)


def test_twin_sums_never_serve_another_target(backend):
  """the unmasked half a masked launch leaves with the predictions ('_wbx_twin') belongs to the member-only
  statistics (spread, variance).  A second evaluation of the SAME predictions object against other, unmasked targets has lanes
  that depend on those targets (skill, unbiased MSE, mean MSE): it must launch its own pass, not read the twin of the first."""
  nlat, nlon, m = 19, 36, 6
  rng = np.random.default_rng(4)
  valid = rng.random((nlat, nlon)) > 0.3
  p, t1, pv, tv1, lat, lon = make_case('lon_fastest', m, nlat, nlon, 2, seed=8, mask=valid)
  tv2 = (tv1 + rng.normal(size=tv1.shape) * 3 + 5).astype(np.float32)
  t2 = xr.DataArray(tv2, dims=t1.dims, coords={k: t1.coords[k].values for k in t1.dims})
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(REGIONS)], masked=True)
  stats = lane_statistics()
  state1, log1 = run(stats, agg, p, t1)
  assert [(e['kind'], e['flags'] & 1) for e in log1] == [('ens_binned', 1)], log1
  check_against_oracle(state1, stats, pv, tv1, 'lon_fastest', lat, lon, None, ['latitude', 'longitude'], mask=valid)
  only_t = {k: stats[k] for k in ('CRPSSkill', 'UnbiasedEnsembleMeanSquaredError', 'EnsembleMeanSquaredError')}
  state2, log2 = run(only_t, agg, p, t2)  # the same predictions OBJECT
  assert [(e['kind'], e['flags'] & 1) for e in log2] == [('ens_binned', 0)], log2  # its own launch
  pd, td = LAYOUTS['lon_fastest']
  names, masks = O.region_masks(lat, lon, REGIONS)
  w = (O.grid_area_weights(lat), ('latitude',))
  for name, (lane, ldims) in oracle_lanes(pv, pd, tv2, td).items():
    if name not in only_t:
      continue
    sws, _, out_dims = O.aggregate(lane, ldims, ['latitude', 'longitude'], weights=[w],
                                   bin_masks=[('region', masks, ('region', 'latitude', 'longitude'))])
    got = np.asarray(state2.sum_weighted_statistics[stats[name].unique_name]['v'].transpose(*out_dims).values)
    np.testing.assert_allclose(got, sws, rtol=RTOL, err_msg=name)
  # ... while spread / variance of these predictions still come from the twin of the first launch: no launch at all
  mo = {k: stats[k] for k in ('CRPSSpread', 'EnsembleVariance')}
  state3, log3 = run(mo, agg, p, t2)
  assert [e['kind'] for e in log3 if e['kind'].startswith('ens')] in ([], ['ens_binned']), log3
  for name in mo:
    lane, ldims = oracle_lanes(pv, pd, tv2, td)[name]
    sws, _, out_dims = O.aggregate(lane, ldims, ['latitude', 'longitude'], weights=[w],
                                   bin_masks=[('region', masks, ('region', 'latitude', 'longitude'))])
    got = np.asarray(state3.sum_weighted_statistics[stats[name].unique_name]['v'].transpose(*out_dims).values)
    np.testing.assert_allclose(got, sws, rtol=RTOL, err_msg=name)
