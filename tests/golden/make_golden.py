#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from the float64 oracle.

The reference holds no stored arrays for this path (its fixtures are generated in-test) and cannot
be imported here, so the vectors come from oracle/wbx_oracle.py AFTER it passed the restated reference tests
(tests/test_oracle_reference_pins.py).  Inputs are stored too (float32), so the fixtures do not depend on the
NumPy RNG implementation.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import wbx_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def config1():
  """BASELINE.json configs[0]: RMSE + bias, 64x32 grid, 2 variables, 3 lead times (2 inits), area weights,
  regions {global, northern-hemisphere} (evaluation_scripts/run_example_evaluation.py:172-185)."""
  lat = np.linspace(-87.1875, 87.1875, 32)
  lon = np.arange(64) * 5.625
  rp, rt = np.random.default_rng(0), np.random.default_rng(1)
  dims3 = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
  dims2 = ('init_time', 'lead_time', 'latitude', 'longitude')
  data = {}
  for name, dims, shape in (('2m_temperature', dims2, (2, 3, 32, 64)), ('geopotential', dims3, (2, 3, 3, 32, 64))):
    data[name] = (dims, rp.normal(size=shape).astype(np.float32), rt.normal(size=shape).astype(np.float32))
  w = (O.grid_area_weights(lat), ('latitude',))
  regions = {'global': ((-90, 90), (0, 360)), 'northern-hemisphere': ((20, 90), (0, 360))}
  _, masks = O.region_masks(lat, lon, regions)
  bins = ('region', masks, ('region', 'latitude', 'longitude'))
  out = {'latitude': lat, 'longitude': lon}
  for name, (dims, p, t) in data.items():
    out[f'{name}__p'], out[f'{name}__t'] = p, t
    for stat, fn in (('Error', O.error), ('SquaredError', O.squared_error)):
      sws, sw, od = O.aggregate(fn(p, t), dims, ['init_time', 'latitude', 'longitude'], weights=[w], bin_masks=[bins])
      out[f'{name}__{stat}__sws'], out[f'{name}__{stat}__sw'] = sws, sw
    out[f'{name}__rmse'] = O.rmse(out[f'{name}__SquaredError__sws'] / out[f'{name}__SquaredError__sw'])
    out[f'{name}__bias'] = out[f'{name}__Error__sws'] / out[f'{name}__Error__sw']
  np.savez_compressed(os.path.join(HERE, 'config1_rmse_bias.npz'), **out)


def ensembles():
  """19x36 mock grid, M = 4 and 5 (weatherbenchX/metrics/metrics_test.py:610-660 sizes), every ensemble lane."""
  lat = np.linspace(-90, 90, 19)
  out = {'latitude': lat}
  for m in (4, 5):
    rng = np.random.default_rng(100 + m)
    t = (rng.normal(size=(2, 19, 36)) + 280).astype(np.float32)
    p = (t[:, None] + rng.normal(size=(2, m, 19, 36))).astype(np.float32)
    pd, td = ('time', 'realization', 'latitude', 'longitude'), ('time', 'latitude', 'longitude')
    out[f'm{m}__p'], out[f'm{m}__t'] = p, t
    w = (O.grid_area_weights(lat), ('latitude',))
    lanes = {
        'CRPSSkill': O.crps_skill(p, pd, t, td, 'realization')[0],
        'CRPSSpread_fair': O.crps_spread(p, pd, 'realization', fair=True, use_sort=True)[0],
        'CRPSSpread_unfair': O.crps_spread(p, pd, 'realization', fair=False, use_sort=False)[0],
        'EnsembleVariance': O.ensemble_variance(p, pd, 'realization')[0],
        'UnbiasedEnsembleMeanSquaredError': O.unbiased_ensemble_mean_squared_error(p, pd, t, td, 'realization')[0],
        'EnsembleMeanSquaredError': O.ensemble_mean_squared_error(p, pd, t, td, 'realization')[0],
    }
    for k, v in lanes.items():
      sws, sw, _ = O.aggregate(v, td, ['latitude', 'longitude'], weights=[w])
      out[f'm{m}__{k}__mean'] = sws / sw
      out[f'm{m}__{k}__point'] = v[0, 3, :5]
  np.savez_compressed(os.path.join(HERE, 'ensemble_19x36.npz'), **out)


def skipna_ensembles():
  """skipna_ensemble=True with NaNs scattered over members (per-point member counts, probabilistic.py:206-216, :304-314)
  and the CRPS ensemble distance against ensemble-valued targets (probabilistic.py:691-782), 19x36 grid, M = 6 / 4."""
  lat = np.linspace(-90, 90, 19)
  rng = np.random.default_rng(321)
  t = rng.normal(size=(2, 19, 36)).astype(np.float32)
  p = (t[:, None] + rng.normal(size=(2, 6, 19, 36))).astype(np.float32)
  p[rng.random(p.shape) < 0.15] = np.nan
  p[1, :, 3, 4] = np.nan   # no member left
  p[1, 1:, 5, 6] = np.nan  # a single member left
  te = (t[:, None] + rng.normal(size=(2, 4, 19, 36))).astype(np.float32)  # ensemble-valued targets
  q = np.where(np.isnan(p), 0.0, p).astype(np.float32)                     # NaN-free predictions for the distance
  pd, td = ('time', 'realization', 'latitude', 'longitude'), ('time', 'latitude', 'longitude')
  w = (O.grid_area_weights(lat), ('latitude',))
  out = {'latitude': lat, 'p': p, 't': t, 'te': te, 'q': q}
  with np.errstate(invalid='ignore', divide='ignore'):
    lanes = {'CRPSSkill': O.crps_skill(p, pd, t, td, 'realization', skipna_ensemble=True)[0],
             'CRPSSpread_fair': O.crps_spread(p, pd, 'realization', fair=True, skipna_ensemble=True)[0],
             'CRPSSpread_unfair': O.crps_spread(p, pd, 'realization', fair=False, skipna_ensemble=True)[0],
             'EnsembleVariance': O.ensemble_variance(p, pd, 'realization', skipna_ensemble=True)[0],
             'UnbiasedEnsembleMeanSquaredError': O.unbiased_ensemble_mean_squared_error(p, pd, t, td, 'realization',
                                                                                        skipna_ensemble=True)[0]}
    for k, v in lanes.items():
      sws, sw, _ = O.aggregate(v, td, ['latitude', 'longitude'], weights=[w], skipna=True)
      out[f'skipna__{k}__sws'], out[f'skipna__{k}__sw'] = sws, sw
  mean = lambda v: (lambda r: r[0] / r[1])(O.aggregate(v, td, ['latitude', 'longitude'], weights=[w]))
  for fair in (True, False):
    out[f'distance__fair{int(fair)}'] = O.crps_ensemble_distance(
        mean(O.crps_skill(q, pd, te, pd, 'realization')[0]), mean(O.crps_spread(q, pd, 'realization', fair=fair)[0]),
        mean(O.crps_spread(te, pd, 'realization', fair=fair)[0]))
  np.savez_compressed(os.path.join(HERE, 'skipna_ensemble_19x36.npz'), **out)


def regions_and_indicators():
  """32x64 grid, 7 regions x land/sea = 14 bins (the public benchmark's binning, run_benchmark_evaluation.py:369-382,
  at test size) with NaN targets under masked=True; ErrorExceedance / EnsembleErrorExceedance / RankHistogram
  (deterministic.py:262-295, probabilistic.py:836-861, 1306-1343) for a 5-member ensemble."""
  lat = np.linspace(-87.1875, 87.1875, 32)
  lon = np.arange(64) * 5.625
  rng = np.random.default_rng(77)
  regions = {'global': ((-90, 90), (0, 360)), 'tropics': ((-20, 20), (0, 360)), 'nh': ((20, 90), (0, 360)),
             'sh': ((-90, -20), (0, 360)), 'europe': ((35, 75), (-12.5, 42.5)), 'namerica': ((25, 60), (240, 285)),
             'ausnz': ((-45, -12.5), (120, 175))}
  land = rng.random((32, 64)) > 0.6
  names, masks = O.region_masks(lat, lon, regions, land_sea_mask=land)
  dims = ('lead_time', 'latitude', 'longitude')
  p = (rng.normal(size=(3, 32, 64)) + 280).astype(np.float32)
  t = (rng.normal(size=(3, 32, 64)) + 280).astype(np.float32)
  t[rng.random(t.shape) < 0.08] = np.nan
  valid = ~np.isnan(t)
  w = (O.grid_area_weights(lat), ('latitude',))
  bm = [('region', masks, ('region', 'latitude', 'longitude'))]
  out = {'latitude': lat, 'longitude': lon, 'land': land, 'region_names': np.array(names), 'p': p, 't': t,
         'region_lims': np.array([[v[0][0], v[0][1], v[1][0], v[1][1]] for v in regions.values()]),
         'region_keys': np.array(list(regions))}
  for stat, fn in (('SquaredError', O.squared_error), ('AbsoluteError', O.absolute_error)):
    sws, sw, od = O.aggregate(fn(p, t), dims, ['latitude', 'longitude'], weights=[w], bin_masks=bm, mask=valid,
                              mask_dims=dims)
    out[f'{stat}__sws'], out[f'{stat}__sw'] = sws, sw
  m = 5
  pe = (np.nan_to_num(t, nan=280.0)[:, None] + rng.normal(size=(3, m, 32, 64)) * 1.5).astype(np.float32)
  pd = ('lead_time', 'number', 'latitude', 'longitude')
  thresholds = np.array([0.5, 1.0, 2.5])
  out['pe'], out['thresholds'] = pe, thresholds
  for name, (stat, sd) in (('EnsembleErrorExceedance', O.ensemble_error_exceedance(pe, pd, t, dims, thresholds, 'number')),
                           ('RankHistogram', O.rank_histogram(pe, pd, t, dims, 'number')),
                           ('ErrorExceedance', O.error_exceedance(pe[:, 0], dims, t, dims, thresholds))):
    sws, sw, od = O.aggregate(stat, sd, ['latitude', 'longitude'], weights=[w], skipna=True)
    out[f'{name}__sws'], out[f'{name}__sw'] = sws, sw
  np.savez_compressed(os.path.join(HERE, 'regions_indicators.npz'), **out)


def weights_and_spectrum():
  out = {'w721': O.grid_area_weights(np.linspace(-90, 90, 721)),
         'w721_desc_unnorm': O.grid_area_weights(np.linspace(90, -90, 721), normalized=False)}
  rng = np.random.default_rng(4)
  f = rng.normal(size=(16, 32)).astype(np.float32)
  out['spec_field'] = f
  out['spec_power'] = O.zonal_power_spectrum(f)
  np.savez_compressed(os.path.join(HERE, 'weights_spectrum.npz'), **out)


if __name__ == '__main__':
  config1()
  ensembles()
  weights_and_spectrum()
  regions_and_indicators()
  skipna_ensembles()
  print('golden vectors written to', HERE)
