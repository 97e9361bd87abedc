"""Randomised planner / Aggregator checks against the oracle: random dim orders, sizes, reduce sets, weights on
random dims, boolean bins on random dims, masks and skipna -- every combination the two-stage reduction has to
map onto (key, depth, x).  Runs on the NumPy plan interpreter (CPU) and on the HIP library (GPU)."""
import os

import numpy as np
import pytest

from oracle import wbx_oracle as O
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import binning
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.metrics import deterministic

# more seeds per family for a soak run: WBX_FUZZ_SCALE=10 python -m pytest tests/test_fuzz.py [-m gpu]
FUZZ_SCALE = int(os.environ.get('WBX_FUZZ_SCALE', '1'))

ALL_DIMS = ['init_time', 'lead_time', 'level', 'latitude', 'longitude', 'tile']


class VectorWeighting(weighting.Weighting):
  """Weights along one arbitrary dim (a user-defined Weighting plugin)."""

  def __init__(self, dim, values):
    self.dim, self.values = dim, values

  def weights(self, statistic):
    if self.dim not in statistic.dims:
      return xr.DataArray(1.0)
    return xr.DataArray(self.values, dims=(self.dim,))


class RandomBins(binning.Binning):
  """Boolean bins over one or two arbitrary dims (a user-defined Binning plugin)."""

  def __init__(self, name, dims, mask):
    super().__init__(name)
    self.dims, self.mask = dims, mask

  def create_bin_mask(self, statistic):
    return xr.DataArray(self.mask, dims=(self.bin_dim_name,) + tuple(self.dims))


@pytest.mark.parametrize('seed', range(80 * FUZZ_SCALE))
def test_random_layouts_and_aggregators(backend, seed):
  rng = np.random.default_rng(1000 + seed)
  ndim = int(rng.integers(1, 6))
  dims = list(rng.permutation(ALL_DIMS)[:ndim])
  sizes = {d: int(rng.integers(1, 7)) for d in dims}
  if rng.random() < 0.5:
    sizes[dims[-1]] = int(rng.choice([4, 8, 64, 65]))  # exercise the 4-wide / multi-wave paths too
  shape = [sizes[d] for d in dims]
  dtype = np.float32 if rng.random() < 0.7 else np.float64
  p = xr.DataArray(rng.normal(size=shape).astype(dtype), dims=dims)
  # the target may miss some dims (broadcast) and be stored in another order
  tdims = [d for d in dims if rng.random() < 0.8] or dims[:1]
  tperm = list(rng.permutation(tdims))
  t = xr.DataArray(rng.normal(size=[sizes[d] for d in tperm]).astype(dtype), dims=tperm)
  reduce_dims = [d for d in dims if rng.random() < 0.5]
  weights, oracle_w = [], []
  for d in dims:
    if rng.random() < 0.3:
      v = rng.random(sizes[d]) + 0.5
      weights.append(VectorWeighting(d, v))
      oracle_w.append((v, (d,)))
  bins, oracle_b = [], []
  for k in range(int(rng.integers(0, 3))):
    bd = list(rng.permutation(dims)[:int(rng.integers(1, min(2, ndim) + 1))])
    nb = int(rng.choice([2, 3, 7]))
    mask = rng.random([nb] + [sizes[d] for d in bd]) > 0.4
    bins.append(RandomBins(f'bin{k}', bd, mask))
    oracle_b.append((f'bin{k}', mask, (f'bin{k}',) + tuple(bd)))
  mode = rng.choice(['plain', 'nan_plain', 'masked', 'skipna'])
  pv = p.values.copy()
  mask_arr = None
  if mode != 'plain' and pv.size:
    idx = tuple(int(rng.integers(0, s)) for s in shape)
    pv[idx] = np.nan
    p = xr.DataArray(pv, dims=dims)
  if mode == 'masked':
    mdims = [d for d in dims if rng.random() < 0.7] or dims[-1:]
    mask_arr = rng.random([sizes[d] for d in mdims]) > 0.2
    full = np.broadcast_to(O.expand_to(mask_arr, tuple(mdims), tuple(dims)), shape)
    mask_arr = mask_arr & ~np.isnan(np.where(full, pv, 0.0)).any(axis=tuple(i for i, d in enumerate(dims) if d not in mdims))
    p.coords['mask'] = xr.DataArray(mask_arr, dims=mdims)
  agg = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=weights or None, bin_by=bins or None,
                               masked=(mode == 'masked'), skipna=(mode == 'skipna'))
  metrics = {'mse': deterministic.MSE(), 'bias': deterministic.Bias()}
  stats = metrics_base.compute_unique_statistics_for_all_metrics(metrics, {'v': p}, {'v': t})
  state = agg.aggregate_statistics(stats)
  te = O.expand_to(t.values, tuple(t.dims), tuple(dims))
  okw = {}
  if mode == 'masked':
    okw = dict(mask=mask_arr, mask_dims=tuple(p.coords['mask'].dims))
  elif mode == 'skipna':
    okw = dict(skipna=True)
  for name, fn in (('SquaredError', O.squared_error), ('Error', O.error)):
    want = O.aggregate(fn(p.values, te), tuple(dims), reduce_dims, weights=oracle_w, bin_masks=oracle_b, **okw)
    got_s, got_w = state.sum_weighted_statistics[name].get('v'), state.sum_weights[name].get('v')
    assert want is not None and got_s is not None
    sws, sw, out_dims = want
    assert set(got_s.dims) == set(out_dims), (got_s.dims, out_dims)
    np.testing.assert_allclose(got_s.transpose(*out_dims).values, sws, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(got_w.transpose(*out_dims).values, sw, rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize('seed', range(40 * FUZZ_SCALE))
def test_random_ensemble_layouts_and_aggregators(backend, seed):
  """The ensemble family on random layouts: the member dim anywhere in the prediction's dim order, targets in another
  order, random reduce sets, vector weights, boolean bins, masks / skipna -- whatever route the planner picks (x summed,
  x kept, flat folded weights, membership bits) must reproduce the float64 restatement."""
  from weatherbenchx_amd.metrics import probabilistic
  rng = np.random.default_rng(5000 + seed)
  ndim = int(rng.integers(1, 5))
  dims = list(rng.permutation([d for d in ALL_DIMS if d != 'tile'])[:ndim])
  sizes = {d: int(rng.integers(1, 6)) for d in dims}
  if rng.random() < 0.5:
    sizes[dims[-1]] = int(rng.choice([4, 64, 65, 130]))
  m = int(rng.choice([2, 4, 5, 9]))
  pdims = list(dims)
  pdims.insert(int(rng.integers(0, ndim + 1)), 'number')
  psizes = dict(sizes, number=m)
  tperm = list(rng.permutation(dims))
  tv = rng.normal(size=[sizes[d] for d in tperm]).astype(np.float32)
  pv = rng.normal(size=[psizes[d] for d in pdims]).astype(np.float32)
  mode = rng.choice(['plain', 'masked', 'skipna'])
  if mode != 'plain' and tv.size > 1:
    tv.reshape(-1)[int(rng.integers(0, tv.size))] = np.nan
  t = xr.DataArray(tv, dims=tperm)
  p = xr.DataArray(pv, dims=pdims)
  mask_arr = None
  if mode == 'masked':
    mask_arr = ~np.isnan(tv) & (rng.random(tv.shape) > 0.2)
    t.coords['mask'] = xr.DataArray(mask_arr, dims=tperm)
  reduce_dims = [d for d in dims if rng.random() < 0.6]
  weights, oracle_w = [], []
  for d in dims:
    if rng.random() < 0.3:
      v = rng.random(sizes[d]) + 0.5
      weights.append(VectorWeighting(d, v))
      oracle_w.append((v, (d,)))
  bins, oracle_b = [], []
  if rng.random() < 0.5:
    bd = list(rng.permutation(dims)[:int(rng.integers(1, min(2, ndim) + 1))])
    nb = int(rng.choice([2, 6, 9]))
    bm = rng.random([nb] + [sizes[d] for d in bd]) > 0.4
    bins.append(RandomBins('bin0', bd, bm))
    oracle_b.append(('bin0', bm, ('bin0',) + tuple(bd)))
  use_sort = bool(rng.random() < 0.7)
  fair = bool(rng.random() < 0.7)
  stats = {'skill': probabilistic.CRPSSkill(), 'spread': probabilistic.CRPSSpread(use_sort=use_sort, fair=fair),
           'var': probabilistic.EnsembleVariance(), 'uemse': probabilistic.UnbiasedEnsembleMeanSquaredError()}
  agg = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=weights or None, bin_by=bins or None,
                               masked=(mode == 'masked'), skipna=(mode == 'skipna'))
  computed = {k: s.compute({'v': p}, {'v': t}) for k, s in stats.items()}
  state = agg.aggregate_statistics(computed)
  okw = {}
  if mode == 'masked':
    okw = dict(mask=mask_arr, mask_dims=tuple(tperm))
  elif mode == 'skipna':
    okw = dict(skipna=True)
  pd_, td = tuple(pdims), tuple(tperm)
  want = {'skill': O.crps_skill(pv, pd_, tv, td, 'number'), 'spread': O.crps_spread(pv, pd_, 'number', fair=fair, use_sort=use_sort),
          'var': O.ensemble_variance(pv, pd_, 'number'), 'uemse': O.unbiased_ensemble_mean_squared_error(pv, pd_, tv, td, 'number')}
  for k, (vals, vdims) in want.items():
    got_s, got_w = state.sum_weighted_statistics[k].get('v'), state.sum_weights[k].get('v')
    # spread / variance are statistics of the predictions alone: they carry the predictions' coordinates, so the
    # targets' `mask` does not reach them and Aggregator(masked=True) leaves them unmasked like the reference
    # (probabilistic.py:250-273 `predictions.var(...)`, aggregation.py:339-352 `'mask' in stat.coords`)
    kw_k = {} if (mode == 'masked' and k in ('spread', 'var')) else okw
    full_dims = O.union_dims(vdims, td) if 'mask' in kw_k else vdims
    vals_full = np.broadcast_to(O.expand_to(vals, vdims, full_dims), [sizes[d] for d in full_dims])
    ref = O.aggregate(vals_full, full_dims, reduce_dims, weights=oracle_w, bin_masks=oracle_b, **kw_k)
    if ref is None:
      assert got_s is None, k
      continue
    assert got_s is not None, k
    sws, sw, out_dims = ref
    assert set(got_s.dims) == set(out_dims), (k, got_s.dims, out_dims)
    np.testing.assert_allclose(got_s.transpose(*out_dims).values, sws, rtol=1e-6, atol=1e-9, err_msg=k)
    np.testing.assert_allclose(got_w.transpose(*out_dims).values, sw, rtol=1e-6, atol=1e-12, err_msg=k)


@pytest.mark.parametrize('seed', range(24 * FUZZ_SCALE))
def test_random_medium_ensemble_layouts(backend, seed):
  """The ensemble family at sizes and member counts where the register-sorting kernels run several tiles per block and the
  pipelined sweeps their look-ahead (2-64 members incl. the padded buckets 3 / 17 / 50 / 51, x up to 257 points, a few
  hundred rows, the member dim anywhere in the dim order, either spatial dim fastest): skill, spread (rank and pair form,
  fair or not), variance and unbiased ensemble-mean MSE against the float64 oracle."""
  from weatherbenchx_amd.metrics import probabilistic
  rng = np.random.default_rng(15000 + seed)
  ndim = int(rng.integers(2, 5))
  dims = list(rng.permutation([d for d in ALL_DIMS if d != 'tile'])[:ndim])
  sizes = {d: int(rng.integers(1, 4)) for d in dims}
  sizes[dims[-1]] = int(rng.choice([64, 70, 130, 257]))
  sizes[dims[int(rng.integers(0, ndim - 1))]] = int(rng.choice([7, 33, 90]))
  use_sort = bool(rng.random() < 0.7)
  m = int(rng.choice([2, 3, 8, 16, 17, 32, 50, 51, 64] if use_sort else [2, 3, 8, 16, 17]))
  pdims = list(dims)
  pdims.insert(int(rng.integers(0, ndim + 1)), 'number')
  psizes = dict(sizes, number=m)
  tperm = list(rng.permutation(dims))
  offset, spread = float(rng.choice([0.0, 280.0])), float(rng.choice([1.0, 30.0]))
  tv = (rng.normal(size=[sizes[d] for d in tperm]) * spread + offset).astype(np.float32)
  pv = (rng.normal(size=[psizes[d] for d in pdims]) * spread + offset).astype(np.float32)
  mode = rng.choice(['plain', 'plain', 'masked', 'skipna'])
  if mode != 'plain':
    tv[rng.random(tv.shape) < 0.02] = np.nan
  t = xr.DataArray(tv, dims=tperm)
  p = xr.DataArray(pv, dims=pdims)
  mask_arr = None
  if mode == 'masked':
    mask_arr = ~np.isnan(tv) & (rng.random(tv.shape) > 0.2)
    t.coords['mask'] = xr.DataArray(mask_arr, dims=tperm)
  reduce_dims = [d for d in dims if rng.random() < 0.6]
  weights, oracle_w = [], []
  for d in dims:
    if rng.random() < 0.4:
      v = rng.random(sizes[d]) + 0.5
      weights.append(VectorWeighting(d, v))
      oracle_w.append((v, (d,)))
  fair = bool(rng.random() < 0.7)
  stats = {'skill': probabilistic.CRPSSkill(), 'spread': probabilistic.CRPSSpread(use_sort=use_sort, fair=fair),
           'var': probabilistic.EnsembleVariance(), 'uemse': probabilistic.UnbiasedEnsembleMeanSquaredError()}
  agg = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=weights or None, masked=(mode == 'masked'),
                               skipna=(mode == 'skipna'))
  computed = {k: s.compute({'v': p}, {'v': t}) for k, s in stats.items()}
  state = agg.aggregate_statistics(computed)
  okw = {}
  if mode == 'masked':
    okw = dict(mask=mask_arr, mask_dims=tuple(tperm))
  elif mode == 'skipna':
    okw = dict(skipna=True)
  pd_, td = tuple(pdims), tuple(tperm)
  want = {'skill': O.crps_skill(pv, pd_, tv, td, 'number'), 'spread': O.crps_spread(pv, pd_, 'number', fair=fair, use_sort=use_sort),
          'var': O.ensemble_variance(pv, pd_, 'number'), 'uemse': O.unbiased_ensemble_mean_squared_error(pv, pd_, tv, td, 'number')}
  for k, (vals, vdims) in want.items():
    got_s, got_w = state.sum_weighted_statistics[k].get('v'), state.sum_weights[k].get('v')
    kw_k = {} if (mode == 'masked' and k in ('spread', 'var')) else okw  # (see the small-size test above)
    full_dims = O.union_dims(vdims, td) if 'mask' in kw_k else vdims
    vals_full = np.broadcast_to(O.expand_to(vals, vdims, full_dims), [sizes[d] for d in full_dims])
    ref = O.aggregate(vals_full, full_dims, reduce_dims, weights=oracle_w, **kw_k)
    if ref is None:
      assert got_s is None, k
      continue
    assert got_s is not None, k
    sws, sw, out_dims = ref
    assert set(got_s.dims) == set(out_dims), (k, got_s.dims, out_dims)
    # uemse = (mean - t)^2 - var / M is a difference of like-sized terms: 1e-6 of the terms, not of the difference
    atol = 1e-6 * float(np.nanmax(np.abs(sw))) * spread * spread if k == 'uemse' else 1e-9
    np.testing.assert_allclose(got_s.transpose(*out_dims).values, sws, rtol=1e-6, atol=atol, err_msg=k)
    np.testing.assert_allclose(got_w.transpose(*out_dims).values, sw, rtol=1e-6, atol=1e-12, err_msg=k)


@pytest.mark.parametrize('seed', range(30 * FUZZ_SCALE))
def test_random_indicator_layouts_and_aggregators(backend, seed):
  """ErrorExceedance / EnsembleErrorExceedance / RankHistogram on random layouts (member dim anywhere, targets in another
  order), random reduce sets, weights, masks / skipna, NaN members and NaN thresholds, against the oracle."""
  from weatherbenchx_amd.metrics import probabilistic
  rng = np.random.default_rng(9000 + seed)
  ndim = int(rng.integers(1, 4))
  dims = list(rng.permutation(['lead_time', 'level', 'latitude', 'longitude'])[:ndim])
  sizes = {d: int(rng.integers(1, 6)) for d in dims}
  if rng.random() < 0.5:
    sizes[dims[-1]] = int(rng.choice([64, 65, 130]))
  m = int(rng.choice([2, 3, 5, 8]))
  pdims = list(dims)
  pdims.insert(int(rng.integers(0, ndim + 1)), 'number')
  psizes = dict(sizes, number=m)
  tperm = list(rng.permutation(dims))
  tv = rng.normal(size=[sizes[d] for d in tperm]).astype(np.float32)
  pv = (rng.normal(size=[psizes[d] for d in pdims]) * 1.5).astype(np.float32)
  if pv.size > 4:
    pv.reshape(-1)[int(rng.integers(0, pv.size))] = np.nan  # a NaN member somewhere: skipped by the member mean
  mode = rng.choice(['plain', 'masked', 'skipna'])
  if mode != 'plain' and tv.size > 1:
    tv.reshape(-1)[int(rng.integers(0, tv.size))] = np.nan
  t = xr.DataArray(tv, dims=tperm)
  p = xr.DataArray(pv, dims=pdims)
  mask_arr = None
  if mode == 'masked':
    mask_arr = ~np.isnan(tv) & (rng.random(tv.shape) > 0.2)
    t.coords['mask'] = xr.DataArray(mask_arr, dims=tperm)
  reduce_dims = [d for d in dims if rng.random() < 0.6]
  weights, oracle_w = [], []
  for d in dims:
    if rng.random() < 0.3:
      v = rng.random(sizes[d]) + 0.5
      weights.append(VectorWeighting(d, v))
      oracle_w.append((v, (d,)))
  thresholds = [0.3, float('nan'), 1.5] if rng.random() < 0.5 else [0.1, 0.7, 1.2, 2.5, 4.0]
  stats = {'exc': probabilistic.EnsembleErrorExceedance(thresholds), 'rank': probabilistic.RankHistogram()}
  agg = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=weights or None, masked=(mode == 'masked'),
                               skipna=(mode == 'skipna'))
  state = agg.aggregate_statistics({k: s.compute({'v': p}, {'v': t}) for k, s in stats.items()})
  okw = {}
  if mode == 'masked':
    okw = dict(mask=mask_arr, mask_dims=tuple(tperm))
  elif mode == 'skipna':
    okw = dict(skipna=True)
  pd_, td = tuple(pdims), tuple(tperm)
  with np.errstate(invalid='ignore'):
    want = {'exc': O.ensemble_error_exceedance(pv, pd_, tv, td, thresholds, 'number'),
            'rank': O.rank_histogram(pv, pd_, tv, td, 'number')}
  for k, (vals, vdims) in want.items():
    got_s, got_w = state.sum_weighted_statistics[k].get('v'), state.sum_weights[k].get('v')
    ref = O.aggregate(vals, vdims, reduce_dims, weights=oracle_w, **okw)
    assert ref is not None and got_s is not None, k
    sws, sw, out_dims = ref
    assert set(got_s.dims) == set(out_dims), (k, got_s.dims, out_dims)
    np.testing.assert_allclose(got_s.transpose(*out_dims).values, sws, rtol=1e-6, atol=1e-9, err_msg=k)
    np.testing.assert_allclose(got_w.transpose(*out_dims).values, sw, rtol=1e-6, atol=1e-12, err_msg=k)


@pytest.mark.parametrize('seed', range(40 * FUZZ_SCALE))
def test_random_layouts_through_every_bin_route(monkeypatch, seed):
  """Host logic only (NumPy plan interpreter): with >= 5 boolean bins the engine may contract membership bits in stage 2
  or run the one-pass binned route; forcing either must give the oracle's sums on random layouts, reduce sets, weights
  and masks (which dims end up as key / depth / x differs from case to case)."""
  import fake_device
  from weatherbenchx_amd import engine
  fake_device.install(monkeypatch)
  rng = np.random.default_rng(7000 + seed)
  ndim = int(rng.integers(2, 6))
  dims = list(rng.permutation(ALL_DIMS)[:ndim])
  sizes = {d: int(rng.integers(1, 7)) for d in dims}
  sizes[dims[-1]] = int(rng.choice([3, 8, 64, 70]))
  shape = [sizes[d] for d in dims]
  pv = rng.normal(size=shape).astype(np.float32)
  tperm = list(rng.permutation(dims))
  tv = rng.normal(size=[sizes[d] for d in tperm]).astype(np.float32)
  mode = rng.choice(['plain', 'masked', 'skipna'])
  if mode != 'plain':
    pv[tuple(int(rng.integers(0, s)) for s in shape)] = np.nan
  p = xr.DataArray(pv, dims=dims)
  t = xr.DataArray(tv, dims=tperm)
  mask_arr = None
  if mode == 'masked':
    mask_arr = ~np.isnan(pv) & (rng.random(shape) > 0.2)
    p.coords['mask'] = xr.DataArray(mask_arr, dims=dims)
  bd = list(rng.permutation(dims)[:int(rng.integers(1, 3))])
  nb = int(rng.choice([5, 9, 34]))
  bm = rng.random([nb] + [sizes[d] for d in bd]) > 0.5
  reduce_dims = sorted(set(bd) | {d for d in dims if rng.random() < 0.5}, key=dims.index)
  weights, oracle_w = [], []
  for d in dims:
    if rng.random() < 0.4:
      v = rng.random(sizes[d]) + 0.5
      weights.append(VectorWeighting(d, v))
      oracle_w.append((v, (d,)))
  te = O.expand_to(tv, tuple(tperm), tuple(dims))
  okw = {}
  if mode == 'masked':
    okw = dict(mask=mask_arr, mask_dims=tuple(dims))
  elif mode == 'skipna':
    okw = dict(skipna=True)
  want = O.aggregate(O.squared_error(pv, te), tuple(dims), reduce_dims, weights=oracle_w,
                     bin_masks=[('bin0', bm, ('bin0',) + tuple(bd))], **okw)
  sws, sw, out_dims = want
  for route in ('never', 'always'):
    monkeypatch.setattr(engine, 'BINNED_MODE', route)
    engine.clear_caches()
    agg = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=weights or None, bin_by=[RandomBins('bin0', bd, bm)],
                                 masked=(mode == 'masked'), skipna=(mode == 'skipna'))
    stats = metrics_base.compute_unique_statistics_for_all_metrics({'mse': deterministic.MSE()}, {'v': p}, {'v': t})
    state = agg.aggregate_statistics(stats)
    got_s, got_w = state.sum_weighted_statistics['SquaredError']['v'], state.sum_weights['SquaredError']['v']
    np.testing.assert_allclose(got_s.transpose(*out_dims).values, sws, rtol=1e-6, atol=1e-9, err_msg=route)
    np.testing.assert_allclose(got_w.transpose(*out_dims).values, sw, rtol=1e-6, atol=1e-12, err_msg=route)


@pytest.mark.parametrize('seed', range(36 * FUZZ_SCALE))
def test_random_medium_layouts_through_every_bin_route(backend, monkeypatch, seed):
  """The same on BOTH backends and at sizes where the one-pass binned kernels run several x tiles, several 64-row batches and
  row splits per patch (x up to 257 points -- whole and ragged 128-byte lines --, up to ~600 reduced rows, a depth dim under
  the bin dims or not), with few distinct membership words per patch (band-like bins: the atom kernel) or many (random
  bins: the slot kernel), DET3 lanes (MSE + bias + MAE) and float32 / float64 inputs: forced binned route, forced two-stage
  route, float64 oracle."""
  from weatherbenchx_amd import engine
  rng = np.random.default_rng(13000 + seed)
  ndim = int(rng.integers(2, 5))
  dims = list(rng.permutation(ALL_DIMS)[:ndim])
  sizes = {d: int(rng.integers(1, 6)) for d in dims}
  sizes[dims[-1]] = int(rng.choice([3, 64, 70, 96, 130, 200, 257]))
  sizes[dims[int(rng.integers(0, ndim - 1))]] = int(rng.choice([7, 33, 64, 90, 150]))
  shape = [sizes[d] for d in dims]
  dtype = np.float32 if rng.random() < 0.75 else np.float64
  pv = rng.normal(size=shape).astype(dtype)
  tperm = list(rng.permutation(dims))
  tv = rng.normal(size=[sizes[d] for d in tperm]).astype(dtype)
  mode = rng.choice(['plain', 'masked', 'skipna'])
  if mode != 'plain':
    pv[rng.random(shape) < 0.02] = np.nan
  p = xr.DataArray(pv, dims=dims)
  t = xr.DataArray(tv, dims=tperm)
  mask_arr = None
  if mode == 'masked':
    mask_arr = ~np.isnan(pv) & (rng.random(shape) > 0.2)
    p.coords['mask'] = xr.DataArray(mask_arr, dims=dims)
  bd = list(rng.permutation(dims)[:int(rng.integers(1, 3))])
  nb = int(rng.choice([5, 9, 34, 40, 64]))
  bshape = [sizes[d] for d in bd]
  if rng.random() < 0.6:  # bands along the first bin dim, one global bin, a two-valued pattern on the rest: few words per patch
    first = (np.arange(bshape[0]) * nb // max(bshape[0], 1))[None, :] == np.arange(nb)[:, None]
    bm = np.broadcast_to(first.reshape([nb, bshape[0]] + [1] * (len(bd) - 1)), [nb] + bshape).copy()
    bm[0] = True
    if len(bd) > 1:
      bm[1:] &= (rng.random(bshape) > 0.3)
  else:
    bm = rng.random([nb] + bshape) > 0.5
  reduce_dims = sorted(set(bd) | {d for d in dims if rng.random() < 0.4}, key=dims.index)
  weights, oracle_w = [], []
  for d in dims:
    if rng.random() < 0.4:
      v = rng.random(sizes[d]) + 0.5
      weights.append(VectorWeighting(d, v))
      oracle_w.append((v, (d,)))
  te = O.expand_to(tv, tuple(tperm), tuple(dims))
  okw = {}
  if mode == 'masked':
    okw = dict(mask=mask_arr, mask_dims=tuple(dims))
  elif mode == 'skipna':
    okw = dict(skipna=True)
  metrics = {'mse': deterministic.MSE(), 'bias': deterministic.Bias(), 'mae': deterministic.MAE()}
  pd64, td64 = pv.astype(np.float64), te.astype(np.float64)
  lanes = {'SquaredError': O.squared_error(pv, te), 'Error': pd64 - td64, 'AbsoluteError': np.abs(pd64 - td64)}
  for route in ('always', 'never'):
    monkeypatch.setattr(engine, 'BINNED_MODE', route)
    engine.clear_caches()
    agg = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=weights or None, bin_by=[RandomBins('bin0', bd, bm)],
                                 masked=(mode == 'masked'), skipna=(mode == 'skipna'))
    stats = metrics_base.compute_unique_statistics_for_all_metrics(metrics, {'v': p}, {'v': t})
    state = agg.aggregate_statistics(stats)
    for name, vals in lanes.items():
      sws, sw, out_dims = O.aggregate(vals, tuple(dims), reduce_dims, weights=oracle_w,
                                      bin_masks=[('bin0', bm, ('bin0',) + tuple(bd))], **okw)
      got_s, got_w = state.sum_weighted_statistics[name]['v'], state.sum_weights[name]['v']
      scale = np.nanmax(np.abs(sws)) if np.isfinite(sws).any() else 1.0
      np.testing.assert_allclose(got_s.transpose(*out_dims).values, sws, rtol=1e-6, atol=1e-9 * max(scale, 1.0),
                                 err_msg=f'{route} {name}')
      np.testing.assert_allclose(got_w.transpose(*out_dims).values, sw, rtol=1e-6, atol=1e-12, err_msg=f'{route} {name}')


@pytest.mark.parametrize('seed', range(25 * FUZZ_SCALE))
def test_random_spectrum_frames(monkeypatch, seed):
  """Host logic only: longitude anywhere in the dim order, random kept dims, vector weights on row dims -- the row
  weights / group ids handed to the spectrum reduction must reproduce the weighted mean of the per-row spectra."""
  import fake_device
  from weatherbenchx_amd import spectra
  fake_device.install(monkeypatch)
  rng = np.random.default_rng(11000 + seed)
  row_dims = list(rng.permutation(['lead_time', 'level', 'latitude'])[:int(rng.integers(1, 4))])
  dims = list(row_dims)
  dims.insert(int(rng.integers(0, len(dims) + 1)), 'longitude')
  sizes = {d: int(rng.integers(1, 5)) for d in row_dims}
  sizes['longitude'] = int(rng.choice([8, 12, 30]))
  vals = rng.normal(size=[sizes[d] for d in dims]).astype(np.float32)
  coords = {'longitude': np.arange(sizes['longitude']) * (360.0 / sizes['longitude'])}
  if 'latitude' in sizes:
    coords['latitude'] = np.linspace(-60, 60, sizes['latitude']) if sizes['latitude'] > 1 else np.array([10.0])
  f = xr.DataArray(vals, dims=dims, coords=coords)
  reduce_dims = [d for d in row_dims if rng.random() < 0.6] or row_dims[:1]
  weights, oracle_w = [], []
  for d in row_dims:
    if rng.random() < 0.4:
      v = rng.random(sizes[d]) + 0.5
      weights.append(VectorWeighting(d, v))
      oracle_w.append((v, (d,)))
  stat = spectra.ZonalPowerSpectrum().compute({'v': f}, {'v': f})['v']
  state = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=weights or None).aggregate_stat_var(stat)
  lon_ax = dims.index('longitude')
  per_row = np.moveaxis(O.zonal_power_spectrum(vals, lon_axis=lon_ax), lon_ax, -1)  # rows in `dims` order, then k
  rd = tuple(d for d in dims if d != 'longitude') + ('zonal_wavenumber',)
  sws, sw, out_dims = O.aggregate(per_row, rd, reduce_dims, weights=oracle_w)
  np.testing.assert_allclose(state.sum_weighted_statistics.transpose(*out_dims).values, sws, rtol=1e-6, atol=1e-9)
  np.testing.assert_allclose(state.sum_weights.transpose(*out_dims).values, sw, rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize('seed', range(16 * FUZZ_SCALE))
def test_random_ensemble_grids_under_region_bins(backend, seed):
  """The one-pass binned ensemble kernel (wbx_ens_binned) on random grids: latitude / longitude counts that are and are not
  multiples of 32 and 64 (whole-line rows on one-wave blocks, ragged rows on four-wave blocks), enough rows for the tapered
  row splits and too few for them, 2-51 members, the three recorded dim orders, with and without a (latitude, longitude) mask
  coordinate (twin atoms), random land / sea patterns: every bin of all five lanes against the float64 oracle, one launch."""
  import test_ens_binned as EB
  from weatherbenchx_amd import binning, weighting
  rng = np.random.default_rng(31000 + seed)
  layout = str(rng.choice(sorted(EB.LAYOUTS)))
  m = int(rng.choice([2, 5, 16, 33, 50, 51]))
  nlat = int(rng.choice([19, 33, 64, 91, 181]))
  nlon = int(rng.choice([36, 64, 90, 128, 145, 256]))
  nlead = int(rng.integers(1, 4))
  land = rng.random((nlat, nlon)) > rng.uniform(0.2, 0.8)
  with_mask = bool(rng.random() < 0.5)
  valid = (rng.random((nlat, nlon)) > 0.25) if with_mask else None
  p, t, pv, tv, lat, lon = EB.make_case(layout, m, nlat, nlon, nlead, seed=seed, mask=valid, ninit=int(rng.integers(1, 3)))
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  reduce_dims = ['latitude', 'longitude'] + (['init_time'] if layout == 'ifs' else [])
  agg = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(EB.REGIONS, land_sea_mask=lsm)], masked=True)
  stats = EB.lane_statistics()
  state, log = EB.run(stats, agg, p, t)
  assert [e['kind'] for e in log] == ['ens_binned'], (layout, m, nlat, nlon, log)
  EB.check_against_oracle(state, stats, pv, tv, layout, lat, lon, land, reduce_dims, mask=valid)


@pytest.mark.parametrize('seed', range(16 * FUZZ_SCALE))
def test_random_ensemble_grids_under_nan_masks_and_skipna(backend, seed):
  """Round 5's widened one-pass route on random grids: a NaN mask over every dim of the targets (add_nan_mask_to_data,
  data_loaders/base.py:25-56: another hole pattern per init / lead) under Aggregator(masked=True), or Aggregator(skipna=True) with
  NaN targets and NaN members (optionally with a (latitude, longitude) mask coordinate on top): ragged and whole-line rows, 2-51
  members, the three recorded dim orders -- every bin of all five lanes and of their weights, ONE wbx_ens_binned launch."""
  import test_ens_binned as EB
  from weatherbenchx_amd import binning, weighting
  from weatherbenchx_amd import data as wdata
  rng = np.random.default_rng(47000 + seed)
  layout = str(rng.choice(sorted(EB.LAYOUTS)))
  m = int(rng.choice([2, 5, 16, 33, 50, 51]))
  nlat = int(rng.choice([19, 33, 64, 91]))
  nlon = int(rng.choice([36, 64, 90, 128, 145]))
  nlead = int(rng.integers(1, 4))
  land = rng.random((nlat, nlon)) > rng.uniform(0.2, 0.8)
  mode = ['nanmask', 'skipna', 'skipna+mask'][seed % 3]
  if layout == 'ifs' and mode != 'nanmask':
    layout = 'lon_fastest'  # (the skipna route keeps init_time: the IFS case reduces it)
  valid = (rng.random((nlat, nlon)) > 0.25) if mode == 'skipna+mask' else None
  p, t, pv, tv, lat, lon = EB.make_case(layout, m, nlat, nlon, nlead, seed=seed, mask=valid, ninit=int(rng.integers(1, 3)))
  tv[rng.random(tv.shape) < rng.uniform(0.02, 0.3)] = np.nan
  pd, td = EB.LAYOUTS[layout]
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  reduce_dims = ['latitude', 'longitude'] + (['init_time'] if layout == 'ifs' else [])
  stats = EB.lane_statistics()
  if mode == 'nanmask':
    t = wdata.add_nan_mask_to_data({'v': xr.DataArray(tv, dims=td, coords={k: t.coords[k].values for k in td})})['v']
    agg = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=[weighting.GridAreaWeighting()],
                                 bin_by=[binning.Regions(EB.REGIONS, land_sea_mask=lsm)], masked=True)
    state, log = EB.run(stats, agg, p, t)
    assert [(e['kind'], e['flags'] & 1) for e in log] == [('ens_binned', 1)], (layout, m, nlat, nlon, log)
    EB._check_lanes(state, stats, pv, tv, layout, lat, lon, land, reduce_dims, mask=~np.isnan(tv), mask_dims=td)  # pylint: disable=protected-access
    return
  pv[rng.random(pv.shape) < 0.01] = np.nan
  p = xr.DataArray(pv, dims=pd, coords={k: p.coords[k].values for k in pd if k != 'number'})
  t2 = xr.DataArray(tv, dims=td, coords={k: t.coords[k].values for k in td})
  t = t2.assign_coords(mask=t.coords['mask']) if valid is not None else t2
  agg = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(EB.REGIONS, land_sea_mask=lsm)], masked=valid is not None, skipna=True)
  state, log = EB.run(stats, agg, p, t)
  assert [e['kind'] for e in log] == ['ens_binned'], (layout, m, nlat, nlon, log)
  EB._check_lanes(state, stats, pv, tv, layout, lat, lon, land, reduce_dims, mask=valid, mask_dims=('latitude', 'longitude'), skipna=True)  # pylint: disable=protected-access
