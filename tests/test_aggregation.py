"""Aggregator / AggregationState behaviour, restating weatherbenchX/aggregation_test.py:69-270 against the
drop-in classes.  Runs on the NumPy plan interpreter here and on the HIP library on the GPU box."""
import numpy as np
import pytest

import mock_data
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import binning
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd import xarray_tree
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.metrics import deterministic


def _test_data():
  template = mock_data.mock_prediction_data(time_start='2020-01-01T00', time_stop='2020-01-03T00',
                                            lead_start_days=0, lead_stop_days=1).rename(
                                                {'time': 'init_time', 'prediction_timedelta': 'lead_time'})
  return xr.zeros_like(template), xr.ones_like(template)


def _aggregate(metrics, predictions, targets, **kw):
  stats = metrics_base.compute_unique_statistics_for_all_metrics(metrics, predictions, targets)
  return aggregation.Aggregator(**kw).aggregate_statistics(stats)


from weatherbenchx_amd.data import add_nan_mask_to_data as add_nan_mask  # data_loaders/base.py:25-56


def test_expected_output(backend):
  predictions, targets = _test_data()
  metrics = {'rmse': deterministic.RMSE()}
  state = _aggregate(metrics, predictions, targets, reduce_dims=['init_time', 'latitude', 'longitude'])
  actual = state.metric_values(metrics)
  summed = (state + state).metric_values(metrics)
  assert set(actual) == {'rmse.2m_temperature', 'rmse.geopotential'}
  assert actual['rmse.2m_temperature'].dims == ('lead_time',)
  assert set(actual['rmse.geopotential'].dims) == {'lead_time', 'level'}
  for v in actual:
    np.testing.assert_allclose(actual[v].values, 1.0)
    np.testing.assert_allclose(summed[v].values, 1.0)
  np.testing.assert_array_equal(actual['rmse.geopotential']['level'].values, [500, 700, 850])


def test_missing_reduce_dims(backend):
  predictions, targets = _test_data()
  metrics = {'rmse': deterministic.RMSE()}
  values = _aggregate(metrics, predictions, targets, reduce_dims=['level', 'latitude', 'longitude']).metric_values(metrics)
  assert list(values) == ['rmse.geopotential']  # 2m_temperature has no level dim: dropped


def test_nan_handling(backend):
  predictions, targets = _test_data()
  targets = targets.where(targets['geopotential']['latitude'] > 0)
  targets = add_nan_mask(targets)
  predictions = dict(predictions)
  metrics = {'rmse': deterministic.RMSE()}
  rd = ['init_time', 'latitude', 'longitude']
  actual = _aggregate(metrics, predictions, targets, reduce_dims=rd).metric_values(metrics)
  assert np.isnan(actual['rmse.geopotential'].values).all()
  actual = _aggregate(metrics, predictions, targets, reduce_dims=rd, masked=True).metric_values(metrics)
  assert not np.isnan(actual['rmse.geopotential'].values).any()
  np.testing.assert_allclose(actual['rmse.geopotential'].values, 1.0)
  actual = _aggregate(metrics, predictions, targets, reduce_dims=rd, skipna=True).metric_values(metrics)
  assert not np.isnan(actual['rmse.geopotential'].values).any()
  # mask on one variable only
  targets['2m_temperature'] = targets['2m_temperature'].drop_vars('mask')
  actual = _aggregate(metrics, predictions, targets, reduce_dims=rd, masked=True).metric_values(metrics)
  assert not np.isnan(actual['rmse.geopotential'].values).any()
  assert np.isnan(actual['rmse.2m_temperature'].values).any()


def test_weighting_two_times_two(backend):
  predictions, targets = _test_data()
  metrics = {'rmse': deterministic.RMSE()}

  class TestWeighting(weighting.Weighting):
    def weights(self, statistic):
      return xr.ones_like(statistic) * 2

  rd = ['init_time', 'latitude', 'longitude']
  s1 = _aggregate(metrics, predictions, targets, reduce_dims=rd)
  s4 = _aggregate(metrics, predictions, targets, reduce_dims=rd, weigh_by=[TestWeighting(), TestWeighting()])
  for stat in s1.sum_weighted_statistics:
    for var in s1.sum_weighted_statistics[stat]:
      xr.assert_allclose(s1.sum_weighted_statistics[stat][var] * 4, s4.sum_weighted_statistics[stat][var])
      xr.assert_allclose(s1.sum_weights[stat][var] * 4, s4.sum_weights[stat][var])
  a, b = s1.metric_values(metrics), s4.metric_values(metrics)
  for v in a:
    xr.assert_allclose(a[v], b[v])


def test_binning_two_region_sets(backend):
  predictions, targets = _test_data()
  metrics = {'rmse': deterministic.RMSE()}
  bin_by = [binning.Regions({'north': ((0, 90), (0, 360)), 'south': ((-90, 0), (0, 360))}, bin_dim_name='bins1'),
            binning.Regions({'east': ((-90, 90), (0, 180)), 'west': ((-90, 90), (180, 360))}, bin_dim_name='bins2')]
  state = _aggregate(metrics, predictions, targets, reduce_dims=['init_time', 'latitude', 'longitude'], bin_by=bin_by)
  actual = state.metric_values(metrics)
  assert set(actual['rmse.geopotential'].dims) == {'bins1', 'bins2', 'lead_time', 'level'}
  assert list(actual['rmse.geopotential']['bins1'].values) == ['north', 'south']
  np.testing.assert_allclose(actual['rmse.geopotential'].values, 1.0)


def test_duplicate_bin_names_raise(backend):
  predictions, targets = _test_data()
  bin_by = [binning.Regions({'a': ((0, 90), (0, 360))}), binning.Regions({'b': ((-90, 0), (0, 360))})]
  with pytest.raises(ValueError, match='Bin dimension names must be unique'):
    _aggregate({'rmse': deterministic.RMSE()}, predictions, targets, reduce_dims=['latitude'], bin_by=bin_by)


def _example_state():
  return aggregation.AggregationState(
      sum_weighted_statistics={'stat_name': {'var1': xr.DataArray([1, 2], dims=['x']),
                                             'var2': xr.DataArray([3, 4], dims=['x'])}},
      sum_weights={'stat_name': {'var1': xr.DataArray([5, 6], dims=['x']), 'var2': xr.DataArray([7, 8], dims=['x'])}})


def test_state_round_trips():
  state = _example_state()
  for back in (aggregation.AggregationState.from_data_tree(state.to_data_tree()),
               aggregation.AggregationState.from_dataset(state.to_dataset())):
    xarray_tree.map_structure(xr.assert_allclose, (state.sum_weighted_statistics, state.sum_weights),
                              (back.sum_weighted_statistics, back.sum_weights))
  assert set(state.to_dataset()) == {'stat_name#var1#sum_weighted_statistics', 'stat_name#var1#sum_weights',
                                     'stat_name#var2#sum_weighted_statistics', 'stat_name#var2#sum_weights'}


def test_state_algebra():
  state = _example_state()
  assert aggregation.AggregationState.sum([aggregation.AggregationState.zero()]).sum_weights is None
  doubled = state + aggregation.AggregationState.zero() + state
  np.testing.assert_allclose(doubled.sum_weights['stat_name']['var2'].values, [14, 16])
  np.testing.assert_allclose(state.mean_statistics()['stat_name']['var1'].values, [0.2, 2 / 6])
  np.testing.assert_allclose(state.sum_along_dims(['x']).sum_weights['stat_name']['var1'].values, 11)
  with pytest.raises(ValueError, match='zero AggregationState'):
    aggregation.AggregationState.zero().map(lambda x: x)
  # outer-join, zero-filled sum of non-aligned coordinates (aggregation.py:27-60)
  a = xr.DataArray([1.0, 2.0], dims=['x'], coords={'x': [0, 1]})
  b = xr.DataArray([10.0, 20.0], dims=['x'], coords={'x': [1, 2]})
  s = aggregation.combining_sum([a, b])
  np.testing.assert_allclose(s.values, [1, 12, 20])
  np.testing.assert_array_equal(s['x'].values, [0, 1, 2])


def test_aggregators_metrics_and_chunks_pickle_without_device_state(backend):
  """Beam pickles DoFns (metrics, aggregators) and chunks to workers (beam_pipeline.py:140-160): caches that hold
  device buffers must not travel, and everything must work again after unpickling."""
  import pickle
  predictions, targets = _test_data()
  predictions, targets = dict(predictions), dict(targets)
  metrics = {'rmse': deterministic.RMSE()}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions({'n': ((0, 90), (0, 360)), 's': ((-90, 0), (0, 360))})])
  before = _values(agg, metrics, predictions, targets)
  agg2, metrics2, p2, t2 = pickle.loads(pickle.dumps((agg, metrics, predictions, targets)))
  assert not any(k.startswith('_w_') for k in agg2.__dict__)
  assert not any(k.startswith('_wbx_') for da in p2.values() for k in da.__dict__)
  after = _values(agg2, metrics2, p2, t2)
  for k in before:
    np.testing.assert_allclose(before[k].values, after[k].values)
  # a lazy statistic pickles as its materialised values
  stat = deterministic.SquaredError().compute(predictions, targets)['geopotential']
  back = pickle.loads(pickle.dumps(stat))
  np.testing.assert_allclose(back.values, 1.0)


def _values(agg, metrics, predictions, targets):
  stats = metrics_base.compute_unique_statistics_for_all_metrics(metrics, predictions, targets)
  return agg.aggregate_statistics(stats).metric_values(metrics)
