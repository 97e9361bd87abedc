"""The error behaviour of the reference's surface, host side only: the same exception types and
messages, the same silent drops.  Cases with their own tests elsewhere: 'Failed to compute statistic' wrapping and
M < 2 (test_metrics.py), duplicate bin names and missing reduce dims (test_aggregation.py), skipna_ensemble with
use_sort (test_metrics.py)."""
import numpy as np
import pytest

import fake_device
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import deterministic
from weatherbenchx_amd.metrics import probabilistic
from weatherbenchx_amd.metrics import wrappers


@pytest.fixture(autouse=True)
def _emulated(monkeypatch):
  fake_device.install(monkeypatch)


def _field(dims=('time', 'latitude', 'longitude'), lat=None):
  lat = np.linspace(-80, 80, 5) if lat is None else lat
  shape = {'time': 2, 'number': 3, 'latitude': len(lat), 'longitude': 8}
  rng = np.random.default_rng(0)
  return xr.DataArray(rng.normal(size=[shape[d] for d in dims]).astype(np.float32), dims=dims,
                      coords={'latitude': lat, 'longitude': np.arange(8) * 45.0})


def test_non_monotonic_latitude_is_an_assertion_error():
  # weighting.py:115-117
  bad = _field(lat=np.array([-80.0, -40.0, -60.0, 0.0, 40.0]))
  with pytest.raises(AssertionError, match='strictly monotonic'):
    weighting.GridAreaWeighting().weights(bad)
  # a statistic without a latitude dim is simply not weighted (weighting.py:110-111)
  w = weighting.GridAreaWeighting().weights(xr.DataArray(np.zeros(3), dims=('time',)))
  assert float(np.asarray(w.values)) == 1.0 and w.dims == ()


def test_missing_ensemble_dimension_raises_value_error():
  p, t = {'v': _field()}, {'v': _field()}
  # probabilistic.py:316-318 -- wrapped as 'Failed to compute statistic' by the statistic loop (metrics/base.py:263-269)
  with pytest.raises(ValueError, match='Dimension number not found'):
    probabilistic.UnbiasedEnsembleMeanSquaredError().compute(p, t)
  # probabilistic.py:63-66: every array under an EnsembleAveragedStatistic needs the ensemble dim
  avg = probabilistic.EnsembleAveragedMetric(deterministic.RMSE())
  with pytest.raises(ValueError, match='not found in'):
    for s in avg.statistics.values():
      s.compute(p, t)
  # wrappers.py:116-148 (EnsembleMean)
  with pytest.raises(ValueError, match='not found in'):
    wrappers.EnsembleMean(which='predictions').transform_fn(p['v'])
  with pytest.raises(ValueError, match='Failed to compute statistic'):
    aggregation.compute_metric_values_for_single_chunk({'m': probabilistic.UnbiasedEnsembleMeanRMSE()},
                                                       aggregation.Aggregator(reduce_dims=['latitude', 'longitude']), p, t)


def test_zero_state_cannot_be_mapped_but_adds_as_identity():
  # aggregation.py:92-110, 190-191
  zero = aggregation.AggregationState.zero()
  with pytest.raises(ValueError, match='Cannot map a zero AggregationState'):
    zero.map(lambda x: x)
  p = {'v': _field()}
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'])
  state = agg.aggregate_statistics({'SquaredError': deterministic.SquaredError().compute(p, p)})
  total = aggregation.AggregationState.sum([zero, state, zero])
  np.testing.assert_array_equal(total.sum_weights['SquaredError']['v'].values, state.sum_weights['SquaredError']['v'].values)
