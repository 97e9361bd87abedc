"""A metric that compares a model with a baseline on the same experimental units (counterpart of
weatherbenchX/statistical_inference/baseline_comparison.py:31-179)."""
from __future__ import annotations

from typing import Callable, Hashable, Mapping

from weatherbenchx_amd import aggregation
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.metrics import wrappers

MetricResult = Mapping[Hashable, xr.DataArray]
Comparison = Callable[[MetricResult, MetricResult], MetricResult]

_MAIN, _BASELINE = 'main_', 'baseline_'


def difference(main_result: MetricResult, baseline_result: MetricResult) -> MetricResult:
  return {k: main_result[k] - baseline_result[k] for k in main_result if k in baseline_result}


class BaselineComparison(metrics_base.Metric):
  """`comparison(metric, baseline_metric)` as a Metric over the union of both metrics' statistics, renamed `main_*` /
  `baseline_*` (unique names too) so that they can live in one AggregationState (`combine_aggregation_states`).  Built after the
  fact, for inference methods; a difference of means is a mean of differences, which is why the default works with the t-tests."""

  def __init__(self, metric: metrics_base.Metric, baseline_metric: metrics_base.Metric | None = None, comparison: Comparison = difference):
    self.metric = metric
    self.baseline_metric = baseline_metric or metric
    self._comparison = comparison

  @property
  def statistics(self) -> Mapping[str, metrics_base.Statistic]:
    out = {}
    for prefix, metric in ((_MAIN, self.metric), (_BASELINE, self.baseline_metric)):
      for name, stat in metric.statistics.items():
        out[prefix + name] = wrappers.RenamedStatistic(stat, prefix + stat.unique_name)
    return out

  def values_from_mean_statistics(self, statistic_values):
    side = lambda prefix: {k[len(prefix):]: v for k, v in statistic_values.items() if k.startswith(prefix)}
    return self._comparison(self.metric.values_from_mean_statistics(side(_MAIN)),
                            self.baseline_metric.values_from_mean_statistics(side(_BASELINE)))


BaselineComparisonAggregationState = aggregation.AggregationState


def combine_aggregation_states(aggregation_state: aggregation.AggregationState,
                               baseline_aggregation_state: aggregation.AggregationState) -> BaselineComparisonAggregationState:
  both = lambda field: {**{_MAIN + k: v for k, v in getattr(aggregation_state, field).items()},
                        **{_BASELINE + k: v for k, v in getattr(baseline_aggregation_state, field).items()}}
  return aggregation.AggregationState(sum_weighted_statistics=both('sum_weighted_statistics'), sum_weights=both('sum_weights'))


def for_metrics(metrics: Mapping[str, metrics_base.Metric], baseline_metrics: Mapping[str, metrics_base.Metric] | None = None,
                comparison: Comparison = difference) -> Mapping[str, BaselineComparison]:
  baseline_metrics = metrics if baseline_metrics is None else baseline_metrics
  return {name: BaselineComparison(metrics[name], baseline_metrics[name], comparison) for name in metrics if name in baseline_metrics}
