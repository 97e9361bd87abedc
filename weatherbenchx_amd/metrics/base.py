"""Metric / Statistic plugin protocol (counterpart of weatherbenchX/metrics/base.py:23-415).

Same class names, abstract methods and helper functions as the reference so user plugins port
unchanged.  Built-in per-point statistics return `lazy.LazyStatistic` DataArrays (see lazy.py); plugin
statistics may return any DataArray (numpy or torch payload) and are reduced by the same HIP path.
"""
from __future__ import annotations

import abc
from collections.abc import Iterator, Mapping
from typing import Hashable, final

import numpy as np

from weatherbenchx_amd import lazy
from weatherbenchx_amd import xarray_lite as xr


class Metric(abc.ABC):
  """A set of statistics plus a map from their (weighted) means to the metric value (base.py:23-82)."""

  @property
  @abc.abstractmethod
  def statistics(self) -> Mapping[str, 'Statistic']:
    """internal name -> Statistic whose mean the metric needs."""

  @abc.abstractmethod
  def values_from_mean_statistics(
      self, statistic_values: Mapping[str, Mapping[Hashable, xr.DataArray]]
  ) -> Mapping[Hashable, xr.DataArray]:
    """Metric values per variable from mean statistics keyed by the internal names."""


_CLIM_REF_CACHE: dict = {}  # (id(climatology), its mutation count, time labels) -> (climatology, dims, index tables)


class Statistic(Metric):
  """Function of a (predictions, targets) chunk, aggregated by a weighted mean (base.py:85-173)."""

  @property
  def unique_name(self) -> str:
    # Deduplication key across metrics; subclasses with parameters must extend it (base.py:120-133).
    return type(self).__name__

  @abc.abstractmethod
  def compute(self, predictions: Mapping[Hashable, xr.DataArray],
              targets: Mapping[Hashable, xr.DataArray]) -> Mapping[Hashable, xr.DataArray]:
    """Per-point statistic per variable."""

  @final
  @property
  def statistics(self) -> Mapping[str, 'Statistic']:
    return {'self': self}

  @final
  def values_from_mean_statistics(self, statistic_values):
    return statistic_values['self']


class PerVariableStatistic(Statistic):
  """Computed independently for every variable present in BOTH inputs; `None` results are dropped and the
  result is always a plain dict (base.py:176-205, base_test.py:79-91)."""

  @final
  def compute(self, predictions, targets):
    out = {}
    for name in predictions.keys():
      if name not in targets.keys():
        continue
      value = self._compute_per_variable(_named(predictions[name], name), _named(targets[name], name))
      if value is not None:
        out[name] = value
    return out

  @abc.abstractmethod
  def _compute_per_variable(self, predictions: xr.DataArray, targets: xr.DataArray) -> xr.DataArray | None:
    ...


def _converted(mapping):
  """{name: DataArray} with foreign labeled arrays converted (kept by the caller for the duration of its loop); anything that is
  not a plain mapping of arrays (a Dataset, a user object) is passed through untouched."""
  if isinstance(mapping, dict) and not all(isinstance(v, xr.DataArray) for v in mapping.values()):
    return {k: (xr.as_dataarray(v) if (hasattr(v, 'dims') and hasattr(v, 'coords') and hasattr(v, 'values')) else v)
            for k, v in mapping.items()}
  return mapping


def _named(da, name):
  da = xr.as_dataarray(da)
  if da.name is None:
    da.name = name
  return da


class PerVariableMetric(Metric):
  """Metric evaluated per variable over the variables common to all of its statistics (base.py:208-243)."""

  @final
  def values_from_mean_statistics(self, statistic_values):
    names = list(self.statistics)
    common = set(statistic_values[names[0]])
    for s in names[1:]:
      common &= set(statistic_values[s])
    return {v: self._values_from_mean_statistics_per_variable({s: statistic_values[s][v] for s in names})
            for v in common}

  @abc.abstractmethod
  def _values_from_mean_statistics_per_variable(self, statistic_values: Mapping[str, xr.DataArray]) -> xr.DataArray:
    ...


# Kept for source compatibility: a Statistic already is a Metric (base.py:246-249).
NoOpMetric = lambda statistic: statistic  # pylint: disable=invalid-name


def generate_unique_statistics_for_all_metrics(metrics, predictions, targets) -> Iterator[tuple[str, Mapping]]:
  """Yields (unique_name, values) once per distinct statistic (base.py:252-269)."""
  # foreign labeled arrays (a real xr.DataArray ...) are converted ONCE here: every statistic of a (predictions, targets) pair
  # then meets the same DataArray objects, and with them the same fused group and the same uploaded copies
  predictions, targets = _converted(predictions), _converted(targets)
  unique = {}
  for metric in metrics.values():
    for stat in metric.statistics.values():
      unique[stat.unique_name] = stat
  for name, stat in unique.items():
    try:
      yield name, stat.compute(predictions, targets)
    except Exception as e:  # pylint: disable=broad-except
      raise ValueError(f'Failed to compute statistic {name}={stat} from:\n{predictions=}\n{targets=}') from e


def compute_unique_statistics_for_all_metrics(metrics, predictions, targets):
  return dict(generate_unique_statistics_for_all_metrics(metrics, predictions, targets))


def compute_metric_from_statistics(metric: Metric, statistic_values):
  """Re-keys statistics from unique to internal names, then evaluates the metric (base.py:294-315)."""
  renamed = {internal: statistic_values[stat.unique_name] for internal, stat in metric.statistics.items()}
  return metric.values_from_mean_statistics(renamed)


def compute_metrics_from_statistics(metrics, statistic_values):
  return {name: compute_metric_from_statistics(m, statistic_values) for name, m in metrics.items()}


def _aligned_ref(climatology, over_dims, positions, prefetch=False):
  """A climatology in HBM (or small enough to be uploaded whole) is gathered in place; a host-resident one -- a memory map, an
  array beyond `climatology_cache.AUTO_RESIDENT_BYTES`, anything `climatology_cache.cached()` was called on -- goes through
  its slab pool: only the (dayofyear, hour) slabs this chunk names are (or become) device resident (base.py:396-403 computes
  exactly those from a lazily backed dataset)."""
  from weatherbenchx_amd import climatology_cache  # pylint: disable=g-import-not-at-top
  cache = climatology_cache.cache_for(climatology)
  if cache is not None:
    return cache.ref(over_dims, positions, prefetch=prefetch)
  return lazy.ClimatologyRef(climatology, over_dims, positions)


class PerVariableStatisticWithClimatology(Statistic):
  """Statistics of (prediction, target, climatology at valid_time) (base.py:338-415).

  The reference materialises `climatology.sel(dayofyear=..., hour=...)` three times per variable
  (base.py:403).  Here the selection is turned into an index table (ClimatologyRef) that the stage-1
  kernel uses to gather whole 2-D climatology fields in place -- no aligned copy is ever made.
  """

  def __init__(self, climatology):
    self._climatology = climatology

  @final
  def compute(self, predictions, targets):
    predictions, targets = dict(predictions), dict(targets)
    out = {}
    for name in predictions:
      out[name] = self._compute_per_variable(_named(predictions[name], name), _named(targets[name], name),
                                             xr.as_dataarray(self._climatology[name]))
    return out

  @final
  def _compute_per_variable(self, predictions, targets, climatology):
    # ACC asks for the same alignment three times per variable (base.py:403): resolve it once per
    # (predictions object, climatology variable).
    ref = self.resolve_climatology(predictions, climatology)
    return self._compute_per_variable_with_aligned_climatology(predictions, targets, ref)

  @classmethod
  def resolve_climatology(cls, predictions, climatology, prefetch=False) -> 'lazy.ClimatologyRef':
    """The alignment of `climatology` with these predictions, resolved once per (predictions object, climatology variable)."""
    cache = predictions.__dict__.setdefault('_wbx_clim_refs', {})
    hit = cache.get(id(climatology))
    version = climatology.__dict__.get('_mutations', 0)
    if hit is None or hit[0] is not climatology or hit[2] != version:
      hit = (climatology, cls._climatology_ref(predictions, climatology, prefetch=prefetch), version)
      cache[id(climatology)] = hit
    return hit[1]

  @staticmethod
  def _climatology_ref(predictions, climatology, prefetch=False) -> 'lazy.ClimatologyRef':
    # The index tables depend on the time labels only: chunks that carry the same (init_time, lead_time) / valid_time values
    # against the same (unmodified) climatology object reuse them (datetime arithmetic and dayofyear / hour extraction
    # were ~0.1 ms of every chunk's host time).
    key = None
    if 'valid_time' in predictions.coords or 'valid_time' in predictions.dims:
      names = ('valid_time',)
    elif (('init_time' in predictions.coords or 'init_time' in predictions.dims)
          and ('lead_time' in predictions.coords or 'lead_time' in predictions.dims)):
      names = ('init_time', 'lead_time')
    else:
      raise ValueError('Predictions should have either valid_time or init/lead_time dimensions.')
    try:
      labels_key = tuple((n, predictions[n].dims, np.asarray(predictions[n].values).tobytes()) for n in names)
      key = (id(climatology), climatology.__dict__.get('_mutations', 0), labels_key)
      hit = _CLIM_REF_CACHE.get(key)
      if hit is not None and hit[0] is climatology:
        return _aligned_ref(climatology, hit[1], hit[2], prefetch)
    except (TypeError, ValueError):
      key = None
    if names == ('valid_time',):
      valid_time = predictions['valid_time']
    else:
      valid_time = predictions['init_time'] + predictions['lead_time']
    if 'time' in climatology.dims:
      labels = {'time': valid_time}
    else:
      labels = {'dayofyear': valid_time.dt.dayofyear}
      if 'hour' in climatology.dims:
        labels['hour'] = valid_time.dt.hour
    positions = {d: climatology._index_positions(d, lab.values) for d, lab in labels.items()}  # pylint: disable=protected-access
    if key is not None:
      if len(_CLIM_REF_CACHE) > 64:
        _CLIM_REF_CACHE.clear()
      _CLIM_REF_CACHE[key] = (climatology, tuple(valid_time.dims), positions)  # (holds the object: its id cannot be recycled)
    return _aligned_ref(climatology, tuple(valid_time.dims), positions, prefetch)

  @abc.abstractmethod
  def _compute_per_variable_with_aligned_climatology(self, predictions, targets, aligned_climatology):
    """`aligned_climatology` is a lazy.ClimatologyRef; call `.aligned_view()` for a DataArray."""
