"""Aggregator / AggregationState (counterpart of weatherbenchX/aggregation.py:27-435).

Same dataclasses, methods and error behaviour as the reference; the arithmetic of
`Aggregator.aggregate_stat_var` -- the two `xr.dot` passes of aggregation.py:357-362 -- runs on the
GPU: stage 1 (fused statistic + unweighted partial sums, csrc/wbx_s1.hpp) then stage 2
(partial x W contraction, csrc/wbx_s2.hip).  Accumulators are always float64 (the reference follows
the input dtype).  Output dims: surviving statistic dims in their original order, then the
bin dims in `bin_by` order (unpinned by the reference, SURVEY F11: compare by name).
"""
from __future__ import annotations

import dataclasses
import weakref
from typing import Any, Callable, Collection, Hashable, Iterable, Mapping, Sequence

import numpy as np

from weatherbenchx_amd import _hip
from weatherbenchx_amd import binning
from weatherbenchx_amd import engine
from weatherbenchx_amd import lazy
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd import xarray_tree
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd import spectra


def combining_sum(data_arrays: Sequence[xr.DataArray]) -> xr.DataArray:
  """Sum with a zero-filled OUTER join of the coordinates (aggregation.py:27-60)."""
  if not data_arrays:
    return sum([])
  total = data_arrays[0]
  for nxt in data_arrays[1:]:
    a, b = xr.align(total, nxt, join='outer', fill_value=0)
    total = a + b
  return total


@dataclasses.dataclass
class AggregationState:
  """sum_weighted_statistics / sum_weights accumulator pair (aggregation.py:63-265)."""

  sum_weighted_statistics: Any
  sum_weights: Any
  # Set by an Aggregator running inside engine.deferred_results(): the arrays above are views of page-locked memory
  # that the GPU is still filling.  Every method below waits on it first; code that reaches into the two trees
  # directly must call wait() itself.
  _fence: Any = dataclasses.field(default=None, repr=False, compare=False)

  def wait(self) -> 'AggregationState':
    """Blocks until the sums of a deferred aggregation have arrived (no-op otherwise)."""
    if self._fence is not None:
      self._fence.wait()
      self._fence = None
    return self

  def __getstate__(self):
    self.wait()
    return {'sum_weighted_statistics': self.sum_weighted_statistics, 'sum_weights': self.sum_weights, '_fence': None}

  @classmethod
  def zero(cls) -> 'AggregationState':
    return cls(sum_weighted_statistics=None, sum_weights=None)

  def __add__(self, other: 'AggregationState') -> 'AggregationState':
    return self.sum([self, other])

  @classmethod
  def sum(cls, aggregation_states: Iterable['AggregationState']) -> 'AggregationState':
    aggregation_states = [s.wait() for s in aggregation_states]
    pairs = [(s.sum_weighted_statistics, s.sum_weights) for s in aggregation_states
             if s.sum_weighted_statistics is not None]
    if not pairs:
      return cls.zero()
    sws, sw = xarray_tree.map_structure(lambda *leaves: combining_sum(leaves), *pairs)
    return cls(sws, sw)

  def mean_statistics(self) -> Any:
    self.wait()

    def mean(num, den):
      # numerator and denominator of one reduction sit on the identical frame: one numpy division, no joins
      if (type(num) is xr.DataArray and type(den) is xr.DataArray and isinstance(num._data, np.ndarray)  # pylint: disable=protected-access,unidiomatic-typecheck
          and isinstance(den._data, np.ndarray) and num._data.shape == den._data.shape and xr._same_frame(num, den)):  # pylint: disable=protected-access
        return num._replace(data=np.true_divide(num._data, den._data))  # pylint: disable=protected-access
      return num / den
    with np.errstate(all='ignore'):
      return xarray_tree.map_structure(mean, self.sum_weighted_statistics, self.sum_weights)

  def metric_values(self, metrics: Mapping[str, metrics_base.Metric]) -> xr.Dataset:
    """Dataset of `<metric>.<variable>` values (aggregation.py:122-148)."""
    per_metric = metrics_base.compute_metrics_from_statistics(metrics, self.mean_statistics())
    out = xr.Dataset()
    for metric_name, per_var in per_metric.items():
      for var_name, da in per_var.items():
        out[f'{metric_name}.{var_name}'] = da
    return out

  def sum_along_dims(self, dims: Collection[str]) -> 'AggregationState':
    if self.sum_weighted_statistics is None:
      return self
    return self.map(lambda x: x.sum(list(dims), skipna=False))

  def dot(self, *arrays: xr.DataArray, dim) -> 'AggregationState':
    return self.map(lambda x: xr.dot(x, *arrays, dim=dim))

  @classmethod
  def map_multi(cls, func: Callable[..., xr.DataArray], *agg_states: 'AggregationState') -> 'AggregationState':
    if any(a.wait().sum_weighted_statistics is None for a in agg_states):
      raise ValueError('Cannot map a zero AggregationState.')
    return cls(xarray_tree.map_structure(func, *[a.sum_weighted_statistics for a in agg_states]),
               xarray_tree.map_structure(func, *[a.sum_weights for a in agg_states]))

  def map(self, func: Callable[[xr.DataArray], xr.DataArray]) -> 'AggregationState':
    return self.map_multi(func, self)

  # -- persistence: nested dict ("data tree") and flat '#'-separated Dataset (aggregation.py:203-265) ----
  def to_data_tree(self) -> dict:
    """Nested dict mirror of xr.DataTree: leaves are {'sum_weighted_statistics': da, 'sum_weights': da}."""
    self.wait()
    if isinstance(self.sum_weighted_statistics, xr.DataArray):
      return {'sum_weighted_statistics': self.sum_weighted_statistics, 'sum_weights': self.sum_weights}
    if isinstance(self.sum_weighted_statistics, Mapping):
      return {k: AggregationState(self.sum_weighted_statistics[k], self.sum_weights[k]).to_data_tree()
              for k in self.sum_weighted_statistics.keys()}
    raise TypeError('Bad type for AggregationState.sum_weighted_statistics.')

  @classmethod
  def from_data_tree(cls, data_tree: Mapping, name=None) -> 'AggregationState':
    if set(data_tree.keys()) == {'sum_weighted_statistics', 'sum_weights'} and isinstance(
        data_tree['sum_weights'], xr.DataArray):
      return cls(data_tree['sum_weighted_statistics'].rename(name), data_tree['sum_weights'].rename(name))
    children = {k: cls.from_data_tree(v, name=k) for k, v in data_tree.items()}
    return cls({k: v.sum_weighted_statistics for k, v in children.items()},
               {k: v.sum_weights for k, v in children.items()})

  def to_dataset(self, separator='#') -> xr.Dataset:
    flat = {}

    def walk(node, path):
      if isinstance(node.get('sum_weights'), xr.DataArray) and len(node) == 2:
        for leaf in ('sum_weighted_statistics', 'sum_weights'):
          flat[separator.join(path + [leaf])] = node[leaf]
      else:
        for k, v in node.items():
          walk(v, path + [str(k)])
    walk(self.to_data_tree(), [])
    return xr.Dataset(flat)

  @classmethod
  def from_dataset(cls, dataset: Mapping, separator='#') -> 'AggregationState':
    tree: dict = {}
    for full, da in dataset.items():
      *path, leaf = str(full).split(separator)
      node = tree
      for part in path:
        node = node.setdefault(part, {})
      node[leaf] = da
    return cls.from_data_tree(tree)


def _fenced(state):
  d = engine.deferred_active()
  if d is not None and state is not None:
    state._fence = d.mark()  # pylint: disable=protected-access
  return state


def _weight_product(stat: xr.DataArray, weigh_by, bin_by):
  """-> (W as a labeled array or None, bin dim names) or None when a bin mask needs dims the statistic lacks."""
  stat_dims = set(stat.dims)
  names = [b.bin_dim_name for b in bin_by or []]
  if len(set(names)) != len(names):
    raise ValueError('Bin dimension names must be unique.')
  w = None
  for method in weigh_by or []:
    wi = xr.as_dataarray(method.weights(stat)).astype(np.float64)
    w = wi if w is None else w * wi
  masks, all_bool = None, True
  for method in bin_by or []:
    mask = xr.as_dataarray(method.create_bin_mask(stat))
    if not (set(mask.dims) - {method.bin_dim_name}) <= stat_dims:
      return None  # cannot bin on dims that are not evaluation-unit dims (aggregation.py:320-330)
    all_bool = all_bool and mask.dtype == np.bool_
    masks = mask if masks is None else (masks & mask if all_bool else masks.astype(np.float64) * mask.astype(np.float64))
  if masks is None:
    return w, tuple(names)
  product = masks.astype(np.float64) if w is None else w * masks.astype(np.float64)
  if all_bool:
    # boolean masks: the engine may contract with the (weights, membership bits) factors instead of the dense product
    product.__dict__['_wbx_factors'] = (w, masks)
  return product, tuple(names)


class _PendingLinear(xr.LazyPickleMixin, xr.DataArray):
  """sum_i coeff_i * term_i of result arrays whose numbers are still on their way (read-back in flight, or living in a
  device accumulator): the combination is evaluated when `.data` is first read -- AggregationState waits on its fence
  before it looks -- and an Accumulation records the terms instead (the reduction is linear, so the coefficients can be
  applied after the chunks have been summed and all-reduced)."""

  def __init__(self, terms, name=None, attrs=None):
    first = terms[0][1]
    self._data = None
    self._dims = first.dims
    self.name = name
    self.attrs = dict(attrs or {})
    self._coords = dict(first._coords)  # pylint: disable=protected-access
    self._linear_terms = [(float(c), t) for c, t in terms]

  @property
  def data(self):
    if self._data is None:
      total = None
      for c, t in self._linear_terms:
        v = np.asarray(t.transpose(*self._dims).values, dtype=np.float64)
        v = v if c == 1.0 else v * c
        total = v if total is None else total + v
      self._data = total
    return self._data

  @property
  def shape(self):
    return self._linear_terms[0][1].transpose(*self._dims).shape

  @property
  def dtype(self):
    return np.dtype(np.float64)


class _Token:
  """Hashable by identity; kept alive by whatever cache key holds it."""
  __slots__ = ()


@dataclasses.dataclass
class Aggregator:
  """Weighted / binned reduction over `reduce_dims` (aggregation.py:268-408); see the reference for the
  NaN note: with skipna=False a NaN anywhere in the reduced set makes every bin NaN."""

  reduce_dims: Collection[str]
  bin_by: Sequence[binning.Binning] | None = None
  weigh_by: Sequence[weighting.Weighting] | None = None
  masked: bool = False
  skipna: bool = False

  def __getstate__(self):
    # cached weight products hold device buffers: drop them when the aggregator is pickled to a worker
    return {k: v for k, v in self.__dict__.items() if not k.startswith('_w_')}

  def __setstate__(self, state):
    self.__dict__.update(state)

  # ---- reference-compatible single-array entry point -------------------------------------------------
  def aggregation_fn(self, stat: xr.DataArray) -> xr.DataArray | None:
    """sum over reduce_dims of stat * weights * bin masks (aggregation.py:297-335)."""
    with engine.synchronous_results():  # a bare DataArray cannot carry a fence
      state = self._aggregate(xr.as_dataarray(stat), use_mask=False, skipna=False)
    return None if state is None else state.sum_weighted_statistics

  def aggregate_stat_var(self, stat: xr.DataArray) -> AggregationState | None:
    """One statistic of one variable -> AggregationState, or None if a reduce/bin dim is missing
    (aggregation.py:337-366)."""
    return _fenced(self._stat_var(stat))

  def aggregate_stat_vars(self, stats: Mapping[Hashable, xr.DataArray]) -> AggregationState:
    return _fenced(self._stat_vars(stats))

  def aggregate_statistics(self, statistics: Mapping[str, Mapping[Hashable, xr.DataArray]]) -> AggregationState:
    self.note_statistics(statistics)
    # (the statistics of one call usually sit on the same frame -- the very same coordinate arrays: their weight product
    # is looked up once per call by identity instead of once per statistic by content hash)
    self.__dict__['_w_call_memo'] = {}
    try:
      per_stat = {name: self._stat_vars(stats) for name, stats in statistics.items()}
    finally:
      self.__dict__.pop('_w_call_memo', None)
    return _fenced(AggregationState({k: v.sum_weighted_statistics for k, v in per_stat.items()},
                                    {k: v.sum_weights for k, v in per_stat.items()}))

  @staticmethod
  def note_statistics(statistics: Mapping[str, Mapping[Hashable, xr.DataArray]]) -> None:
    """Look ahead over the statistics that are about to be aggregated: the spread lane is the only ensemble lane that
    depends on (algorithm, fair).  Whatever lane of a (p, t) group comes first launches the kernel for ALL five lanes, so it
    has to launch the variant the group's CRPSSpread statistic will ask for -- otherwise CRPSEnsemble reads the ensemble
    twice (skill first, spread with other parameters second: round 2's 0.82 ms per 1.73 GB for the reference-default
    CRPSEnsemble()).  Called by aggregate_statistics and by the chunk loop (pipeline._consume)."""
    by_p = {}
    for stats in statistics.values():
      for s in stats.values():
        if isinstance(s, lazy.LazyStatistic) and s.is_lazy and s._group.kind == 'ens':  # pylint: disable=protected-access
          by_p.setdefault(id(s._group.p), set()).add(s._group)  # pylint: disable=protected-access
          if s._lane == lazy.ENS_LANE['CRPSSpread'] and s._ens_params:  # pylint: disable=protected-access
            s._group.spread_params = dict(s._ens_params)  # pylint: disable=protected-access
    # Statistics of the predictions alone (spread, variance) live in a group of their own -- no mask coordinate -- next to the
    # group of the same predictions against the masked targets.  One launch can serve both (engine.ENS_TWIN_MASK): the unmasked
    # group learns who its masked sibling is, so that whichever statistic comes first runs that one launch.
    for groups in by_p.values():
      masked = [g for g in groups if 'mask' in g.coords]
      if len(masked) == 1:
        for g in groups:
          if g is not masked[0] and 'mask' not in g.coords:
            g.twin_sibling = weakref.ref(masked[0])
            if getattr(g, 'spread_params', None) and not getattr(masked[0], 'spread_params', None):
              masked[0].spread_params = dict(g.spread_params)

  def _stat_var(self, stat: xr.DataArray) -> AggregationState | None:
    stat = xr.as_dataarray(stat)
    return self._aggregate(stat, use_mask=self.masked and 'mask' in stat.coords, skipna=self.skipna)

  def _stat_vars(self, stats: Mapping[Hashable, xr.DataArray]) -> AggregationState:
    per_var = {name: self._stat_var(s) for name, s in stats.items() if s is not None}
    per_var = {k: v for k, v in per_var.items() if v is not None}
    return AggregationState({k: v.sum_weighted_statistics for k, v in per_var.items()},
                            {k: v.sum_weights for k, v in per_var.items()})

  # ---- implementation ---------------------------------------------------------------------------------
  def _aggregate(self, stat: xr.DataArray, *, use_mask: bool, skipna: bool) -> AggregationState | None:
    reduce_set = set(self.reduce_dims)
    if not reduce_set <= set(stat.dims):
      return None  # variables without every reduce dim are dropped (aggregation.py:305-309)
    wp = self._cached_weight_product(stat)
    if wp is None:
      return None
    w_da, bin_dims = wp

    if isinstance(stat, lazy.LinearCombination) and stat.is_lazy and not use_mask and not skipna:
      parts = [self._aggregate(term, use_mask=False, skipna=False) for term in stat._terms]  # pylint: disable=protected-access
      coeffs = [c * stat._scale for c in stat._coeffs]  # pylint: disable=protected-access
      if engine.deferred_active() is not None:  # combined once the sums have arrived / been accumulated
        sws = _PendingLinear([(c, p.sum_weighted_statistics) for c, p in zip(coeffs, parts)], name=stat.name)
        return AggregationState(sws, parts[0].sum_weights)
      sws = None
      for p, c in zip(parts, coeffs):
        term = p.sum_weighted_statistics if c == 1.0 else p.sum_weighted_statistics * c
        sws = term if sws is None else sws + term
      return AggregationState(sws, parts[0].sum_weights)

    if (isinstance(stat, lazy.LazyCategorical) and stat.is_lazy and not set(stat.cat_dims) & reduce_set
        and (w_da is None or not set(stat.cat_dims) & set(w_da.dims))):
      return self._reduce_categorical(stat, w_da, bin_dims, use_mask, skipna)

    if isinstance(stat, spectra.LazySpectrum) and stat.is_lazy and not use_mask and not skipna:
      fused = self._reduce_spectrum(stat, w_da, bin_dims)
      if fused is not None:
        return fused

    if isinstance(stat, lazy.LazyStatistic) and stat.is_lazy:
      values, counts, out_dims, frame_coords, lane, scale = self._reduce_lazy(stat, w_da, bin_dims, use_mask, skipna)
    else:
      values, counts, out_dims = self._reduce_materialised(stat, w_da, bin_dims, use_mask, skipna)
      frame_coords, lane, scale = stat._coords, 0, 1.0  # pylint: disable=protected-access

    final_dims = tuple(d for d in stat.dims if d in out_dims) + tuple(bin_dims)
    # (a `mask` coordinate whose dims all survive stays on the result, like every other coordinate: xr.dot keeps it,
    #  aggregation.py:335 -- RelativeIntensity's per-slice mask reaches the metric values that way, deterministic_test.py:87-89)
    coords = {k: v for k, v in frame_coords.items() if set(v[0]) <= set(final_dims)}
    if w_da is not None:
      for k, v in w_da._coords.items():  # pylint: disable=protected-access
        if set(v[0]) <= set(final_dims):
          coords.setdefault(k, v)

    pending = engine.deferred_active() is not None

    perm = tuple(out_dims.index(d) for d in final_dims)

    def wrap(arr):  # (`values[lane, ...]`: the ellipsis keeps a 0-d result a view instead of a scalar copy)
      view = np.asarray(arr, dtype=np.float64).transpose(perm)
      # deferred: keep the (possibly strided) view of the buffer the GPU is still writing / accumulating -- no reads here
      data = view if pending else np.array(view, order="C", copy=True)
      return xr.DataArray._assemble(data, final_dims, coords, name=stat.name, attrs=stat.attrs)  # pylint: disable=protected-access

    if scale != 1.0:
      if pending:  # the factor is applied when the numbers are read (or after the accumulators have been reduced)
        return AggregationState(_PendingLinear([(scale, wrap(values[lane, ...]))], name=stat.name, attrs=stat.attrs),
                                _PendingLinear([(scale, wrap(counts[lane, ...]))], name=stat.name, attrs=stat.attrs))
      return AggregationState(wrap(values[lane, ...] * scale), wrap(counts[lane, ...] * scale))
    return AggregationState(wrap(values[lane, ...]), wrap(counts[lane, ...]))

  def __setattr__(self, name, value):
    # the cached weight products below belong to the configuration they were built for
    if not name.startswith('_w_'):
      self.__dict__.pop('_w_products', None)
      self.__dict__.pop('_w_dep_hints', None)
    object.__setattr__(self, name, value)

  def _cached_weight_product(self, stat: xr.DataArray):
    """W = prod(weights) * prod(bin masks) is rebuilt by the reference for every (statistic, variable)
    call (aggregation.py:311-330).  For the built-in coordinate-only plugins it is cached per
    (plugin objects, statistic dims, coordinates of the dims W depends on); assigning a new `weigh_by` / `bin_by`
    drops the cache (`__setattr__`), and the plugin objects themselves are part of the key (a list edited in place)."""
    known = (weighting.GridAreaWeighting, binning.Regions, binning.LandSea)
    plugins = tuple(self.weigh_by or ()) + tuple(self.bin_by or ())
    if not all(isinstance(m, known) for m in plugins):
      return _weight_product(stat, self.weigh_by, self.bin_by)  # (a user plugin may look at the statistic's values)
    memo = self.__dict__.get('_w_call_memo')  # set for the duration of one aggregate_statistics call
    if memo is None:
      return self._weight_product_by_content(stat, plugins)
    mkey = (stat.dims, stat.shape, tuple((k, id(v[1])) for k, v in stat._coords.items()))  # pylint: disable=protected-access
    if mkey not in memo:
      memo[mkey] = self._weight_product_by_content(stat, plugins)
    return memo[mkey]

  def _weight_product_by_content(self, stat: xr.DataArray, plugins):
    cache = self.__dict__.setdefault('_w_products', {})
    hints = self.__dict__.setdefault('_w_dep_hints', {})
    # ids stay unique while the cache entry holds the objects (stored next to the product)
    who = (len(self.weigh_by or ()),) + tuple(id(m) for m in plugins)
    frame = (who, stat.dims, stat.shape)

    def key_for(dep):
      return (frame, tuple((d, hash(np.asarray(stat._coords[d][1]).tobytes()) if d in stat._coords else None)  # pylint: disable=protected-access
                           for d in dep))
    dep = hints.get(frame)
    if dep is not None:
      hit = cache.get(key_for(dep))
      if hit is not None:
        return hit[0]
    wp = _weight_product(stat, self.weigh_by, self.bin_by)
    if wp is None:
      return None
    w_da, bin_dims = wp
    dep = tuple(d for d in (w_da.dims if w_da is not None else ()) if d not in bin_dims)
    if len(cache) > 16:
      cache.clear()
      hints.clear()
    hints[frame] = dep
    cache[key_for(dep)] = (wp, plugins)
    return wp

  @staticmethod
  def _w_token(w_da):
    """Identity of a weight product for result caches: a token object that lives and dies with `w_da` (an id() could be
    reused by a later, different W; a content hash of a 721 x 1440 x 34 product would cost more than the reduction)."""
    if w_da is None:
      return None
    tok = w_da.__dict__.get('_wbx_token')
    if tok is None:
      tok = w_da.__dict__['_wbx_token'] = _Token()
    return tok

  def _cache_key(self, w_da, bin_dims, use_mask, skipna, extra=()):
    """Key of a finished fused reduction on its FusedGroup: everything the sums depend on besides the group's inputs.
    W enters through its token, so an aggregator whose weigh_by / bin_by changed, or another aggregator that happens to
    sit at a recycled address, can never be served sums computed with a different W."""
    return (tuple(sorted(self.reduce_dims, key=str)), use_mask, skipna, tuple(bin_dims), self._w_token(w_da), extra)

  def _reduce_lazy(self, stat: lazy.LazyStatistic, w_da, bin_dims, use_mask, skipna):
    grp = stat._group  # pylint: disable=protected-access
    mean_dims = stat._mean_dims  # pylint: disable=protected-access
    ens_params = stat._ens_params  # pylint: disable=protected-access
    # (the family a deterministic group launches grows from DET3 to DET6 when a climatology statistic joins it: sums
    # cached before that hold three lanes only)
    family = grp.inputs_and_func()[1] if grp.kind == 'det' else 0
    extra = (tuple(sorted(ens_params.items())) if ens_params else (), mean_dims, family)
    key = self._cache_key(w_da, bin_dims, use_mask, skipna, extra)
    hit = grp.cache.get(key)
    if hit is None and grp.kind == 'ens' and stat._lane != lazy.ENS_LANE['CRPSSpread']:  # pylint: disable=protected-access
      # only the spread lane depends on (algorithm, fair): every other lane is served by whatever ensemble
      # launch already ran for this aggregator, else by the cheaper rank-form kernel.
      want_skip = bool(ens_params and ens_params.get('skipna'))
      for k2, v2 in grp.cache.items():
        if k2[:-1] == key[:-1] and k2[-1][1] == mean_dims and dict(k2[-1][0]).get('skipna', False) == want_skip:
          hit = v2
          break
      if hit is None:
        ens_params = {'algo': _hip.ENS_PAIRWISE if want_skip else _hip.ENS_SORT, 'fair': True, 'skipna': want_skip}
        ahead = getattr(grp, 'spread_params', None)  # (aggregate_statistics: the variant the group's spread lane will want)
        if ahead is not None and bool(ahead.get('skipna')) == want_skip:
          ens_params = dict(ahead)
        key = self._cache_key(w_da, bin_dims, use_mask, skipna, (tuple(sorted(ens_params.items())), mean_dims, family))
        hit = grp.cache.get(key)
    if hit is None:
      sib = getattr(grp, 'twin_sibling', None)
      sib = sib() if sib is not None else None
      if sib is not None and grp.kind == 'ens' and not use_mask and self.masked and engine.ENS_TWIN_MASK:
        # the masked sibling's launch also yields this group's (unmasked) sums: run it first if it has not run yet
        sparams = dict(getattr(sib, 'spread_params', None) or ens_params or {'algo': _hip.ENS_SORT, 'fair': True, 'skipna': False})
        skey = self._cache_key(w_da, bin_dims, True, skipna, (tuple(sorted(sparams.items())), mean_dims, 0))
        if skey not in sib.cache and not sparams.get('skipna'):
          sib.cache[skey] = sib.reduce(self.reduce_dims, w_da, bin_dims, use_mask=True, skipna=skipna, ens_params=sparams,
                                       extra_reduce=mean_dims)
      hit = grp.reduce(self.reduce_dims, w_da, bin_dims, use_mask=use_mask, skipna=skipna, ens_params=ens_params,
                       extra_reduce=mean_dims)
      grp.cache[key] = hit
    values, counts, out_dims = hit
    scale = 1.0
    for d in mean_dims:  # mean over d == sum over d / n; the count carries the same factor
      scale /= grp.sizes[d]
    names = stat._coord_names  # pylint: disable=protected-access
    frame_coords = grp.coords if names is None else {k: v for k, v in grp.coords.items() if k in names}
    return values, counts, out_dims, frame_coords, stat._lane, scale  # pylint: disable=protected-access

  def _reduce_categorical(self, stat: 'lazy.LazyCategorical', w_da, bin_dims, use_mask, skipna):
    """Indicator statistics: every category is a lane of ONE launch (wbx_cat_partial); the categories come back as
    the statistic's trailing dimension."""
    grp = stat._group  # pylint: disable=protected-access
    cat_dim = stat._cat_dim  # pylint: disable=protected-access
    # every category is a lane of one launch: `values` / `counts` are (category,) + out_dims views of its output
    values, counts, out_dims = grp.reduce(self.reduce_dims, w_da, bin_dims, use_mask=use_mask, skipna=skipna)
    # NaN thresholds (deterministic.py:293-294) are NaN indicators inside the kernel already: poisoned sums, or counted
    # out under skipna, exactly where a valid point is involved
    dims_in = (cat_dim,) + tuple(out_dims)
    kernel_dims = tuple(d for d in tuple(grp.dims) + (cat_dim,) if d in dims_in) + tuple(bin_dims)
    cat_axis = kernel_dims.index(cat_dim)
    # (a threshold field that adds no dim, or several: the category axis is dropped / split into them -- views either way)
    final_dims = kernel_dims[:cat_axis] + tuple(stat.cat_dims) + kernel_dims[cat_axis + 1:]
    coords = {k: x for k, x in stat._coords.items() if set(x[0]) <= set(final_dims) and k != 'mask'}  # pylint: disable=protected-access
    if w_da is not None:
      for k, x in w_da._coords.items():  # pylint: disable=protected-access
        if set(x[0]) <= set(final_dims):
          coords.setdefault(k, x)
    order = [dims_in.index(d) for d in kernel_dims]
    pending = engine.deferred_active() is not None

    def mk(a):
      a = np.transpose(np.asarray(a, dtype=np.float64), order)
      a = stat.split_categories(a if pending else np.ascontiguousarray(a), cat_axis)
      return xr.DataArray(a, dims=final_dims, coords=coords, name=stat.name,
                          attrs=stat.attrs, _raw_coords=True)
    return AggregationState(mk(values), mk(counts))

  def _reduce_spectrum(self, stat: 'spectra.LazySpectrum', w_da, bin_dims):
    """Weighted mean of zonal spectra over rows (time, latitude, ...) without materialising per-row spectra:
    the row weights ride into the |F|^2 reduction kernel (csrc/wbx_spectrum.hip)."""
    k_dim = stat._k_dim  # pylint: disable=protected-access
    if k_dim in set(self.reduce_dims):
      return None  # summing over wavenumber: use the generic path on the materialised spectrum
    row_dims = [d for d in stat.dims if d != k_dim]
    if w_da is not None and (k_dim in w_da.dims or stat._lon_dim in w_da.dims):  # pylint: disable=protected-access
      if stat._lon_dim in w_da.dims:  # pylint: disable=protected-access
        raise ValueError('bin masks / weights that depend on longitude cannot be applied to a zonal spectrum')
      return None
    kept = [d for d in row_dims if d not in set(self.reduce_dims)]
    w = w_da if w_da is not None else xr.DataArray(np.float64(1.0))
    bins = [dict(zip(bin_dims, idx)) for idx in np.ndindex(*[w.sizes[b] for b in bin_dims])] if bin_dims else [{}]
    vals, cnts = [], []
    sizes = {d: stat.sizes[d] for d in row_dims}
    if len(bins) > 1 and engine.deferred_active() is not None:  # the per-bin results are stacked on the host right away
      with engine.synchronous_results():
        return self._reduce_spectrum(stat, w_da, bin_dims)
    for sel in bins:
      wb = w.isel(sel) if sel else w
      arr, dims = stat.reduce_rows(wb, kept)
      vals.append(arr)
      # sum of the row weights per kept row: data independent, kept with the weight object like the device tables
      ckey = (tuple(row_dims), tuple(sizes[d] for d in row_dims), tuple(kept))
      cstore = wb.__dict__.setdefault('_wbx_spectrum_counts', {})
      c = cstore.get(ckey)
      if c is None:
        full = np.broadcast_to(xr._bcast_data(wb.astype(np.float64), row_dims, sizes), [sizes[d] for d in row_dims])  # pylint: disable=protected-access
        c = full.sum(axis=tuple(i for i, d in enumerate(row_dims) if d not in kept))
        if len(cstore) > 8:
          cstore.clear()
        cstore[ckey] = c
      cnts.append(np.broadcast_to(c[..., None], arr.shape))
    out_dims = tuple(dims) + tuple(bin_dims)
    bshape = [w.sizes[b] for b in bin_dims]
    if len(vals) == 1:  # a view: under deferred_results() the numbers arrive when the state's fence is waited on
      v = vals[0].reshape(list(vals[0].shape) + bshape)
    else:
      v = np.stack(vals, axis=-1).reshape(list(vals[0].shape) + bshape)
    # the stacked sums of row weights are data independent too: one read-only array per (weights, frame) while results are
    # being accumulated in HBM (the chunk loop then counts how often it met that very array instead of copying and adding
    # 4 MB per statistic and chunk, engine.Accumulation.capture); a private copy otherwise
    fkey = ('stacked', tuple(row_dims), tuple(sizes[d] for d in row_dims), tuple(kept), tuple(bin_dims), tuple(vals[0].shape))
    fstore = w.__dict__.setdefault('_wbx_spectrum_counts', {})
    c = fstore.get(fkey)
    if c is None:
      c = np.array(np.stack(cnts, axis=-1).reshape(list(vals[0].shape) + bshape), dtype=np.float64)
      c.flags.writeable = False
      fstore[fkey] = c
    if engine.accumulation_active() is None:
      c = c.copy()
    coords = {k: x for k, x in stat._coords.items() if set(x[0]) <= set(out_dims)}  # pylint: disable=protected-access
    if w_da is not None:
      for k, x in w_da._coords.items():  # pylint: disable=protected-access
        if set(x[0]) <= set(out_dims):
          coords.setdefault(k, x)
    mk = lambda a: xr.DataArray(a, dims=out_dims, coords=coords, name=stat.name, _raw_coords=True)
    return AggregationState(mk(np.asarray(v, dtype=np.float64)), mk(c))

  def _reduce_materialised(self, stat: xr.DataArray, w_da, bin_dims, use_mask, skipna):
    """Any DataArray (user-defined statistics, numpy or torch payload): the PASS1 family."""
    mask = stat.coords['mask'] if use_mask else None
    if mask is not None:
      mask = xr.DataArray(mask.data if xr._is_torch(mask.data) else mask.values.astype(bool), dims=mask.dims)  # pylint: disable=protected-access
    plain = xr.DataArray(stat.data, dims=stat.dims)
    if not xr._is_float(plain.data):  # pylint: disable=protected-access
      plain = plain.astype(np.float64)
    return engine.reduce_statistics('det', [plain], stat.dims, stat.sizes, self.reduce_dims, w_da, bin_dims,
                                    func=_hip.PASS1, mask=mask, skipna=skipna)


def compute_metric_values_for_single_chunk(metrics, aggregator: Aggregator, predictions, targets) -> xr.Dataset:
  """statistics -> aggregate -> metric values for one chunk (aggregation.py:411-435)."""
  statistics = metrics_base.compute_unique_statistics_for_all_metrics(metrics, predictions, targets)
  return aggregator.aggregate_statistics(statistics).metric_values(metrics)
