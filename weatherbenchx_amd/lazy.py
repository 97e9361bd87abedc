"""Deferred per-point statistics: how an unfused plugin API is lowered to fused kernels.

The reference contract is "Statistic.compute returns a full-resolution array, the Aggregator reduces it"
(weatherbenchX/metrics/base.py:135-158, weatherbenchX/aggregation.py:337-366).  Materialising that
array on the GPU would double HBM traffic, so the built-in statistics return a `LazyStatistic`: a
DataArray (dims, coords, mask coordinate all present) whose payload is only computed -- by the HIP map
kernel -- if somebody actually reads `.data`/`.values`.  Statistics built from the same
(predictions, targets[, climatology]) arrays share one `FusedGroup`; the Aggregator asks the group for
all lanes at once, which is one stage-1 launch that reads every input exactly once.
"""
from __future__ import annotations

import os
import weakref
from typing import Sequence

import numpy as np

from weatherbenchx_amd import _hip
from weatherbenchx_amd import engine
from weatherbenchx_amd import planner
from weatherbenchx_amd import replay
from weatherbenchx_amd import xarray_lite as xr

DET_LANE = {'Error': 0, 'AbsoluteError': 1, 'SquaredError': 2, 'SquaredPredictionAnomaly': 3,
            'SquaredTargetAnomaly': 4, 'AnomalyCovariance': 5}
ENS_LANE = {'CRPSSkill': 0, 'CRPSSpread': 1, 'EnsembleVariance': 2, 'UnbiasedEnsembleMeanSquaredError': 3,
            'EnsembleMeanSquaredError': 4}


_frame_memo: list = [None]  # (weakref p, weakref t, mutations, frame without drop_dims): the last (p, t) frame that was computed


def _stat_frame(arrays: Sequence[xr.DataArray], drop_dims=()):
  """dims / sizes / coords of the broadcast of `arrays` (what `a - b` would carry), without computing it."""
  if len(arrays) == 2:
    # `_aligned(p, t)` checks the frame of a pair and the FusedGroup built right after asks for it again (minus the member
    # dim): one walk over the coordinates serves both
    memo = _frame_memo[0]
    a, b = arrays
    muts = (a.__dict__.get('_mutations', 0), b.__dict__.get('_mutations', 0))
    if memo is not None and memo[0]() is a and memo[1]() is b and memo[2] == muts:
      dims, sizes, coords = memo[3]
    elif drop_dims:
      return _stat_frame_walk(arrays, drop_dims)  # (nobody has looked at the whole frame of this pair: it may not even exist)
    else:
      dims, sizes, coords = _stat_frame_walk(arrays)
      _frame_memo[0] = (weakref.ref(a), weakref.ref(b), muts, (dims, sizes, coords))
    if not drop_dims:
      return dims, dict(sizes), dict(coords)
    drop = set(drop_dims)
    if drop & set(dims):
      # a dropped dim changes which coordinates collide: walk again with it (the member dim is on p only: not the case here)
      if any(d in b.dims for d in drop) or any(d in a.dims and d in b.dims for d in drop):
        return _stat_frame_walk(arrays, drop_dims)
    return (tuple(d for d in dims if d not in drop), {d: n for d, n in sizes.items() if d not in drop},
            {k: v for k, v in coords.items() if not (set(v[0]) & drop)})
  return _stat_frame_walk(arrays, drop_dims)


def _stat_frame_walk(arrays: Sequence[xr.DataArray], drop_dims=()):
  dims, sizes = [], {}
  for a in arrays:
    for d, n in a.sizes.items():
      if d in drop_dims:
        continue
      if d not in dims:
        dims.append(d)
        sizes[d] = n
      elif sizes[d] != n:
        if all(d in x._coords for x in arrays if d in x.dims):  # pylint: disable=protected-access
          raise _NeedsAlignment(d)  # labeled dims of different length: inner join like xarray arithmetic
        raise ValueError(f'cannot broadcast: size mismatch along {d!r} ({sizes[d]} vs {n})')
  coords = {}
  dropped = set()
  for a in arrays:
    for k, (cd, cv) in a._coords.items():  # pylint: disable=protected-access
      if set(cd) & set(drop_dims) or k in dropped:
        continue
      if k in coords:
        d0, v0 = coords[k]
        if d0 != cd or not xr._values_equal(v0, cv):  # pylint: disable=protected-access
          if k in dims:
            raise _NeedsAlignment(k)
          del coords[k]
          dropped.add(k)
      else:
        coords[k] = (cd, cv)
  return tuple(dims), sizes, coords


class _NeedsAlignment(Exception):
  pass


class ClimatologyRef(xr.LazyPickleMixin, xr.DataArray):
  """The climatology aligned with valid_time (metrics/base.py:382-403) -- as an index table, not a copy.

  Built-in statistics only read `.source` / `.positions` and let the stage-1 kernel gather whole climatology fields
  in place.  For user-defined `PerVariableStatisticWithClimatology` plugins it still IS the aligned DataArray the
  reference would hand over: touching `.data` / doing arithmetic materialises the gather (on the device for torch
  payloads)."""

  origin = None    # (host climatology, its positions) when `source` is a slab pool (climatology_cache.SlabCache.ref)
  activate = None  # slab pool: the launch streams wait for the slabs still on their way (called before the launches)

  def __init__(self, source: xr.DataArray, over_dims: tuple, positions: dict):
    self.source = source          # dims e.g. (dayofyear, hour, level, latitude, longitude)
    self.over_dims = tuple(over_dims)    # statistic dims the selection varies over, e.g. (init_time, lead_time)
    self.positions = positions    # climatology dim -> int positions, shape = sizes of over_dims
    self._data = None
    self._dims = self.aligned_dims()
    self.name = source.name
    self.attrs = dict(source.attrs)
    self._coords = {k: v for k, v in source._coords.items() if not set(v[0]) & set(positions)}  # pylint: disable=protected-access

  def aligned_dims(self):
    return self.over_dims + tuple(d for d in self.source.dims if d not in self.positions)

  @property
  def shape(self):
    first = np.asarray(next(iter(self.positions.values())))
    return tuple(first.shape) + tuple(self.source.sizes[d] for d in self.source.dims if d not in self.positions)

  @property
  def dtype(self):
    return self.source.dtype

  @property
  def data(self):
    if self._data is None:
      self._data = self.aligned_view().data
    return self._data

  def aligned_view(self) -> xr.DataArray:
    """The aligned climatology as a plain labeled array (a gather of whole fields)."""
    # (behind a slab pool: gathered from the HOST climatology the pool caches -- what a user-defined statistic is handed is
    #  the reference's `climatology.sel(...).compute()`, metrics/base.py:396-403)
    src, positions = self.origin if self.origin is not None else (self.source, self.positions)
    order = [src.dims.index(d) for d in positions] + [i for i, d in enumerate(src.dims) if d not in positions]
    data = xr._transpose(src.data, order)  # pylint: disable=protected-access
    gather = tuple(np.asarray(positions[d]).reshape(-1) for d in positions)
    if xr._is_torch(data):  # pylint: disable=protected-access
      import torch  # pylint: disable=g-import-not-at-top
      gather = tuple(torch.as_tensor(g, device=data.device) for g in gather)
    g = data[gather]
    shape = tuple(np.asarray(next(iter(positions.values()))).shape)
    g = g.reshape(shape + tuple(xr._shape(g)[1:]))  # pylint: disable=protected-access
    coords = {k: v for k, v in src._coords.items() if not set(v[0]) & set(positions)}  # pylint: disable=protected-access
    return xr.DataArray(g, dims=self.aligned_dims(), coords=coords, _raw_coords=True)


class FusedGroup:
  """All fused statistics over one (predictions, targets[, climatology]) triple."""

  def __init__(self, kind: str, p: xr.DataArray, t: xr.DataArray, ens=None, cat=None):
    self.kind = kind
    self.p, self.t = p, t
    self.ens = ens  # {'member_dim', 'M'}
    self.cat = cat  # indicator statistics: {'func', 'ncat', 'thresholds', 'member_dim', 'M'}
    self.clim: ClimatologyRef | None = None
    self._clim_key = None
    drop = (ens['member_dim'],) if ens else ((cat['member_dim'],) if cat and cat.get('member_dim') else ())
    self.dims, self.sizes, self.coords = _stat_frame([p, t], drop_dims=drop)
    self.cache: dict = {}

  def attach_climatology(self, ref: ClimatologyRef, key) -> bool:
    if self.clim is None:
      # frame check: aligned climatology dims must already be statistic dims (it only broadcasts)
      for d in ref.aligned_dims():
        if d not in self.dims:
          return False
      for d in ref.source.dims:
        if d in self.dims and d in ref.source._coords and d in self.coords:  # pylint: disable=protected-access
          if not xr._values_equal(ref.source._coords[d][1], self.coords[d][1]):  # pylint: disable=protected-access
            return False
      self.clim, self._clim_key = ref, key
      return True
    return self._clim_key == key

  @property
  def mask(self) -> xr.DataArray | None:
    if 'mask' in self.coords:
      cd, cv = self.coords['mask']
      # a mask that was built in HBM (data.add_nan_mask_to_data on device payloads) is consumed there
      return xr.DataArray(cv if xr._is_torch(cv) else np.asarray(cv), dims=cd)  # pylint: disable=protected-access
    return None

  # -- execution ---------------------------------------------------------------------------------
  def inputs_and_func(self):
    if self.kind in ('ens', 'cat', 'ens2'):
      return [self.p, self.t], 0
    if self.clim is not None:
      return [self.p, self.t, self.clim.source], _hip.DET6
    return [self.p, self.t], _hip.DET3

  def reduce(self, reduce_dims, w_da, bin_dims, *, use_mask: bool, skipna: bool, ens_params=None, extra_reduce=()):
    inputs, func = self.inputs_and_func()
    mask = self.mask if use_mask else None
    ens = None
    if self.kind == 'ens':
      ens = dict(self.ens, **(ens_params or {}))
    if self.kind == 'ens2':
      return engine.reduce_statistics('ens2', inputs, self.dims, self.sizes, tuple(reduce_dims) + tuple(extra_reduce),
                                      w_da, bin_dims, mask=mask, skipna=skipna, ens=self.ens)
    if self.kind == 'cat':
      return engine.reduce_statistics('cat', inputs, self.dims, self.sizes, tuple(reduce_dims) + tuple(extra_reduce),
                                      w_da, bin_dims, mask=mask, skipna=skipna, cat=self.cat)
    return _reduce_with_gather(self.kind, inputs, self.dims, self.sizes, tuple(reduce_dims) + tuple(extra_reduce), w_da,
                               bin_dims, func=func, mask=mask, skipna=skipna, clim=self.clim, ens=ens)

  def materialise(self, lane: int, ens_params=None) -> np.ndarray:
    if self.kind == 'ens2':  # stage 1 with every dim kept IS the per-point statistic
      with engine.synchronous_results():
        values, _, out_dims = self.reduce((), None, (), use_mask=False, skipna=False)
      arr = np.asarray(values, np.float64)[lane]
      return np.ascontiguousarray(np.transpose(arr, [out_dims.index(d) for d in self.dims]))
    inputs, func = self.inputs_and_func()
    ens = dict(self.ens, **(ens_params or {})) if self.kind == 'ens' else None
    gather, inputs = _gather_spec(self.clim, inputs, self.dims)
    return engine.materialise(self.kind, inputs, self.dims, self.sizes, lane, func=func, gather=gather, ens=ens)


def gather_from_ref(clim: ClimatologyRef, dtype_code: int) -> planner.GatherSpec:
  """Element-offset gather table of an aligned climatology from its device layout (the climatology itself is uploaded once and
  cached on its DataArray)."""
  ctx = _hip.default_context()
  if clim.origin is not None and str(clim.source.dtype) != ('float32' if dtype_code == _hip.F32 else 'float64'):
    raise ValueError(f'the slab-cached climatology is {clim.source.dtype} but the statistic is evaluated in '
                     f"{'float32' if dtype_code == _hip.F32 else 'float64'}: cast the climatology (or the fields) to one dtype")
  if clim.activate is not None:
    clim.activate()  # slab pool: the launch streams wait for slabs that are still on the copy stream (asked for a chunk ahead)
  dev = engine._to_device(ctx, clim.source, dtype_code)  # pylint: disable=protected-access
  table = np.zeros(tuple(np.asarray(next(iter(clim.positions.values()))).shape), dtype=np.int64)
  for d, pos in clim.positions.items():
    table = table + np.asarray(pos, dtype=np.int64) * dev.layout.stride(d)
  return planner.GatherSpec(dims=tuple(clim.over_dims), table=table)


def _gather_spec(clim: ClimatologyRef | None, inputs, dims):
  """Element-offset gather table for the climatology input, from its device layout."""
  if clim is None:
    return None, inputs
  datas = [i.data for i in inputs]
  dtype_code = engine._common_dtype(datas)  # pylint: disable=protected-access
  return gather_from_ref(clim, dtype_code), inputs


def _regather(p_new, g):
  """replay.ChunkRecord: the plan variant of a recorded launch for the chunk whose predictions are `p_new` -- the climatology
  positions of ITS time labels (metrics/base.py: `_climatology_ref`, cached per label set), the gather table from them, the
  cached plan with that table swapped in."""
  from weatherbenchx_amd.metrics import base as metrics_base  # pylint: disable=g-import-not-at-top
  try:
    ref = metrics_base.PerVariableStatisticWithClimatology._climatology_ref(p_new, g['source'])  # pylint: disable=protected-access
    if tuple(ref.over_dims) != tuple(g['over_dims']):
      return None
    return g['replan'](gather_from_ref(ref, g['dtype_code']))
  except (ValueError, KeyError):
    return None


def _reduce_with_gather(kind, inputs, dims, sizes, reduce_dims, w_da, bin_dims, *, func, mask, skipna, clim, ens):
  gather, inputs = _gather_spec(clim, inputs, dims)
  rec = replay.active() if clim is not None else None
  if rec is not None:  # a chunk that is being recorded: what its gather plans have to be rebuilt from for the next chunk
    rec.gather_context = {'source': clim.origin[0] if clim.origin is not None else clim.source, 'p': inputs[0], 'over_dims': tuple(clim.over_dims),
                          'dtype_code': engine._common_dtype([i.data for i in inputs]), 'regather': _regather}  # pylint: disable=protected-access
  try:
    return engine.reduce_statistics(kind, inputs, dims, sizes, reduce_dims, w_da, bin_dims, func=func, mask=mask,
                                    skipna=skipna, gather=gather, ens=ens)
  except ValueError as e:
    if clim is None or 'innermost' not in str(e):
      raise
    # the selection runs along the contiguous dim: align the climatology first, then fuse as usual
    aligned = clim.aligned_view()
    inputs = list(inputs[:2]) + [aligned]
    return engine.reduce_statistics(kind, inputs, dims, sizes, reduce_dims, w_da, bin_dims, func=func, mask=mask,
                                    skipna=skipna, gather=None, ens=ens)
  finally:
    if rec is not None:
      rec.gather_context = None


def _group_for(kind: str, p: xr.DataArray, t: xr.DataArray, ens=None, clim_key=None, cat=None) -> FusedGroup:
  """The FusedGroup shared by every statistic built from these very (p, t) objects."""
  # (the table holds the group WEAKLY: the group holds p, and a strong p -> table -> group -> p cycle kept a chunk's result
  # buffers -- views of pooled page-locked / device memory cached on the group -- away from their pools until the cyclic
  # collector happened to run; a chunk loop then allocated fresh memory job after job: 2.5 instead of 1.0 ms per chunk)
  table = p.__dict__.setdefault('_wbx_groups', {})
  key = (kind, id(t), t.__dict__.get('_mutations', 0), ens['member_dim'] if ens else None, clim_key)
  hit = table.get(key)
  if hit is not None and hit[0]() is t:
    grp = hit[1]()
    if grp is not None:
      return grp
  grp = FusedGroup(kind, p, t, ens=ens, cat=cat)
  if len(table) > 8:  # entries of groups that have died
    for k in [k for k, v in table.items() if v[1]() is None]:
      del table[k]
  table[key] = (weakref.ref(t), weakref.ref(grp))
  return grp


class LazyStatistic(xr.LazyPickleMixin, xr.DataArray):
  """A per-point statistic that is a DataArray in every respect, evaluated on demand by a HIP kernel."""

  def __init__(self, group: FusedGroup, lane: int, name=None, ens_params=None, mean_dims=(), coord_names=None):
    # deliberately no super().__init__: the payload does not exist yet
    self._data = None
    self._dims = tuple(d for d in group.dims if d not in mean_dims)
    self.name = name
    self.attrs = {}
    dset = set(self._dims)
    # coord_names: statistics of the predictions alone (EnsembleVariance, CRPSSpread) carry the predictions' coordinates
    # only -- in particular not the targets' `mask` (probabilistic.py:250-273: `predictions.var(...)`)
    self._coords = {k: v for k, v in group.coords.items()
                    if set(v[0]) <= dset and (coord_names is None or k in coord_names)}
    self._group = group
    self._lane = lane
    self._ens_params = ens_params
    self._mean_dims = tuple(mean_dims)
    self._coord_names = coord_names

  @property
  def is_lazy(self) -> bool:
    return self._data is None

  @property
  def data(self):
    if self._data is None:
      arr = self._group.materialise(self._lane, self._ens_params)
      if self._mean_dims:
        axes = tuple(self._group.dims.index(d) for d in self._mean_dims)
        arr = arr.mean(axis=axes)
      self._data = arr
    return self._data

  @property
  def shape(self):
    return tuple(self._group.sizes[d] for d in self._dims)

  @property
  def dtype(self):
    return np.dtype(np.float64)

  def mean(self, dim=None, skipna=None, **kw):
    """Mean over a dim that nothing else depends on stays lazy (EnsembleAveragedStatistic,
    probabilistic.py:35-69): it becomes one more reduced dim of the fused launch."""
    dims = (dim,) if isinstance(dim, str) else tuple(dim or ())
    # only an explicit skipna=False stays lazy: the default (None) drops NaNs like DataArray.mean / xarray do
    if (self.is_lazy and dims and skipna is False and all(d in self._dims for d in dims)
        and self._group.kind == 'det'):
      return LazyStatistic(self._group, self._lane, name=self.name, ens_params=self._ens_params,
                           mean_dims=self._mean_dims + dims, coord_names=self._coord_names)
    return super().mean(dim, skipna=skipna, **kw)


class LazyEnsembleMean(xr.LazyPickleMixin, xr.DataArray):
  """`predictions.mean(ensemble_dim)` (wrappers.py:116-148) that remembers where it came from, so
  SquaredError of it is served by the ensemble kernel's lane 4 without a second pass over the members."""

  def __init__(self, source: xr.DataArray, ensemble_dim: str):
    self._data = None
    self._dims = tuple(d for d in source.dims if d != ensemble_dim)
    self.name = source.name
    self.attrs = dict(source.attrs)
    dset = set(self._dims)
    self._coords = {k: v for k, v in source._coords.items() if set(v[0]) <= dset}  # pylint: disable=protected-access
    self._source = source
    self._ensemble_dim = ensemble_dim

  @property
  def is_lazy(self) -> bool:
    return self._data is None

  @property
  def data(self):
    if self._data is None:
      self._data = xr.DataArray.mean(self._source, self._ensemble_dim, skipna=False).data
    return self._data

  @property
  def shape(self):
    return tuple(self._source.sizes[d] for d in self._dims)

  @property
  def dtype(self):
    return self._source.dtype


# ---- constructors used by the metric classes ---------------------------------------------------------
def _aligned(p: xr.DataArray, t: xr.DataArray):
  try:
    _stat_frame([p, t])
    return p, t
  except _NeedsAlignment:
    return xr.align(p, t, join='inner')


def det_statistic(stat_name: str, p, t, climatology_ref: ClimatologyRef | None = None, clim_key=None) -> xr.DataArray:
  p, t = xr.as_dataarray(p), xr.as_dataarray(t)
  if isinstance(p, LazyEnsembleMean) and p.is_lazy and stat_name == 'SquaredError':
    return ens_statistic('EnsembleMeanSquaredError', p._source, t, p._ensemble_dim)  # pylint: disable=protected-access
  table = p.__dict__.get('_wbx_groups')
  if not (table and any(k[0] == 'det' and k[1] == id(t) and k[2] == t.__dict__.get('_mutations', 0) and v[0]() is t
                        and v[1]() is not None for k, v in table.items())):
    p, t = _aligned(p, t)  # (a group of these very objects exists: an earlier statistic has checked their frames)
  # one group per (p, t): Error/AbsoluteError/SquaredError and the anomaly statistics of the FIRST
  # climatology share a launch; a second, different climatology gets its own group.
  grp = _group_for('det', p, t)
  if climatology_ref is not None and not grp.attach_climatology(climatology_ref, clim_key):
    grp = _group_for('det', p, t, clim_key=clim_key)
    if not grp.attach_climatology(climatology_ref, clim_key):
      raise ValueError('climatology does not broadcast against predictions/targets '
                       f'(climatology dims {climatology_ref.aligned_dims()}, statistic dims {grp.dims})')
  return LazyStatistic(grp, DET_LANE[stat_name], name=p.name)


def _same_mask(p: xr.DataArray, t: xr.DataArray) -> bool:
  pm, tm = p._coords.get('mask'), t._coords.get('mask')  # pylint: disable=protected-access
  if tm is None:
    return True
  return pm is not None and pm[0] == tm[0] and xr._values_equal(pm[1], tm[1])  # pylint: disable=protected-access


# True: CRPSSpread(use_sort=False) launches the register-tiled O(M^2) pair kernel (EnsOpF32<..., PAIRWISE>) instead of the
# rank-form kernel.  A measurement switch (bench.py's `pairwise_form`, tools/kbench.py): the results agree to ~1e-7.
PAIR_FORM_KERNEL = os.environ.get('WBX_ENS_PAIR_FORM', '0') == '1'


def ens_statistic(stat_name: str, p, t, ensemble_dim: str, *, use_sort=False, fair=True,
                  skipna_ensemble=False, member_only=False) -> xr.DataArray:
  """`member_only`: the statistic looks at `p` alone (EnsembleVariance, CRPSSpread); `t` is only the kernel's companion
  operand.  Its frame and coordinates are then those of `p` -- the reference computes it from the predictions, so a
  `mask` coordinate or an extra dim of the targets must not leak into it (Aggregator(masked=True) leaves such a statistic
  unmasked, aggregation.py:339-352).  While `t` has the frame of `p` the (p, t) launch of the other lanes is shared."""
  p, t = xr.as_dataarray(p), xr.as_dataarray(t)
  if ensemble_dim not in p.dims:
    raise ValueError(f'Dimension {ensemble_dim} not found in {p.dims}')
  if ensemble_dim in t.dims:
    raise ValueError(f'targets must not carry {ensemble_dim!r} here (select a member or use target_members())')
  coord_names = None
  if member_only:
    if not (set(t.dims) <= set(p.dims) and _same_mask(p, t)):
      t = first_member(p, ensemble_dim)  # a companion with exactly the predictions' frame
    coord_names = frozenset(p._coords)  # pylint: disable=protected-access
  table = p.__dict__.get('_wbx_groups')
  if not (table and any(k[0] == 'ens' and k[1] == id(t) and k[2] == t.__dict__.get('_mutations', 0) and k[3] == ensemble_dim
                        and v[0]() is t and v[1]() is not None for k, v in table.items())):
    p, t = _aligned(p, t)  # (a group of these very objects exists: an earlier statistic has checked their frames)
  m = p.sizes[ensemble_dim]
  grp = _group_for('ens', p, t, ens={'member_dim': ensemble_dim, 'M': m})
  # use_sort=False (the reference's default, probabilistic.py:644) asks for the O(M^2) pair form of the SAME number -- the
  # reference's own CRPSSpread.unique_name leaves use_sort out (probabilistic.py:189-192), i.e. it treats the two forms as
  # one statistic.  On this hardware the rank form is the cheaper way to that number (51 members: 0.31 against 0.51 ms per
  # 1.73 GB) and all five lanes of a (p, t) pair come out of ONE launch, so both settings run the rank-form kernel; the
  # register-tiled pair kernel stays reachable through PAIR_FORM_KERNEL (or WBX_ENS_PAIR_FORM=1) for measurements.
  # skipna_ensemble needs per-point member counts: the generic pair-form kernel (WBX_FLAG_SKIPNA_ENS).
  pair = bool(skipna_ensemble) or (not use_sort and PAIR_FORM_KERNEL)
  params = {'algo': _hip.ENS_PAIRWISE if pair else _hip.ENS_SORT, 'fair': bool(fair), 'skipna': bool(skipna_ensemble)}
  return LazyStatistic(grp, ENS_LANE[stat_name], name=p.name, ens_params=params, coord_names=coord_names)


class LazyCategorical(xr.LazyPickleMixin, xr.DataArray):
  """An indicator statistic (ErrorExceedance, EnsembleErrorExceedance, RankHistogram): the frame of (p, t) plus ONE new
  trailing dimension of categories.  The Aggregator reduces all categories in one launch (wbx_cat_partial); reading
  `.data` evaluates the same kernel without reducing anything.

  `split` = (dims, shape, coords): the kernel's category axis stands for these trailing dims -- none (a threshold field that
  adds no dimension: one category, squeezed out) or several (a field that adds two or more: stacked for the kernel)."""

  def __init__(self, group: FusedGroup, cat_dim: str, cat_coord, name=None, split=None):
    self._data = None
    self._split = None if split is None else (tuple(split[0]), tuple(int(n) for n in split[1]), dict(split[2]))
    self._dims = tuple(group.dims) + ((cat_dim,) if split is None else self._split[0])
    self.name = name
    self.attrs = {}
    self._coords = dict(group.coords)
    if cat_coord is not None and split is None:
      self._coords[cat_dim] = ((cat_dim,), np.asarray(cat_coord))
    if split is not None:
      for d, c in self._split[2].items():
        self._coords[d] = ((d,), np.asarray(c))
    self._group = group
    self._cat_dim = cat_dim

  @property
  def is_lazy(self) -> bool:
    return self._data is None

  @property
  def ncat(self) -> int:
    return int(self._group.cat['ncat'])

  @property
  def cat_dims(self) -> tuple:
    """The statistic's dims that the kernel's category axis stands for."""
    return (self._cat_dim,) if self._split is None else self._split[0]

  def split_categories(self, a: np.ndarray, axis: int) -> np.ndarray:
    """`a` with its category axis replaced by the dims it stands for -- always a view (an axis is split or dropped)."""
    if self._split is None:
      return a
    v = a.reshape(a.shape[:axis] + self._split[1] + a.shape[axis + 1:])
    if a.size and not np.may_share_memory(v, a):
      raise RuntimeError('internal: category split copied a pending result')
    return v

  @property
  def data(self):
    if self._data is None:
      grp = self._group
      with engine.synchronous_results():
        values, _, out_dims = grp.reduce((), None, (), use_mask=False, skipna=False)
      arr = np.moveaxis(np.asarray(values, np.float64), 0, -1)
      arr = np.transpose(arr, [out_dims.index(d) for d in grp.dims] + [len(out_dims)])
      self._data = self.split_categories(np.ascontiguousarray(arr), arr.ndim - 1)
    return self._data

  @property
  def shape(self):
    return tuple(self._group.sizes[d] for d in self._group.dims) + ((self.ncat,) if self._split is None else self._split[1])

  @property
  def dtype(self):
    return np.dtype(np.float64)


ENS2_LANE = {'CRPSSkill': 0, 'UnbiasedEnsembleMeanSquaredError': 1}


def ens2_statistic(stat_name: str, p, t, ensemble_dim: str, *, skipna_ensemble: bool) -> xr.DataArray:
  """Statistics of an ensemble of predictions against an ensemble of TARGETS with per-point member counts on both sides
  (wbx_ens2_partial): both lanes of a (p, t) pair come out of one launch."""
  p, t = xr.as_dataarray(p), xr.as_dataarray(t)
  for da in (p, t):
    if ensemble_dim not in da.dims:
      raise ValueError(f'Dimension {ensemble_dim} not found in {da.dims}')
  try:  # frames without the member axes (the two ensembles have their own sizes and member labels)
    _stat_frame([p.isel({ensemble_dim: 0}, drop=True), t.isel({ensemble_dim: 0}, drop=True)])
  except _NeedsAlignment:
    p0, t0 = xr.align(p.isel({ensemble_dim: 0}, drop=True), t.isel({ensemble_dim: 0}, drop=True), join='inner')
    p = p.sel({d: p0[d].values for d in p0.dims if d in p0.coords})
    t = t.sel({d: t0[d].values for d in t0.dims if d in t0.coords})
  ens = {'member_dim': ensemble_dim, 'M': p.sizes[ensemble_dim], 'N': t.sizes[ensemble_dim], 'skipna': bool(skipna_ensemble)}
  grp = _group_for('ens2', p, t, ens=ens, clim_key=('ens2', bool(skipna_ensemble)))
  return LazyStatistic(grp, ENS2_LANE[stat_name], name=p.name)


def cat_statistic(func: int, p, t, cat_dim: str, cat_coord, *, thresholds=None, ensemble_dim=None, threshold_field=None,
                  split=None) -> xr.DataArray:
  p, t = xr.as_dataarray(p), xr.as_dataarray(t)
  if ensemble_dim is not None:
    if ensemble_dim not in p.dims:
      raise ValueError(f'Dimension {ensemble_dim} not found in {p.dims}')
    if ensemble_dim in t.dims:
      raise ValueError(f'targets must not carry {ensemble_dim!r} here')
  p, t = _aligned(p, t)
  if cat_dim in p.dims or cat_dim in t.dims:
    raise ValueError(f'{cat_dim!r} is already a dimension of the inputs')
  thr = None if thresholds is None else np.asarray(thresholds, np.float64).reshape(-1)
  if threshold_field is not None:
    # thresholds that depend on the statistic's dims (wbx_cat_exceed_field): a float64 DataArray over (some of) the frame's dims
    # + `cat_dim`, consumed in place through its strides
    threshold_field = xr.as_dataarray(threshold_field)
    extra = [d for d in threshold_field.dims if d != cat_dim and d not in p.dims and d not in t.dims]
    if extra or cat_dim not in threshold_field.dims:
      raise ValueError(f'thresholds over {threshold_field.dims} do not broadcast against the statistic (extra dims {extra})')
    for d in threshold_field.dims:  # same labels on the shared dims (the reference's comparison aligns on them)
      ref = p if d in p.dims else t
      if d != cat_dim and d in threshold_field.coords and d in ref.coords and not xr._values_equal(threshold_field.coords[d].values, ref.coords[d].values):  # pylint: disable=protected-access
        raise ValueError(f'thresholds and inputs carry different {d!r} coordinates: select the common labels first')
    if not xr._is_float(threshold_field.data) or str(threshold_field.dtype) != 'float64':  # pylint: disable=protected-access
      threshold_field = threshold_field.astype(np.float64)
    thr = None
  ncat = (p.sizes[ensemble_dim] + 1) if func == _hip.CAT_RANK else (int(thr.size) if thr is not None else threshold_field.sizes[cat_dim])
  cat = {'func': int(func), 'ncat': ncat, 'thresholds': thr, 'member_dim': ensemble_dim,
         'M': p.sizes[ensemble_dim] if ensemble_dim else 1, 'thr_field': threshold_field, 'cat_dim': cat_dim}
  key = ('cat', int(func), cat_dim, None if thr is None else thr.tobytes(), None if threshold_field is None else id(threshold_field))
  grp = _group_for('cat', p, t, ens=None, clim_key=(key, ensemble_dim), cat=cat)
  return LazyCategorical(grp, cat_dim, cat_coord, name=p.name, split=split)


def first_member(p: xr.DataArray, ensemble_dim: str) -> xr.DataArray:
  """Member 0 of `p` as a (cached) view: the companion operand of statistics that only look at the predictions.  (Not
  `target_members(p)[0]`: that builds a view of EVERY member -- 51 of them, 1.2 ms of host time per chunk on the public
  probabilistic configuration with a mask.)"""
  cache = p.__dict__.setdefault('_wbx_member0', {})
  if ensemble_dim not in cache:
    # (the general isel walks every axis and every coordinate: ~40 us per chunk of the masked public configuration; a view of
    #  member 0 needs one slice and the coordinates that do not live on the member dim)
    ax = p.dims.index(ensemble_dim)
    data = p.data[(slice(None),) * ax + (0,)]
    coords = {k: v for k, v in p._coords.items() if ensemble_dim not in v[0]}  # pylint: disable=protected-access
    cache[ensemble_dim] = xr.DataArray._assemble(data, p.dims[:ax] + p.dims[ax + 1:], coords, name=p.name, attrs=p.attrs)  # pylint: disable=protected-access
  return cache[ensemble_dim]


def target_members(t: xr.DataArray, ensemble_dim: str):
  """The members of an ensemble-valued target as separate (cached) views: statistics that are linear in the target
  member index are evaluated per member and averaged (LinearCombination)."""
  cache = t.__dict__.setdefault('_wbx_members', {})
  if ensemble_dim not in cache:
    cache[ensemble_dim] = [t.isel({ensemble_dim: j}, drop=True) for j in range(t.sizes[ensemble_dim])]
  return cache[ensemble_dim]


class LinearCombination(xr.LazyPickleMixin, xr.DataArray):
  """scale * sum(coeff_i * term_i) of lazy statistics on the same frame.  The weighted reduction is linear, so the
  Aggregator reduces every term with its own fused launch and combines the accumulators (WindVectorSquaredError =
  SE(u) + SE(v), deterministic.py:174-219; CRPSSkill / UnbiasedEnsembleMeanSquaredError against an ensemble of
  targets, probabilistic.py:134-145, 320-336)."""

  def __init__(self, terms, scale: float = 1.0, name=None, coeffs=None):
    first = terms[0]
    self._coeffs = [1.0] * len(terms) if coeffs is None else [float(c) for c in coeffs]
    if len(self._coeffs) != len(terms):
      raise ValueError('one coefficient per term')
    self._data = None
    self._dims = first.dims
    self.name = name
    self.attrs = {}
    self._coords = dict(first._coords)  # pylint: disable=protected-access
    self._terms = list(terms)
    self._scale = float(scale)

  @property
  def is_lazy(self):
    return self._data is None

  @property
  def data(self):
    if self._data is None:
      total = self._terms[0].data * self._coeffs[0] if self._coeffs[0] != 1.0 else self._terms[0].data
      for t, c in zip(self._terms[1:], self._coeffs[1:]):
        total = total + (t.data * c if c != 1.0 else t.data)
      self._data = total * self._scale if self._scale != 1.0 else total
    return self._data

  @property
  def shape(self):
    return self._terms[0].shape

  @property
  def dtype(self):
    return np.dtype(np.float64)
