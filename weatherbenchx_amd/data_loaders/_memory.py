"""In-memory chunk loaders (re-exported by the package and its `base` / `xarray_loaders` / `latency_wrappers` modules) with the reference's names and `load_chunk` contract (counterpart of
weatherbenchX/data_loaders/base.py:58-170 and weatherbenchX/data_loaders/xarray_loaders.py:58-460).

The reference's loaders read zarr / NetCDF through xarray + dask, neither of which exists in this image; what a scoring job
needs from them is the CONTRACT -- (init_times, lead_times) -> {variable: DataArray} whose prediction and target chunks
broadcast against each other -- and that is what these classes restate over an `xarray_lite.Dataset` that is already open
(`ds=`; host arrays or tensors resident in HBM, in which case every selection below is a device gather and the chunk never
visits the host).  For fields in files use `loaders.PredictionsFromFiles` / `loaders.TargetsFromFiles` (page-locked reads
feeding the copy stream); `path=` here raises and says so.

  PredictionsFromXarray               `.sel(init_time=.., lead_time=..)`                         xarray_loaders.py:176-221
  TargetsFromXarray                   gather at valid_time = init_time + lead_time               xarray_loaders.py:224-275
  ClimatologyFromXarray               gather at (dayofyear, hour) of the valid time              xarray_loaders.py:278-330
  PersistenceFromXarray               the analysis at init_time repeated along lead_time         xarray_loaders.py:333-352
  ProbabilisticClimatologyFromXarray  one member per year at the same day-of-year / hour         xarray_loaders.py:355-432
  ConstantLoader                      the same dataset for every chunk                           xarray_loaders.py:435-449
  ConstantLatencyWrapper, XarrayConstantLatencyWrapper, MultipleConstantLatencyWrapper
                                      forecasts as they were AVAILABLE at the queried time       latency_wrappers.py:25-336
"""
from __future__ import annotations

from typing import Any, Callable, Hashable, Iterable, Mapping, Optional, Union

import numpy as np

from weatherbenchx_amd import data as wdata
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd import xarray_tree

add_nan_mask_to_data = wdata.add_nan_mask_to_data  # (data_loaders/base.py:25-56)

LeadTimes = Optional[Union[np.ndarray, slice]]


class DataLoader:
  """Shared `load_chunk`: source -> process_chunk_fn -> interpolation -> nan mask -> values as coordinate, in the
  reference's order (data_loaders/base.py:119-170).  `compute` is accepted for signature parity; nothing here is lazy."""

  def __init__(self, interpolation=None, compute: bool = True, add_nan_mask: bool = False,
               process_chunk_fn: Optional[Callable[[Mapping[Hashable, xr.DataArray]], Mapping[Hashable, xr.DataArray]]] = None,
               add_values_to_coords: bool = False):
    self._interpolation = interpolation
    self._compute = compute
    self._add_nan_mask = add_nan_mask
    self._process_chunk_fn = process_chunk_fn
    self._add_values_to_coords = add_values_to_coords

  def _load_chunk_from_source(self, init_times: np.ndarray, lead_times: LeadTimes = None):
    raise NotImplementedError()

  def load_chunk(self, init_times: np.ndarray, lead_times: LeadTimes = None,
                 reference: Optional[Mapping[Hashable, xr.DataArray]] = None) -> Mapping[Hashable, xr.DataArray]:
    chunk = self._load_chunk_from_source(init_times, lead_times)
    if self._process_chunk_fn is not None:
      chunk = self._process_chunk_fn(chunk)
    if self._interpolation is not None:
      chunk = self._interpolation.interpolate(chunk, reference)
    if self._add_nan_mask:
      chunk = add_nan_mask_to_data(chunk)
    if self._add_values_to_coords:
      chunk = xarray_tree.map_structure(_with_values_coord, chunk)
    return chunk


def _with_values_coord(da: xr.DataArray) -> xr.DataArray:
  out = da._replace()  # pylint: disable=protected-access
  out._coords['values_as_coord'] = (tuple(da.dims), da.data)  # pylint: disable=protected-access
  return out


def _rename_dataset(ds: xr.Dataset, rename_dimensions: Optional[Union[Mapping[str, str], str]] = 'ecmwf',
                    rename_variables: Optional[Mapping[str, str]] = None,
                    convert_lat_lon_to_latitude_longitude: bool = True) -> xr.Dataset:
  """Dimension / variable renaming of xarray_loaders.py:26-55: 'ecmwf' maps time -> init_time and prediction_timedelta ->
  lead_time on a forecast dataset, time -> valid_time on an analysis."""
  names = ds.coords
  if convert_lat_lon_to_latitude_longitude and 'lat' in names and 'lon' in names:
    ds = ds.rename({'lat': 'latitude', 'lon': 'longitude'})
  if isinstance(rename_dimensions, str):
    if rename_dimensions != 'ecmwf':
      raise ValueError('rename_dimensions must be either "ecmwf", a dict or None.')
    if 'prediction_timedelta' in names:
      ds = ds.rename({'time': 'init_time', 'prediction_timedelta': 'lead_time'})
    else:
      ds = ds.rename({'time': 'valid_time'})
  elif isinstance(rename_dimensions, Mapping):
    ds = ds.rename(dict(rename_dimensions))
  elif rename_dimensions is not None:
    raise ValueError('rename_dimensions must be either "ecmwf", a dict or None.')
  if rename_variables is not None:
    ds = ds.rename(dict(rename_variables))
  return ds


class XarrayDataLoader(DataLoader):
  """Base of the dataset-backed loaders (xarray_loaders.py:58-158).  The dataset is prepared once, on the first chunk:
  preprocessing_fn -> renaming -> variable subset -> sel_kwargs."""

  def __init__(self, path: Optional[str] = None, ds: Optional[xr.Dataset] = None, variables: Optional[Iterable[str]] = None,
               sel_kwargs: Optional[Mapping[str, Any]] = None,
               rename_dimensions: Optional[Union[Mapping[str, str], str]] = 'ecmwf',
               automatically_convert_lat_lon_to_latitude_longitude: bool = True,
               rename_variables: Optional[Mapping[str, str]] = None,
               preprocessing_fn: Optional[Callable[[xr.Dataset], xr.Dataset]] = None, **kwargs):
    if path is not None and ds is not None:
      raise ValueError('Only one of path or ds can be specified, not both.')
    if path is None and ds is None:
      raise ValueError('Either path or ds must be specified.')
    if path is not None:
      raise NotImplementedError(
          f'{type(self).__name__}(path=...): zarr / xarray readers are not part of this build; open the data yourself and '
          'pass ds=, or use loaders.PredictionsFromFiles / loaders.TargetsFromFiles for .npy and NetCDF-3 files.')
    self._ds = ds if isinstance(ds, xr.Dataset) else xr.Dataset(dict(ds))
    self._variables = variables
    self._sel_kwargs = sel_kwargs
    self._rename_dimensions = rename_dimensions
    self._automatically_convert_lat_lon_to_latitude_longitude = automatically_convert_lat_lon_to_latitude_longitude
    self._rename_variables = rename_variables
    self._preprocessing_fn = preprocessing_fn
    self._preprocessed = False
    super().__init__(**kwargs)

  def maybe_prepare_dataset(self):
    if self._preprocessed:
      return
    ds = self._ds
    if self._preprocessing_fn is not None:
      ds = self._preprocessing_fn(ds)
    ds = _rename_dataset(ds, self._rename_dimensions, self._rename_variables,
                         self._automatically_convert_lat_lon_to_latitude_longitude)
    if self._variables is not None:
      ds = ds[list(self._variables)]
    if self._sel_kwargs is not None:
      ds = ds.sel(**self._sel_kwargs)
    self._ds = ds
    self._preprocessed = True

  def load_chunk(self, init_times, lead_times=None, reference=None):
    self.maybe_prepare_dataset()
    return super().load_chunk(init_times, lead_times, reference)


def _ns(times, kind: str) -> np.ndarray:
  return np.asarray(times, dtype=f'{"datetime64" if kind == "M" else "timedelta64"}[ns]')


def _time_axes(init_times, lead_times):
  """init_time [I], lead_time [L] and valid_time [I, L] as DataArrays (what `xr.DataArray(init) + xr.DataArray(lead)`
  builds at xarray_loaders.py:259-262)."""
  init, lead = _ns(init_times, 'M'), _ns(lead_times, 'm')
  valid = xr.DataArray(init[:, None] + lead[None, :], dims=('init_time', 'lead_time'),
                       coords={'init_time': init, 'lead_time': lead})
  return init, lead, valid


def _exact(lead_times) -> bool:
  return lead_times is not None and not isinstance(lead_times, slice)


class PredictionsFromXarray(XarrayDataLoader):
  """Forecasts with init_time and lead_time dims.  Exact lead times or an (inclusive, by label) lead-time slice; None returns
  every lead time (xarray_loaders.py:209-221)."""

  def _load_chunk_from_source(self, init_times, lead_times=None):
    init = _ns(init_times, 'M')
    if lead_times is None:
      return self._ds.sel(init_time=init)
    if isinstance(lead_times, slice):
      return self._ds.sel(init_time=init, lead_time=lead_times)
    return self._ds.sel(init_time=init, lead_time=_ns(lead_times, 'm'))


class TargetsFromXarray(XarrayDataLoader):
  """Analyses / observations with ONE time dim `valid_time`.  With exact lead times the chunk is gathered at init + lead and has
  dims (init_time, lead_time, ...) plus the 2-D coordinate valid_time; without lead times the init times are the valid times
  (xarray_loaders.py:249-275)."""

  def _load_chunk_from_source(self, init_times, lead_times=None):
    if isinstance(lead_times, slice):
      raise ValueError('Lead time slice not supported for target data loaders.')
    if lead_times is None:
      return self._ds.sel(valid_time=_ns(init_times, 'M'))
    _, _, valid = _time_axes(init_times, lead_times)
    return self._ds.sel(valid_time=valid)


class ClimatologyFromXarray(XarrayDataLoader):
  """A climatology indexed by `climatology_time_coords` (default dayofyear, hour) served as if it were a forecast: each
  (init, lead) takes the entry of its valid time (xarray_loaders.py:278-330)."""

  def __init__(self, climatology_time_coords: Iterable[str] = ('dayofyear', 'hour'),
               rename_dimensions: Optional[Union[Mapping[str, str], str]] = None, **kwargs):
    super().__init__(rename_dimensions=rename_dimensions, **kwargs)
    self._climatology_time_coords = tuple(climatology_time_coords)

  def _load_chunk_from_source(self, init_times, lead_times=None):
    if isinstance(lead_times, slice):
      raise ValueError('Lead time slice not yet supported for climatology data loaders.')
    if lead_times is None:
      init = _ns(init_times, 'M')
      when = xr.DataArray(init, dims=('init_time',), coords={'init_time': init})
    else:
      _, _, when = _time_axes(init_times, lead_times)
    return self._ds.sel({c: getattr(when.dt, c) for c in self._climatology_time_coords})


class PersistenceFromXarray(XarrayDataLoader):
  """The analysis valid at init_time repeated for every lead time (xarray_loaders.py:333-352)."""

  def _load_chunk_from_source(self, init_times, lead_times=None):
    if not _exact(lead_times):
      raise ValueError('Exact lead times must be specified for persistence data loader.')
    chunk = self._ds.sel(valid_time=_ns(init_times, 'M')).expand_dims({'lead_time': _ns(lead_times, 'm')})
    return chunk.rename({'valid_time': 'init_time'})


class ProbabilisticClimatologyFromXarray(XarrayDataLoader):
  """Every year in [start_year, end_year] is one ensemble member: for each valid time, the analysis at the same day-of-year
  and hour of that year.  The time of member `y` is 1 January of y plus (dayofyear - 1) days plus the hour, so day 366 of a
  leap year lands on 1 January of y + 1 for a non-leap y (xarray_loaders.py:355-432)."""

  def __init__(self, start_year: int, end_year: int, ensemble_dim: str = 'number', **kwargs):
    super().__init__(**kwargs)
    self._start_year = start_year
    self._end_year = end_year
    self._ensemble_dim = ensemble_dim

  def _load_chunk_from_source(self, init_times, lead_times=None):
    if not _exact(lead_times):
      raise ValueError('Exact lead times must be specified for persistence data loader.')
    init, lead, valid = _time_axes(init_times, lead_times)
    hours = (valid.dt.dayofyear.values - 1) * 24 + valid.dt.hour.values  # [I, L]
    offset = hours.astype('timedelta64[h]').astype('timedelta64[ns]')
    years = np.arange(self._start_year, self._end_year + 1)
    starts = np.array([np.datetime64(str(y)) for y in years]).astype('datetime64[ns]')
    member_times = xr.DataArray(starts[:, None, None] + offset[None], dims=(self._ensemble_dim, 'init_time', 'lead_time'),
                                coords={self._ensemble_dim: np.arange(years.size), 'init_time': init, 'lead_time': lead})
    return self._ds.sel(valid_time=member_times)


class ConstantLoader(DataLoader):
  """Returns the same dataset whatever the times (xarray_loaders.py:435-449)."""

  def __init__(self, constant_ds):
    super().__init__()
    self._constant_ds = constant_ds

  def _load_chunk_from_source(self, init_times, lead_times=None):
    return self._constant_ds


# ---- operational latency (weatherbenchX/data_loaders/latency_wrappers.py:25-336) -----------------------------------------

def _concat_chunks(chunks, dim):
  return xarray_tree.map_structure(lambda *parts: xr.concat(list(parts), dim=dim), *chunks)


class ConstantLatencyWrapper(DataLoader):
  """Serves a forecast the way it would have been AVAILABLE: a forecast with nominal init time t0 is issued at t0 + latency,
  and a query for (init_time, lead_time) is answered from the most recent nominal init whose issue time is <= init_time, at
  lead_time + (init_time - t0); the chunk is relabelled to the queried init / lead times and the per-init pieces are joined along
  `concat_dim` ('index' for sparse data).  latency_wrappers.py:25-190."""

  def __init__(self, data_loader: DataLoader, latency: np.timedelta64, nominal_init_times: np.ndarray,
               concat_dim: str = 'init_time'):
    self.data_loader = data_loader
    self.latency = latency
    self.nominal_init_times = nominal_init_times
    self._concat_dim = concat_dim
    # pylint: disable=protected-access
    super().__init__(interpolation=data_loader._interpolation, compute=data_loader._compute,
                     add_nan_mask=data_loader._add_nan_mask, process_chunk_fn=data_loader._process_chunk_fn)
    # pylint: enable=protected-access

  def get_available_init_time(self, init_time: np.datetime64):
    """The latest nominal init time already issued at `init_time`, or None."""
    nominal = _ns(self.nominal_init_times, 'M')
    if nominal.size == 0:
      return None
    issued = nominal + np.asarray(self.latency).astype('timedelta64[ns]') <= np.asarray(init_time).astype('datetime64[ns]')
    if not issued.any():
      return None
    return nominal[issued].max()

  def _load_chunk_from_source(self, init_times, lead_times=None):
    if isinstance(self.data_loader, XarrayDataLoader):
      self.data_loader.maybe_prepare_dataset()  # (we call its _load_chunk_from_source, not load_chunk)
    if lead_times is None:
      raise ValueError('Latency adjustement is only valid with lead times.')
    if isinstance(lead_times, slice):
      raise ValueError('Latency adjustment needs exact lead times, not a slice.')
    lead_times = _ns(lead_times, 'm')
    pieces = []
    for init_time in _ns(init_times, 'M'):  # one by one: the offset depends on the init time
      available = self.get_available_init_time(init_time)
      if available is None:
        raise ValueError(f'No available init time found for init time {init_time}.')
      offset = init_time - available  # >= latency
      raw = self.data_loader._load_chunk_from_source(np.array([available]), lead_times + offset)  # pylint: disable=protected-access

      def relabel(x, offset=offset):
        x = xr.as_dataarray(x)
        # by offset, so that this also holds for sparse data, whose init / lead times are coordinates over `index`
        return x.assign_coords(init_time=x.coords['init_time'] + offset).assign_coords(lead_time=x.coords['lead_time'] - offset)

      pieces.append(xarray_tree.map_structure(relabel, raw))
    return _concat_chunks(pieces, self._concat_dim)


class XarrayConstantLatencyWrapper(ConstantLatencyWrapper):
  """ConstantLatencyWrapper whose nominal init times are the `init_time_dim` coordinate of the wrapped loader's dataset
  (latency_wrappers.py:193-236)."""

  def __init__(self, data_loader: XarrayDataLoader, latency: np.timedelta64, init_time_dim: str = 'init_time',
               concat_dim: str = 'init_time'):
    self._init_time_dim = init_time_dim
    self._nominal_init_times_set = False
    super().__init__(data_loader, latency, nominal_init_times=np.array([], dtype='datetime64[ns]'), concat_dim=concat_dim)

  def maybe_set_nominal_init_times(self):
    if self._nominal_init_times_set:
      return
    self.data_loader.maybe_prepare_dataset()
    self.nominal_init_times = np.asarray(self.data_loader._ds[self._init_time_dim].values)  # pylint: disable=protected-access
    self._nominal_init_times_set = True

  def _load_chunk_from_source(self, init_times, lead_times=None):
    self.maybe_set_nominal_init_times()
    return super()._load_chunk_from_source(init_times, lead_times)

  def get_available_init_time(self, init_time):
    self.maybe_set_nominal_init_times()
    return super().get_available_init_time(init_time)


class MultipleConstantLatencyWrapper(DataLoader):
  """Several latency-wrapped loaders with different nominal init times (00/12 UTC and 06/18 UTC archives, say): each queried
  init time is answered by the loader with the most recent available nominal init; on a tie the LARGER latency wins (more
  lookahead).  The pieces come from the wrapped loaders' `load_chunk`, i.e. already interpolated.
  latency_wrappers.py:239-336."""

  def __init__(self, data_loaders: list, concat_dim: str = 'init_time'):
    super().__init__()
    self._data_loaders = list(data_loaders)
    self._concat_dim = concat_dim

  def _load_chunk_from_source(self, init_times, lead_times=None):
    raise NotImplementedError('This should only be called for the individual data loaders.')

  def _get_data_loader(self, init_time):
    best, best_key = None, None
    for loader in self._data_loaders:
      available = loader.get_available_init_time(init_time)
      if available is None:
        continue
      key = (np.asarray(init_time - available).astype('timedelta64[ns]').astype(np.int64),
             -np.asarray(loader.latency).astype('timedelta64[ns]').astype(np.int64))
      if best_key is None or key < best_key:  # strict: the first of fully equal loaders, like a stable argsort
        best, best_key = loader, key
    if best is None:
      raise ValueError('No available init time found for init time %s.' % init_time)
    return best

  def load_chunk(self, init_times, lead_times=None, reference=None):
    pieces = []
    for init_time in _ns(init_times, 'M'):
      pieces.append(self._get_data_loader(init_time).load_chunk(np.array([init_time]), lead_times, reference))
    return _concat_chunks(pieces, self._concat_dim)
