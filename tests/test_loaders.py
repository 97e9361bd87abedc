"""File-backed loaders (weatherbenchx_amd/loaders.py): `.npy` memory maps and NetCDF-3 files read chunk by chunk into
(page-locked) buffers, predictions as (init, lead) blocks, targets gathered at valid_time = init + lead -- the semantics of
PredictionsFromXarray / TargetsFromXarray (data_loaders/xarray_loaders.py:176-316) -- and a chunked evaluation over them
against the oracle on the whole arrays."""
import os

import numpy as np
import pytest

from oracle import wbx_oracle as O
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import loaders
from weatherbenchx_amd import pipeline
from weatherbenchx_amd import time_chunks
from weatherbenchx_amd import weighting
from weatherbenchx_amd.metrics import deterministic

RTOL = 1e-6


def _write_zarr(path, arr, chunks, compressor, sep='.', dtype='<f4', skip=()):
  """A zarr v2 array directory: .zarray + one file per chunk (full-size chunks at the edges, as zarr writes them)."""
  import itertools
  import json
  os.makedirs(path, exist_ok=True)
  meta = {'zarr_format': 2, 'shape': list(arr.shape), 'chunks': list(chunks), 'dtype': dtype, 'compressor': compressor,
          'fill_value': 'NaN', 'order': 'C', 'filters': None}
  if sep != '.':
    meta['dimension_separator'] = sep
  json.dump(meta, open(os.path.join(path, '.zarray'), 'w'))
  grid = [range(-(-n // c)) for n, c in zip(arr.shape, chunks)]
  for idx in itertools.product(*grid):
    if idx in skip:
      continue
    block = np.full(chunks, np.nan, np.dtype(dtype))
    sl = tuple(slice(i * c, min((i + 1) * c, n)) for i, c, n in zip(idx, chunks, arr.shape))
    block[tuple(slice(0, s.stop - s.start) for s in sl)] = arr[sl]
    raw = block.tobytes()
    if compressor is not None:
      import gzip
      import zlib
      raw = zlib.compress(raw, 1) if compressor['id'] == 'zlib' else gzip.compress(raw, 1)
    f = os.path.join(path, *[str(i) for i in idx]) if sep == '/' else os.path.join(path, '.'.join(str(i) for i in idx))
    os.makedirs(os.path.dirname(f), exist_ok=True)
    open(f, 'wb').write(raw)


def _write(tmp_path, fmt, nlat=19, nlon=36):
  rng = np.random.default_rng(5)
  ninit, nlead, nlev = 5, 3, 2
  init_times = np.datetime64('2020-01-01T00', 'ns') + np.arange(ninit) * np.timedelta64(12, 'h')
  lead_times = (np.arange(nlead) * 12).astype('timedelta64[h]').astype('timedelta64[ns]')
  times = np.datetime64('2020-01-01T00', 'ns') + np.arange(ninit + nlead) * np.timedelta64(12, 'h')
  lat, lon, level = np.linspace(-90, 90, nlat), np.arange(nlon) * (360.0 / nlon), np.array([500, 850])
  pv = (rng.normal(size=(ninit, nlead, nlev, nlon, nlat)) + 280).astype(np.float32)   # latitude fastest, like the archives
  tv = (rng.normal(size=(times.size, nlev, nlon, nlat)) + 280).astype(np.float32)
  tv[3, 1, 4, 5] = np.nan
  dims, coords = ('level', 'longitude', 'latitude'), {'level': level, 'longitude': lon, 'latitude': lat}
  if fmt == 'npy':
    pp, tp = os.path.join(tmp_path, 'p.npy'), os.path.join(tmp_path, 't.npy')
    np.save(pp, pv)
    np.save(tp, tv)
    src_p, src_t = {'z': pp}, {'z': tp}
  elif fmt.startswith('zarr'):
    # zarr v2 stores written by hand (the package is not in the image): a group directory per loader with the variable inside;
    # chunks that do not divide the grid (edge chunks), one (init, lead) / one time per chunk like the public archives
    comp = {'zarr': None, 'zarr_zlib': {'id': 'zlib', 'level': 1}, 'zarr_slash': {'id': 'gzip', 'level': 1}}[fmt]
    sep = '/' if fmt == 'zarr_slash' else '.'
    pp, tp = os.path.join(tmp_path, 'p.zarr'), os.path.join(tmp_path, 't.zarr')
    _write_zarr(os.path.join(pp, 'z'), pv, (1, 2, 1, 25, nlat), comp, sep, dtype='<f4')
    _write_zarr(os.path.join(tp, 'z'), tv, (1, nlev, nlon, 7), comp, sep, dtype='>f4' if fmt == 'zarr_zlib' else '<f4')
    src_p, src_t = {'z': (pp, 'z')}, {'z': (tp, 'z')}
  else:
    from scipy.io import netcdf_file
    pp, tp = os.path.join(tmp_path, 'p.nc'), os.path.join(tmp_path, 't.nc')
    for path, arr, names in ((pp, pv, ('init_time', 'lead_time') + dims), (tp, tv, ('time',) + dims)):
      f = netcdf_file(path, 'w', version=2)
      for n, s in zip(names, arr.shape):
        f.createDimension(n, s)
      v = f.createVariable('z', np.float32, names)
      v[:] = arr
      f.close()
    src_p, src_t = {'z': (pp, 'z')}, {'z': (tp, 'z')}
  return src_p, src_t, init_times, lead_times, times, dims, coords, pv, tv


@pytest.mark.parametrize('fmt', ['npy', 'nc', 'zarr', 'zarr_zlib', 'zarr_slash'])
def test_chunks_from_files_have_the_reference_frames(tmp_path, fmt):
  src_p, src_t, init_times, lead_times, times, dims, coords, pv, tv = _write(str(tmp_path), fmt)
  lp = loaders.PredictionsFromFiles(src_p, init_times, lead_times, dims, coords, pinned=False)
  lt = loaders.TargetsFromFiles(src_t, times, dims, coords, pinned=False, add_nan_mask=True)
  p = lp.load_chunk(init_times[[3, 1]], lead_times[1:])['z']
  assert p.dims == ('init_time', 'lead_time') + dims
  np.testing.assert_array_equal(p.values, pv[[3, 1]][:, 1:])
  np.testing.assert_array_equal(p['init_time'].values, init_times[[3, 1]])
  t = lt.load_chunk(init_times[[3, 1]], lead_times[1:])['z']
  assert t.dims == ('init_time', 'lead_time') + dims
  for a, i in enumerate((3, 1)):
    for b, l in enumerate((1, 2)):
      np.testing.assert_array_equal(t.values[a, b], tv[i + l])       # valid_time = init + lead (12 h steps)
  np.testing.assert_array_equal(t.coords['valid_time'].values, init_times[[3, 1]][:, None] + lead_times[1:][None, :])
  np.testing.assert_array_equal(np.asarray(t.coords['mask'].values), ~np.isnan(t.values))  # data_loaders/base.py:25-56
  assert p.values.dtype == np.float32 and p.values.dtype.isnative
  with pytest.raises(KeyError):
    lp.load_chunk(np.array(['2031-01-01'], dtype='datetime64[ns]'), lead_times)
  with pytest.raises(ValueError, match='slice'):
    lt.load_chunk(init_times[:1], slice(None))
  assert lp.timings['chunks'] == 1 and lp.timings['bytes'] == p.values.nbytes


@pytest.mark.parametrize('fmt', ['npy', 'nc', 'zarr_zlib'])
def test_chunked_evaluation_from_files_against_the_oracle(backend, tmp_path, fmt):
  src_p, src_t, init_times, lead_times, times, dims, coords, pv, tv = _write(str(tmp_path), fmt)
  lp = loaders.PredictionsFromFiles(src_p, init_times, lead_times, dims, coords, pinned=backend == 'hip')
  lt = loaders.TargetsFromFiles(src_t, times, dims, coords, pinned=backend == 'hip', add_nan_mask=True)
  metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE()}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()], masked=True)
  tc = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=2, lead_time_chunk_size=2)
  got = pipeline.evaluate_chunks(tc, loaders.load_chunk_fn(lp, lt), metrics, agg, prefetch=1 if backend == 'hip' else 0)[None].metric_values(metrics)
  valid_idx = (np.arange(init_times.size)[:, None] + np.arange(lead_times.size)[None, :])  # 12 h init step == 12 h lead step
  tfull = tv[valid_idx]
  fdims = ('init_time', 'lead_time') + dims
  w = (O.grid_area_weights(coords['latitude']), ('latitude',))
  ok = ~np.isnan(tfull)
  for name, lane in (('rmse', O.squared_error(pv, tfull)), ('mae', O.absolute_error(pv, tfull))):
    sws, sw, od = O.aggregate(lane, fdims, ['init_time', 'latitude', 'longitude'], weights=[w], mask=ok, mask_dims=fdims)
    want = np.sqrt(sws / sw) if name == 'rmse' else sws / sw
    np.testing.assert_allclose(np.asarray(got[f'{name}.z'].transpose(*od).values), want, rtol=RTOL, err_msg=name)


def test_lead_time_slice_is_selected_by_label_with_inclusive_bounds(tmp_path):
  """`TimeChunks` hands lead-time INTERVALS over as slices of timedelta64 (time_chunks.py:54-61, 97-108) and the
  reference resolves them with `.sel(lead_time=slice)` -- by label, both ends included (xarray_loaders.py:199-200)."""
  src_p, _, init_times, lead_times, _, dims, coords, pv, _ = _write(str(tmp_path), 'npy')
  lp = loaders.PredictionsFromFiles(src_p, init_times, lead_times, dims, coords, pinned=False)
  tc = time_chunks.TimeChunks(init_times, slice(np.timedelta64(12, 'h'), np.timedelta64(24, 'h')), init_time_chunk_size=2)
  init_chunk, lead_chunk = next(iter(tc))
  assert isinstance(lead_chunk, slice)
  p = lp.load_chunk(init_chunk, lead_chunk)['z']
  np.testing.assert_array_equal(p['lead_time'].values, lead_times[1:3])  # 12 h and 24 h: the stop label is part of the slice
  np.testing.assert_array_equal(p.values, pv[:2, 1:3])
  p = lp.load_chunk(init_chunk, slice(None, np.timedelta64(13, 'h')))['z']  # open start, a stop between two labels
  np.testing.assert_array_equal(p['lead_time'].values, lead_times[:2])
  p = lp.load_chunk(init_chunk, slice(np.timedelta64(1, 'h'), None))['z']
  np.testing.assert_array_equal(p['lead_time'].values, lead_times[1:])
  assert lp.load_chunk(init_chunk, slice(np.timedelta64(100, 'h'), None))['z'].shape[1] == 0


@pytest.mark.parametrize('fmt', ['npy', 'nc', 'zarr', 'zarr_zlib'])
def test_device_layout_transposes_on_the_way_into_the_chunk(tmp_path, fmt):
  """`device_layout='lon_fastest'` over a [.., longitude, latitude] archive: the chunk arrives [.., latitude, longitude],
  C-contiguous, value for value the stored field transposed (wbx_host_transpose: grids that are no multiple of the 32 x 32
  tile or the 8 x 8 block, big-endian NetCDF decoded on the way, threads on)."""
  src_p, src_t, init_times, lead_times, times, dims, coords, pv, tv = _write(str(tmp_path), fmt, nlat=43, nlon=70)
  for threads in (1, 3):
    lp = loaders.PredictionsFromFiles(src_p, init_times, lead_times, dims, coords, pinned=False, device_layout='lon_fastest', threads=threads)
    lt = loaders.TargetsFromFiles(src_t, times, dims, coords, pinned=False, add_nan_mask=True, device_layout='lon_fastest', threads=threads)
    lp._gather_swapped.__func__  # noqa: B018
    p = lp.load_chunk(init_times[[3, 1]], lead_times[1:])['z']
    assert p.dims == ('init_time', 'lead_time', 'level', 'latitude', 'longitude') and p.values.flags['C_CONTIGUOUS']
    np.testing.assert_array_equal(p.values, np.swapaxes(pv[[3, 1]][:, 1:], -1, -2))
    t = lt.load_chunk(init_times[[3, 1]], lead_times[1:])['z']
    assert t.dims == ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
    for a, i in enumerate((3, 1)):
      for b, l in enumerate((1, 2)):
        np.testing.assert_array_equal(t.values[a, b], np.swapaxes(tv[i + l], -1, -2))
    np.testing.assert_array_equal(np.asarray(t.coords['mask'].values), ~np.isnan(t.values))
    t0 = lt.load_chunk(times[2:4])['z']  # init times as valid times
    np.testing.assert_array_equal(t0.values, np.swapaxes(tv[2:4], -1, -2))
  # a layout the store already has is left alone; one it cannot be brought to by exchanging the last two dims is refused
  same = loaders.PredictionsFromFiles(src_p, init_times, lead_times, dims, coords, pinned=False, device_layout='lat_fastest')
  assert same.load_chunk(init_times[:1], lead_times)['z'].dims == ('init_time', 'lead_time') + dims
  with pytest.raises(ValueError, match='do not end in'):
    loaders.PredictionsFromFiles(src_p, init_times, lead_times, ('longitude', 'level', 'latitude'), coords, device_layout='lon_fastest')
  with pytest.raises(ValueError, match='device_layout'):
    loaders.PredictionsFromFiles(src_p, init_times, lead_times, dims, coords, device_layout='tiled')


def test_host_transpose_entry_point():
  """The raw C entry point: float32 and float64, extents around the tile / block sizes, argument checks."""
  import ctypes as C
  from weatherbenchx_amd import _hip
  lib = _hip.load_library()
  rng = np.random.default_rng(0)
  for dt in (np.float32, np.float64):
    for batch, rows, cols in ((1, 1, 1), (2, 7, 9), (1, 32, 32), (3, 33, 31), (1, 64, 8), (2, 40, 100), (1, 1440 // 8, 721 // 7)):
      src = rng.standard_normal((batch, rows, cols)).astype(dt)
      dst = np.full((batch, cols, rows), np.nan, dt)
      assert lib.wbx_host_transpose(dst.ctypes.data, src.ctypes.data, batch, rows, cols, src.itemsize) == 0
      np.testing.assert_array_equal(dst, np.swapaxes(src, 1, 2))
  a = np.zeros(64, np.float32)
  assert lib.wbx_host_transpose(a.ctypes.data, a.ctypes.data, 1, 8, 8, 4) != 0 and b'overlap' in lib.wbx_last_error()
  assert lib.wbx_host_transpose(a.ctypes.data, None, 1, 8, 8, 4) != 0
  assert lib.wbx_host_transpose(a.ctypes.data, a.ctypes.data + 128, 1, 4, 4, 2) != 0 and b'elem_bytes' in lib.wbx_last_error()
  assert lib.wbx_host_transpose(None, None, 0, 8, 8, 4) == 0


@pytest.mark.parametrize('layout', [None, 'lon_fastest'])
def test_both_device_layouts_from_one_latitude_fastest_file(backend, tmp_path, layout):
  """One [.., longitude, latitude] archive, evaluated as it is stored and through the transposing loader: RMSE / MAE and the
  zonal spectra of predictions and targets agree with the oracle on the whole arrays either way."""
  from weatherbenchx_amd import spectra
  src_p, src_t, init_times, lead_times, times, dims, coords, pv, tv = _write(str(tmp_path), 'npy', nlat=19, nlon=36)
  tv = np.nan_to_num(tv, nan=280.0)
  np.save(src_t['z'], tv)
  lp = loaders.PredictionsFromFiles(src_p, init_times, lead_times, dims, coords, pinned=backend == 'hip', device_layout=layout)
  lt = loaders.TargetsFromFiles(src_t, times, dims, coords, pinned=backend == 'hip', device_layout=layout)
  metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE()}
  spec = {'sp': spectra.ZonalPowerSpectrum('predictions'), 'st': spectra.ZonalPowerSpectrum('targets')}
  area = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  zonal = aggregation.Aggregator(reduce_dims=['init_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
  tc = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=2)
  load = loaders.load_chunk_fn(lp, lt)
  out = pipeline.evaluate_passes(tc, [('det', load, metrics, area), ('spec', load, spec, zonal)], prefetch=1 if backend == 'hip' else 0)
  got = out['det'][None].metric_values(metrics)
  valid_idx = (np.arange(init_times.size)[:, None] + np.arange(lead_times.size)[None, :])
  tfull = tv[valid_idx]
  fdims = ('init_time', 'lead_time') + dims
  w = (O.grid_area_weights(coords['latitude']), ('latitude',))
  for name, lane in (('rmse', O.squared_error(pv, tfull)), ('mae', O.absolute_error(pv, tfull))):
    sws, sw, od = O.aggregate(lane, fdims, ['init_time', 'latitude', 'longitude'], weights=[w])
    want = np.sqrt(sws / sw) if name == 'rmse' else sws / sw
    np.testing.assert_allclose(np.asarray(got[f'{name}.z'].transpose(*od).values), want, rtol=RTOL, err_msg=name)
  sgot = out['spec'][None].metric_values(spec)
  for key, field in (('sp.z', pv), ('st.z', tfull)):
    s = O.zonal_power_spectrum(np.swapaxes(field, -1, -2).astype(np.float64), lon_axis=-1)  # [init, lead, level, lat, k]
    wl = O.grid_area_weights(coords['latitude'])[None, None, None, :, None]
    want = (s * wl).sum(axis=(0, 3)) / (np.ones_like(s) * wl).sum(axis=(0, 3))
    res = sgot[key]
    kdim = [d for d in res.dims if d not in ('lead_time', 'level')][0]
    np.testing.assert_allclose(np.asarray(res.transpose('lead_time', 'level', kdim).values), want, rtol=2e-4, atol=2e-6 * want.max(), err_msg=key)


def test_zarr_reader_edges(tmp_path):
  """ZarrArray: a missing chunk reads as the fill value, leading indices select lazily, codecs that need numcodecs are refused
  with the way out, zarr v3 / filters are refused."""
  import json
  rng = np.random.default_rng(1)
  a = rng.standard_normal((3, 5, 7)).astype(np.float32)
  path = os.path.join(str(tmp_path), 'a')
  _write_zarr(path, a, (2, 2, 4), {'id': 'zlib', 'level': 1}, skip=((1, 1, 0),))
  z = loaders.ZarrArray(path)
  assert z.shape == (3, 5, 7) and z.dtype == np.float32 and z.nbytes == a.nbytes
  want = a.copy()
  want[2:3, 2:4, 0:4] = np.nan  # chunk (1, 1, 0) covers [2:4, 2:4, 0:4]; the array ends at 3 along axis 0
  np.testing.assert_array_equal(z.read(), want)
  np.testing.assert_array_equal(np.asarray(z[1]), a[1])
  np.testing.assert_array_equal(z[2][4].read(), a[2, 4])
  np.testing.assert_array_equal(z[-1][0].read(), a[2, 0])
  out = np.empty((5, 7), np.float64)
  z[0].read(out)
  np.testing.assert_array_equal(out, a[0].astype(np.float64))
  with pytest.raises(IndexError):
    z[3]
  with pytest.raises(TypeError):
    z[0:2]
  for comp, text in (({'id': 'blosc', 'cname': 'lz4'}, 'numcodecs'), ({'id': 'zstd'}, 'numcodecs')):
    p2 = os.path.join(str(tmp_path), comp['id'])
    _write_zarr(p2, a, (2, 2, 4), None)
    meta = json.load(open(os.path.join(p2, '.zarray')))
    meta['compressor'] = comp
    json.dump(meta, open(os.path.join(p2, '.zarray'), 'w'))
    with pytest.raises(ValueError, match=text):
      loaders.ZarrArray(p2)
  meta['compressor'], meta['zarr_format'] = None, 3
  json.dump(meta, open(os.path.join(p2, '.zarray'), 'w'))
  with pytest.raises(ValueError, match='zarr v2'):
    loaders.ZarrArray(p2)
  with pytest.raises(FileNotFoundError):
    loaders.ZarrArray(str(tmp_path))


def _write_baselines(tmp_path, fmt, nlat=19, nlon=36):
  """A [dayofyear, hour, level, lon, lat] climatology and a 6-hourly analysis series around new year 2020/21 (a leap year:
  day 366 is used), as .npy / NetCDF / zarr."""
  rng = np.random.default_rng(11)
  nlev, nhour = 2, 4
  clim = (rng.normal(size=(366, nhour, nlev, nlon, nlat)) + 270).astype(np.float32)
  times = np.datetime64('2020-12-29T00', 'ns') + np.arange(28) * np.timedelta64(6, 'h')
  tv = (rng.normal(size=(times.size, nlev, nlon, nlat)) + 270).astype(np.float32)
  lat, lon, level = np.linspace(-90, 90, nlat), np.arange(nlon) * (360.0 / nlon), np.array([500, 850])
  dims, coords = ('level', 'longitude', 'latitude'), {'level': level, 'longitude': lon, 'latitude': lat}
  if fmt == 'npy':
    cp, tp = os.path.join(tmp_path, 'c.npy'), os.path.join(tmp_path, 't.npy')
    np.save(cp, clim)
    np.save(tp, tv)
    src_c, src_t = {'z': cp}, {'z': tp}
  elif fmt == 'zarr':
    cp, tp = os.path.join(tmp_path, 'c.zarr'), os.path.join(tmp_path, 't.zarr')
    _write_zarr(os.path.join(cp, 'z'), clim, (1, 1, nlev, nlon, nlat), {'id': 'zlib', 'level': 1}, '.', dtype='<f4')
    _write_zarr(os.path.join(tp, 'z'), tv, (1, nlev, nlon, 7), None, '.', dtype='<f4')
    src_c, src_t = {'z': (cp, 'z')}, {'z': (tp, 'z')}
  else:
    from scipy.io import netcdf_file
    cp, tp = os.path.join(tmp_path, 'c.nc'), os.path.join(tmp_path, 't.nc')
    for path, arr, names in ((cp, clim, ('dayofyear', 'hour') + dims), (tp, tv, ('time',) + dims)):
      f = netcdf_file(path, 'w', version=2)
      for n, s in zip(names, arr.shape):
        f.createDimension(n, s)
      v = f.createVariable('z', np.float32, names)
      v[:] = arr
      f.close()
    src_c, src_t = {'z': (cp, 'z')}, {'z': (tp, 'z')}
  return src_c, src_t, times, dims, coords, clim, tv


def _doy_hour(valid):
  day = valid.astype('datetime64[D]')
  return (day - valid.astype('datetime64[Y]').astype('datetime64[D]')).astype(int), (valid - day).astype('timedelta64[h]').astype(int) // 6


@pytest.mark.parametrize('fmt', ['npy', 'nc', 'zarr'])
@pytest.mark.parametrize('layout', [None, 'lon_fastest'])
def test_climatology_and_persistence_read_as_predictions(tmp_path, fmt, layout):
  """ClimatologyFromXarray / PersistenceFromXarray (xarray_loaders.py:266-337; their tests xarray_loaders_test.py:60-112):
  frames, values at dayofyear / hour of the valid time across a leap-year end, the reference's refusals."""
  src_c, src_t, times, dims, coords, clim, tv = _write_baselines(str(tmp_path), fmt)
  kw = dict(pinned=False, device_layout=layout)
  lc = loaders.ClimatologyFromFiles(src_c, dims, coords, **kw)
  lp = loaders.PersistenceFromFiles(src_t, times, dims, coords, **kw)
  out_dims = dims if layout is None else ('level', 'latitude', 'longitude')
  swap = (lambda a: a) if layout is None else (lambda a: np.swapaxes(a, -1, -2))
  init_times = np.array(['2020-12-30T12', '2020-12-29T00', '2020-12-31T18'], dtype='datetime64[ns]')
  lead_times = np.array([0, 18, 48], dtype='timedelta64[h]').astype('timedelta64[ns]')
  c = lc.load_chunk(init_times, lead_times)['z']
  assert c.dims == ('init_time', 'lead_time') + out_dims
  valid = init_times[:, None] + lead_times[None, :]
  d, h = _doy_hour(valid)
  assert d.max() == 365 and d.min() == 0            # 2020-12-31 is day 366; 2021-01-01 wraps to day 1
  np.testing.assert_array_equal(c.values, swap(clim[d, h]))
  np.testing.assert_array_equal(c.coords['dayofyear'].values, d + 1)
  np.testing.assert_array_equal(c.coords['hour'].values, h * 6)
  np.testing.assert_array_equal(c['lead_time'].values, lead_times)
  # no lead times: the init times are the valid times
  c0 = lc.load_chunk(init_times)['z']
  assert c0.dims == ('init_time',) + out_dims
  d0, h0 = _doy_hour(init_times)
  np.testing.assert_array_equal(c0.values, swap(clim[d0, h0]))
  with pytest.raises(ValueError, match='Lead time slice not yet supported for climatology data loaders'):
    lc.load_chunk(init_times, slice(None))
  with pytest.raises(KeyError):  # 03 UTC is not one of the climatology's hours
    lc.load_chunk(np.array(['2020-12-30T03'], dtype='datetime64[ns]'))

  p = lp.load_chunk(init_times, lead_times)['z']
  assert p.dims == ('init_time', 'lead_time') + out_dims
  ti = ((init_times - times[0]) // np.timedelta64(6, 'h')).astype(int)
  for b in range(lead_times.size):
    np.testing.assert_array_equal(p.values[:, b], swap(tv[ti]))
  np.testing.assert_array_equal(p['init_time'].values, init_times)
  for bad in (None, slice(None)):
    with pytest.raises(ValueError, match='Exact lead times must be specified for persistence data loader'):
      lp.load_chunk(init_times, bad)
  one = lp.load_chunk(init_times[:1], lead_times[:1])['z']
  np.testing.assert_array_equal(one.values[0, 0], swap(tv[ti[0]]))


def test_daily_climatology_without_an_hour_axis(tmp_path):
  rng = np.random.default_rng(2)
  clim = rng.normal(size=(366, 3, 8)).astype(np.float32)
  path = os.path.join(str(tmp_path), 'c.npy')
  np.save(path, clim)
  lc = loaders.ClimatologyFromFiles({'t': path}, ('longitude', 'latitude'), {'longitude': np.arange(3.0), 'latitude': np.arange(8.0)},
                                    hours=False, pinned=False)
  init = np.array(['2021-03-01T06', '2020-03-01T18'], dtype='datetime64[ns]')
  c = lc.load_chunk(init, np.array([0, 24], dtype='timedelta64[h]'))['t']
  np.testing.assert_array_equal(c.values[:, :, ...], clim[np.array([[59, 60], [60, 61]])])  # 1 Mar = day 60 (2021) / 61 (2020)
  assert 'hour' not in c.coords
  with pytest.raises(ValueError, match='climatology file holds'):
    loaders.ClimatologyFromFiles({'t': path}, ('longitude', 'latitude'), {}, pinned=False).load_chunk(init)


def test_climatology_and_persistence_baselines_scored_against_the_oracle(backend, tmp_path):
  """The two baselines of the reference's evaluation scripts as PREDICTIONS of a chunked evaluation, against the oracle on
  the whole arrays."""
  src_c, src_t, times, dims, coords, clim, tv = _write_baselines(str(tmp_path), 'npy')
  pinned = backend == 'hip'
  lt = loaders.TargetsFromFiles(src_t, times, dims, coords, pinned=pinned)
  init_times = times[:8:2]
  lead_times = np.array([0, 12, 36], dtype='timedelta64[h]').astype('timedelta64[ns]')
  valid = init_times[:, None] + lead_times[None, :]
  vi = ((valid - times[0]) // np.timedelta64(6, 'h')).astype(int)
  d, h = _doy_hour(valid)
  ii = ((init_times - times[0]) // np.timedelta64(6, 'h')).astype(int)
  truth = tv[vi]
  fdims = ('init_time', 'lead_time') + dims
  w = (O.grid_area_weights(coords['latitude']), ('latitude',))
  metrics = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE()}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  tc = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=3, lead_time_chunk_size=2)
  for loader, pred in ((loaders.ClimatologyFromFiles(src_c, dims, coords, pinned=pinned), clim[d, h]),
                       (loaders.PersistenceFromFiles(src_t, times, dims, coords, pinned=pinned),
                        np.broadcast_to(tv[ii][:, None], truth.shape))):
    got = pipeline.evaluate_chunks(tc, loaders.load_chunk_fn(loader, lt), metrics, agg)[None].metric_values(metrics)
    for name, lane in (('rmse', O.squared_error(pred, truth)), ('mae', O.absolute_error(pred, truth))):
      sws, sw, od = O.aggregate(lane, fdims, ['init_time', 'latitude', 'longitude'], weights=[w])
      want = np.sqrt(sws / sw) if name == 'rmse' else sws / sw
      np.testing.assert_allclose(np.asarray(got[f'{name}.z'].transpose(*od).values), want, rtol=RTOL, err_msg=f'{type(loader).__name__} {name}')


def test_years_as_ensemble_members_scored_with_crps(backend, tmp_path):
  """ProbabilisticClimatologyFromXarray (xarray_loaders.py:340-409; its test xarray_loaders_test.py:114-141): members are the
  same dayofyear / hour of 2015..2019, day 366 of a non-leap year is 1 January of the next; CRPS of that ensemble through a
  chunked evaluation against the oracle."""
  from weatherbenchx_amd.metrics import probabilistic
  rng = np.random.default_rng(4)
  nlon, nlat = 12, 7
  times = np.arange('2015-01-01T00', '2021-01-03T00', np.timedelta64(12, 'h'), dtype='datetime64[ns]')
  tv = (rng.normal(size=(times.size, nlon, nlat)) + 280).astype(np.float32)
  path = os.path.join(str(tmp_path), 't.npy')
  np.save(path, tv)
  dims = ('longitude', 'latitude')
  coords = {'longitude': np.arange(nlon) * 30.0, 'latitude': np.linspace(-90, 90, nlat)}
  pinned = backend == 'hip'
  lp = loaders.ProbabilisticClimatologyFromFiles({'t2m': path}, times, dims, coords, start_year=2015, end_year=2019, pinned=pinned)
  lt = loaders.TargetsFromFiles({'t2m': path}, times, dims, coords, pinned=pinned)
  init_times = np.arange('2020-12-30T00', '2021-01-01T00', np.timedelta64(24, 'h'), dtype='datetime64[ns]')
  lead_times = np.arange(0, 3, 1, dtype='timedelta64[D]').astype('timedelta64[ns]')
  chunk = lp.load_chunk(init_times, lead_times)['t2m']
  assert chunk.dims == ('number', 'init_time', 'lead_time') + dims and chunk.sizes['number'] == 5
  # the restatement: valid time -> (dayofyear, hour) -> that offset from 1 January of each year
  valid = init_times[:, None] + lead_times[None, :]
  doy0, _ = _doy_hour(valid)
  members = np.stack([np.datetime64(str(y), 'ns') + doy0 * np.timedelta64(24, 'h') for y in range(2015, 2020)])
  assert members[0, 1, 0] == np.datetime64('2016-01-01')       # 2020-12-31 = day 366 -> 2015 has none
  assert members[1, 1, 0] == np.datetime64('2016-12-31')       # ... 2016 has
  np.testing.assert_array_equal(chunk.coords['valid_time'].values, members)
  pv = tv[((members - times[0]) // np.timedelta64(12, 'h')).astype(int)]
  np.testing.assert_array_equal(chunk.values, pv)
  with pytest.raises(ValueError, match='Exact lead times'):
    lp.load_chunk(init_times, None)

  truth = tv[((valid - times[0]) // np.timedelta64(12, 'h')).astype(int)]
  metrics = {'crps': probabilistic.CRPSEnsemble(ensemble_dim='number')}
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  tc = time_chunks.TimeChunks(init_times, lead_times, init_time_chunk_size=1, lead_time_chunk_size=2)
  got = pipeline.evaluate_chunks(tc, loaders.load_chunk_fn(lp, lt), metrics, agg)[None].metric_values(metrics)
  pdims, tdims = ('number', 'init_time', 'lead_time') + dims, ('init_time', 'lead_time') + dims
  w = (O.grid_area_weights(coords['latitude']), ('latitude',))
  sk = O.aggregate(O.crps_skill(pv, pdims, truth, tdims, 'number')[0], tdims, ['init_time', 'latitude', 'longitude'], weights=[w])
  sp = O.aggregate(O.crps_spread(pv, pdims, 'number')[0], tdims, ['init_time', 'latitude', 'longitude'], weights=[w])
  np.testing.assert_allclose(np.asarray(got['crps.t2m'].values), O.crps(sk[0] / sk[1], sp[0] / sp[1]), rtol=RTOL)
