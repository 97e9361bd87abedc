"""Zonal spectra.  The reference snapshot has no spectrum code or test, so parity is
UNPINNED for this component; these tests pin the build's own definition analytically (Parseval, constant field,
single sinusoid, linearity in the row weights) and against the float64 numpy.fft oracle.  Tolerance 1e-5
relative: the FFT itself is single precision (rocFFT), only the |F|^2 accumulation is fp64."""
import os

import numpy as np
import pytest

from oracle import wbx_oracle as O
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import binning
from weatherbenchx_amd import spectra
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as metrics_base

RTOL = 1e-5


def bound_1440(want):
  """What the 1440-point fp32 kernels are held to against the float64 oracle (r3: rows are shifted by an estimate of their
  mean before the transform and F_0 is put back in fp64, csrc/wbx_zspec1440.hpp, so the error no longer scales with the mean):
  |dF_k| <~ eps (|F_k| + max_{k >= 1} |F_k|), i.e. |dS_k| <= 2e-6 S_k + 1e-6 sqrt(S'_max S_k) with S'_max = max_{k >= 1} S_k;
  S_0 to 1e-6 of its value.  Measured (tests/measure_spectrum_error.py, N(0, 1) and N(280, 1) rows, both layouts): median
  1.4e-7, 99.9th percentile 5e-6 of S_k per row, |dS_k| / sqrt(S'_max S_k) <= 5.5e-7; 1e-7 after a mean over 200 rows."""
  rest = want[..., 1:].max(axis=-1, keepdims=True)
  bound = 2e-6 * want + 1e-6 * np.sqrt(rest * want)
  bound[..., 0] = 1e-6 * want[..., 0] + 1e-6 * np.sqrt(rest[..., 0] * want[..., 0])
  return bound
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _field(vals, dims, lat=None, lon=None):
  coords = {}
  if lat is not None:
    coords['latitude'] = lat
  if lon is not None:
    coords['longitude'] = lon
  return xr.DataArray(np.asarray(vals, np.float32), dims=dims, coords=coords)


def test_golden_spectrum_fixture(backend):
  g = np.load(os.path.join(G, 'weights_spectrum.npz'))
  f = _field(g['spec_field'], ('latitude', 'longitude'))
  s = spectra.ZonalPowerSpectrum().compute({'v': f}, {'v': f})['v']
  assert s.dims == ('latitude', 'zonal_wavenumber') and s.shape == (16, 17)
  np.testing.assert_allclose(s.values, g['spec_power'], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('team', [None, '64', '128', '256'])
@pytest.mark.parametrize('nlon', [32, 45, 1440])
def test_parseval_constant_and_sinusoid(backend, monkeypatch, nlon, team):
  if team is not None:  # threads per row pair of the fused FFT kernel (csrc/wbx_spectrum.hip)
    monkeypatch.setenv('WBX_SPECTRUM_TEAM', team)
  rng = np.random.default_rng(nlon)
  x = 2 * np.pi * np.arange(nlon) / nlon
  rows = np.stack([rng.normal(size=nlon), np.full(nlon, 3.0), 1.5 * np.cos(5 * x + 0.3), 2.0 + np.sin(x)])
  f = _field(rows, ('row', 'longitude'))
  s = spectra.ZonalPowerSpectrum().compute({'v': f}, {'v': f})['v'].values
  r32 = rows.astype(np.float32).astype(np.float64)
  # Parseval: sum_k S_k = mean(f^2) (+ the Nyquist power once more for even nlon: every k >= 1 is doubled)
  total = s.sum(axis=1)
  extra = s[:, -1] / 2 if nlon % 2 == 0 else 0.0
  np.testing.assert_allclose(total - extra, (r32 ** 2).mean(axis=1), rtol=RTOL)
  # constant field: only k = 0
  np.testing.assert_allclose(s[1, 0], 9.0, rtol=RTOL)
  assert np.all(np.abs(s[1, 1:]) < 1e-9)
  # A cos(5x + phi): all power A^2/2 in bin 5;  2 + sin(x): 4 in bin 0, 1/2 in bin 1
  np.testing.assert_allclose(s[2, 5], 1.5 ** 2 / 2, rtol=RTOL)
  assert np.abs(np.delete(s[2], 5)).max() < 1e-6
  np.testing.assert_allclose(s[3, :2], [4.0, 0.5], rtol=RTOL)
  np.testing.assert_allclose(s, O.zonal_power_spectrum(r32), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('nlon', [48, 45, 50])  # fused (4*2*3), odd -> library route, fused with radix 5
@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
def test_area_weighted_mean_spectrum_matches_oracle(backend, layout, nlon):
  """lat_fastest = lead x level slabs of adjacent latitude rows with strided longitude: one wbx_zonal_spectrum_slabs
  call (rows transposed tile by tile into the fused kernel, or one library batch per slab for odd lengths)."""
  rng = np.random.default_rng(0)
  lat, lon = np.linspace(-80, 80, 9), np.arange(nlon) * (360.0 / nlon)
  dims = ('lead_time', 'level', 'latitude', 'longitude') if layout == 'lon_fastest' else \
      ('lead_time', 'level', 'longitude', 'latitude')
  shape = {'lead_time': 3, 'level': 2, 'latitude': 9, 'longitude': nlon}
  vals = rng.normal(size=[shape[d] for d in dims]).astype(np.float32)
  f = _field(vals, dims, lat=lat, lon=lon)
  metrics = {'spec': spectra.ZonalPowerSpectrum(), 'espec': spectra.ZonalEnergySpectrum()}
  agg = aggregation.Aggregator(reduce_dims=['lead_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
  stats = metrics_base.compute_unique_statistics_for_all_metrics(metrics, {'v': f}, {'v': f})
  res = agg.aggregate_statistics(stats).metric_values(metrics)
  lon_ax = dims.index('longitude')
  per_row = np.moveaxis(O.zonal_power_spectrum(vals, lon_axis=lon_ax), lon_ax, -1)  # [..., k]
  rd = tuple(d for d in dims if d != 'longitude')
  w = O.grid_area_weights(lat)
  wv = O.expand_to(w, ('latitude',), rd)[..., None]
  red = tuple(rd.index(d) for d in ('lead_time', 'latitude'))
  want = (per_row * wv).sum(axis=red) / (wv * np.ones_like(per_row)).sum(axis=red)
  got = res['spec.v']
  assert set(got.dims) == {'level', 'zonal_wavenumber'}
  np.testing.assert_allclose(got.transpose('level', 'zonal_wavenumber').values, want, rtol=1e-4, atol=1e-8)
  circ = 2 * np.pi * spectra.EARTH_RADIUS_M * np.cos(np.deg2rad(lat))
  cv = O.expand_to(circ, ('latitude',), rd)[..., None]
  want_e = (per_row * wv * cv).sum(axis=red) / (wv * np.ones_like(per_row)).sum(axis=red)
  np.testing.assert_allclose(res['espec.v'].transpose('level', 'zonal_wavenumber').values, want_e, rtol=1e-4, atol=1e-3)


def test_latitude_band_bins_and_longitude_dependent_masks(backend):
  rng = np.random.default_rng(1)
  lat, lon = np.linspace(-80, 80, 9), np.arange(32) * 11.25
  f = _field(rng.normal(size=(2, 9, 32)), ('time', 'latitude', 'longitude'), lat=lat, lon=lon)
  stat = spectra.ZonalPowerSpectrum().compute({'v': f}, {'v': f})
  bands = binning.Regions({'north': ((0, 90), (0, 360)), 'south': ((-90, 0), (0, 360))})
  # Regions needs a longitude coordinate, which a spectrum no longer has -> rejected loudly (KeyError), as the
  # reference's `statistic.longitude` lookup would (binning.py:186)
  with pytest.raises((KeyError, ValueError)):
    aggregation.Aggregator(reduce_dims=['time', 'latitude'], bin_by=[bands]).aggregate_stat_var(stat['v'])
  # summing over wavenumber goes through the generic (materialised) path and equals Parseval's total
  st = aggregation.Aggregator(reduce_dims=['zonal_wavenumber']).aggregate_stat_var(stat['v'])
  vals = f.values.astype(np.float64)
  np.testing.assert_allclose(st.sum_weighted_statistics.values,
                             (vals ** 2).mean(axis=-1) + 0.5 * stat['v'].values[..., -1], rtol=1e-4)


def test_spectra_under_deferred_results(backend):
  """engine.deferred_results(): the spectrum read-back is enqueued like every other result; a state launched for
  chunk k + 1 before chunk k is looked at still yields chunk k's numbers (own page-locked block per read-back)."""
  from weatherbenchx_amd import engine
  rng = np.random.default_rng(5)
  lat, lon = np.linspace(-80, 80, 9), np.arange(48) * 7.5
  dims = ('lead_time', 'latitude', 'longitude')
  fields = [_field(rng.normal(size=(3, 9, 48)).astype(np.float32), dims, lat=lat, lon=lon) for _ in range(3)]
  metrics = {'spec': spectra.ZonalPowerSpectrum()}
  agg = aggregation.Aggregator(reduce_dims=['lead_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])

  def launch(f):
    return agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, {'v': f}, {'v': f}))
  want = [launch(f).metric_values(metrics)['spec.v'].values.copy() for f in fields]
  with engine.deferred_results():
    states = [launch(f) for f in fields]  # all launched before any is read
    got = [s.metric_values(metrics)['spec.v'].values.copy() for s in states]
  for g, w in zip(got, want):
    np.testing.assert_allclose(g, w, rtol=1e-10)  # fp64 atomics: the order of the adds differs from run to run


@pytest.mark.parametrize('nlon', [96, 250, 540, 600, 750, 972, 1000, 1458, 2048, 2046])
def test_row_lengths_exercise_every_first_pass_radix_and_table_layout(backend, nlon):
  """Row lengths whose half length starts with each radix (4: 600, 1000, 2048; 2: 540, 972; 5: 750; 3: 1458), short rows
  on one-wave / two-wave teams (96, 250), the longest supported row (2048: 3 mirrored wavenumbers per thread) and a
  non-smooth length on the library route (2046 = 2 * 3 * 11 * 31): per-row spectra of 5 rows (odd: a lone last row)
  against numpy's float64 rfft."""
  rng = np.random.default_rng(nlon)
  vals = (rng.normal(size=(5, nlon)) + 3.0).astype(np.float32)
  f = xr.DataArray(vals, dims=('latitude', 'longitude'),
                   coords={'latitude': np.linspace(-40, 40, 5), 'longitude': np.arange(nlon) * (360.0 / nlon)})
  stat = spectra.ZonalPowerSpectrum().compute({'v': f}, {'v': f})['v']
  got = np.asarray(stat.values)
  want = O.zonal_power_spectrum(vals)
  assert got.shape == (5, nlon // 2 + 1)
  np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-6 * want.max())


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
@pytest.mark.parametrize('nlat,mean', [(7, 0.0), (10, 0.0), (7, 280.0), (50, 280.0)])
def test_1440_point_rows_one_wave_kernel(backend, layout, nlat, mean):
  """0.25 degree rows (csrc/wbx_zspec1440.hpp: one wave per row pair, 720 = 12 x 5 x 12; latitude-fastest fields through
  the block-staged variant, 24 adjacent rows per step): groups of 7 rows (every other pair straddles a group boundary, the
  lone last row), of 10, and of 50 (three runs of rows per slab, the last team of a run with a lone row) against the
  float64 numpy.fft oracle.  The fp32 transform's error is relative to the largest coefficient of the SHIFTED row (the mean
  is taken out in front of the transform): bound_1440, the same for white noise and for a mean of 280 (S_0 = 78400 against
  1e-3 per wave; held to 2e-5 S_k + 4e-7 sqrt(S_0 S_k) -- per-row errors of 1e-5 in the median -- before the shift)."""
  rng = np.random.default_rng(nlat)
  nlon = 1440
  lat, lon = np.linspace(-80, 80, nlat), np.arange(nlon) * 0.25
  dims = ('lead_time', 'level', 'latitude', 'longitude') if layout == 'lon_fastest' else \
      ('lead_time', 'level', 'longitude', 'latitude')
  shape = {'lead_time': 3, 'level': 3, 'latitude': nlat, 'longitude': nlon}
  vals = (rng.normal(size=[shape[d] for d in dims]) + mean).astype(np.float32)
  f = _field(vals, dims, lat=lat, lon=lon)
  metrics = {'spec': spectra.ZonalPowerSpectrum()}
  agg = aggregation.Aggregator(reduce_dims=['lead_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
  res = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, {'v': f}, {'v': f})).metric_values(metrics)
  lon_ax = dims.index('longitude')
  per_row = np.moveaxis(O.zonal_power_spectrum(vals, lon_axis=lon_ax), lon_ax, -1)
  rd = tuple(d for d in dims if d != 'longitude')
  wv = O.expand_to(O.grid_area_weights(lat), ('latitude',), rd)[..., None]
  red = tuple(rd.index(d) for d in ('lead_time', 'latitude'))
  want = (per_row * wv).sum(axis=red) / (wv * np.ones_like(per_row)).sum(axis=red)
  got = res['spec.v'].transpose('level', 'zonal_wavenumber').values

  def check(g, w):
    worst = float(np.max(np.abs(g - w) / bound_1440(w)))
    assert worst <= 1.0, worst
    if mean != 0.0:
      np.testing.assert_allclose(g[..., 0], w[..., 0], rtol=1e-6)
  check(got, want)
  # per-row spectra (no reduction): rows as their own groups
  stat = spectra.ZonalPowerSpectrum().compute({'v': f}, {'v': f})['v'].transpose(*rd, 'zonal_wavenumber')
  check(np.asarray(stat.values), per_row)


@pytest.mark.parametrize('seed', range(8))
def test_1440_point_rows_random_shapes_and_reductions(backend, seed):
  """Randomised: 1-3 leads x 1-4 levels x 1-60 latitudes of 1440-point rows in either layout, any subset of
  (lead_time, level, latitude) reduced (rows as their own groups, groups per level, one group for everything), with and
  without area weights -- the row-pair / run / slab bookkeeping of both 1440-point kernels against the float64 oracle."""
  rng = np.random.default_rng(1000 + seed)
  nlon = 1440
  shape = {'lead_time': int(rng.integers(1, 4)), 'level': int(rng.integers(1, 5)), 'latitude': int(rng.integers(1, 61)),
           'longitude': nlon}
  layout = ('lon_fastest', 'lat_fastest')[seed % 2]
  dims = ('lead_time', 'level', 'latitude', 'longitude') if layout == 'lon_fastest' else \
      ('lead_time', 'level', 'longitude', 'latitude')
  lat = np.linspace(-85, 85, shape['latitude']) if shape['latitude'] > 1 else np.array([10.0])
  vals = (rng.normal(size=[shape[d] for d in dims]) * rng.uniform(0.5, 3.0) + rng.uniform(-5, 5)).astype(np.float32)
  f = _field(vals, dims, lat=lat, lon=np.arange(nlon) * 0.25)
  row_dims = ('lead_time', 'level', 'latitude')
  reduce_dims = [d for d in row_dims if rng.random() < 0.5]
  weighted = bool(rng.random() < 0.5) and shape['latitude'] > 1
  metrics = {'spec': spectra.ZonalPowerSpectrum()}
  agg = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=[weighting.GridAreaWeighting()] if weighted else [])
  res = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, {'v': f}, {'v': f})).metric_values(metrics)
  lon_ax = dims.index('longitude')
  per_row = np.moveaxis(O.zonal_power_spectrum(vals, lon_axis=lon_ax), lon_ax, -1)
  rd = tuple(d for d in dims if d != 'longitude')
  w = O.grid_area_weights(lat) if weighted else np.ones(shape['latitude'])
  wv = O.expand_to(w, ('latitude',), rd)[..., None]
  red = tuple(rd.index(d) for d in reduce_dims)
  want = (per_row * wv).sum(axis=red) / (wv * np.ones_like(per_row)).sum(axis=red)
  kept = [d for d in rd if d not in reduce_dims]
  got = res['spec.v'].transpose(*kept, 'zonal_wavenumber').values
  worst = float(np.max(np.abs(got - want) / bound_1440(want)))
  assert got.shape == want.shape and worst <= 1.0, (shape, layout, reduce_dims, weighted, worst)


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
@pytest.mark.parametrize('nlon,nlat,mean', [(64, 32, 0.0), (64, 32, 280.0), (240, 121, 0.0), (240, 121, 280.0), (256, 9, 5.0e4),
                                            (360, 181, 280.0), (360, 33, 0.0), (512, 7, 5.0e4), (720, 37, 280.0), (1024, 5, 5.0e4),
                                            (2048, 3, 280.0)])
def test_short_rows_are_shifted_by_their_mean_too(backend, layout, nlon, nlat, mean):
  """The other grids of the public configs (64 x 32 and 240 x 121; 256 = the longest row of a one-wave team) and the lengths
  that run on teams of two or four waves (1 degree = 360, 0.5 degree = 720, 512 / 1024 / 2048: first pass with and without
  the register prefetch) go through the generic fused kernel, which shifts a row by the mean of its even points in front of
  the fp32 transform like the 1440-point kernels (csrc/wbx_spectrum.hip, team_pass / team_total_f32): the same bound against
  the float64 oracle whatever the mean -- N(280, 1) like a temperature field, N(5e4, 1) like geopotential --, odd row counts
  (a lone last row), both layouts."""
  rng = np.random.default_rng(nlon + nlat)
  lat, lon = np.linspace(-85, 85, nlat), np.arange(nlon) * (360.0 / nlon)
  dims = ('lead_time', 'latitude', 'longitude') if layout == 'lon_fastest' else ('lead_time', 'longitude', 'latitude')
  shape = {'lead_time': 3, 'latitude': nlat, 'longitude': nlon}
  vals = (rng.normal(size=[shape[d] for d in dims]) + mean).astype(np.float32)
  f = _field(vals, dims, lat=lat, lon=lon)
  lon_ax = dims.index('longitude')
  per_row = np.moveaxis(O.zonal_power_spectrum(vals, lon_axis=lon_ax), lon_ax, -1)
  stat = spectra.ZonalPowerSpectrum().compute({'v': f}, {'v': f})['v'].transpose('lead_time', 'latitude', 'zonal_wavenumber')
  got = np.asarray(stat.values)
  worst = float(np.max(np.abs(got - per_row) / bound_1440(per_row)))
  assert worst <= 1.0, worst
  if mean != 0.0:  # (S_0 of a zero-mean row is one more small coefficient)
    np.testing.assert_allclose(got[..., 0], per_row[..., 0], rtol=1e-6)


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
def test_1440_point_rows_nan_stays_in_its_row(backend, layout):
  """A NaN anywhere in a row makes that row's whole spectrum NaN (as numpy.fft does) and nothing else: the two rows of a
  pair share every packed instruction, their values must not mix."""
  rng = np.random.default_rng(7)
  nlat, nlon = 9, 1440
  vals = rng.normal(size=(nlat, nlon)).astype(np.float32)
  vals[2, 77] = np.nan   # row 2 = row A of the pair (2, 3)
  vals[5, 1439] = np.nan  # row 5 = row B of the pair (4, 5)
  dims = ('latitude', 'longitude') if layout == 'lon_fastest' else ('longitude', 'latitude')
  arr = vals if layout == 'lon_fastest' else np.ascontiguousarray(vals.T)
  f = _field(arr, dims, lat=np.linspace(-80, 80, nlat), lon=np.arange(nlon) * 0.25)
  s = np.asarray(spectra.ZonalPowerSpectrum().compute({'v': f}, {'v': f})['v'].transpose('latitude', 'zonal_wavenumber').values)
  bad = np.isnan(s).all(axis=1)
  assert list(np.nonzero(bad)[0]) == [2, 5] and not np.isnan(s[~bad]).any()
  want = O.zonal_power_spectrum(vals[~bad])
  np.testing.assert_allclose(s[~bad], want, rtol=2e-4, atol=1e-6 * want.max())


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
def test_1440_point_rows_agree_with_the_rocfft_route(backend, monkeypatch, layout):
  """An independent implementation of the same definition: the batched R2C rocFFT route (WBX_SPECTRUM_PATH=rocfft) against
  the one-wave kernels on the same 1440-point field; both are single precision, so they agree to the fp32 bound of either."""
  rng = np.random.default_rng(11)
  shape = {'lead_time': 2, 'level': 2, 'latitude': 13, 'longitude': 1440}
  dims = ('lead_time', 'level', 'latitude', 'longitude') if layout == 'lon_fastest' else \
      ('lead_time', 'level', 'longitude', 'latitude')
  vals = (rng.normal(size=[shape[d] for d in dims]) + 1.0).astype(np.float32)
  f = _field(vals, dims, lat=np.linspace(-80, 80, 13), lon=np.arange(1440) * 0.25)
  rd = tuple(d for d in dims if d != 'longitude')

  def run():
    return np.asarray(spectra.ZonalPowerSpectrum().compute({'v': f}, {'v': f})['v'].transpose(*rd, 'zonal_wavenumber').values)
  fused = run()
  monkeypatch.setenv('WBX_SPECTRUM_PATH', 'rocfft')
  library = run()
  bound = 4e-5 * library + 8e-7 * np.sqrt(library.max(axis=-1, keepdims=True) * library)
  assert float(np.max(np.abs(fused - library) / bound)) <= 1.0
