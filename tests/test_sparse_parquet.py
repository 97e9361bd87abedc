"""Station observations from time-partitioned Parquet (weatherbenchx_amd/data_loaders/sparse_parquet.py): the METAR month file the
reference's own tests read (weatherbenchX/test_data/metar-timeNominal-by-month, kept as data under tests/golden/) through
METARFromParquet as binning_test.py:62-97, 185-265 does, every selection checked against plain pandas on the same file; a
synthetic day-partitioned archive for exact times, tolerances, duplicates and files read once per chunk; and a station evaluation:
gridded forecasts interpolated to the stations, binned by lead time, against the oracle."""
import os

import numpy as np
import pandas as pd
import pytest

from oracle import wbx_oracle as O
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import binning
from weatherbenchx_amd import interpolations
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.data_loaders import sparse_parquet
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.metrics import deterministic

METAR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'metar-timeNominal-by-month')
H = np.timedelta64(1, 'h')


def _metar_frame():
  return pd.read_parquet(os.path.join(METAR, 'year=2020', 'month=1', '2020-01.parquet'))


def test_partition_file_names():
  t = np.datetime64('2020-03-07T05')
  assert sparse_parquet.parquet_filename_for_time('/d', t, 'M') == '/d/year=2020/month=3/2020-03.parquet'
  assert sparse_parquet.parquet_filename_for_time('/d', t, 'D') == '/d/year=2020/month=3/day=7/2020-03-07.parquet'
  assert sparse_parquet.parquet_filename_for_time('/d', t, 'h') == '/d/year=2020/month=3/day=7/hour=5/2020-03-07T05.parquet'
  files = sparse_parquet.get_parquet_files_subset('/d', np.datetime64('2020-02-28T23'), np.datetime64('2020-03-01T01'), 'day')
  assert [os.path.basename(f) for f in files] == ['2020-02-28.parquet', '2020-02-29.parquet', '2020-03-01.parquet']
  assert len(sparse_parquet.get_parquet_files_subset('/d', np.datetime64('2020-01-31T23'), np.datetime64('2020-02-01T00'), 'month')) == 2
  with pytest.raises(NotImplementedError):
    sparse_parquet.get_parquet_files_subset('/d', t, t, 'year')
  with pytest.raises(ValueError, match='Unsupported partitioned_by'):
    sparse_parquet.SparseObservationsFromParquet('/d', 'year', 'time', ['x'])


def test_metar_exact_lead_times():
  loader = sparse_parquet.METARFromParquet(path=METAR, variables=['2m_temperature', '10m_wind_speed'], partitioned_by='month',
                                           split_variables=True, dropna=True, time_dim='timeNominal')
  init_times = np.array(['2020-01-02T00', '2020-01-02T12'], dtype='datetime64[ns]')
  lead_times = np.array([6, 12], dtype='timedelta64[h]')
  chunk = loader.load_chunk(init_times, lead_times)
  raw = _metar_frame()
  bad = ('Z', 'B', 'X', 'Q', 'k')
  for ours, theirs in (('2m_temperature', 'temperature'), ('10m_wind_speed', 'windSpeed')):
    da = chunk[ours]
    assert da.dims == ('index',) and {'latitude', 'longitude', 'elevation', 'stationName', 'valid_time', 'init_time', 'lead_time'} <= set(da.coords)
    want = []
    for it in init_times:
      for lt in lead_times:
        rows = raw[raw.timeNominal == pd.Timestamp(it + lt)]
        values = rows[theirs].where(~rows[theirs + 'DD'].isin(bad))
        want.append(pd.DataFrame({'v': values, 'station': rows.stationName, 'init': it, 'lead': lt.astype('timedelta64[ns]'),
                                  'lon': np.mod(rows.longitude, 360)}))
    want = pd.concat(want, ignore_index=True)
    want = want[want.v.notna()]
    np.testing.assert_array_equal(da.values, want.v.to_numpy())
    np.testing.assert_array_equal(da.coords['stationName'].values, want.station.to_numpy().astype(str))
    np.testing.assert_array_equal(da.coords['init_time'].values, want.init.to_numpy())
    np.testing.assert_array_equal(da.coords['lead_time'].values, want.lead.to_numpy())
    np.testing.assert_allclose(da.coords['longitude'].values, want.lon.to_numpy())
    np.testing.assert_array_equal(da.coords['valid_time'].values, (want.init + want.lead).to_numpy())
    np.testing.assert_array_equal(da.coords['index'].values, want.index.to_numpy())            # positions before the NaNs were dropped
    assert (da.coords['longitude'].values >= 0).all() and not np.isnan(da.values).any() and da.size > 50


def test_metar_lead_time_slice_and_the_binnings_on_it():
  """binning_test.py:62-97, 185-265: lead times from a slice, then ByExactCoord / ByCoordBins / BySets on the chunk."""
  loader = sparse_parquet.METARFromParquet(path=METAR, variables=['2m_temperature'], partitioned_by='month', split_variables=True,
                                           dropna=True, time_dim='timeObs', file_tolerance=H)
  init_times = np.array(['2020-01-02T00', '2020-01-02T12'], dtype='datetime64[ns]')
  stat = loader.load_chunk(init_times, slice(1 * H, 6 * H))['2m_temperature']
  raw = _metar_frame()
  n = 0
  for it in init_times:
    inside = (raw.timeObs >= pd.Timestamp(it - 1 * H)) & (raw.timeObs < pd.Timestamp(it + 6 * H)) & raw.temperature.notna() & ~raw.temperatureDD.isin(
        ('Z', 'B', 'X', 'Q', 'k'))
    n += int(inside.sum())
  assert stat.size == n > 100
  lead = stat.coords['lead_time'].values
  np.testing.assert_array_equal(lead, stat.coords['valid_time'].values - stat.coords['init_time'].values)
  assert lead.min() >= -1 * H and lead.max() < 6 * H                    # the window opens slice.start BEFORE the init time (sparse_parquet.py:229)
  mask = binning.ByCoordBins('lead_time', np.arange(1, 7) * H.astype('timedelta64[ns]')).create_bin_mask(stat)
  assert (np.asarray(mask.values).mean(axis=1) > 0).all()               # binning_test.py:207-211
  names = np.asarray(stat.coords['stationName'].values)
  sets = binning.BySets({'set1': names[:10], 'set2': names[10:20], 'scalar_set': names[0], 'empty_set': [], 'wrong_set': [1, 2, 3, 4]},
                        coord_name='stationName', bin_dim_name='station_subset', add_global_bin=True).create_bin_mask(stat)
  assert np.asarray(sets.values)[-1].all() and not np.asarray(sets.values)[3].any() and not np.asarray(sets.values)[4].any()
  by_station = binning.ByExactCoord('stationName').create_bin_mask(stat)
  assert by_station.shape == (np.unique(names).size, stat.size)
  with pytest.raises(FileNotFoundError):                                 # the files are assumed complete: December 2019 is not there
    loader.load_chunk(np.array(['2020-01-01T00'], dtype='datetime64[ns]'), slice(1 * H, 6 * H))


def _write_days(root, days=3, stations=6, seed=0):
  rng = np.random.default_rng(seed)
  frames = []
  for d in range(days):
    day = np.datetime64('2021-05-01') + np.timedelta64(d, 'D')
    rows = []
    for h in range(24):
      nominal = day.astype('datetime64[ns]') + h * H
      for s in range(stations):
        for rep in range(2 if (s == 0 and h % 6 == 0) else 1):           # station 0 reports twice at the main hours
          rows.append({'station': f'S{s}', 'lat': -50.0 + 20 * s, 'lon': -170.0 + 60 * s, 'nominal': nominal,
                       'obs': nominal - np.timedelta64(int(rng.integers(0, 20)) + 25 * rep, 'm'),
                       't2m': float(rng.normal(280, 5)) if rng.random() > 0.1 else np.nan, 'wind': float(rng.gamma(2.0))})
    df = pd.DataFrame(rows)
    stamp = day.item()
    path = os.path.join(root, f'year={stamp.year}', f'month={stamp.month}', f'day={stamp.day}')
    os.makedirs(path)
    df.to_parquet(os.path.join(path, f'{stamp.year}-{stamp.month:02d}-{stamp.day:02d}.parquet'))
    frames.append(df)
  return pd.concat(frames, ignore_index=True)


def test_day_partitions_tolerance_duplicates_and_one_read_per_file(tmp_path, monkeypatch):
  everything = _write_days(str(tmp_path))
  reads = []
  real = pd.read_parquet
  monkeypatch.setattr(pd, 'read_parquet', lambda fn, *a, **k: (reads.append(os.path.basename(fn)), real(fn, *a, **k))[1])
  kw = dict(path=str(tmp_path), partitioned_by='day', variables=['t2m', 'wind'], coordinate_variables=['station', 'lat', 'lon', 'obs'])
  init_times = np.array(['2021-05-02T00', '2021-05-02T12'], dtype='datetime64[ns]')
  lead_times = np.array([0, 6, 24], dtype='timedelta64[h]')
  # exact nominal times, everything in one Dataset, rows with a NaN in any variable dropped
  ds = sparse_parquet.SparseObservationsFromParquet(time_dim='nominal', dropna=True, file_tolerance=np.timedelta64(0, 'h'), **kw).load_chunk(
      init_times, lead_times)
  valid = (init_times[:, None] + lead_times[None, :]).ravel()
  want = pd.concat([everything[everything.nominal == pd.Timestamp(v)] for v in valid], ignore_index=True)
  want = want[want.t2m.notna()]
  assert isinstance(ds, xr.Dataset) and ds['t2m'].size == ds['wind'].size == len(want)
  np.testing.assert_array_equal(ds['wind'].values, want.wind.to_numpy())
  assert sorted(set(reads)) == ['2021-05-02.parquet', '2021-05-03.parquet'] and len(reads) == 2   # six windows, two files, two reads
  # observation times within +-30 min of the valid time; duplicates resolved to the report closest to it
  reads.clear()
  loader = sparse_parquet.SparseObservationsFromParquet(time_dim='obs', tolerance=np.timedelta64(30, 'm'), remove_duplicates=True,
                                                        pick_closest_duplicate_by='obs', observation_dim='station',
                                                        coordinate_variables=['station', 'lat', 'lon', 'nominal'], path=str(tmp_path),
                                                        partitioned_by='day', variables=['t2m', 'wind'], split_variables=True)
  out = loader.load_chunk(init_times, np.array([0], dtype='timedelta64[h]'))['wind']
  for it in init_times:
    window = everything[(everything.obs >= pd.Timestamp(it - 30 * np.timedelta64(1, 'm'))) & (everything.obs < pd.Timestamp(it + 30 * np.timedelta64(1, 'm')))]
    best = window.assign(d=(window.obs - pd.Timestamp(it)).abs()).sort_values('d').drop_duplicates('station', keep='first')
    mine = out.isel(index=np.nonzero(out.coords['init_time'].values == it)[0])
    assert sorted(mine.coords['station'].values.tolist()) == sorted(best.station.tolist()) and mine.size == best.station.nunique()
    got = dict(zip(mine.coords['station'].values.tolist(), mine.values.tolist()))
    for _, row in best.iterrows():
      assert got[row.station] == row.wind
  assert len(reads) == len(set(reads))                                   # still once per file
  with pytest.raises(ValueError, match='non-empty'):
    sparse_parquet.SparseObservationsFromParquet(time_dim='obs', tolerance=(np.timedelta64(1, 'h'), np.timedelta64(1, 'h')), **kw)
  with pytest.raises(ValueError, match='station_dim must be specified'):
    sparse_parquet.SparseObservationsFromParquet(time_dim='obs', remove_duplicates=True, **kw)
  no_leads = sparse_parquet.SparseObservationsFromParquet(time_dim='nominal', **kw).load_chunk(init_times)
  assert 'init_time' not in no_leads['t2m'].coords and no_leads['t2m'].size == int(everything.nominal.isin(pd.to_datetime(init_times)).sum())


def test_station_evaluation_of_gridded_forecasts(backend):
  """Forecast grids -> stations (InterpolateToReferenceCoords), squared errors binned by lead time, the Aggregator's reduction
  over `index` -- against the same thing written out with pandas and the oracle's interpolation-free arithmetic."""
  del backend
  loader = sparse_parquet.METARFromParquet(path=METAR, variables=['2m_temperature'], partitioned_by='month', split_variables=True,
                                           dropna=True, time_dim='timeNominal')
  init_times = np.array(['2020-01-03T00'], dtype='datetime64[ns]')
  lead_times = np.array([6, 12, 18], dtype='timedelta64[h]')
  targets = loader.load_chunk(init_times, lead_times)
  obs = targets['2m_temperature']
  lat, lon = np.linspace(-90, 90, 37), np.arange(0, 360, 5.0)
  field = 288.0 - 40.0 * np.abs(np.sin(np.deg2rad(lat)))[:, None] + 2.0 * np.cos(np.deg2rad(lon))[None, :]
  grid = xr.DataArray(field, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon}, name='2m_temperature')
  to_stations = interpolations.InterpolateToReferenceCoords(method='linear', dims=['latitude', 'longitude'], wrap_longitude=True)
  predictions = to_stations.interpolate({'2m_temperature': grid}, targets)
  p = predictions['2m_temperature']
  assert p.dims == ('index',) and p.size == obs.size
  metrics = {'rmse': deterministic.RMSE(), 'bias': deterministic.Bias()}
  stats = metrics_base.compute_unique_statistics_for_all_metrics(metrics, predictions, targets)
  out = aggregation.Aggregator(reduce_dims=['index'], bin_by=[binning.ByExactCoord('lead_time')]).aggregate_statistics(stats).metric_values(metrics)
  err = np.asarray(p.values, dtype=np.float64) - np.asarray(obs.values, dtype=np.float64)
  lt = obs.coords['lead_time'].values
  for k, lead in enumerate(np.unique(lt)):
    sel = lt == lead
    np.testing.assert_allclose(float(np.asarray(out['rmse.2m_temperature'].values)[k]), np.sqrt(O.squared_error(err[sel], 0 * err[sel]).mean()), rtol=1e-5)
    np.testing.assert_allclose(float(np.asarray(out['bias.2m_temperature'].values)[k]), err[sel].mean(), rtol=1e-4, atol=1e-6)
  np.testing.assert_array_equal(out['rmse.2m_temperature'].coords['lead_time'].values, np.unique(lt))
