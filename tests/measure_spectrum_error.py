"""Measured error of the 1440-point spectrum kernels against the float64 numpy.fft oracle (per row: median / 99.9th percentile
/ maximum relative error and the maximum of |dS_k| / sqrt(S_max S_k); aggregated over 200 rows), white noise and a mean of
280, both layouts -- where the bound stated in tests/test_spectra.py comes from.  A checker like the tests next to it (it calls
the oracle), not collected by pytest.  usage (GPU box): python tests/measure_spectrum_error.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import wbx_oracle as O
from weatherbenchx_amd import aggregation, spectra, weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as mb
for layout in ('lon_fastest', 'lat_fastest'):
  for mean in (0.0, 280.0):
    rng = np.random.default_rng(1)
    nlat, nlon = 50, 1440
    lat, lon = np.linspace(-80, 80, nlat), np.arange(nlon) * 0.25
    dims = ('lead_time', 'level', 'latitude', 'longitude') if layout == 'lon_fastest' else ('lead_time', 'level', 'longitude', 'latitude')
    shape = {'lead_time': 4, 'level': 3, 'latitude': nlat, 'longitude': nlon}
    vals = (rng.normal(size=[shape[d] for d in dims]) + mean).astype(np.float32)
    f = xr.DataArray(vals, dims=dims, coords={'latitude': lat, 'longitude': lon})
    lon_ax = dims.index('longitude')
    per_row = np.moveaxis(O.zonal_power_spectrum(vals, lon_axis=lon_ax), lon_ax, -1)
    rd = tuple(d for d in dims if d != 'longitude')
    stat = np.asarray(spectra.ZonalPowerSpectrum().compute({'v': f}, {'v': f})['v'].transpose(*rd, 'zonal_wavenumber').values)
    rel = np.abs(stat - per_row) / per_row
    relmax = np.abs(stat - per_row) / np.sqrt(per_row.max(-1, keepdims=True) * per_row)
    print(layout, mean, 'per-row: median rel', np.median(rel), 'p99.9 rel', np.quantile(rel, 0.999), 'max rel', rel.max(), 'max |d|/sqrt(Smax S)', relmax.max())
    metrics = {'spec': spectra.ZonalPowerSpectrum()}
    agg = aggregation.Aggregator(reduce_dims=['lead_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
    res = agg.aggregate_statistics(mb.compute_unique_statistics_for_all_metrics(metrics, {'v': f}, {'v': f})).metric_values(metrics)
    wv = O.expand_to(O.grid_area_weights(lat), ('latitude',), rd)[..., None]
    red = tuple(rd.index(d) for d in ('lead_time', 'latitude'))
    want = (per_row * wv).sum(axis=red) / (wv * np.ones_like(per_row)).sum(axis=red)
    got = res['spec.v'].transpose('level', 'zonal_wavenumber').values
    rel = np.abs(got - want) / want
    print('   aggregated over 200 rows: median rel', np.median(rel), 'max rel', rel.max())
