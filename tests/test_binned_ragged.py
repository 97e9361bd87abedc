"""Binned (region) aggregation on rows that are not whole 128-byte lines, with patches of BOTH binned kernels inside one block
of the atom kernel (wbx_det_binned): against the float64 oracle on the emulated backend (host logic) and on the GPU."""
import numpy as np
import pytest

from oracle import wbx_oracle as O
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import engine
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import base as metrics_base
from weatherbenchx_amd.metrics import deterministic

RTOL = 1e-6


class _MaskBins:
  """A user binning handed as plain boolean layers [bin, latitude, longitude] (binning.py:25-49: any Binning may do this)."""

  def __new__(cls, layers, lat, lon):
    from weatherbenchx_amd import binning

    class _B(binning.Binning):
      def create_bin_mask(self, statistic):
        return xr.DataArray(layers, dims=('zone', 'latitude', 'longitude'),
                            coords={'zone': np.arange(layers.shape[0]), 'latitude': lat, 'longitude': lon})
    return _B('zone')


@pytest.mark.parametrize('layout', ['lat_fastest', 'lon_fastest'])
@pytest.mark.parametrize('nlat,nlon,kind', [(150, 90, 'half'), (150, 90, 'all'), (721, 70, 'half'), (97, 130, 'stripes'),
                                            (64, 90, 'half'), (300, 200, 'none')])
def test_ragged_rows_with_patches_of_both_binned_kernels_in_one_block(backend, monkeypatch, layout, nlat, nlon, kind):
  """On rows that are not whole 128-byte lines (150 / 721 / 97 / 300 latitudes, latitude-fastest) a block of the atom kernel
  is four waves on adjacent x tiles that meet at a barrier every 64 rows.  Waves leave that block early when their tile lies
  beyond the row (97 points: two tiles in a block of four) or when their patch has more than 32 distinct membership words
  and belongs to the slot kernel: 40 user bins that are random point by point south of the equator ('half': the southern
  tiles are declined, the northern ones are not), everywhere ('all'), in longitude stripes ('stripes': whole row ranges
  declined) or nowhere ('none').  MSE, bias and the masked counts of every bin against the float64 oracle, rtol 1e-6; the
  two-stage route on the same inputs to 1e-9."""
  rng = np.random.default_rng(nlat * 1000 + nlon + len(kind))
  lat, lon = np.linspace(-90, 90, nlat), np.arange(nlon) * (360.0 / nlon)
  sp = ('latitude', 'longitude') if layout == 'lon_fastest' else ('longitude', 'latitude')
  dims = ('init_time', 'lead_time') + sp
  sizes = {'init_time': 2, 'lead_time': 3, 'latitude': nlat, 'longitude': nlon}
  coords = {'init_time': np.datetime64('2020-01-01T00', 'ns') + np.arange(2) * np.timedelta64(12, 'h'),
            'lead_time': np.arange(3) * np.timedelta64(6, 'h'), 'latitude': lat, 'longitude': lon}
  shape = tuple(sizes[d] for d in dims)
  pv = (rng.normal(size=shape) + 280).astype(np.float32)
  tv = (rng.normal(size=shape) + 280).astype(np.float32)
  tv[rng.random(shape) < 0.03] = np.nan
  nbin = 40
  bands = (np.arange(nlat)[None, :, None] * nbin // nlat == np.arange(nbin)[:, None, None]) | (np.arange(nbin)[:, None, None] == 0)
  layers = np.broadcast_to(bands, (nbin, nlat, nlon)).copy()  # latitude bands + one global bin: a handful of words per patch
  noise = rng.random((nbin, nlat, nlon)) < 0.5
  if kind == 'half':
    layers[:, :nlat // 2, :] = noise[:, :nlat // 2, :]
  elif kind == 'all':
    layers = noise
  elif kind == 'stripes':
    stripe = (np.arange(nlon) // 20) % 2 == 1
    layers[:, :, stripe] = noise[:, :, stripe]
  p = xr.DataArray(pv, dims=dims, coords=coords)
  t = xr.DataArray(tv, dims=dims, coords=coords)
  t.coords['mask'] = ~np.isnan(t)
  metrics = {'mse': deterministic.MSE(), 'bias': deterministic.Bias()}
  results, states = {}, {}
  for binned in ('always', 'never'):
    monkeypatch.setattr(engine, 'BINNED_MODE', binned)
    engine.clear_caches()
    agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                                 bin_by=[_MaskBins(layers, lat, lon)], masked=True)
    stats = metrics_base.compute_unique_statistics_for_all_metrics(metrics, {'z': p}, {'z': t})
    states[binned] = agg.aggregate_statistics(stats)
    results[binned] = states[binned].metric_values(metrics)
  for k in results['never']:
    np.testing.assert_allclose(results['always'][k].values, results['never'][k].values, rtol=1e-9, atol=1e-12, equal_nan=True)
  w = (O.grid_area_weights(lat), ('latitude',))
  ok = ~np.isnan(tv)
  for name, stat in (('mse', O.squared_error(pv, tv)), ('bias', pv.astype(np.float64) - tv)):
    sws, sw, od = O.aggregate(stat, dims, ['init_time', 'latitude', 'longitude'], weights=[w],
                              bin_masks=[('zone', layers, ('zone', 'latitude', 'longitude'))], mask=ok, mask_dims=dims)
    with np.errstate(all='ignore'):
      want = sws / sw
    got = results['always'][f'{name}.z'].transpose(*od).values
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-9, equal_nan=True)
  got_w = states['always'].sum_weights['SquaredError']['z'].transpose(*od).values
  np.testing.assert_allclose(got_w, sw, rtol=1e-12)
