"""Round-5 GPU parity tests at the sizes that run: wbx_ens_binned under the masks the reference itself builds -- the per-point
NaN mask of add_nan_mask_to_data (data_loaders/base.py:25-56), another hole at every lead time -- and under
Aggregator(skipna=True) (aggregation.py:339-357), at M = 51 on the full 0.25 degree grid with the public benchmark's 34 bins:
every bin of all five lanes and of their weights against the float64 oracle, ONE launch.  Tolerance: rtol 1e-6 (north_star)."""
import ctypes as C

import numpy as np
import pytest

from oracle import wbx_oracle as O
from weatherbenchx_amd import _hip
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import binning
from weatherbenchx_amd import data as wdata
from weatherbenchx_amd import engine
from weatherbenchx_amd import planner
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
import test_ens_binned as EB
import test_gpu_round4 as R4

pytestmark = pytest.mark.gpu
RTOL = 1e-6
NLAT, NLON = 721, 1440


@pytest.fixture(scope='module')
def ctx():
  assert _hip.is_available(), 'gpu tests need libwbx_hip.so and a HIP device'
  return _hip.default_context(0)


def _holes(tv, td, lat, lon, rng):
  """NaN targets the way observations / a regional analysis leave them: a block of latitudes that moves with the lead time, a
  longitude band that differs per lead, and scattered single points."""
  sizes = dict(zip(td, tv.shape))
  la = np.abs(lat)[:, None] * np.ones(lon.size)[None, :]
  for l in range(sizes['lead_time']):
    hole = (la > 70 - 6 * l) & (np.cos(np.deg2rad(lon) * (3 + l))[None, :] > 0.3)
    hole |= rng.random(hole.shape) < 0.002
    sp = tuple(d for d in td if d in ('latitude', 'longitude'))
    hv = hole if sp == ('latitude', 'longitude') else hole.T
    idx = tuple(l if d == 'lead_time' else slice(None) for d in td)
    tv[idx][np.broadcast_to(hv, tv[idx].shape)] = np.nan
  return tv


def _check(state, stats, pv, tv, layout, lat, lon, land, reduce_dims, **kw):
  saved = EB.REGIONS
  EB.REGIONS = R4.REGIONS17
  try:
    EB._check_lanes(state, stats, pv, tv, layout, lat, lon, land, reduce_dims, **kw)  # pylint: disable=protected-access
  finally:
    EB.REGIONS = saved


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest', 'ifs'])
def test_nan_mask_per_lead_time_full_grid_one_launch(ctx, layout):
  """M = 51, 721 x 1440, 34 bins, `mask = ~isnan(targets)` with another hole at every lead time, masked=True:
  the masked statistics AND spread / variance of the predictions alone out of ONE wbx_ens_binned launch."""
  lat, lon = np.linspace(-90, 90, NLAT), np.linspace(0, 360, NLON, endpoint=False)
  land = R4._land(lat, lon)  # pylint: disable=protected-access
  rng = np.random.default_rng(5)
  p, t, pv, tv, lat, lon = EB.make_case(layout, 51, NLAT, NLON, 2, seed=77, ninit=1)
  tv = _holes(tv, t.dims, lat, lon, rng)
  t = wdata.add_nan_mask_to_data({'v': xr.DataArray(tv, dims=t.dims, coords={k: t.coords[k].values for k in t.dims})})['v']
  assert tuple(t.coords['mask'].dims) == tuple(t.dims)
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  reduce_dims = ['latitude', 'longitude'] + (['init_time'] if layout == 'ifs' else [])
  agg = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(R4.REGIONS17, land_sea_mask=lsm)], masked=True)
  stats = EB.lane_statistics()
  state, log = EB.run(stats, agg, p, t)
  assert [(e['kind'], e['flags'] & 1) for e in log] == [('ens_binned', 1)], log
  assert not (log[0]['w_flags'] & _hip.BINNED_MASK_ON_W)  # the per-point route
  _check(state, stats, pv, tv, layout, lat, lon, land, reduce_dims, mask=~np.isnan(tv), mask_dims=EB.LAYOUTS[layout][1])


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest'])
@pytest.mark.parametrize('masked', [False, True])
def test_skipna_full_grid_one_launch(ctx, layout, masked):
  """Aggregator(skipna=True) at M = 51 on the full grid: NaN targets (holes that move with the lead time) and NaN members
  (scattered), with and without a (latitude, longitude) mask coordinate on top; every bin of the five lanes and of the five
  per-statistic weight sums."""
  lat, lon = np.linspace(-90, 90, NLAT), np.linspace(0, 360, NLON, endpoint=False)
  land = R4._land(lat, lon)  # pylint: disable=protected-access
  valid = ~((np.abs(lat)[:, None] > 80) & (np.cos(np.deg2rad(lon) * 5)[None, :] > 0.2))
  rng = np.random.default_rng(6)
  p, t, pv, tv, lat, lon = EB.make_case(layout, 51, NLAT, NLON, 2, seed=78, mask=valid if masked else None)
  tv = _holes(tv, t.dims, lat, lon, rng)
  pv[rng.random(pv.shape) < 2e-4] = np.nan  # ~1 % of the points lose a member
  pd, td = EB.LAYOUTS[layout]
  p = xr.DataArray(pv, dims=pd, coords={k: p.coords[k].values for k in pd if k != 'number'})
  t2 = xr.DataArray(tv, dims=td, coords={k: t.coords[k].values for k in td})
  t = t2.assign_coords(mask=t.coords['mask']) if masked else t2
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(R4.REGIONS17, land_sea_mask=lsm)], masked=masked, skipna=True)
  stats = EB.lane_statistics()
  state, log = EB.run(stats, agg, p, t)
  assert [e['kind'] for e in log] == ['ens_binned'], log
  _check(state, stats, pv, tv, layout, lat, lon, land, ['latitude', 'longitude'], mask=valid if masked else None,
         mask_dims=('latitude', 'longitude'), skipna=True)


def test_raw_c_abi_point_mask_and_skipna(ctx):
  """wbx_ens_binned through raw pointers with a mask that has a stride along A (no WBX_BINNED_MASK_ON_W) and with
  WBX_FLAG_SKIPNA: the 6- / 12- / 10- / 20-lane outputs against NumPy."""
  rng = np.random.default_rng(2)
  nlead, m, nlat, nlon, nbin = 3, 8, 40, 200, 9
  tv = rng.normal(size=(nlead, nlat, nlon)).astype(np.float32)
  pv = (tv[:, None] + rng.normal(size=(nlead, m, nlat, nlon))).astype(np.float32)
  mask = rng.random((nlead, nlat, nlon)) > 0.3
  tv_nan = tv.copy()
  tv_nan[rng.random(tv.shape) < 0.1] = np.nan
  pv_nan = pv.copy()
  pv_nan[rng.random(pv.shape) < 0.01] = np.nan
  dims = ('lead_time', 'latitude', 'longitude')
  sizes = {'lead_time': nlead, 'latitude': nlat, 'longitude': nlon}
  lay_p = planner.InputLayout(strides={'lead_time': m * nlat * nlon, 'latitude': nlon, 'longitude': 1}, itemsize=4, base_alignment=256)
  lay_t = planner.InputLayout(strides={'lead_time': nlat * nlon, 'latitude': nlon, 'longitude': 1}, itemsize=4, base_alignment=256)
  lay_m = planner.InputLayout(strides={'lead_time': nlat * nlon, 'latitude': nlon, 'longitude': 1}, itemsize=1, base_alignment=256)
  boxy = np.zeros((nlat, nlon, nbin), bool)
  for b in range(nbin - 1):
    boxy[(b * 4) % nlat:(b * 4) % nlat + 18, (b * 23) % nlon:(b * 23) % nlon + 90, b] = True
  bits = np.zeros((nlat, nlon), np.uint64)
  for b in range(nbin):
    bits |= boxy[..., b].astype(np.uint64) << np.uint64(b)
  wrow = rng.random(nlat) + 0.5
  w_flags = _hip.BINNED_W_ON_X | _hip.BINNED_WT_ROW_ONLY
  order = ['CRPSSkill', 'CRPSSpread', 'EnsembleVariance', 'UnbiasedEnsembleMeanSquaredError', 'EnsembleMeanSquaredError']
  bits_buf, w_buf, m_buf = ctx.upload(bits), ctx.upload(wrow), ctx.upload(mask.astype(np.uint8))
  member = boxy.astype(np.float64)

  def run(pvals, tvals, flags, wf, nl):
    plan = planner.build_s1_plan(dims, sizes, [lay_p, lay_t, None, lay_m if flags & _hip.FLAG_MASKED else None],
                                 ('latitude', 'longitude'), wdep_dims={'latitude', 'longitude'}, flags=_hip.FLAG_FAIR | flags,
                                 allow_vec4=False)
    dplan = engine._PlanOnDevice(ctx, plan)  # pylint: disable=protected-access
    out = ctx.alloc(nlead * nl * nbin * 8)
    pb, tb = ctx.upload(pvals), ctx.upload(tvals)
    _hip.check(R4._raw_call(ctx, plan, dplan, m, nlat * nlon, pb, tb, m_buf if flags & _hip.FLAG_MASKED else None, w_buf, bits_buf,  # pylint: disable=protected-access
                            nlead, 1, nlat, wf, nbin, None, out), 'wbx_ens_binned')
    return ctx.download(out.ptr, (nlead, nl, nbin), np.float64)

  def sums(lane, ok):
    with np.errstate(invalid='ignore'):
      return np.einsum('ayx,y,yxb->ab', np.where(ok, lane, 0.0), wrow, member), np.einsum('ayx,y,yxb->ab', ok.astype(np.float64), wrow, member)

  pdims = ('lead_time', 'number', 'latitude', 'longitude')
  # (1) a mask with a stride along A, no skipna: 6 lanes, and 12 with the twin flag
  lanes = EB.oracle_lanes(pv, pdims, tv, dims)
  for wf, nl in ((w_flags, 6), (w_flags | _hip.BINNED_TWIN_MASK, 12)):
    got = run(pv, tv, _hip.FLAG_MASKED, wf, nl)
    for l, name in enumerate(order):
      want, cnt = sums(lanes[name][0], mask)
      np.testing.assert_allclose(got[:, l], want, rtol=RTOL, atol=1e-9, err_msg=name)
      np.testing.assert_allclose(got[:, 5], cnt, rtol=1e-12)
      if nl == 12:
        want, cnt = sums(lanes[name][0], np.ones_like(mask))
        np.testing.assert_allclose(got[:, 6 + l], want, rtol=RTOL, atol=1e-9, err_msg=name)
        np.testing.assert_allclose(got[:, 11], cnt, rtol=1e-12)
  # (2) skipna, with NaN targets and NaN members: 10 lanes; with mask + twin: 20
  lanes = EB.oracle_lanes(pv_nan, pdims, tv_nan, dims)
  got = run(pv_nan, tv_nan, _hip.FLAG_SKIPNA, w_flags, 10)
  for l, name in enumerate(order):
    want, cnt = sums(lanes[name][0], ~np.isnan(lanes[name][0]))
    np.testing.assert_allclose(got[:, l], want, rtol=RTOL, atol=1e-9, err_msg=name)
    np.testing.assert_allclose(got[:, 5 + l], cnt, rtol=1e-12, err_msg=name)
  got = run(pv_nan, tv_nan, _hip.FLAG_SKIPNA | _hip.FLAG_MASKED, w_flags, 10)
  for l, name in enumerate(order):
    want, cnt = sums(lanes[name][0], mask & ~np.isnan(lanes[name][0]))
    np.testing.assert_allclose(got[:, l], want, rtol=RTOL, atol=1e-9, err_msg=name)
    np.testing.assert_allclose(got[:, 5 + l], cnt, rtol=1e-12, err_msg=name)
  got = run(pv_nan, tv_nan, _hip.FLAG_SKIPNA | _hip.FLAG_MASKED, w_flags | _hip.BINNED_TWIN_MASK, 20)
  for l, name in enumerate(order):
    if l in (1, 2):
      assert np.isnan(got[:, l]).all()
      want, cnt = sums(lanes[name][0], ~np.isnan(lanes[name][0]))
      np.testing.assert_allclose(got[:, 10 + l], want, rtol=RTOL, atol=1e-9, err_msg=name)
      np.testing.assert_allclose(got[:, 15 + l], cnt, rtol=1e-12, err_msg=name)
    else:
      assert np.isnan(got[:, 10 + l]).all()
      want, cnt = sums(lanes[name][0], mask & ~np.isnan(lanes[name][0]))
      np.testing.assert_allclose(got[:, l], want, rtol=RTOL, atol=1e-9, err_msg=name)
      np.testing.assert_allclose(got[:, 5 + l], cnt, rtol=1e-12, err_msg=name)


def test_cross_xcd_hand_off_stress(ctx):
  """the in-kernel sums over a cell's patches hand atom tables and records from the wave that wrote them to
  the wave that arrives last -- relaxed agent-scope stores + `s_waitcnt vmcnt(0)` + a relaxed counter on the writer, plain
  `buffer_load ... sc1` on the reader (csrc/wbx_ens_atoms.hpp).  Geometry that makes every hand-off cross XCDs and leaves no
  slack: 2048 cells of 64 x 32 points -- one x tile, a few short row splits -- in cell-fastest block order, so the patches of a
  cell are dealt to DIFFERENT XCDs (workgroup i runs on XCD i mod 8 and an XCD walks a contiguous eighth of the patch list) and
  finish within microseconds of each other.  1000 launches, each compared BIT FOR BIT with the first (a stale or torn read of a
  published table changes the sums), and the first against float64 NumPy."""
  rng = np.random.default_rng(11)
  ncell, m, nlat, nlon, nbin = 2048, 8, 32, 64, 6
  tv = rng.normal(size=(ncell, nlat, nlon)).astype(np.float32)
  pv = (tv[:, None] + rng.normal(size=(ncell, m, nlat, nlon))).astype(np.float32)
  mask = rng.random((nlat, nlon)) > 0.2
  dims = ('lead_time', 'latitude', 'longitude')
  sizes = {'lead_time': ncell, 'latitude': nlat, 'longitude': nlon}
  lay_p = planner.InputLayout(strides={'lead_time': m * nlat * nlon, 'latitude': nlon, 'longitude': 1}, itemsize=4, base_alignment=256)
  lay_t = planner.InputLayout(strides={'lead_time': nlat * nlon, 'latitude': nlon, 'longitude': 1}, itemsize=4, base_alignment=256)
  lay_m = planner.InputLayout(strides={'lead_time': 0, 'latitude': nlon, 'longitude': 1}, itemsize=1, base_alignment=256)
  boxy = np.zeros((nlat, nlon, nbin), bool)
  for b in range(nbin):
    boxy[(b * 5) % nlat:(b * 5) % nlat + 14, (b * 11) % nlon:(b * 11) % nlon + 30, b] = True
  boxy[..., 0] = True
  bits = np.zeros((nlat, nlon), np.uint64)
  for b in range(nbin):
    bits |= boxy[..., b].astype(np.uint64) << np.uint64(b)
  wrow = rng.random(nlat) + 0.5
  bits_buf, w_buf, m_buf = ctx.upload(bits), ctx.upload(wrow), ctx.upload(mask.astype(np.uint8))
  pb, tb = ctx.upload(pv), ctx.upload(tv)
  pdims = ('lead_time', 'number', 'latitude', 'longitude')
  lanes = EB.oracle_lanes(pv, pdims, tv, dims)
  order = ['CRPSSkill', 'CRPSSpread', 'EnsembleVariance', 'UnbiasedEnsembleMeanSquaredError', 'EnsembleMeanSquaredError']
  for flags, wf, nl in ((0, _hip.BINNED_W_ON_X | _hip.BINNED_WT_ROW_ONLY, 6),
                        (_hip.FLAG_MASKED, _hip.BINNED_W_ON_X | _hip.BINNED_WT_ROW_ONLY | _hip.BINNED_MASK_ON_W | _hip.BINNED_TWIN_MASK, 12)):
    plan = planner.build_s1_plan(dims, sizes, [lay_p, lay_t, None, lay_m if flags else None], ('latitude', 'longitude'),
                                 wdep_dims={'latitude', 'longitude'}, flags=_hip.FLAG_FAIR | flags, allow_vec4=False)
    dplan = engine._PlanOnDevice(ctx, plan)  # pylint: disable=protected-access
    ring = [ctx.alloc(ncell * nl * nbin * 8) for _ in range(50)]
    first = None
    for rep in range(20):
      for out in ring:
        _hip.check(R4._raw_call(ctx, plan, dplan, m, nlat * nlon, pb, tb, m_buf if flags else None, w_buf, bits_buf,  # pylint: disable=protected-access
                                ncell, 1, nlat, wf, nbin, None, out), 'wbx_ens_binned')
      for k, out in enumerate(ring):
        got = ctx.download(out.ptr, (ncell, nl, nbin), np.float64)
        if first is None:
          first = got
          valid = mask if flags else np.ones_like(mask)
          for l, name in enumerate(order):
            want = np.einsum('ayx,y,yxb->ab', np.where(valid, lanes[name][0], 0.0), wrow, boxy.astype(np.float64))
            np.testing.assert_allclose(got[:, l], want, rtol=RTOL, atol=1e-9, err_msg=name)
        else:
          bad = np.argwhere(got.view(np.uint64) != first.view(np.uint64))
          assert bad.size == 0, f'launch {rep * 50 + k} (flags {flags}) differs from the first at (cell, lane, bin) {bad[:5].tolist()}'


@pytest.mark.parametrize('case', ['1440_lon', '1440_lat', '1440_lon_rows', '240_g64', '512_g256', '1000_rocfft', '1440_rocfft'])
def test_spectra_are_bit_reproducible(ctx, case, monkeypatch):
  """the spectra's sums over rows no longer depend on the order in which teams arrive (rounds 1-4 added them
  with fp64 atomics: 9-14 % of the outputs differed in the last bits from run to run).  Every route -- the one-wave 1440-point
  kernel, its latitude-fastest (block-staged) variant, the generic in-LDS kernel with one-wave and whole-block teams, the rocFFT
  route -- twenty times on the same field: bit-identical outputs (groups of 721 rows: an odd count, so pairs straddle group
  boundaries; `_rows`: every row its own group), and the float64 numpy.fft oracle for the first."""
  from weatherbenchx_amd import spectra
  from weatherbenchx_amd.metrics import base as metrics_base
  import torch
  nlon = {'1440': 1440, '240': 240, '512': 512, '1000': 1000}[case.split('_')[0]]
  nlat, nlead, nlev = (721, 3, 4) if nlon == 1440 else (91, 3, 5)
  if 'rocfft' in case:
    monkeypatch.setenv('WBX_SPECTRUM_PATH', 'rocfft')
    nlat, nlead, nlev = 61, 2, 3
  lat, lon = np.linspace(-90, 90, nlat), np.arange(nlon) * (360.0 / nlon)
  dims = ('lead_time', 'level', 'longitude', 'latitude') if case == '1440_lat' else ('lead_time', 'level', 'latitude', 'longitude')
  shape = {'lead_time': nlead, 'level': nlev, 'latitude': nlat, 'longitude': nlon}
  g = torch.Generator(device='cuda')
  g.manual_seed(3)
  vals = torch.randn([shape[d] for d in dims], generator=g, device='cuda') * 3 + 280
  f = xr.DataArray(vals, dims=dims, coords={'latitude': lat, 'longitude': lon})
  metrics = {'spec': spectra.ZonalPowerSpectrum()}
  reduce_dims = ['lead_time'] if case.endswith('_rows') else ['latitude']
  agg = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=[weighting.GridAreaWeighting()] if 'latitude' in reduce_dims else None)

  def run():
    fresh = xr.DataArray(vals, dims=dims, coords={'latitude': lat, 'longitude': lon})
    res = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, {'v': fresh}, {'v': fresh}))
    return np.ascontiguousarray(np.asarray(res.metric_values(metrics)['spec.v'].values))
  first = run()
  for i in range(19):
    again = run()
    bad = np.argwhere(again.view(np.uint64) != first.view(np.uint64))
    assert bad.size == 0, f'{case}: run {i + 2} differs from the first in {len(bad)} of {first.size} outputs, e.g. at {bad[:3].tolist()}'
  # ... and the numbers are the oracle's
  host = vals.cpu().numpy()
  lon_ax = dims.index('longitude')
  per_row = np.moveaxis(O.zonal_power_spectrum(host, lon_axis=lon_ax), lon_ax, -1)
  rd = tuple(d for d in dims if d != 'longitude')
  if 'latitude' in reduce_dims:
    wv = O.expand_to(O.grid_area_weights(lat), ('latitude',), rd)[..., None]
    ax = rd.index('latitude')
    want = (per_row * wv).sum(axis=ax) / (wv * np.ones_like(per_row)).sum(axis=ax)
    res_dims = [d for d in rd if d != 'latitude']
  else:
    want = per_row.mean(axis=rd.index('lead_time'))
    res_dims = [d for d in rd if d != 'lead_time']
  fresh = xr.DataArray(vals, dims=dims, coords={'latitude': lat, 'longitude': lon})
  res = agg.aggregate_statistics(metrics_base.compute_unique_statistics_for_all_metrics(metrics, {'v': fresh}, {'v': fresh}))
  got = np.asarray(res.metric_values(metrics)['spec.v'].transpose(*res_dims, 'zonal_wavenumber').values)
  np.testing.assert_allclose(got[..., 1:], want[..., 1:], rtol=2e-4, atol=2e-6 * want[..., 1:].max())
  np.testing.assert_allclose(got[..., 0], want[..., 0], rtol=1e-6 if nlon == 1440 or nlon <= 256 else 2e-4)


def test_fused_det_spectra_are_bit_reproducible(ctx, monkeypatch):
  """The same for the spectra that come out of the deterministic sweep (wbx_det_spectrum: records of 2 x 721 values), both
  layouts' routes: five jobs of four chunks, bit-identical accumulators."""
  import torch
  from weatherbenchx_amd import pipeline, spectra, time_chunks
  from weatherbenchx_amd.metrics import deterministic
  nlat, nlon, nlead, nlev, n = 45, 1440, 2, 3, 4
  g = torch.Generator(device='cuda')
  g.manual_seed(9)
  lat, lon = np.linspace(-66, 66, nlat), np.arange(nlon) * 0.25
  lead = (np.arange(nlead) * 6).astype('timedelta64[h]').astype('timedelta64[ns]')
  inits = np.datetime64('2020-01-01T00', 'ns') + np.arange(n) * np.timedelta64(24, 'h')
  for layout in ('lon_fastest', 'lat_fastest'):
    sp = ('latitude', 'longitude') if layout == 'lon_fastest' else ('longitude', 'latitude')
    zd = ('init_time', 'lead_time', 'level') + sp
    shp = (1, nlead, nlev) + tuple({'latitude': nlat, 'longitude': nlon}[d] for d in sp)
    pool = [(torch.randn(shp, generator=g, device='cuda') + 280, torch.randn(shp, generator=g, device='cuda') + 280) for _ in range(2)]
    index = {int(t.astype('int64')): i for i, t in enumerate(inits)}

    def load(ic, lc):
      i = index[int(ic[0].astype('int64'))]
      cs = {'init_time': ic, 'lead_time': lead, 'level': np.arange(nlev), 'latitude': lat, 'longitude': lon}
      return {'z': xr.DataArray(pool[i % 2][0], dims=zd, coords=cs)}, {'z': xr.DataArray(pool[i % 2][1], dims=zd, coords=cs)}
    det = {'rmse': deterministic.RMSE(), 'bias': deterministic.Bias()}
    spec = {'sp': spectra.ZonalPowerSpectrum('predictions'), 'st': spectra.ZonalPowerSpectrum('targets')}
    area = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
    zonal = aggregation.Aggregator(reduce_dims=['init_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
    passes = [('deterministic', load, det, area), ('spectra', load, spec, zonal)]
    if layout == 'lat_fastest':
      monkeypatch.setattr(engine, 'FUSE_DET_SPECTRA_LATFAST', True)

    def run():
      engine.S1_EVENT_LOG = []
      try:
        out = pipeline.evaluate_passes(time_chunks.TimeChunks(inits, lead, init_time_chunk_size=1), passes)
        kinds = {e['kind'] for e in engine.S1_EVENT_LOG}
      finally:
        engine.S1_EVENT_LOG = None
      vals = out['spectra'][None].metric_values(spec)
      return {k: np.ascontiguousarray(np.asarray(v.values)) for k, v in vals.items()}, kinds
    first, kinds = run()
    assert 'det_spectrum' in kinds, kinds
    for _ in range(4):
      again, _ = run()
      for k in first:
        assert np.array_equal(again[k].view(np.uint64), first[k].view(np.uint64)), (layout, k)


# ---- ordered spectrum sums under arbitrary group tables (the records' store, the closing kernel's rounds) ----------------------
def _spectrum_reference(rows64, group, scale, ngroup):
  """sum over the rows of a group of scale[row] * S_k(row): S = |rfft / n|^2 x (1 for k = 0, else 2) in float64."""
  n = rows64.shape[-1]
  f = np.fft.rfft(rows64, axis=-1) / n
  s = (f.real ** 2 + f.imag ** 2) * np.where(np.arange(n // 2 + 1) == 0, 1.0, 2.0)
  out = np.zeros((ngroup, n // 2 + 1))
  np.add.at(out, group, s * scale[:, None])
  return out


@pytest.mark.parametrize('seed', range(12))
def test_raw_spectrum_with_random_group_tables(ctx, seed, monkeypatch):
  """wbx_zonal_spectrum / wbx_zonal_spectrum_slabs through the raw C ABI with group tables the labeled-array layer never builds:
  unsorted groups, a change at every row (one record per row: more than 2048 records of one group, the closing kernel's further
  rounds), groups nobody belongs to, one group for everything, zero scales; rows of 1440 points in both layouts, the generic
  fused kernel and the rocFFT route; accumulate = 0 and 1.  Twice each: bit-identical, and the float64 numpy.fft oracle."""
  rng = np.random.default_rng(52000 + seed)
  route = ['lon1440', 'lat1440', 'generic', 'rocfft'][seed % 4]
  if route == 'rocfft':
    monkeypatch.setenv('WBX_SPECTRUM_PATH', 'rocfft')
  nlon = {'lon1440': 1440, 'lat1440': 1440, 'generic': int(rng.choice([64, 240, 360, 512])), 'rocfft': int(rng.choice([90, 250, 1000]))}[route]
  pattern = ['alternate', 'random', 'blocks', 'single'][(seed // 4 + seed) % 4]
  if route == 'lat1440':
    nslab, rps = int(rng.integers(2, 7)), int(rng.choice([24, 47, 121, 721]))
    nrows = nslab * rps
  else:
    nrows = int(rng.choice([1, 2, 97, 2000, 5001])) if pattern != 'alternate' else 5001
    nslab, rps = 1, nrows
  ngroup = 1 if pattern == 'single' else int(rng.integers(2, 9))
  if pattern == 'alternate':
    group = (np.arange(nrows) % 2 * (ngroup - 1)).astype(np.int32)  # ~2500 one-row records in each of two groups, the rest empty
  elif pattern == 'random':
    group = rng.integers(0, ngroup, nrows).astype(np.int32)
  elif pattern == 'blocks':
    group = np.sort(rng.integers(0, ngroup, nrows)).astype(np.int32)[::-1].copy()
  else:
    group = np.zeros(nrows, np.int32)
  scale = rng.random(nrows) + 0.5
  scale[rng.random(nrows) < 0.1] = 0.0
  field = (rng.normal(size=(nrows, nlon)) * 2 + rng.normal() * 50).astype(np.float32)
  if route == 'lat1440':
    stored = np.ascontiguousarray(field.reshape(nslab, rps, nlon).transpose(0, 2, 1))  # [slab][lon][row]
    lon_stride, row_stride = rps, 1
    offs = (np.arange(nslab) * rps * nlon).astype(np.int64)
  else:
    stored, lon_stride, row_stride, offs = field, 1, nlon, np.zeros(1, np.int64)
  dev, g_dev, s_dev = ctx.upload(stored), ctx.upload(group), ctx.upload(scale)
  nk = nlon // 2 + 1
  want = _spectrum_reference(field.astype(np.float64), group, scale, ngroup)
  seedv = rng.normal(size=(ngroup, nk))
  results = []
  for accumulate in (0, 1, 0, 1):
    out = ctx.upload(seedv.copy())
    if route == 'lat1440':
      _hip.check(ctx.lib.wbx_zonal_spectrum_slabs(ctx.handle, C.c_void_p(dev.ptr), lon_stride, row_stride, rps, nslab,
                                                  offs.ctypes.data_as(C.c_void_p), nlon, C.c_void_p(g_dev.ptr), C.c_void_p(s_dev.ptr),
                                                  ngroup, accumulate, C.c_void_p(out.ptr)), 'wbx_zonal_spectrum_slabs')
    else:
      _hip.check(ctx.lib.wbx_zonal_spectrum(ctx.handle, C.c_void_p(dev.ptr), lon_stride, row_stride, nrows, nlon, C.c_void_p(g_dev.ptr),
                                            C.c_void_p(s_dev.ptr), ngroup, accumulate, C.c_void_p(out.ptr)), 'wbx_zonal_spectrum')
    results.append(ctx.download(out.ptr, (ngroup, nk), np.float64))
  for accumulate, got in zip((0, 1), results[:2]):
    ref = want + (seedv if accumulate else 0.0)
    tol = 2e-4 if (route == 'rocfft' or (route == 'generic' and nlon > 256)) else 5e-6
    np.testing.assert_allclose(got[:, 1:], ref[:, 1:], rtol=tol, atol=tol * np.abs(want).max() + 1e-12, err_msg=f'{route} {pattern} accumulate={accumulate}')
    np.testing.assert_allclose(got[:, 0], ref[:, 0], rtol=1e-5 if tol < 1e-4 else 2e-4, atol=1e-9, err_msg=f'{route} {pattern} k = 0')
  assert np.array_equal(results[0].view(np.uint64), results[2].view(np.uint64)), (route, pattern)
  assert np.array_equal(results[1].view(np.uint64), results[3].view(np.uint64)), (route, pattern)


def test_raw_binned_launches_add_into_out(ctx):
  """WBX_BINNED_ACCUMULATE through the raw C ABI: wbx_ens_binned (plain, twin, skipna) and wbx_det_binned with the flag leave
  out = seed + result -- bit for bit the sum of the seed and the result of the same launch without the flag -- and an empty
  reduction leaves `out` untouched."""
  rng = np.random.default_rng(3)
  nlead, m, nlat, nlon, nbin = 3, 8, 40, 200, 9
  tv = rng.normal(size=(nlead, nlat, nlon)).astype(np.float32)
  pv = (tv[:, None] + rng.normal(size=(nlead, m, nlat, nlon))).astype(np.float32)
  tv[rng.random(tv.shape) < 0.05] = np.nan
  mask = rng.random((nlat, nlon)) > 0.2
  dims = ('lead_time', 'latitude', 'longitude')
  sizes = {'lead_time': nlead, 'latitude': nlat, 'longitude': nlon}
  lay_p = planner.InputLayout(strides={'lead_time': m * nlat * nlon, 'latitude': nlon, 'longitude': 1}, itemsize=4, base_alignment=256)
  lay_t = planner.InputLayout(strides={'lead_time': nlat * nlon, 'latitude': nlon, 'longitude': 1}, itemsize=4, base_alignment=256)
  lay_m = planner.InputLayout(strides={'lead_time': 0, 'latitude': nlon, 'longitude': 1}, itemsize=1, base_alignment=256)
  boxy = np.zeros((nlat, nlon, nbin), bool)
  for b in range(nbin):
    boxy[(b * 5) % nlat:(b * 5) % nlat + 14, (b * 37) % nlon:(b * 37) % nlon + 60, b] = True
  bits = np.zeros((nlat, nlon), np.uint64)
  for b in range(nbin):
    bits |= boxy[..., b].astype(np.uint64) << np.uint64(b)
  wrow = rng.random(nlat) + 0.5
  bits_buf, w_buf, m_buf = ctx.upload(bits), ctx.upload(wrow), ctx.upload(mask.astype(np.uint8))
  pb, tb = ctx.upload(pv), ctx.upload(tv)
  base = _hip.BINNED_W_ON_X | _hip.BINNED_WT_ROW_ONLY
  for flags, wf, nl in ((0, base, 6), (_hip.FLAG_MASKED, base | _hip.BINNED_MASK_ON_W | _hip.BINNED_TWIN_MASK, 12),
                        (_hip.FLAG_SKIPNA, base, 10)):
    plan = planner.build_s1_plan(dims, sizes, [lay_p, lay_t, None, lay_m if flags & _hip.FLAG_MASKED else None], ('latitude', 'longitude'),
                                 wdep_dims={'latitude', 'longitude'}, flags=_hip.FLAG_FAIR | flags, allow_vec4=False)
    dplan = engine._PlanOnDevice(ctx, plan)  # pylint: disable=protected-access
    seed = rng.normal(size=(nlead, nl, nbin))
    plain, added = ctx.alloc(seed.nbytes), ctx.upload(seed.copy())
    for wflags, out in ((wf, plain), (wf | _hip.BINNED_ACCUMULATE, added)):
      _hip.check(R4._raw_call(ctx, plan, dplan, m, nlat * nlon, pb, tb, m_buf if flags & _hip.FLAG_MASKED else None, w_buf, bits_buf,  # pylint: disable=protected-access
                              nlead, 1, nlat, wflags, nbin, None, out), 'wbx_ens_binned')
    res, got = ctx.download(plain.ptr, seed.shape, np.float64), ctx.download(added.ptr, seed.shape, np.float64)
    assert np.array_equal((seed + res).view(np.uint64), got.view(np.uint64)), (flags, np.argwhere((seed + res) != got)[:4])
  # the deterministic launch
  tclean = np.nan_to_num(tv, nan=0.0)
  t2 = ctx.upload(tclean)
  p2 = ctx.upload(np.ascontiguousarray(pv[:, 0]))
  plan = planner.build_s1_plan(dims, sizes, [lay_t, lay_t, None, None], ('latitude', 'longitude'), wdep_dims={'latitude', 'longitude'},
                               flags=0, allow_vec4=False)
  dplan = engine._PlanOnDevice(ctx, plan)  # pylint: disable=protected-access
  wfull = ctx.upload(np.ascontiguousarray(np.broadcast_to(wrow[:, None], (nlat, nlon))))
  seed = rng.normal(size=(nlead, 3, nbin))
  plain, added = ctx.alloc(seed.nbytes), ctx.upload(seed.copy())
  for wflags, out in ((_hip.BINNED_W_ON_X, plain), (_hip.BINNED_W_ON_X | _hip.BINNED_ACCUMULATE, added)):
    _hip.check(ctx.lib.wbx_det_binned(ctx.handle, C.byref(dplan.struct), _hip.DET3, _hip.F32, C.c_void_p(p2.ptr), C.c_void_p(t2.ptr), None, None,
                                      C.c_void_p(wfull.ptr), C.c_void_p(bits_buf.ptr), nlead, 1, nlat, wflags, nbin, None,
                                      C.c_void_p(out.ptr)), 'wbx_det_binned')
  res, got = ctx.download(plain.ptr, seed.shape, np.float64), ctx.download(added.ptr, seed.shape, np.float64)
  want = np.einsum('ayx,y,yxb->ab', pv[:, 0].astype(np.float64) - tclean, wrow, boxy.astype(np.float64))
  np.testing.assert_allclose(res[:, 0], want, rtol=1e-9, atol=1e-9)
  assert np.array_equal((seed + res).view(np.uint64), got.view(np.uint64))
