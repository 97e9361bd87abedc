"""Round-4 GPU parity tests at the sizes that run: the one-pass binned + masked ensemble kernel (wbx_ens_binned) on the
public benchmark's probabilistic configuration -- M = 51, 721 x 1440, Regions(17) x land/sea = 34 bins, GridAreaWeighting,
masked=True with a (latitude, longitude) validity mask -- every bin of all five lanes against the float64 oracle, both
layouts and the recorded IFS-ENS layout.  Tolerance: rtol 1e-6 (north_star)."""
import ctypes as C

import numpy as np
import pytest

from oracle import wbx_oracle as O
from weatherbenchx_amd import _hip
from weatherbenchx_amd import aggregation
from weatherbenchx_amd import binning
from weatherbenchx_amd import engine
from weatherbenchx_amd import planner
from weatherbenchx_amd import weighting
from weatherbenchx_amd import xarray_lite as xr
import test_ens_binned as EB

pytestmark = pytest.mark.gpu
RTOL = 1e-6
NLAT, NLON = 721, 1440
# the 17 evaluation regions of public_benchmark/run_benchmark_evaluation.py:110-131 (coordinates are data)
REGIONS17 = {
    'global': ((-90, 90), (0, 360)), 'tropics': ((-20, 20), (0, 360)), 'northern-hemisphere': ((20, 90), (0, 360)),
    'southern-hemisphere': ((-90, -20), (0, 360)), 'europe': ((35, 75), (-12.5, 42.5)),
    'north-america': ((25, 60), (360 - 120, 360 - 75)), 'north-atlantic': ((25, 65), (360 - 70, 360 - 10)),
    'north-pacific': ((25, 60), (145, 360 - 130)), 'east-asia': ((25, 60), (102.5, 150)),
    'ausnz': ((-45, -12.5), (120, 175)), 'arctic': ((60, 90), (0, 360)), 'antarctic': ((-90, -60), (0, 360)),
    'northern-africa': ((5, 32.5), (-12.5, 37.5)), 'southern-africa': ((-30, 5), (12.5, 37.5)),
    'south-america': ((-40, 5), (-75, -45)), 'west-asia': ((15, 60), (42.5, 102.5)),
    'south-east-asia': ((-12.5, 25), (95, 125)),
}


@pytest.fixture(scope='module')
def ctx():
  assert _hip.is_available(), 'gpu tests need libwbx_hip.so and a HIP device'
  return _hip.default_context(0)


def _land(lat, lon):
  """A coast-like land mask: blobs a few degrees wide, so patches see land / sea alternate the way real coasts do."""
  return (np.sin(np.deg2rad(lon) * 3)[None, :] * np.cos(np.deg2rad(lat) * 2.5)[:, None]
          + 0.3 * np.sin(np.deg2rad(lon) * 17)[None, :] * np.sin(np.deg2rad(lat) * 13)[:, None]) > 0.35


@pytest.mark.parametrize('layout', ['lon_fastest', 'lat_fastest', 'ifs'])
@pytest.mark.parametrize('m', [51, 50])
def test_public_probabilistic_chunk_every_bin_every_lane(ctx, layout, m):
  """M = 51 (and the 50 perturbed members of IFS-ENS) on the full 0.25 degree grid, 34 bins, masked=True with a (latitude,
  longitude) validity mask: the masked statistics AND the statistics of the predictions alone (spread, variance: no mask
  coordinate) out of ONE wbx_ens_binned launch -- and nothing else."""
  if m == 50 and layout != 'ifs':
    pytest.skip('M = 50 runs on the layout it comes in')
  lat, lon = np.linspace(-90, 90, NLAT), np.linspace(0, 360, NLON, endpoint=False)
  land = _land(lat, lon)
  valid = ~((np.abs(lat)[:, None] > 80) & (np.cos(np.deg2rad(lon) * 5)[None, :] > 0.2))  # a NaN-mask-like hole near the poles
  p, t, pv, tv, lat, lon = EB.make_case(layout, m, NLAT, NLON, 1, seed=100 + m, mask=valid, ninit=1)
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  reduce_dims = ['latitude', 'longitude'] + (['init_time'] if layout == 'ifs' else [])
  agg = aggregation.Aggregator(reduce_dims=reduce_dims, weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(REGIONS17, land_sea_mask=lsm)], masked=True)
  stats = EB.lane_statistics()
  state, log = EB.run(stats, agg, p, t)
  assert [(e['kind'], e['flags'] & 1) for e in log] == [('ens_binned', 1)], log
  EB.check_against_oracle(state, stats, pv, tv, layout, lat, lon, land, reduce_dims, mask=valid, regions=REGIONS17)


def test_two_stage_route_gives_the_same_numbers(ctx, monkeypatch):
  """A/B of the routes on one chunk: wbx_ens_binned against the x-kept ensemble kernel + wbx_contract_bits."""
  lat, lon = np.linspace(-90, 90, 181), np.linspace(0, 360, 360, endpoint=False)
  land = _land(lat, lon)
  p, t, pv, tv, lat, lon = EB.make_case('lat_fastest', 51, 181, 360, 2, seed=4)
  lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
  agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                               bin_by=[binning.Regions(REGIONS17, land_sea_mask=lsm)])
  stats = EB.lane_statistics()
  one, log1 = EB.run(stats, agg, p, t)
  monkeypatch.setattr(engine, 'ENS_BINNED', False)
  engine.clear_caches()
  p2 = xr.DataArray(pv, dims=p.dims, coords={k: p.coords[k].values for k in p.dims if k != 'number'})
  t2 = xr.DataArray(tv, dims=t.dims, coords={k: t.coords[k].values for k in t.dims})
  two, log2 = EB.run(stats, agg, p2, t2)
  assert [e['kind'] for e in log1] == ['ens_binned'] and 'ens_binned' not in [e['kind'] for e in log2]
  for name, s in stats.items():
    a = np.asarray(one.sum_weighted_statistics[s.unique_name]['v'].values)
    b = np.asarray(two.sum_weighted_statistics[s.unique_name]['v'].values)
    np.testing.assert_allclose(a, b, rtol=RTOL, atol=1e-9 * np.abs(b).max(), err_msg=name)


def _raw_call(ctx, plan, dplan, m, mstride, p_buf, t_buf, mask_buf, wt_buf, bits_buf, nA, nBk, nBr, w_flags, nbin, atoms, out):
  return ctx.lib.wbx_ens_binned(ctx.handle, C.byref(dplan.struct), _hip.F32, m, mstride, _hip.ENS_SORT, C.c_void_p(p_buf.ptr),
                                C.c_void_p(t_buf.ptr), C.c_void_p(mask_buf.ptr) if mask_buf is not None else None,
                                C.c_void_p(wt_buf.ptr) if wt_buf is not None else None, C.c_void_p(bits_buf.ptr), nA, nBk, nBr,
                                w_flags, nbin, C.c_void_p(atoms.ptr) if atoms is not None else None, C.c_void_p(out.ptr))


def test_raw_c_abi_call_and_its_error_returns(ctx):
  """wbx_ens_binned through raw pointers (what a binding would do): result == oracle with and without prepared atom tables;
  the combinations it does not take come back as WBX_ERR_INVALID with a message, and bins that are no boxes (more than 32
  distinct membership words in a patch) are reported by wbx_ens_binned_atoms and turn the cell NaN instead of wrong."""
  rng = np.random.default_rng(1)
  nlead, m, nlat, nlon, nbin = 2, 8, 40, 200, 9
  tv = rng.normal(size=(nlead, nlat, nlon)).astype(np.float32)
  pv = (tv[:, None] + rng.normal(size=(nlead, m, nlat, nlon))).astype(np.float32)
  dims = ('lead_time', 'latitude', 'longitude')
  sizes = {'lead_time': nlead, 'latitude': nlat, 'longitude': nlon}
  lay_p = planner.InputLayout(strides={'lead_time': m * nlat * nlon, 'latitude': nlon, 'longitude': 1}, itemsize=4, base_alignment=256)
  lay_t = planner.InputLayout(strides={'lead_time': nlat * nlon, 'latitude': nlon, 'longitude': 1}, itemsize=4, base_alignment=256)
  plan = planner.build_s1_plan(dims, sizes, [lay_p, lay_t, None, None], ('latitude', 'longitude'),
                               wdep_dims={'latitude', 'longitude'}, flags=_hip.FLAG_FAIR, allow_vec4=False)
  assert plan.a_dims == ('lead_time',) and plan.br_dims == ('latitude',) and plan.x_dim == 'longitude'
  dplan = engine._PlanOnDevice(ctx, plan)  # pylint: disable=protected-access
  member = rng.random((nlat, nlon, nbin)) < 0.4
  member[:20, :, 0] = True
  member[:, :, nbin - 1] = False
  # box-like bins: a handful of distinct words per patch
  boxy = np.zeros((nlat, nlon, nbin), bool)
  for b in range(nbin - 1):
    boxy[(b * 4) % nlat:(b * 4) % nlat + 18, (b * 23) % nlon:(b * 23) % nlon + 90, b] = True
  wrow = rng.random(nlat) + 0.5
  bufs = {'p': ctx.upload(pv), 't': ctx.upload(tv), 'w': ctx.upload(wrow)}
  w_flags = _hip.BINNED_W_ON_X | _hip.BINNED_WT_ROW_ONLY
  nA, nBk, nBr = nlead, 1, nlat
  lanes = EB.oracle_lanes(pv, ('lead_time', 'number', 'latitude', 'longitude'), tv, dims)
  order = ['CRPSSkill', 'CRPSSpread', 'EnsembleVariance', 'UnbiasedEnsembleMeanSquaredError', 'EnsembleMeanSquaredError']

  def pack(mem):
    bits = np.zeros((nlat, nlon), np.uint64)
    for b in range(nbin):
      bits |= mem[..., b].astype(np.uint64) << np.uint64(b)
    return bits

  def tables(bits_buf):
    nbytes, overflow = C.c_int64(0), C.c_int64(-1)
    _hip.check(ctx.lib.wbx_ens_binned_atoms_size(C.byref(dplan.struct), nA, nBk, nBr, _hip.BINNED_W_ON_X, C.byref(nbytes)), 'size')
    atoms = ctx.alloc(int(nbytes.value))
    _hip.check(ctx.lib.wbx_ens_binned_atoms(ctx.handle, C.byref(dplan.struct), nA, nBk, nBr, _hip.BINNED_W_ON_X,
                                            C.c_void_p(bits_buf.ptr), C.c_void_p(atoms.ptr), C.byref(overflow)), 'atoms')
    return atoms, int(overflow.value)

  out = ctx.alloc(nA * nBk * 6 * nbin * 8)
  bits_buf = ctx.upload(pack(boxy))
  atoms, overflow = tables(bits_buf)
  assert overflow == 0
  for prepared in (atoms, None):
    _hip.check(_raw_call(ctx, plan, dplan, m, nlat * nlon, bufs['p'], bufs['t'], None, bufs['w'], bits_buf, nA, nBk, nBr, w_flags,
                         nbin, prepared, out), 'wbx_ens_binned')
    got = ctx.download(out.ptr, (nA, nBk, 6, nbin), np.float64)
    for l, name in enumerate(order):
      want = np.einsum('ayx,y,yxb->ab', lanes[name][0], wrow, boxy.astype(np.float64))
      np.testing.assert_allclose(got[:, 0, l], want, rtol=RTOL, atol=1e-9, err_msg=name)
    np.testing.assert_allclose(got[:, 0, 5], np.broadcast_to(np.einsum('y,yxb->b', wrow, boxy.astype(np.float64)), (nA, nbin)), rtol=1e-12)
    assert (got[:, 0, :, nbin - 1] == 0).all()
  # random membership: patches with more than 32 distinct words are counted, and their cells come back NaN
  rbits = ctx.upload(pack(member))
  atoms_r, overflow_r = tables(rbits)
  assert overflow_r > 0
  _hip.check(_raw_call(ctx, plan, dplan, m, nlat * nlon, bufs['p'], bufs['t'], None, bufs['w'], rbits, nA, nBk, nBr, w_flags, nbin,
                       atoms_r, out), 'wbx_ens_binned')
  assert np.isnan(ctx.download(out.ptr, (nA, nBk, 6, nbin), np.float64)).all()
  # what it does not take
  def refused(rc, text):
    assert rc == -1 and text in ctx.lib.wbx_last_error().decode(), (rc, ctx.lib.wbx_last_error())
  refused(_raw_call(ctx, plan, dplan, 65, nlat * nlon, bufs['p'], bufs['t'], None, bufs['w'], bits_buf, nA, nBk, nBr, w_flags, nbin,
                    atoms, out), '2..64 members')
  refused(_raw_call(ctx, plan, dplan, m, nlat * nlon, bufs['p'], bufs['t'], None, bufs['w'], bits_buf, nA, nBk, nBr,
                    _hip.BINNED_W_ON_X, nbin, atoms, out), 'factored weights')
  import dataclasses
  plan_m = dataclasses.replace(plan, flags=plan.flags | _hip.FLAG_MASKED)
  dplan_m = engine._PlanOnDevice(ctx, plan_m)  # pylint: disable=protected-access
  mask_buf = ctx.upload(np.ones((nlat, nlon), np.uint8))
  # (ABI 11: a mask with strides along any dim and WBX_FLAG_SKIPNA are taken -- tests/test_gpu_round5.py; what is left:)
  refused(_raw_call(ctx, plan_m, dplan_m, m, nlat * nlon, bufs['p'], bufs['t'], mask_buf, bufs['w'], bits_buf, nA, nBk, nBr,
                    _hip.BINNED_WT_ROW_ONLY, nbin, atoms, out), 'WBX_BINNED_W_ON_X')
  plan_s = dataclasses.replace(plan, flags=plan.flags | _hip.FLAG_SKIPNA_ENS)
  dplan_s = engine._PlanOnDevice(ctx, plan_s)  # pylint: disable=protected-access
  refused(_raw_call(ctx, plan_s, dplan_s, m, nlat * nlon, bufs['p'], bufs['t'], None, bufs['w'], bits_buf, nA, nBk, nBr, w_flags,
                    nbin, atoms, out), 'skipna_ensemble')


# ---- spectra fused into the deterministic launch: several variables, pass order, prefetch -------------------------
def _two_variable_case():
  from weatherbenchx_amd import time_chunks
  rng = np.random.default_rng(18)
  nlat, nlon, ninit, nlead, nlev = 19, 1440, 3, 2, 2
  lat, lon = np.linspace(-81, 81, nlat), np.arange(nlon) * 0.25
  init_times = np.datetime64('2021-06-01T00', 'ns') + np.arange(ninit) * np.timedelta64(24, 'h')
  lead_time = (np.arange(nlead) * 12).astype('timedelta64[h]').astype('timedelta64[ns]')
  level = np.array([500, 850])
  shape = (ninit, nlead, nlev, nlat, nlon)
  data = {v: ((rng.normal(size=shape) * (2 + i) + 270 + 10 * i).astype(np.float32), (rng.normal(size=shape) * 2 + 270 + 10 * i).astype(np.float32))
          for i, v in enumerate(('z', 't'))}
  calls = []

  def load(inits, leads):
    calls.append(1)
    i = [int(np.where(init_times == x)[0][0]) for x in inits]
    cs = {'init_time': inits, 'lead_time': lead_time, 'level': level, 'latitude': lat, 'longitude': lon}
    dims = ('init_time', 'lead_time', 'level', 'latitude', 'longitude')
    return ({v: xr.DataArray(data[v][0][i], dims=dims, coords=cs) for v in data}, {v: xr.DataArray(data[v][1][i], dims=dims, coords=cs) for v in data})
  times = time_chunks.TimeChunks(init_times, lead_time, init_time_chunk_size=1)
  return times, load, calls, ninit


def _run_passes(order, fuse, monkeypatch, prefetch=0):
  from weatherbenchx_amd import pipeline, spectra
  from weatherbenchx_amd.metrics import deterministic
  times, load, calls, ninit = _two_variable_case()
  det = {'rmse': deterministic.RMSE(), 'mae': deterministic.MAE()}
  spec = {'sp': spectra.ZonalPowerSpectrum('predictions'), 'st': spectra.ZonalPowerSpectrum('targets')}
  area = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  zonal = aggregation.Aggregator(reduce_dims=['init_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
  passes = {'det': ('det', load, det, area), 'spec': ('spec', load, spec, zonal)}
  monkeypatch.setattr(engine, 'FUSE_DET_SPECTRA', fuse)
  engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], 1
  try:
    out = pipeline.evaluate_passes(times, [passes[n] for n in order], prefetch=prefetch)
    kinds = [e['kind'] for e in engine.S1_EVENT_LOG]
  finally:
    engine.S1_EVENT_LOG = None
  return out['det'][None].metric_values(det), out['spec'][None].metric_values(spec), kinds, len(calls), ninit


def test_fused_spectra_of_two_variables_do_not_share_a_buffer(ctx, monkeypatch):
  """Two variables on the 1440-point grid through one loader: every deterministic launch of a chunk runs before the spectra
  pass reads the first fused spectrum, so each launch needs its own result buffers (round 3 parked a pointer into ONE scratch
  slot per context: every variable then accumulated the last variable's spectrum)."""
  d1, s1, k1, _, ninit = _run_passes(('det', 'spec'), True, monkeypatch)
  d0, s0, k0, _, _ = _run_passes(('det', 'spec'), False, monkeypatch)
  assert k1.count('det_spectrum') == 2 * ninit and 'spectrum' not in k1
  assert 'det_spectrum' not in k0 and k0.count('spectrum') == 4 * ninit
  assert not np.allclose(s0['sp.z'].values, s0['sp.t'].values, rtol=1e-3)  # the variables really differ
  for k in d0:  # (r6: the fused launch folds stage 2 in: the same sums in another order, tests/test_gpu_round3.py has the bitwise case)
    np.testing.assert_allclose(d1[k].values, d0[k].values, rtol=1e-13, err_msg=k)
  for k in s0:
    np.testing.assert_allclose(s1[k].values, s0[k].values, rtol=1e-12, err_msg=k)


def test_spectra_pass_in_front_of_the_deterministic_pass_does_not_fuse(ctx, monkeypatch):
  """The fused launch only pays when the deterministic pass comes first; the other order runs separate launches, leaves
  nothing parked on the arrays and gives the same numbers."""
  d1, s1, k1, _, ninit = _run_passes(('spec', 'det'), True, monkeypatch)
  d0, s0, _, _, _ = _run_passes(('det', 'spec'), False, monkeypatch)
  assert 'det_spectrum' not in k1 and k1.count('spectrum') == 4 * ninit
  assert not engine._fusion_parks and not engine._fusion_requests  # pylint: disable=protected-access
  for k in d0:
    np.testing.assert_array_equal(d1[k].values, d0[k].values, err_msg=k)
  for k in s0:
    np.testing.assert_allclose(s1[k].values, s0[k].values, rtol=1e-12, err_msg=k)


def test_passes_share_one_feeder_per_loader(ctx, monkeypatch):
  """prefetch > 0: ONE chunk feeder per loader, so the loader runs once per chunk and the passes still fuse."""
  d1, s1, k1, ncalls, ninit = _run_passes(('det', 'spec'), True, monkeypatch, prefetch=1)
  d0, s0, _, _, _ = _run_passes(('det', 'spec'), False, monkeypatch)
  assert ncalls == ninit and k1.count('det_spectrum') == 2 * ninit
  for k in d0:  # (r6: stage 2 folded into the fused launch: another summation order)
    np.testing.assert_allclose(d1[k].values, d0[k].values, rtol=1e-13, err_msg=k)
  for k in s0:
    np.testing.assert_allclose(s1[k].values, s0[k].values, rtol=1e-12, err_msg=k)


# ---- parity at the sizes that run: the latitude-fastest (FLAT) pipelined sweep on the full grid -----------------------------
@pytest.mark.parametrize('rows_per_chunk', [1, 2, 3])
def test_flat_pipelined_sweep_full_grid_every_row_group(ctx, rows_per_chunk):
  """ens_pipe_kernel<51, true, SORT, FLAT> on [level, member, 1440 longitudes, 721 latitudes] with the latitude weights
  folded in -- the kernel of bench.py's lat_fastest main line -- through the raw C ABI, so that EVERY partial it writes is
  seen: one value per (level, run of `rows_per_chunk` longitude rows, lane) against the float64 oracle.  721-float rows put
  every chunk boundary inside a 64-element tile; 2-row chunks are the 37-level field's geometry."""
  rng = np.random.default_rng(40 + rows_per_chunk)
  nlev, m, nlon, nlat = 2, 51, 1440, 721
  lat = np.linspace(-90, 90, nlat)
  tv = (rng.normal(size=(nlev, nlon, nlat)) + 280).astype(np.float32)
  pv = (tv[:, None] + rng.normal(size=(nlev, m, nlon, nlat))).astype(np.float32)
  tv = (tv + rng.normal(size=tv.shape)).astype(np.float32)
  dims = ('level', 'longitude', 'latitude')
  sizes = {'level': nlev, 'longitude': nlon, 'latitude': nlat}
  lay_p = planner.InputLayout(strides={'level': m * nlon * nlat, 'longitude': nlat, 'latitude': 1}, itemsize=4, base_alignment=256)
  lay_t = planner.InputLayout(strides={'level': nlon * nlat, 'longitude': nlat, 'latitude': 1}, itemsize=4, base_alignment=256)
  plan = planner.build_s1_plan(dims, sizes, [lay_p, lay_t, None, None], ('latitude', 'longitude'), wdep_dims=set(),
                               flags=_hip.FLAG_FAIR, allow_vec4=False, fold_x='point64')
  import dataclasses
  plan = dataclasses.replace(plan, depth_chunk=rows_per_chunk, nchunk=-(-nlon // rows_per_chunk))  # (the planner cuts 1 .. 4 rows)
  wt = O.grid_area_weights(lat)
  plan.x_weights = np.ascontiguousarray(wt)
  assert plan.block_threads == 64 and plan.plane_rows == nlon and not plan.x_kept and plan.depth_chunk == rows_per_chunk, plan
  dplan = engine._PlanOnDevice(ctx, plan)  # pylint: disable=protected-access
  bufs = ctx.upload(pv), ctx.upload(tv)
  out = ctx.alloc(plan.nkey * plan.nchunk * 5 * 8)
  _hip.check(ctx.lib.wbx_ens_partial(ctx.handle, C.byref(dplan.struct), _hip.F32, m, nlon * nlat, _hip.ENS_SORT, C.c_void_p(bufs[0].ptr),
                                     C.c_void_p(bufs[1].ptr), None, C.c_void_p(out.ptr)), 'wbx_ens_partial')
  got = ctx.download(out.ptr, (nlev, plan.nchunk, 5), np.float64)
  lanes = EB.oracle_lanes(pv, ('level', 'number', 'longitude', 'latitude'), tv, dims)
  order = ['CRPSSkill', 'CRPSSpread', 'EnsembleVariance', 'UnbiasedEnsembleMeanSquaredError', 'EnsembleMeanSquaredError']
  for l, name in enumerate(order):
    rows = (lanes[name][0] * wt).sum(axis=-1)  # [level, longitude]
    pad = plan.nchunk * rows_per_chunk - nlon
    want = np.pad(rows, ((0, 0), (0, pad))).reshape(nlev, plan.nchunk, rows_per_chunk).sum(axis=-1)
    np.testing.assert_allclose(got[:, :, l], want, rtol=RTOL, err_msg=name)


def test_ifs_layout_m50_full_grid_through_the_api(ctx):
  """The recorded IFS-ENS chunk layout (init_time, number, lead_time, longitude, latitude), docs/source/how_to/
  metric_wrappers.ipynb:955-964 -- a member's planes are NOT adjacent (member stride = lead x lon x lat) -- with the 50
  perturbed members on the full 0.25 degree grid and the default aggregator: ONE launch of the pipelined one-wave sweep over
  contiguous planes with folded latitude weights (asserted from the event log), results == oracle."""
  rng = np.random.default_rng(50)
  ninit, m, nlead, nlon, nlat = 1, 50, 2, 1440, 721
  lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
  pd, td = ('init_time', 'number', 'lead_time', 'longitude', 'latitude'), ('init_time', 'lead_time', 'longitude', 'latitude')
  tv = (rng.normal(size=(ninit, nlead, nlon, nlat)) * 3 + 5.5e4).astype(np.float32)  # geopotential-like magnitudes
  pv = (tv[:, None] + rng.normal(size=(ninit, m, nlead, nlon, nlat)) * 30).astype(np.float32)
  tv = (tv + rng.normal(size=tv.shape) * 30).astype(np.float32)
  coords = {'init_time': np.array(['2020-01-01'], dtype='datetime64[ns]'), 'lead_time': (np.arange(nlead) * 12).astype('timedelta64[h]').astype('timedelta64[ns]'),
            'latitude': lat, 'longitude': lon}
  p = xr.DataArray(pv, dims=pd, coords=coords)
  t = xr.DataArray(tv, dims=td, coords=coords)
  stats = EB.lane_statistics()
  agg = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  state, log = EB.run(stats, agg, p, t)
  ens = [e for e in log if e['kind'] == 'ens']
  assert len(log) == 1 and len(ens) == 1 and ens[0]['flat'] and ens[0]['block'] == 64 and ens[0]['algo'] == 0, log
  means = state.mean_statistics()
  wt = O.grid_area_weights(lat)
  for name, (lane, ldims) in EB.oracle_lanes(pv, pd, tv, td).items():
    assert ldims == td
    want = (lane * wt).sum(axis=(0, 2, 3)) / (wt.sum() * ninit * nlon)
    got = np.asarray(means[stats[name].unique_name]['v'].values)
    np.testing.assert_allclose(got, want, rtol=RTOL, err_msg=name)


# ---- the fp32 chain sums of the pipelined kernels where they are weakest ----------------------
def _row_means(ctx, pv, tv, lat_rows):
  """All five lanes reduced over `longitude` only (64 points = ONE tile of the pipelined sweep per output value)."""
  nrow, nx = tv.shape
  coords = {'latitude': np.linspace(-80, 80, nrow), 'longitude': np.arange(nx) * (360.0 / nx)}
  p = xr.DataArray(pv, dims=('number', 'latitude', 'longitude'), coords=coords)
  t = xr.DataArray(tv, dims=('latitude', 'longitude'), coords=coords)
  stats = EB.lane_statistics()
  agg = aggregation.Aggregator(reduce_dims=['longitude'])
  state, log = EB.run(stats, agg, p, t)
  assert [e['kind'] for e in log] == ['ens'] and log[0]['block'] == 64 and not log[0]['x_kept'], log  # the pipelined sweep
  means = state.mean_statistics()
  return {name: np.asarray(means[s.unique_name]['v'].values) for name, s in stats.items()}


@pytest.mark.parametrize('m', [50, 51])
@pytest.mark.parametrize('case', ['geopotential', 'overconfident', 'near_perfect_mean'])
def test_fp32_chain_sums_where_they_are_weakest(ctx, case, m):
  """ens_pipe_kernel sums e = x - median in fp32 chains (wbx_ens_impl.hpp, stats32).  Held here on rows of 64 points (one tile
  per output, nothing averages out) and on the whole field:
    geopotential       5.5e4 +- 30: large magnitude, tight ensemble
    overconfident      spread = 0.05 x error: lane 3 ~ lane 4, variance tiny
    near_perfect_mean  the target sits at the ensemble mean + noise of variance var / M: lane 3 = (mean - t)^2 - var / M has a
                       mean ~ 1e-3 x lane 4 -- the difference of two nearly equal terms
  Lanes 0, 1, 2, 4: 1e-6 relative per row.  Lane 3 is a difference; its bound is absolute, in units of the two terms it is
  the difference of: |d lane3| <= 1e-6 (lane 4 + lane 2 / M) per row (include/wbx.h)."""
  rng = np.random.default_rng({'geopotential': 1, 'overconfident': 2, 'near_perfect_mean': 3}[case] * 100 + m)
  nrow, nx = 2048, 64
  if case == 'geopotential':
    tv = (rng.normal(size=(nrow, nx)) * 300 + 5.5e4).astype(np.float32)
    pv = (tv[None] + rng.normal(size=(m, nrow, nx)) * 30).astype(np.float32)
    tv = (tv + rng.normal(size=tv.shape) * 30).astype(np.float32)
  elif case == 'overconfident':
    tv = (rng.normal(size=(nrow, nx)) * 5 + 280).astype(np.float32)
    pv = (tv[None] + rng.normal(size=(1, nrow, nx)) + rng.normal(size=(m, nrow, nx)) * 0.05).astype(np.float32)
  else:
    base = rng.normal(size=(nrow, nx)) * 5 + 280
    pv = (base[None] + rng.normal(size=(m, nrow, nx))).astype(np.float32)
    p64 = pv.astype(np.float64)
    tv = (p64.mean(axis=0) + rng.normal(size=(nrow, nx)) * np.sqrt(p64.var(axis=0, ddof=1) / m * 1.001)).astype(np.float32)
  got = _row_means(ctx, pv, tv, nrow)
  pd, td = ('number', 'latitude', 'longitude'), ('latitude', 'longitude')
  want = {k: v[0].mean(axis=-1) for k, v in EB.oracle_lanes(pv, pd, tv, td).items()}
  for name in ('CRPSSkill', 'CRPSSpread', 'EnsembleVariance', 'EnsembleMeanSquaredError'):
    np.testing.assert_allclose(got[name], want[name], rtol=RTOL, err_msg=f'{case} M={m} {name}')
  scale3 = want['EnsembleMeanSquaredError'] + want['EnsembleVariance'] / m
  d3 = np.abs(got['UnbiasedEnsembleMeanSquaredError'] - want['UnbiasedEnsembleMeanSquaredError'])
  assert (d3 <= 1e-6 * scale3).all(), (case, m, float((d3 / scale3).max()))
  # the whole field (131072 points): every lane to 1e-6 relative where its mean is not itself a cancellation; lane 3 of the
  # near-perfect case to its absolute bound
  for name in got:
    g, w = got[name].mean(), want[name].mean()
    if name == 'UnbiasedEnsembleMeanSquaredError':
      assert abs(g - w) <= 1e-6 * scale3.mean(), (case, m, name, g, w)
      if case != 'near_perfect_mean':
        assert abs(g / w - 1) < 1e-6, (case, m, name, g, w)
    else:
      assert abs(g / w - 1) < 1e-6, (case, m, name, g, w)
  if case == 'near_perfect_mean':
    ratio = want['UnbiasedEnsembleMeanSquaredError'].mean() / want['EnsembleMeanSquaredError'].mean()
    assert abs(ratio) < 0.05, ratio  # the case is what it says


def test_crps_ensemble_with_few_points_per_output_cell(ctx):
  """CRPS = skill - spread / 2 cancels by a factor 3-4 on a calibrated ensemble: per output of 64 points (not a mean over 10^6)
  the metric itself -- not only its two statistics -- holds 1e-6 against the oracle."""
  rng = np.random.default_rng(77)
  m, nrow, nx = 51, 4096, 64
  tv = (rng.normal(size=(nrow, nx)) * 8 + 285).astype(np.float32)
  pv = (tv[None] + rng.normal(size=(m, nrow, nx))).astype(np.float32)
  tv = (tv + rng.normal(size=tv.shape)).astype(np.float32)
  got = _row_means(ctx, pv, tv, nrow)
  pd, td = ('number', 'latitude', 'longitude'), ('latitude', 'longitude')
  lanes = EB.oracle_lanes(pv, pd, tv, td)
  want = O.crps(lanes['CRPSSkill'][0].mean(axis=-1), lanes['CRPSSpread'][0].mean(axis=-1))
  crps = got['CRPSSkill'] - 0.5 * got['CRPSSpread']
  np.testing.assert_allclose(crps, want, rtol=RTOL)
  assert np.abs(crps / want - 1).max() < 5e-7


# ---- spectra + deterministic lanes in one sweep over LATITUDE-FASTEST fields (wbx_det_spectrum_slabs) ------------------------
@pytest.mark.parametrize('func,nlat', [('DET6', 37), ('DET3', 37), ('DET6', 8), ('DET6', 721)])
def test_det_spectrum_slabs_entry_point_against_the_oracle(ctx, func, nlat):
  """wbx_det_spectrum_slabs with raw pointers on [.., longitude, latitude] fields (the archives' layout,
  data_loaders/xarray_loaders.py:185-188): the plan's x is the STRIDED longitude, a slab = the `nlat` adjacent rows of one
  (lead, level); 37 rows = runs of 8 + a short one with a lone last row, 8 rows = one run, 721 = the real grid.  The partial
  buffer against wbx_det_partial on the same plan and against the float64 oracle, both spectra against numpy.fft within
  the transform's bound; the climatology through a gather table (another slot per lead)."""
  from test_spectra import bound_1440
  rng = np.random.default_rng(5)
  nlead, nlev, nlon = (3, 2, 1440) if nlat < 721 else (2, 1, 1440)
  dims = ('init_time', 'lead_time', 'level', 'longitude', 'latitude')
  shape = (1, nlead, nlev, nlon, nlat)
  pv = (rng.normal(size=shape) * 3 + 280).astype(np.float32)
  tv = (rng.normal(size=shape) * 3 + 280).astype(np.float32)
  nslot = 5
  cv = (rng.normal(size=(nslot, nlev, nlon, nlat)) * 10 + 280).astype(np.float32)
  slot_of_lead = np.array([4, 0, 2])[:nlead]
  p, t = xr.DataArray(pv, dims=dims), xr.DataArray(tv, dims=dims)
  c = xr.DataArray(cv, dims=('slot', 'level', 'longitude', 'latitude'))
  devs = [engine._to_device(ctx, a, _hip.F32) for a in (p, t, c)] + [None]
  lays = [d.layout for d in devs[:3]] + [None]
  sizes = dict(zip(dims, shape))
  table = (slot_of_lead * devs[2].layout.stride('slot')).reshape(1, nlead).astype(np.int64)
  gather = planner.GatherSpec(dims=('init_time', 'lead_time'), table=table) if func == 'DET6' else None
  if func == 'DET3':
    lays[2] = None
  plan = planner.build_s1_plan(dims, sizes, lays, ['init_time', 'latitude', 'longitude'], wdep_dims=['latitude'], gather=gather,
                               force_x_dim='longitude', allow_vec4=False)
  assert plan.ndepth == 1 and plan.nchunk == 1 and plan.nkey == nlead * nlev * nlat and plan.key_dims == ('lead_time', 'level', 'latitude')
  assert plan.xstride[0] == nlat and not plan.x_kept
  dplan = engine._device_plan(ctx, plan)
  code = getattr(_hip, func)
  nl = _hip.DET_LANES[code]
  nrows, ngroup, nk = plan.nkey, nlead * nlev, nlon // 2 + 1
  w = np.cos(np.deg2rad(np.linspace(-88, 88, nlat)))
  group = np.repeat(np.arange(ngroup, dtype=np.int32), nlat)
  scale = np.tile(w, ngroup)
  g_dev, s_dev = ctx.upload(group), ctx.upload(scale)
  part_a, part_b = ctx.alloc(nrows * nl * 8), ctx.alloc(nrows * nl * 8)
  pw_p, pw_t = ctx.alloc(ngroup * nk * 8), ctx.alloc(ngroup * nk * 8)
  ptr = lambda d: C.c_void_p(d.ptr) if d is not None else None
  cdev = devs[2] if func == 'DET6' else None
  _hip.check(ctx.lib.wbx_det_partial(ctx.handle, C.byref(dplan.struct), code, _hip.F32, ptr(devs[0]), ptr(devs[1]), ptr(cdev), None,
                                     ptr(part_a)), 'wbx_det_partial')
  for _ in range(2):  # (twice: the second launch must not see anything of the first)
    _hip.check(ctx.lib.wbx_det_spectrum_slabs(ctx.handle, C.byref(dplan.struct), code, _hip.F32, ptr(devs[0]), ptr(devs[1]), ptr(cdev),
                                              nlat, ptr(g_dev), ptr(s_dev), ngroup, ptr(part_b), ptr(pw_p), ptr(pw_t)),
               'wbx_det_spectrum_slabs')
  ctx.synchronize()
  a = ctx.download(part_a.ptr, (nrows, nl)).copy()
  b = ctx.download(part_b.ptr, (nrows, nl)).copy()
  np.testing.assert_allclose(b, a, rtol=1e-13, atol=1e-9)
  p64, t64 = np.swapaxes(pv.astype(np.float64)[0], -1, -2), np.swapaxes(tv.astype(np.float64)[0], -1, -2)  # [lead, level, lat, lon]
  c64 = np.swapaxes(cv.astype(np.float64)[slot_of_lead], -1, -2)
  want = [O.error(p64, t64), O.absolute_error(p64, t64), O.squared_error(p64, t64)]
  if func == 'DET6':
    want += [O.squared_prediction_anomaly(p64, c64), O.squared_target_anomaly(t64, c64), O.anomaly_covariance(p64, t64, c64)]
  for lane, wv in enumerate(want):
    np.testing.assert_allclose(b[:, lane], wv.sum(axis=-1).reshape(-1), rtol=1e-11, atol=1e-6, err_msg=f'lane {lane}')
  for buf, f64 in ((pw_p, p64), (pw_t, t64)):
    got = ctx.download(buf.ptr, (ngroup, nk)).copy()
    ref = (O.zonal_power_spectrum(f64) * w[None, None, :, None]).sum(axis=2).reshape(ngroup, nk)
    assert np.all(np.abs(got - ref) <= bound_1440(ref))
  # misuse is refused, not mis-run: a slab length that does not divide the keys, a statistic family without spectra
  assert ctx.lib.wbx_det_spectrum_slabs(ctx.handle, C.byref(dplan.struct), code, _hip.F32, ptr(devs[0]), ptr(devs[1]), ptr(cdev),
                                        nlat + 1, ptr(g_dev), ptr(s_dev), ngroup, ptr(part_b), ptr(pw_p), ptr(pw_t)) == -1
  assert ctx.lib.wbx_det_spectrum_slabs(ctx.handle, C.byref(dplan.struct), _hip.PASS1, _hip.F32, ptr(devs[0]), ptr(devs[1]), ptr(cdev),
                                        nlat, ptr(g_dev), ptr(s_dev), ngroup, ptr(part_b), ptr(pw_p), ptr(pw_t)) == -1


def test_latitude_fastest_chunks_fuse_spectra_into_the_deterministic_sweep(ctx, monkeypatch):
  """The chunk loop on latitude-fastest arrays with engine.FUSE_DET_SPECTRA_LATFAST: RMSE / ACC / MAE per (lead, level) under
  the area aggregator and the zonal spectra of predictions and targets under another -- ONE launch per chunk (det_spectrum
  with slab_rows = nlat), results equal
  to the separate launches (deterministic: same fp64 formulas, other summation order; spectra: same transform) and to the
  same data stored longitude-fastest."""
  from weatherbenchx_amd import pipeline, spectra, time_chunks
  from weatherbenchx_amd.metrics import deterministic
  rng = np.random.default_rng(18)
  nlat, nlon, ninit, nlead, nlev = 27, 1440, 3, 2, 3
  lat, lon = np.linspace(-84, 84, nlat), np.arange(nlon) * 0.25
  init_times = np.datetime64('2021-06-01T00', 'ns') + np.arange(ninit) * np.timedelta64(24, 'h')
  lead_time = (np.arange(nlead) * 12).astype('timedelta64[h]').astype('timedelta64[ns]')
  level = np.array([500, 700, 850])
  pv = (rng.normal(size=(ninit, nlead, nlev, nlon, nlat)) * 2 + 270).astype(np.float32)
  tv = (rng.normal(size=(ninit, nlead, nlev, nlon, nlat)) * 2 + 270).astype(np.float32)
  cvals = (rng.normal(size=(8, 2, nlev, nlon, nlat)) * 5 + 270).astype(np.float32)

  def clim_of(latfast):
    dims = ('dayofyear', 'hour', 'level') + (('longitude', 'latitude') if latfast else ('latitude', 'longitude'))
    vals = cvals if latfast else np.ascontiguousarray(np.swapaxes(cvals, -1, -2))
    return xr.Dataset({'z': xr.DataArray(vals, dims=dims, coords={'dayofyear': np.arange(152, 160), 'hour': np.array([0, 12]),
                                                                 'level': level, 'latitude': lat, 'longitude': lon})})

  def loader(latfast):
    def load(inits, leads):
      i = [int(np.where(init_times == x)[0][0]) for x in inits]
      cs = {'init_time': inits, 'lead_time': lead_time, 'level': level, 'latitude': lat, 'longitude': lon}
      dims = ('init_time', 'lead_time', 'level') + (('longitude', 'latitude') if latfast else ('latitude', 'longitude'))
      a, b = (pv[i], tv[i]) if latfast else (np.ascontiguousarray(np.swapaxes(pv[i], -1, -2)), np.ascontiguousarray(np.swapaxes(tv[i], -1, -2)))
      return {'z': xr.DataArray(a, dims=dims, coords=cs)}, {'z': xr.DataArray(b, dims=dims, coords=cs)}
    return load
  spec = {'sp': spectra.ZonalPowerSpectrum('predictions'), 'st': spectra.ZonalPowerSpectrum('targets')}
  area = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
  zonal = aggregation.Aggregator(reduce_dims=['init_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
  times = time_chunks.TimeChunks(init_times, lead_time, init_time_chunk_size=1)

  def run(latfast, fuse):
    det = {'rmse': deterministic.RMSE(), 'acc': deterministic.ACC(clim_of(latfast)), 'mae': deterministic.MAE()}
    monkeypatch.setattr(engine, 'FUSE_DET_SPECTRA', fuse)
    monkeypatch.setattr(engine, 'FUSE_DET_SPECTRA_LATFAST', fuse)  # (opt-in: measured slower than the launches it replaces)
    engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], 1
    try:
      load = loader(latfast)
      out = pipeline.evaluate_passes(times, [('det', load, det, area), ('spec', load, spec, zonal)])
      log = list(engine.S1_EVENT_LOG)
    finally:
      engine.S1_EVENT_LOG = None
    return out['det'][None].metric_values(det), out['spec'][None].metric_values(spec), log
  d1, s1, l1 = run(True, True)
  d0, s0, l0 = run(True, False)
  d2, s2, l2 = run(False, True)
  k1, k0 = [e['kind'] for e in l1], [e['kind'] for e in l0]
  assert k1.count('det_spectrum') == ninit and 'spectrum' not in k1 and 'det' not in k1, k1
  assert all(e['slab_rows'] == nlat for e in l1 if e['kind'] == 'det_spectrum')
  assert all(e.get('slab_rows', 0) == 0 for e in l2 if e['kind'] == 'det_spectrum')
  assert 'det_spectrum' not in k0 and k0.count('spectrum') == 2 * ninit
  for k in d0:
    np.testing.assert_allclose(d1[k].values, d0[k].values, rtol=1e-12, err_msg=k)
    np.testing.assert_allclose(d1[k].values, d2[k].values, rtol=1e-12, err_msg=k)
  for k in s0:
    # (same transform; the mean shift is estimated from other samples of the row: fp32 rounding of another shifted row)
    np.testing.assert_allclose(s1[k].values, s0[k].values, rtol=2e-6, err_msg=k)
    np.testing.assert_allclose(s1[k].values, s2[k].values, rtol=2e-6, err_msg=k)


# ---- skipna_ensemble: the register-resident rank form over the valid members (WBX_ENS_SKIPNA_SORT) ---------------------------
@pytest.mark.parametrize('m', [51, 50, 13, 33, 3])
def test_skipna_ensemble_register_kernel_against_the_oracle(ctx, m):
  """skipna_ensemble=True on float32 ensembles (probabilistic.py:139-145, 206-216, 303-336): per-point member counts from 0 to
  M -- points without a member, with one, with two, with all -- an infinite member, a NaN and an infinite target; every lane
  per POINT (the map path) and under a (latitude, longitude) reduction with skipna, against the float64 restatement."""
  from weatherbenchx_amd.metrics import probabilistic
  rng = np.random.default_rng(100 + m)
  nlat, nlon = 24, 160
  lat, lon = np.linspace(-90, 90, nlat), np.arange(nlon) * (360.0 / nlon)
  pv = (rng.normal(size=(2, m, nlat, nlon)) * 3 + 280).astype(np.float32)
  tv = (rng.normal(size=(2, nlat, nlon)) * 3 + 280).astype(np.float32)
  pv[rng.random(pv.shape) < 0.2] = np.nan
  pv[0, :, 3, 4] = np.nan                 # no member
  pv[0, 1:, 5, 6] = np.nan                # one member
  pv[0, 2:, 5, 7] = np.nan                # two members
  pv[1, :, 7, 8] = rng.normal(size=m)     # all members
  pv[1, m // 2, 9, 10] = np.inf           # an infinite member: the generic operator's point
  tv[1, 11, 12] = np.nan
  tv[1, 11, 13] = np.inf
  pdims, tdims = ('time', 'number', 'latitude', 'longitude'), ('time', 'latitude', 'longitude')
  coords = {'latitude': lat, 'longitude': lon}
  p = {'v': xr.DataArray(pv, dims=pdims, coords=coords)}
  t = {'v': xr.DataArray(tv, dims=tdims, coords=coords)}
  stats = {'skill': probabilistic.CRPSSkill(skipna_ensemble=True),
           'spread': probabilistic.CRPSSpread(use_sort=False, fair=True, skipna_ensemble=True),
           'var': probabilistic.EnsembleVariance(skipna_ensemble=True),
           'uemse': probabilistic.UnbiasedEnsembleMeanSquaredError(skipna_ensemble=True)}
  with np.errstate(all='ignore'):
    want = {'skill': O.crps_skill(pv, pdims, tv, tdims, 'number', skipna_ensemble=True),
            'spread': O.crps_spread(pv, pdims, 'number', fair=True, skipna_ensemble=True),
            'var': O.ensemble_variance(pv, pdims, 'number', skipna_ensemble=True),
            'uemse': O.unbiased_ensemble_mean_squared_error(pv, pdims, tv, tdims, 'number', skipna_ensemble=True)}
  engine.S1_EVENT_LOG, engine.S1_EVENT_REPEAT = [], 1
  try:
    for k, s in stats.items():
      got = np.asarray(s.compute(p, t)['v'].values)  # per point: wbx_ens_map
      ref, rdims = want[k]
      ref = np.asarray(ref)
      assert got.shape == ref.shape, (k, got.shape, ref.shape)
      with np.errstate(all='ignore'):
        # a difference of two nearly equal numbers (unbiased MSE of a point with two members close to the target) is held
        # absolutely to the size of its terms
        scale = np.nan_to_num(np.abs(ref), nan=0.0, posinf=0.0) + (np.nan_to_num(want['var'][0], nan=0.0, posinf=0.0) if k == 'uemse' else 0.0)
      ok = np.isclose(got, ref, rtol=RTOL, atol=0.0, equal_nan=True) | (np.abs(got - ref) <= RTOL * scale)
      assert ok.all(), (k, np.argwhere(~ok)[:5], got[~ok][:5], ref[~ok][:5])
    agg = aggregation.Aggregator(reduce_dims=['latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()], skipna=True)
    red = aggregation.compute_metric_values_for_single_chunk(stats, agg, p, t)
    log = list(engine.S1_EVENT_LOG)
  finally:
    engine.S1_EVENT_LOG = None
  assert any(e['kind'] == 'ens' and e['flags'] & _hip.FLAG_SKIPNA_ENS for e in log)
  w = (O.grid_area_weights(lat), ('latitude',))
  with np.errstate(all='ignore'):
    for k, (vals, dims) in want.items():
      sws, sw, _ = O.aggregate(vals, dims, ['latitude', 'longitude'], weights=[w], skipna=True)
      np.testing.assert_allclose(red[f'{k}.v'].values, sws / sw, rtol=RTOL, err_msg=k)
