"""ms per chunk of the chunk loop (pipeline.evaluate_chunks, accumulators in HBM) with chunk records on and off, on the public
benchmark's chunks at full size: deterministic (12 lead x 13 level, RMSE/MSE/bias/ACC/activity, 34 bins, masked), probabilistic
(8 lead x 51 members; no mask / (lat, lon) mask / NaN mask per lead) and a spectra + deterministic composite.
usage: bench_replay.py [det|ens|ens_mask|ens_nan|spec] [lon_fastest|lat_fastest|ifs] [n=60]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from weatherbenchx_amd import aggregation, binning, engine, pipeline, replay, spectra, time_chunks, weighting
from weatherbenchx_amd import data as wdata
from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.metrics import deterministic, probabilistic, wrappers
from wb_regions import REGIONS

args = sys.argv[1:]
which = next((a for a in args if a in ('det', 'ens', 'ens_mask', 'ens_nan', 'spec')), 'det')
layout = next((a for a in args if a in ('lon_fastest', 'lat_fastest', 'ifs')), 'lon_fastest')
n = int(next((a[2:] for a in args if a.startswith('n=')), 60))
nlat, nlon = 721, 1440
lat, lon = np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)
sp = ('latitude', 'longitude') if layout == 'lon_fastest' else ('longitude', 'latitude')
shp = tuple({'latitude': nlat, 'longitude': nlon}[d] for d in sp)
inits = np.datetime64('2020-01-01T00', 'ns') + np.arange(n) * np.timedelta64(24, 'h')
index = {int(t.astype('int64')): i for i, t in enumerate(inits)}
land = (np.sin(np.deg2rad(lon) * 3)[None, :] * np.cos(np.deg2rad(lat) * 2.5)[:, None]
        + 0.3 * np.sin(np.deg2rad(lon) * 17)[None, :] * np.sin(np.deg2rad(lat) * 13)[:, None]) > 0.35
lsm = xr.DataArray(land, dims=('latitude', 'longitude'), coords={'latitude': lat, 'longitude': lon})
area = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()],
                              bin_by=[binning.Regions(REGIONS, land_sea_mask=lsm)], masked=True)
NPOOL = 2

if which in ('det', 'spec'):
  nlead, nlev = (12, 13) if which == 'det' else (8, 37)
  lead = (np.arange(nlead) * 12).astype('timedelta64[h]').astype('timedelta64[ns]')
  level = np.arange(nlev)
  dims = ('init_time', 'lead_time', 'level') + sp
  pool = [(torch.randn((1, nlead, nlev) + shp, device='cuda') + 280, torch.randn((1, nlead, nlev) + shp, device='cuda') + 280) for _ in range(NPOOL)]
  clim = xr.Dataset({'z': xr.DataArray(torch.randn((14, 4, nlev) + shp, device='cuda') * 10 + 280, dims=('dayofyear', 'hour', 'level') + sp,
                                       coords={'dayofyear': np.arange(1, 15), 'hour': np.array([0, 6, 12, 18]), 'level': level,
                                               'latitude': lat, 'longitude': lon})})
  ring = np.datetime64('2020-01-01T00', 'ns') + np.arange(6) * np.timedelta64(24, 'h')

  def load(ic, lc):
    i = index[int(ic[0].astype('int64'))]
    cs = {'init_time': ic, 'lead_time': lead, 'level': level, 'latitude': lat, 'longitude': lon,
          'valid_time': (('init_time', 'lead_time'), ring[i % 6] + lead[None, :])}
    return {'z': xr.DataArray(pool[i % NPOOL][0], dims=dims, coords=cs)}, {'z': xr.DataArray(pool[i % NPOOL][1], dims=dims, coords=cs)}
  det = {'rmse': deterministic.RMSE(), 'mse': deterministic.MSE(), 'bias': deterministic.Bias(), 'acc': deterministic.ACC(clim),
         'prediction_activity': deterministic.PredictionActivity(clim)}
  if which == 'det':
    passes = [('det', load, det, area)]
    nbytes = nlead * nlev * nlat * nlon * 12
  else:
    plain = aggregation.Aggregator(reduce_dims=['init_time', 'latitude', 'longitude'], weigh_by=[weighting.GridAreaWeighting()])
    zonal = aggregation.Aggregator(reduce_dims=['init_time', 'latitude'], weigh_by=[weighting.GridAreaWeighting()])
    spec = {'spectrum_p': spectra.ZonalPowerSpectrum('predictions'), 'spectrum_t': spectra.ZonalPowerSpectrum('targets')}
    passes = [('det', load, det, plain), ('spectra', load, spec, zonal)]
    nbytes = nlead * nlev * nlat * nlon * 12
else:
  nlead, m = 8, 51
  lead = (np.arange(nlead) * 12).astype('timedelta64[h]').astype('timedelta64[ns]')
  if layout == 'ifs':
    sp = ('longitude', 'latitude')
    shp = (nlon, nlat)
    pd, td = ('init_time', 'number', 'lead_time') + sp, ('init_time', 'lead_time') + sp
  else:
    pd, td = ('init_time', 'lead_time', 'number') + sp, ('init_time', 'lead_time') + sp
  pool = []
  for k in range(NPOOL):
    t_t = torch.randn((1, nlead) + shp, device='cuda') + 280
    p_t = (t_t[:, None] if layout == 'ifs' else t_t[:, :, None]) + torch.randn(((1, m, nlead) if layout == 'ifs' else (1, nlead, m)) + shp, device='cuda')
    mask_da = None
    if which == 'ens_nan':
      holes = np.stack([(np.abs(lat)[:, None] > 80 - 2 * l) & (np.cos(np.deg2rad(lon) * (5 + l + k))[None, :] > 0.2) for l in range(nlead)])
      hv = holes if sp == ('latitude', 'longitude') else np.ascontiguousarray(np.swapaxes(holes, 1, 2))
      t_t[0][torch.as_tensor(hv, device='cuda')] = float('nan')
      mask_da = wdata.add_nan_mask_to_data({'v': xr.DataArray(t_t, dims=td)})['v'].coords['mask']
    pool.append((p_t, t_t, mask_da))
  valid = ~((np.abs(lat)[:, None] > 80) & (np.cos(np.deg2rad(lon) * 5)[None, :] > 0.2))
  mv = valid if sp == ('latitude', 'longitude') else np.ascontiguousarray(valid.T)
  ll_mask = xr.DataArray(torch.as_tensor(mv, device='cuda'), dims=sp)

  def load(ic, lc):
    i = index[int(ic[0].astype('int64'))]
    p_t, t_t, nan_da = pool[i % NPOOL]
    cs = {'init_time': ic, 'lead_time': lead, 'latitude': lat, 'longitude': lon}
    t = xr.DataArray(t_t, dims=td, coords=cs)
    if which == 'ens_nan':
      t = t.assign_coords(mask=nan_da)
    elif which == 'ens_mask':
      t = t.assign_coords(mask=ll_mask)
    return {'v': xr.DataArray(p_t, dims=pd, coords=cs)}, {'v': t}
  ens = {'crps': probabilistic.CRPSEnsemble(use_sort=True), 'unbiased_spread_skill': probabilistic.UnbiasedSpreadSkillRatio(),
         'unbiased_mean_rmse': probabilistic.UnbiasedEnsembleMeanRMSE(),
         'mean_rmse': wrappers.WrappedMetric(deterministic.RMSE(), [wrappers.EnsembleMean(which='predictions')])}
  passes = [('ens', load, ens, area)]
  nbytes = nlead * nlat * nlon * (m + 1) * 4
torch.cuda.synchronize()
lead_all = lead


def run(k, enabled, alternate=True):
  replay.ENABLED = enabled
  engine.ALTERNATE_CHUNKS = alternate
  replay.reset_stats()
  times = time_chunks.TimeChunks(inits[:k], lead_all, init_time_chunk_size=1)
  t0 = time.perf_counter()
  out = pipeline.evaluate_passes(times, passes)
  first = next(iter(out.values()))[None]
  vals = first.metric_values(passes[0][2])
  dt = time.perf_counter() - t0
  key = sorted(vals)[0]
  return dt / k * 1e3, float(np.asarray(vals[key].values).reshape(-1)[0]), dict(replay.STATS)


res = {'which': which, 'layout': layout, 'chunks': n, 'GB_per_chunk': round(nbytes / 1e9, 3)}
for tag, enabled, alt in (('ordinary_one_stream', False, False), ('replay_one_stream', True, False), ('ordinary', False, True), ('replay', True, True)) * 2:
  run(8, enabled, alt)  # warm (plans, tables, pools)
  ms, check, stats = run(n, enabled, alt)
  res.setdefault(tag + '_ms_per_chunk', []).append(round(ms, 4))
  res[tag + '_check'] = check
  if enabled:
    res['stats'] = {k: v for k, v in stats.items() if k != 'refusals'}
    res['refusals'] = stats['refusals'][:3]
# kernel time per chunk on the ordinary path (HIP event marks, nothing waits)
replay.ENABLED = False
engine.S1_EVENT_LOG, engine.S1_EVENT_MARKS = [], True
times = time_chunks.TimeChunks(inits[:12], lead_all, init_time_chunk_size=1)
pipeline.evaluate_passes(times, passes)
torch.cuda.synchronize()
log = engine.resolve_event_marks(engine.S1_EVENT_LOG)
engine.S1_EVENT_LOG, engine.S1_EVENT_MARKS = None, False
res['kernel_ms_per_chunk'] = round(sum(e['ms'] for e in log) / 12, 4)
res['launch_kinds'] = sorted({e['kind'] for e in log})
res['replay_over_kernel'] = round(min(res['replay_ms_per_chunk']) / res['kernel_ms_per_chunk'], 3)
res['frac_of_hbm_peak_replay'] = round(nbytes / (min(res['replay_ms_per_chunk']) * 1e-3) / 8e12, 4)
print(json.dumps(res))
