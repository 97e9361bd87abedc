"""CPU oracle: a plain-NumPy float64 restatement of WeatherBench-X's scoring hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under weatherbenchx_amd/ imports this module; it is used by tests/,
by __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg, as the checker / the timed CPU baseline.

The reference (google-research/weatherbenchX, pure Python on xarray/NumPy) cannot be imported in the build
container or on the GPU box (no xarray/jax/absl/apache_beam), so parity is pinned the other way
round: this restatement reproduces every analytic / brute-force known answer the reference's own tests hold
for the path (tests/test_oracle_reference_pins.py restates them; list in SURVEY section 8c), and the HIP
path is then compared against this restatement.  Zonal spectra have no reference at all:
"parity unpinned" for that function.

Every function names the reference lines it follows (paths relative to the reference checkout).
Arrays are plain ndarrays plus a tuple of dimension names; all arithmetic is float64 on float64-cast
inputs (the reference's own numerics follow the input dtype).
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------------------------------
# name-based broadcasting helpers (what xarray does for the reference)


def union_dims(*dim_tuples):
  out = []
  for dims in dim_tuples:
    for d in dims:
      if d not in out:
        out.append(d)
  return tuple(out)


def expand_to(arr, dims, out_dims):
  """View of `arr` (named `dims`) broadcastable against an array named `out_dims`."""
  arr = np.asarray(arr)
  perm = [dims.index(d) for d in out_dims if d in dims]
  arr = np.transpose(arr, perm)
  index = tuple(slice(None) if d in dims else None for d in out_dims)
  return arr[index]


def f64(x):
  return np.asarray(x, dtype=np.float64)


# --------------------------------------------------------------------------------------------------
# weights and bin masks


def latitude_cell_bounds(lat_rad):
  """weatherbenchX/weighting.py:62-79: mid-points, ends extended half a cell and clipped to +-pi/2."""
  lat_rad = np.asarray(lat_rad)
  diff = np.diff(lat_rad)
  lower = max(lat_rad[0] - diff[0] / 2, -np.pi / 2)
  upper = min(lat_rad[-1] + diff[-1] / 2, np.pi / 2)
  return np.concatenate([[lower], (lat_rad[:-1] + lat_rad[1:]) / 2, [upper]])


def grid_area_weights(lat_deg, normalized=True):
  """weatherbenchX/weighting.py:82-88,105-130: sin(upper) - sin(lower), descending latitudes allowed."""
  lat_deg = np.asarray(lat_deg, dtype=np.float64)
  d = np.diff(lat_deg)
  assert np.all(d > 0) or np.all(d < 0), 'Points must be strictly monotonic'
  flip = lat_deg[0] > lat_deg[1]
  lat = lat_deg[::-1] if flip else lat_deg
  b = latitude_cell_bounds(np.deg2rad(lat))
  w = np.sin(b[1:]) - np.sin(b[:-1])
  if flip:
    w = w[::-1]
  if normalized:
    w = w / w.mean()
  return w


def region_masks(lat_deg, lon_deg, regions, land_sea_mask=None):
  """weatherbenchX/binning.py:52-89,166-201 -> (names, bool[region, lat, lon]); inclusive bounds,
  longitudes modulo 360 with wrap-around; optional `{name}_land` copies."""
  lat = np.asarray(lat_deg)[:, None]
  lon = np.mod(np.asarray(lon_deg), 360)[None, :]
  names, masks = [], []
  for name, ((lat0, lat1), (lon0, lon1)) in regions.items():
    if lat0 >= lat1:
      raise ValueError('lat_lims[0] must be smaller than lat_lims[1]')
    lat_m = (lat >= lat0) & (lat <= lat1)
    l0, l1 = np.mod(lon0, 360), np.mod(lon1, 360)
    lon_m = ((lon >= l0) & (lon <= l1)) if l1 > l0 else ((lon <= l1) | (lon >= l0))
    names.append(name)
    masks.append(lat_m & lon_m)
  masks = np.stack(masks)
  if land_sea_mask is not None:
    land = masks & np.asarray(land_sea_mask).astype(bool)[None]
    masks = np.concatenate([masks, land])
    names = names + [f'{n}_land' for n in names]
  return names, masks


# --------------------------------------------------------------------------------------------------
# per-point statistics (weatherbenchX/metrics/deterministic.py, probabilistic.py)


def error(p, t):
  """deterministic.py:91-100."""
  return f64(p) - f64(t)


def absolute_error(p, t):
  """deterministic.py:103-112."""
  return np.abs(f64(p) - f64(t))


def squared_error(p, t):
  """deterministic.py:115-123."""
  return (f64(p) - f64(t)) ** 2


def wind_vector_squared_error(pu, pv, tu, tv):
  """deterministic.py:174-219."""
  return (f64(pu) - f64(tu)) ** 2 + (f64(pv) - f64(tv)) ** 2


def squared_prediction_anomaly(p, c):
  """deterministic.py:222-232."""
  return (f64(p) - f64(c)) ** 2


def squared_target_anomaly(t, c):
  """deterministic.py:235-245."""
  return (f64(t) - f64(c)) ** 2


def anomaly_covariance(p, t, c):
  """deterministic.py:248-259."""
  return (f64(p) - f64(c)) * (f64(t) - f64(c))


def align_climatology(clim, clim_dims, valid_time, vt_dims):
  """metrics/base.py:382-403: climatology[dayofyear, (hour), ...] gathered at valid_time.
  `valid_time` datetime64 array named vt_dims; clim_dims starts with ('dayofyear',) or ('dayofyear','hour')
  with dayofyear coordinate 1..366 and hour coordinate 0,6,12,18 spacing 24/nhour."""
  vt = np.asarray(valid_time).astype('datetime64[ns]')
  doy = (vt.astype('datetime64[D]') - vt.astype('datetime64[Y]').astype('datetime64[D]')).astype(np.int64)
  if len(clim_dims) > 1 and clim_dims[1] == 'hour':
    nh = np.asarray(clim).shape[1]
    hour = (vt - vt.astype('datetime64[D]')).astype('timedelta64[h]').astype(np.int64)
    out = np.asarray(clim)[doy, hour // (24 // nh)]
    rest = clim_dims[2:]
  else:
    out = np.asarray(clim)[doy]
    rest = clim_dims[1:]
  return out, tuple(vt_dims) + tuple(rest)


def _move_member_last(p, p_dims, ensemble_dim):
  ax = p_dims.index(ensemble_dim)
  return np.moveaxis(f64(p), ax, -1), tuple(d for d in p_dims if d != ensemble_dim)


def _abs_error_per_member(p, p_dims, t, t_dims, ensemble_dim):
  """|p_m - t| with the member axis last, over the union of the other dims."""
  pm, dims = _move_member_last(p, p_dims, ensemble_dim)
  out_dims = union_dims(dims, t_dims)
  pe = expand_to(pm, dims + (ensemble_dim,), out_dims + (ensemble_dim,))
  return np.abs(pe - expand_to(f64(t), t_dims, out_dims)[..., None]), out_dims


def _nanmean_last(x):
  """xarray's mean(skipna=True): NaNs are left out, an all-NaN slice gives NaN (quietly)."""
  ok = ~np.isnan(x)
  n = ok.sum(axis=-1)
  with np.errstate(invalid='ignore', divide='ignore'):
    return np.where(ok, x, 0.0).sum(axis=-1) / np.where(n > 0, n, np.nan)


def _nanvar_last(x, ddof=1):
  """xarray's var(ddof=1, skipna=True): over the non-NaN members, NaN where fewer than ddof + 1 are left."""
  ok = ~np.isnan(x)
  n = ok.sum(axis=-1)
  mean = _nanmean_last(x)
  with np.errstate(invalid='ignore', divide='ignore'):
    dev2 = np.where(ok, (x - mean[..., None]) ** 2, 0.0).sum(axis=-1)
    return dev2 / np.where(n - ddof > 0, n - ddof, np.nan)


def crps_skill(p, p_dims, t, t_dims, ensemble_dim, skipna_ensemble=False):
  """probabilistic.py:116-145: mean_m |p_m - t|; skipna_ensemble -> the mean skips NaN members (:143-145).  Targets that
  carry the ensemble dim too (:133-142): the mean runs over every (prediction member, target member) pair."""
  if ensemble_dim in t_dims:
    tm, tdims = _move_member_last(t, t_dims, ensemble_dim)
    acc, out_dims = None, None
    cnt = None
    for j in range(tm.shape[-1]):  # one target member at a time: no M x M' x grid temporary
      pe, out_dims = _abs_error_per_member(p, p_dims, tm[..., j], tdims, ensemble_dim)
      if skipna_ensemble:
        ok = ~np.isnan(pe)
        acc = np.where(ok, pe, 0.0).sum(axis=-1) + (0.0 if acc is None else acc)
        cnt = ok.sum(axis=-1) + (0 if cnt is None else cnt)
      else:
        acc = pe.sum(axis=-1) + (0.0 if acc is None else acc)
    n_p = np.shape(p)[p_dims.index(ensemble_dim)]
    with np.errstate(invalid='ignore', divide='ignore'):
      return (acc / np.where(cnt > 0, cnt, np.nan) if skipna_ensemble else acc / (n_p * tm.shape[-1])), out_dims
  pm, dims = _move_member_last(p, p_dims, ensemble_dim)
  out_dims = union_dims(dims, t_dims)
  pe = expand_to(pm, dims + (ensemble_dim,), out_dims + (ensemble_dim,))
  te = expand_to(f64(t), t_dims, out_dims)[..., None]
  ae = np.abs(pe - te)
  return (_nanmean_last(ae) if skipna_ensemble else ae.mean(axis=-1)), out_dims


def rankdata_ordinal(x, axis=-1):
  """probabilistic.py:148-158: ordinal ranks 1..M via argsort + put_along_axis."""
  x = np.swapaxes(np.asarray(x), axis, -1)
  order = np.argsort(x, axis=-1)
  ranks = np.empty(order.shape, dtype=np.int64)
  np.put_along_axis(ranks, order, np.broadcast_to(np.arange(1, x.shape[-1] + 1), x.shape), axis=-1)
  return np.swapaxes(ranks, axis, -1)


def crps_spread(p, p_dims, ensemble_dim, fair=True, use_sort=False, skipna_ensemble=False):
  """probabilistic.py:165-247: rank form (:231-240) or pairwise form (:241-247).  skipna_ensemble (pairwise form only,
  :215-216): pairs with a NaN member add nothing to the double sum, and M becomes the per-point count of non-NaN
  members (:206-207, :243-247)."""
  pm, dims = _move_member_last(p, p_dims, ensemble_dim)
  m = pm.shape[-1]
  if skipna_ensemble:
    if use_sort:
      raise ValueError('skipna_ensemble is not supported with use_sort=True.')
    n = (~np.isnan(pm)).sum(axis=-1).astype(np.float64)
    total = np.zeros(pm.shape[:-1])
    for i in range(m):
      total += np.nansum(np.abs(pm - pm[..., i:i + 1]), axis=-1)
    with np.errstate(invalid='ignore', divide='ignore'):
      return total / (n * (n - int(fair))), dims
  if m < 2:
    raise ValueError('Cannot estimate CRPS spread with n_ensemble < 2.')
  if use_sort:
    rank = rankdata_ordinal(pm, axis=-1)
    return 2 * ((2 * rank - m - 1) * pm).mean(axis=-1) / (m - int(fair)), dims
  total = np.zeros(pm.shape[:-1])
  for i in range(m):  # O(M^2) without the M x M x grid temporary
    total += np.abs(pm - pm[..., i:i + 1]).sum(axis=-1)
  return total / (m * (m - int(fair))), dims


def ensemble_variance(p, p_dims, ensemble_dim, skipna_ensemble=False):
  """probabilistic.py:250-273: var(ddof=1), over the non-NaN members with skipna_ensemble."""
  pm, dims = _move_member_last(p, p_dims, ensemble_dim)
  return (_nanvar_last(pm) if skipna_ensemble else pm.var(axis=-1, ddof=1)), dims


def _ensemble_moments(x, x_dims, ensemble_dim, skipna_ensemble):
  """(mean, var(ddof=1) / n, dims without the member dim) -- probabilistic.py:304-314 for the predictions, :320-330 for
  ensemble-valued targets: with skipna_ensemble mean, variance and n run over the non-NaN members of each point."""
  xm, dims = _move_member_last(x, x_dims, ensemble_dim)
  if skipna_ensemble:
    n = (~np.isnan(xm)).sum(axis=-1).astype(np.float64)
    mean, var = _nanmean_last(xm), _nanvar_last(xm)
  else:
    n = xm.shape[-1]
    with np.errstate(invalid='ignore', divide='ignore'):
      mean, var = xm.mean(axis=-1), (xm.var(axis=-1, ddof=1) if xm.shape[-1] > 1 else np.full(xm.shape[:-1], np.nan))
  with np.errstate(invalid='ignore', divide='ignore'):
    return mean, var / n, dims


def unbiased_ensemble_mean_squared_error(p, p_dims, t, t_dims, ensemble_dim, skipna_ensemble=False):
  """probabilistic.py:276-336: (mean_m p - t)^2 - var_p / M; skipna_ensemble -> mean, variance and M over the non-NaN
  members of each point (:304-314).  Targets that carry the ensemble dim too (:320-330): t becomes the mean of the target
  members and their own bias var_t / N is subtracted as well, (mean p - mean t)^2 - var_p / M - var_t / N (:334-336)."""
  p_mean, p_bias, dims = _ensemble_moments(p, p_dims, ensemble_dim, skipna_ensemble)
  if ensemble_dim in t_dims:
    t_mean, t_bias, tdims = _ensemble_moments(t, t_dims, ensemble_dim, skipna_ensemble)
  else:
    t_mean, t_bias, tdims = f64(t), None, tuple(t_dims)
  out_dims = union_dims(dims, tdims)
  with np.errstate(invalid='ignore', divide='ignore'):
    out = (expand_to(p_mean, dims, out_dims) - expand_to(t_mean, tdims, out_dims)) ** 2 - expand_to(p_bias, dims, out_dims)
    if t_bias is not None:
      out = out - expand_to(t_bias, tdims, out_dims)
  return out, out_dims


def ensemble_mean_squared_error(p, p_dims, t, t_dims, ensemble_dim):
  """wrappers.py:116-148 (EnsembleMean) followed by deterministic.py:115-123."""
  pm, dims = _move_member_last(p, p_dims, ensemble_dim)
  out_dims = union_dims(dims, t_dims)
  return (expand_to(pm.mean(axis=-1), dims, out_dims) - expand_to(f64(t), t_dims, out_dims)) ** 2, out_dims


# --------------------------------------------------------------------------------------------------
# aggregation (weatherbenchX/aggregation.py:297-366)

# ---- indicator statistics (a new trailing dimension of categories) ------------------------------------------
def error_exceedance(p, p_dims, t, t_dims, thresholds, threshold_dim='error_exceedance_thresholds'):
  """deterministic.py:262-295: float(|p - t| > thr_k), NaN where |p - t| or thr_k is NaN."""
  out_dims = union_dims(p_dims, t_dims)
  ae = np.abs(expand_to(f64(p), p_dims, out_dims) - expand_to(f64(t), t_dims, out_dims))[..., None]
  thr = f64(thresholds).reshape((1,) * len(out_dims) + (-1,))
  with np.errstate(invalid='ignore'):
    out = (ae > thr).astype(np.float64)
  out = np.where(np.isnan(ae), np.nan, out)
  out = np.where(np.isnan(thr), np.nan, out)
  return out, out_dims + (threshold_dim,)


def ensemble_error_exceedance(p, p_dims, t, t_dims, thresholds, ensemble_dim,
                              threshold_dim='error_exceedance_thresholds'):
  """probabilistic.py:836-861: the member mean (xarray default skipna=True) of the exceedance indicators."""
  full, dims = error_exceedance(p, p_dims, t, t_dims, thresholds, threshold_dim)
  ax = dims.index(ensemble_dim)
  with np.errstate(invalid='ignore'), __import__('warnings').catch_warnings():
    __import__('warnings').simplefilter('ignore', RuntimeWarning)
    out = np.nanmean(full, axis=ax)
  return out, tuple(d for d in dims if d != ensemble_dim)


def rank_histogram(p, p_dims, t, t_dims, ensemble_dim, rank_dim='rank'):
  """probabilistic.py:1306-1343: one-hot of #{m : p_m < t} over M + 1 ranks (comparisons with NaN are False)."""
  pm, dims = _move_member_last(p, p_dims, ensemble_dim)
  out_dims = union_dims(dims, t_dims)
  pe = expand_to(pm, dims + (ensemble_dim,), out_dims + (ensemble_dim,))
  te = expand_to(f64(t), t_dims, out_dims)[..., None]
  with np.errstate(invalid='ignore'):
    ranks = (pe < te).astype(np.int64).sum(axis=-1)
  m = pm.shape[-1]
  return (ranks[..., None] == np.arange(m + 1)).astype(np.float64), out_dims + (rank_dim,)



def aggregate(stat, dims, reduce_dims, weights=(), bin_masks=(), mask=None, mask_dims=None, skipna=False):
  """Aggregator.aggregate_stat_var: returns (sum_weighted_statistics, sum_weights, out_dims).

  weights:   sequence of (array, dims)              -- product of all (aggregation.py:311-314)
  bin_masks: sequence of (bin_dim_name, bool array, dims incl. the bin dim) (aggregation.py:320-330)
  mask:      optional boolean validity array named mask_dims (`masked=True` with a mask coordinate,
             aggregation.py:339-352); skipna drops NaN statistic values (aggregation.py:353-355).
  Output dims: surviving statistic dims in order, then the bin dims in order.  A variable lacking a reduce dim
  returns None (aggregation.py:305-309)."""
  dims = tuple(dims)
  if not set(reduce_dims) <= set(dims):
    return None
  stat = f64(stat)
  valid = np.ones(stat.shape, dtype=bool)
  if mask is not None:
    valid = valid & np.broadcast_to(expand_to(np.asarray(mask, dtype=bool), tuple(mask_dims), dims), stat.shape)
  if skipna:
    valid = valid & ~np.isnan(stat)
  if mask is not None or skipna:
    stat = np.where(valid, stat, 0.0)
  ones = valid.astype(np.float64)
  letters = {}

  def L(d):
    if d not in letters:
      letters[d] = chr(ord('a') + len(letters))
    return letters[d]

  operands, specs = [], []
  for arr, wd in weights:
    operands.append(f64(arr))
    specs.append(''.join(L(d) for d in wd))
  bin_names = []
  for name, m, md in bin_masks:
    if not (set(md) - {name}) <= set(dims):
      return None
    operands.append(np.asarray(m, dtype=np.float64))
    specs.append(''.join(L(d) for d in md))
    bin_names.append(name)
  out_dims = tuple(d for d in dims if d not in set(reduce_dims)) + tuple(bin_names)
  sspec = ''.join(L(d) for d in dims)
  ospec = ''.join(L(d) for d in out_dims)
  expr = ','.join([sspec] + specs) + '->' + ospec
  with np.errstate(all='ignore'):
    sws = np.einsum(expr, stat, *operands)
    sw = np.einsum(expr, ones, *operands)
  return sws, sw, out_dims


# --------------------------------------------------------------------------------------------------
# metrics from mean statistics (deterministic.py:312-425, probabilistic.py:606-688, 864-1003)


def rmse(mean_se):
  return np.sqrt(mean_se)


def acc(mean_cov, mean_spa, mean_sta):
  return mean_cov / (np.sqrt(mean_spa) * np.sqrt(mean_sta))


def crps(mean_skill, mean_spread):
  return mean_skill - 0.5 * mean_spread


def crps_ensemble_distance(mean_skill, mean_spread, mean_target_spread):
  """probabilistic.py:773-782: E|X - Y| - E|X - X'| / 2 - E|Y - Y'| / 2 from the three mean statistics."""
  return mean_skill - 0.5 * mean_spread - 0.5 * mean_target_spread


def unbiased_spread_skill_ratio(mean_var, mean_uemse):
  return np.sqrt(mean_var / mean_uemse)


# --------------------------------------------------------------------------------------------------
# zonal energy spectrum -- NOT in the reference snapshot: parity unpinned.  Definition of
# the build (WeatherBench-2 lineage): F_k = rfft(f)/n, S_k = |F_k|^2 * (1 if k == 0 else 2).


def relative_intensity(p, t, spatial_axes, mask=None, epsilon=1e-6):
  """RelativeIntensity (metrics/deterministic.py:30-88): |(mean p + eps) / (mean t + eps) - 1| over `spatial_axes`, the means
  with skipna=False.  With `mask` (True = valid, same shape as p): means over the valid points only -- masked-out values count as
  0, sums with skipna=False, divided by the count; a slice without a valid point has both means 0 (result 0).  -> (values, mask
  of the result = count > 0, or None)."""
  p, t = np.asarray(p, np.float64), np.asarray(t, np.float64)
  axes = tuple(spatial_axes)
  with np.errstate(all='ignore'):
    if mask is None:
      pm, tm, out_mask = p.mean(axis=axes), t.mean(axis=axes), None
    else:
      m = np.asarray(mask, bool)
      count = m.sum(axis=axes)
      ps, ts = np.where(m, p, 0.0).sum(axis=axes), np.where(m, t, 0.0).sum(axis=axes)
      pm = np.where(count > 0, ps / np.where(count > 0, count, 1), 0.0)
      tm = np.where(count > 0, ts / np.where(count > 0, count, 1), 0.0)
      out_mask = (count > 0).astype(int)
    return np.abs((pm + epsilon) / (tm + epsilon) - 1), out_mask


# ---- multivariate / distribution scores: the tables of the reference written out (probabilistic.py:339-603, 785-833) ---------
def energy_score_skill(p, t, norm_axes, ensemble_axis):
  """mean_m sqrt(sum_norm (p_m - t)**2); `t` broadcastable to `p` (size 1 along the ensemble axis).  probabilistic.py:480-503."""
  norm_axes = tuple(a % p.ndim for a in np.atleast_1d(norm_axes))
  ensemble_axis %= p.ndim
  norms = np.sqrt(np.sum((p - t) ** 2, axis=norm_axes, keepdims=True))
  return np.squeeze(norms.mean(axis=ensemble_axis, keepdims=True), axis=norm_axes + (ensemble_axis,))


def energy_score_spread(p, norm_axes, ensemble_axis, fair=True):
  """sum over the full M x M table of ||p_m - p_m'|| / (M (M - 1)) or / M**2.  probabilistic.py:506-551."""
  norm_axes = tuple(a % p.ndim for a in np.atleast_1d(norm_axes))
  ensemble_axis %= p.ndim
  m = p.shape[ensemble_axis]
  q = np.moveaxis(p, ensemble_axis, 0)                               # [M, ...]
  moved = tuple(sorted((a + 1 if a < ensemble_axis else a) for a in norm_axes))  # norm axes in q's numbering
  table = q[:, None] - q[None, :]                                    # [M, M', ...]
  norms = np.sqrt(np.sum(table ** 2, axis=tuple(a + 1 for a in moved)))
  with np.errstate(all='ignore'):
    return norms.sum(axis=(0, 1)) / (m * (m - 1) if fair else m * m)


def variogram_score(p, t, pair_axis, ensemble_axis, power=0.5):
  """sum_{i,j} (|t_i - t_j|**power - mean_m |p_i^m - p_j^m|**power)**2 with the full [N, N] tables; `t` has p's axes with size 1
  along the ensemble axis.  probabilistic.py:554-603."""
  pair_axis %= p.ndim
  ensemble_axis %= p.ndim
  q = np.moveaxis(p, (pair_axis, ensemble_axis), (0, 1))             # [N, M, ...]
  u = np.moveaxis(t, (pair_axis, ensemble_axis), (0, 1))[:, 0]       # [N, ...]
  tx = (np.abs(q[:, None] - q[None, :]) ** power).mean(axis=2)       # [N, N', ...]
  ty = np.abs(u[:, None] - u[None, :]) ** power
  return ((ty - tx) ** 2).sum(axis=(0, 1))


def wasserstein_1d(u, v):
  """1-Wasserstein distance of two samples: integral of |F_u - F_v| over the pooled support (what
  scipy.stats.wasserstein_distance evaluates; probabilistic.py:823-832 calls it point by point)."""
  u, v = np.sort(np.asarray(u, dtype=np.float64)), np.sort(np.asarray(v, dtype=np.float64))
  pooled = np.sort(np.concatenate([u, v]))
  total = 0.0
  for a, b in zip(pooled[:-1], pooled[1:]):
    total += abs(np.searchsorted(u, a, side='right') / u.size - np.searchsorted(v, a, side='right') / v.size) * (b - a)
  return total


def ensemble_rps(p_members, t_value, thresholds, fair=True, right_inclusive=True):
  """Ranked probability score of ONE point: p_members [M], scalar target, thresholds [K] (probabilistic.py:339-477):
  sum_k of (mean_m 1[p_m <= b_k] - 1[t <= b_k])**2, minus var_m(1[p_m <= b_k]) / M per threshold when fair."""
  p_members = np.asarray(p_members, dtype=np.float64)
  cmp = np.less_equal if right_inclusive else np.less
  total = 0.0
  for b in thresholds:
    cp, ct = cmp(p_members, b).astype(np.float64), float(cmp(t_value, b))
    total += (cp.mean() - ct) ** 2 - (cp.var(ddof=1) / cp.size if fair else 0.0)
  return total


def contingency_table(decision, event, weights=None):
  """Mean TP, FP, FN, TN of boolean decisions against boolean events over all points (categorical.py:25-101 under a
  weighted mean)."""
  decision, event = np.asarray(decision, dtype=bool).ravel(), np.asarray(event, dtype=bool).ravel()
  w = np.ones(decision.size) if weights is None else np.asarray(weights, dtype=np.float64).ravel()
  mean = lambda x: float((x * w).sum() / w.sum())
  return mean(decision & event), mean(decision & ~event), mean(~decision & event), mean(~decision & ~event)


def relative_economic_value(probability, event, thresholds, cost_loss_ratios):
  """REV [threshold (with 0 and 1 added at the ends), cost_loss_ratio] of probability forecasts of a binary event, one decision
  rule `probability > threshold` per row (probabilistic.py:1062-1303): expenses per unit loss
  forecast = r (TP + FP) + FN, perfect = r (TP + FN), climate = min(r, TP + FN); REV = (climate - forecast) / (climate - perfect)."""
  probability, event = np.asarray(probability, dtype=np.float64).ravel(), np.asarray(event, dtype=bool).ravel()
  rows = [np.ones(event.size, dtype=bool)] + [probability > b for b in thresholds] + [np.zeros(event.size, dtype=bool)]
  out = np.empty((len(rows), len(cost_loss_ratios)))
  for i, decision in enumerate(rows):
    tp, fp, fn, _ = contingency_table(decision, event)
    for j, r in enumerate(cost_loss_ratios):
      climate = min(r, tp + fn)
      with np.errstate(all='ignore'):
        out[i, j] = np.float64(climate - (r * (tp + fp) + fn)) / np.float64(climate - r * (tp + fn))
  return out


def neighborhood_mean(field, n, wrap_longitude=False):
  """n x n neighbourhood mean of ONE [lat, lon] field, the window written out point by point: cyclic in both directions, then
  the (n - 1) / 2 outermost rows (and columns unless wrap_longitude) set to 0; a window holding a NaN is NaN.  spatial.py:24-56."""
  field = np.asarray(field, dtype=np.float64)
  if n == 1:
    return field
  h = (n - 1) // 2
  nlat, nlon = field.shape
  out = np.zeros((nlat, nlon))
  for i in range(nlat):
    for j in range(nlon):
      out[i, j] = np.mean([field[(i + a) % nlat, (j + b) % nlon] for a in range(-h, h + 1) for b in range(-h, h + 1)])
  out[:h] = 0
  out[nlat - h:] = 0
  if not wrap_longitude:
    out[:, :h] = 0
    out[:, nlon - h:] = 0
  return out


def fractions_skill_score(p, t, n, wrap_longitude=False):
  """FSS of binary [..., lat, lon] arrays with plain means over everything (spatial.py:280-339)."""
  p, t = np.asarray(p, dtype=np.float64), np.asarray(t, dtype=np.float64)
  pf = np.stack([neighborhood_mean(f, n, wrap_longitude) for f in p.reshape((-1,) + p.shape[-2:])])
  tf = np.stack([neighborhood_mean(f, n, wrap_longitude) for f in t.reshape((-1,) + t.shape[-2:])])
  return 1 - np.mean((pf - tf) ** 2) / (np.mean(pf ** 2) + np.mean(tf ** 2))


def zonal_power_spectrum(field, lon_axis=-1):
  f = f64(field)
  n = f.shape[lon_axis]
  F = np.fft.rfft(f, axis=lon_axis) / n
  power = (F.real ** 2 + F.imag ** 2)
  factor = np.full(power.shape[lon_axis], 2.0)
  factor[0] = 1.0
  shape = [1] * power.ndim
  shape[lon_axis] = -1
  return power * factor.reshape(shape)


# --------------------------------------------------------------------------------------------------
# "reference structure" CPU path for timing: one pass per statistic with full-size temporaries in the
# input dtype, np.einsum for the weighted reduction and a second einsum on ones_like for sum_weights,
# exactly the shape of work of aggregation.py:337-366 + deterministic.py:91-259.


def reference_structure_deterministic(p, t, c, lat_weights, reduce_axes=(0, 3, 4)):
  """p, t, c: float32 [init, lead, level, lat, lon]; lat_weights float64[lat].  Returns the six mean statistics
  (Error, AbsoluteError, SquaredError, SquaredPredictionAnomaly, SquaredTargetAnomaly, AnomalyCovariance)."""
  del reduce_axes
  out = {}
  stats = {
      'Error': lambda: p - t,
      'AbsoluteError': lambda: abs(p - t),
      'SquaredError': lambda: (p - t) ** 2,
      'SquaredPredictionAnomaly': lambda: (p - c) ** 2,
      'SquaredTargetAnomaly': lambda: (t - c) ** 2,
      'AnomalyCovariance': lambda: (p - c) * (t - c),
  }
  for name, fn in stats.items():
    stat = fn()
    sws = np.einsum('abcde,d->bc', stat, lat_weights)
    sw = np.einsum('abcde,d->bc', np.ones_like(stat), lat_weights)
    out[name] = sws / sw
  return out


def reference_structure_ensemble(p, t, lat_weights):
  """p float32 [member, lat, lon], t float32 [lat, lon]: CRPS (rank form, fair), variance, unbiased MSE,
  ensemble-mean SE -- the public-benchmark ensemble suite (public_benchmark/run_benchmark_evaluation.py:341-353)."""
  m = p.shape[0]
  out = {}
  skill = np.abs(p - t[None]).mean(axis=0)
  rank = rankdata_ordinal(p, axis=0)
  spread = 2 * ((2 * rank - m - 1) * p).mean(axis=0) / (m - 1)
  var = p.var(axis=0, ddof=1)
  mean = p.mean(axis=0)
  stats = {'CRPSSkill': skill, 'CRPSSpread': spread, 'EnsembleVariance': var,
           'UnbiasedEnsembleMeanSquaredError': (mean - t) ** 2 - var / m, 'EnsembleMeanSquaredError': (mean - t) ** 2}
  for name, stat in stats.items():
    sws = np.einsum('de,d->', stat, lat_weights)
    sw = np.einsum('de,d->', np.ones_like(stat), lat_weights)
    out[name] = sws / sw
  return out
