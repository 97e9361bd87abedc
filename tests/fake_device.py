"""NumPy interpreter of the stage-1 / stage-2 plans (TEST INFRASTRUCTURE, CPU only).

Mirrors what csrc/wbx_s1.hpp, wbx_det.hip, wbx_ens_impl.hpp and wbx_s2.hip do with the tables the planner
emits, so that planner + Aggregator + labeled-array logic can be verified without a GPU.  It is installed by
monkeypatching the launch functions of weatherbenchx_amd.engine inside the `backend` fixture only.
"""
import numpy as np

from weatherbenchx_amd import _hip
from weatherbenchx_amd import engine
from weatherbenchx_amd import planner


class _Buf:
  def __init__(self, arr):
    self.ptr = arr


class _Fence:
  def wait(self):
    pass


class FakeCtx:
  device_id = 0

  def upload(self, arr):
    return _Buf(np.array(arr, copy=True))

  def alloc(self, nbytes):
    return _Buf(np.zeros(max(int(nbytes) // 8, 1)))

  def download(self, ptr, shape, dtype=np.float64):
    n = int(np.prod(shape, dtype=np.int64))
    return np.array(np.asarray(ptr).reshape(-1)[:n], dtype=dtype).reshape(shape)

  def download_async(self, ptr, shape):
    return self.download(ptr, shape)

  def fence(self):
    return _Fence()

  def synchronize(self):
    pass


def _acc_add(ctx, dst_buf, dst_off, src_ptr, n, overwrite):
  src = np.asarray(src_ptr, dtype=np.float64).reshape(-1)[:n]
  with np.errstate(all='ignore'):
    dst_buf.ptr[dst_off:dst_off + n] = src if overwrite else dst_buf.ptr[dst_off:dst_off + n] + src


def _to_device(ctx, da, dtype_code):
  cache = da.__dict__.setdefault('_wbx_dev_fake', {})
  if dtype_code not in cache:
    want = np.float32 if dtype_code == _hip.F32 else np.float64
    host = np.ascontiguousarray(np.asarray(da.values), dtype=want)
    st = [int(s // host.itemsize) for s in host.strides] if host.ndim else []
    lay = planner.InputLayout(strides=dict(zip(da.dims, st)), itemsize=host.itemsize, base_alignment=256)
    cache[dtype_code] = engine._Dev(host.reshape(-1), lay, dtype_code, host, host.nbytes)
  return cache[dtype_code]


def _mask_to_device(ctx, mask):
  host = np.ascontiguousarray(np.asarray(mask.values).astype(bool).astype(np.uint8))
  st = [int(s) for s in host.strides] if host.ndim else []
  return engine._Dev(host.reshape(-1), planner.InputLayout(strides=dict(zip(mask.dims, st)), itemsize=1,
                                                           base_alignment=256), 'u8', host, host.nbytes)


def _offsets(plan, i):
  ko = plan.key_off[i] if plan.key_off[i] is not None else np.zeros(plan.nkey, dtype=np.int64)
  do = plan.depth_off[i] if plan.depth_off[i] is not None else np.zeros(plan.ndepth, dtype=np.int64)
  off = ko[:, None, None] + do[None, :, None] + (np.arange(plan.nx, dtype=np.int64) * plan.xstride[i])[None, None, :]
  if i == 2 and plan.gather_tab is not None:
    gk = plan.gather_key if plan.gather_key is not None else np.zeros(plan.nkey, dtype=np.int32)
    gd = plan.gather_depth if plan.gather_depth is not None else np.zeros(plan.ndepth, dtype=np.int32)
    off = off + plan.gather_tab[gk.astype(np.int64)[:, None] * plan.n_gather_depth + gd[None, :]][:, :, None]
  return off


def _det_lanes(func, v):
  p = v[0].astype(np.float64)
  if func == _hip.PASS1:
    return [p]
  t = v[1].astype(np.float64)
  e = p - t
  lanes = [e, np.abs(e), e * e]
  if func == _hip.DET6:
    c = v[2].astype(np.float64)
    lanes += [(p - c) ** 2, (t - c) ** 2, (p - c) * (t - c)]
  return lanes


def _ens_lanes(plan, devs, ens, flags):
  m, mstride, algo = ens
  off_p = _offsets(plan, 0)
  members = np.stack([devs[0].ptr[off_p + k * mstride] for k in range(m)], axis=-1).astype(np.float64)
  t = devs[1].ptr[_offsets(plan, 1)].astype(np.float64)
  fair = 1.0 if flags & _hip.FLAG_FAIR else 0.0
  if flags & _hip.FLAG_SKIPNA_ENS:
    with np.errstate(all='ignore'):
      n = (~np.isnan(members)).sum(axis=-1).astype(np.float64)
      d = members - t[..., None]
      skill = np.nansum(np.abs(d), axis=-1) / n
      spread = np.nansum(np.abs(members[..., :, None] - members[..., None, :]), axis=(-1, -2)) / (n * (n - fair))
      mean = np.nansum(members, axis=-1) / n
      var = np.where(n > 0, np.nansum((members - mean[..., None]) ** 2, axis=-1) / (n - 1), np.nan)  # no member: NaN
      md = mean - t
    return [skill, spread, var, md * md - var / n, md * md]
  d = members - t[..., None]
  skill = np.abs(d).mean(axis=-1)
  if algo == _hip.ENS_SORT:
    srt = np.sort(members, axis=-1)
    coef = 2 * np.arange(1, m + 1) - m - 1
    spread = 2 * (srt * coef).sum(axis=-1) / (m * (m - fair))
    spread = np.where(np.isnan(members).any(axis=-1), np.nan, spread)
  else:
    spread = np.abs(members[..., :, None] - members[..., None, :]).sum(axis=(-1, -2)) / (m * (m - fair))
  with np.errstate(all='ignore'):
    var = members.var(axis=-1, ddof=1) if m > 1 else np.full(skill.shape, np.nan)
  md = d.mean(axis=-1)
  return [skill, spread, var, md * md - var / m, md * md]


def _chunked(plan, arr):
  """[nkey, D, nx] -> [nkey, nchunk, nj]"""
  if getattr(plan, 'x_weights', None) is not None:  # weights on x folded into stage 1 (value and count lanes alike)
    arr = arr * np.asarray(plan.x_weights)[None, None, :]
  out = np.zeros((plan.nkey, plan.nchunk, plan.nj))
  for c in range(plan.nchunk):
    d0, d1 = c * plan.depth_chunk, min((c + 1) * plan.depth_chunk, plan.ndepth)
    blk = arr[:, d0:d1, :]
    out[:, c, :] = blk.sum(axis=1) if plan.x_kept else blk.sum(axis=(1, 2))[:, None]
  return out


def _ens2_lanes(plan, devs, ens, flags):
  m, mstride, n_t, tstride = ens
  off_p, off_t = _offsets(plan, 0), _offsets(plan, 1)
  pm = np.stack([devs[0].ptr[off_p + k * mstride] for k in range(m)], axis=-1).astype(np.float64)
  tm = np.stack([devs[1].ptr[off_t + k * tstride] for k in range(n_t)], axis=-1).astype(np.float64)
  with np.errstate(all='ignore'):
    if flags & _hip.FLAG_SKIPNA_ENS:
      d = np.abs(pm[..., :, None] - tm[..., None, :])
      npair = (~np.isnan(d)).sum(axis=(-1, -2)).astype(np.float64)
      skill = np.where(npair > 0, np.nansum(d, axis=(-1, -2)) / npair, np.nan)
      cnt = lambda x: (~np.isnan(x)).sum(axis=-1).astype(np.float64)
      mean = lambda x: np.where(cnt(x) > 0, np.nansum(x, axis=-1) / cnt(x), np.nan)
      var = lambda x: np.where(cnt(x) > 1, np.nansum((x - mean(x)[..., None]) ** 2, axis=-1) / (cnt(x) - 1), np.nan)
      uemse = (mean(pm) - mean(tm)) ** 2 - var(pm) / cnt(pm) - var(tm) / cnt(tm)
    else:
      skill = np.abs(pm[..., :, None] - tm[..., None, :]).mean(axis=(-1, -2))
      pv = pm.var(axis=-1, ddof=1) if m > 1 else np.full(skill.shape, np.nan)
      tv = tm.var(axis=-1, ddof=1) if n_t > 1 else np.full(skill.shape, np.nan)
      uemse = (pm.mean(axis=-1) - tm.mean(axis=-1)) ** 2 - pv / m - tv / n_t
  return [skill, uemse]


def _cat_lanes(plan, devs, cat):
  cfunc, ncat, m, mstride, thr, cstride = cat
  off_p = _offsets(plan, 0)
  members = np.stack([devs[0].ptr[off_p + k * mstride] for k in range(m)], axis=-1).astype(np.float64)
  t = devs[1].ptr[_offsets(plan, 1)].astype(np.float64)[..., None]
  if cfunc == _hip.CAT_RANK:
    r = (members < t).sum(axis=-1)
    return [(r == k).astype(np.float64) for k in range(ncat)]
  ae = np.abs(members - t)
  n = (~np.isnan(ae)).sum(axis=-1).astype(np.float64)
  if cstride is not None:  # a threshold field: threshold k of every point through input 2's offsets
    off = _offsets(plan, 2)
    fields = [np.asarray(devs[2].ptr, dtype=np.float64)[off + k * cstride] for k in range(ncat)]
    return [np.where(np.isnan(f), np.nan, (ae > f[..., None]).sum(axis=-1) / np.where(n > 0, n, np.nan)) for f in fields]
  thresholds = np.asarray(thr.ptr, dtype=np.float64)
  return [np.where(np.isnan(thresholds[k]), np.nan, (ae > thresholds[k]).sum(axis=-1) / np.where(n > 0, n, np.nan))
          for k in range(ncat)]  # NaN threshold: NaN indicator (deterministic.py:293-294)


def _run_s1(ctx, kind, dplan, plan, devs, dtype_code, nlanes_total, func=0, ens=None, cat=None, inputs=None, fold=None):  # pylint: disable=unused-argument
  if engine.S1_EVENT_LOG is not None:
    engine.S1_EVENT_LOG.append({'kind': kind, 'flags': int(plan.flags), 'ms': 0.0, 'x_kept': plan.x_kept, 'block': plan.block_threads})
  with np.errstate(all='ignore'):
    if kind == 'det':
      nin = {_hip.DET3: 2, _hip.DET6: 3, _hip.PASS1: 1}[func]
      vals = [devs[i].ptr[_offsets(plan, i)] for i in range(nin)]
      lanes = _det_lanes(func, vals)
    elif kind == 'cat':
      lanes = _cat_lanes(plan, devs, cat)
    elif kind == 'ens2':
      lanes = _ens2_lanes(plan, devs, ens, plan.flags)
    else:
      lanes = _ens_lanes(plan, devs, ens, plan.flags)
    counted = bool(plan.flags & 3)
    cols = []
    if counted:
      valid = np.ones(lanes[0].shape, dtype=bool)
      if plan.flags & _hip.FLAG_MASKED:
        valid = devs[3].ptr[_offsets(plan, 3)] != 0
      if plan.flags & _hip.FLAG_SKIPNA:
        oks = [valid & ~np.isnan(l) for l in lanes]
        cols = [_chunked(plan, np.where(ok, l, 0.0)) for ok, l in zip(oks, lanes)]
        cols += [_chunked(plan, ok.astype(np.float64)) for ok in oks]
      else:  # mask only: one shared count lane
        cols = [_chunked(plan, np.where(valid, l, 0.0)) for l in lanes]
        cols.append(_chunked(plan, np.broadcast_to(valid, lanes[0].shape).astype(np.float64)))
    else:
      cols = [_chunked(plan, l) for l in lanes]
    partial = np.stack(cols, axis=2)  # [nkey, nchunk, lane, nj]
  assert partial.shape[2] == nlanes_total
  return _Buf(partial.reshape(plan.partial_shape(nlanes_total)))


def _run_binned(ctx, dplan, plan, devs, dtype_code, nl_total, func, w_buf):
  """wbx_det_binned: weights and membership applied per point (no unweighted partials)."""
  nA, nBk, nBr = plan.n(plan.a_dims), plan.n(plan.bk_dims), plan.n(plan.br_dims)
  nbin = w_buf.shape[-1]
  nj = plan.nj if plan.x_kept else 1
  with np.errstate(all='ignore'):
    nin = {_hip.DET3: 2, _hip.DET6: 3, _hip.PASS1: 1}[func]
    lanes = _det_lanes(func, [devs[i].ptr[_offsets(plan, i)] for i in range(nin)])
    if plan.flags & 3:
      valid = np.ones(lanes[0].shape, dtype=bool)
      if plan.flags & _hip.FLAG_MASKED:
        valid = devs[3].ptr[_offsets(plan, 3)] != 0
      if plan.flags & _hip.FLAG_SKIPNA:
        oks = [valid & ~np.isnan(l) for l in lanes]
        lanes = [np.where(ok, l, 0.0) for ok, l in zip(oks, lanes)] + [ok.astype(np.float64) for ok in oks]
      else:
        lanes = [np.where(valid, l, 0.0) for l in lanes] + [np.broadcast_to(valid, lanes[0].shape).astype(np.float64)]
    assert len(lanes) == nl_total
    wt = np.asarray(w_buf.bufs[0].ptr).reshape(nBk, nBr, nj)
    if w_buf.factored is not None and engine.SEPARABLE_BINNED_WEIGHTS:  # the factored form the kernel would be handed
      flag, buf = w_buf.factored
      fac = np.asarray(buf.ptr)
      wt = np.broadcast_to(fac.reshape(nBk, 1, nj) if flag == _hip.BINNED_WT_X_ONLY else fac.reshape(nBk, nBr, 1),
                           (nBk, nBr, nj))
    bits = np.asarray(w_buf.bufs[1].ptr).reshape(nBk, nBr, nj)
    member = ((bits[..., None] >> np.arange(nbin, dtype=np.uint64)) & np.uint64(1)).astype(np.float64)
    out = np.empty((nA, nBk, nl_total, 1, nbin))
    for l, v in enumerate(lanes):
      v = v.reshape(nA, nBk, nBr, plan.ndepth, plan.nx) * wt[None, :, :, None, :]
      out[:, :, l, 0, :] = np.einsum('abrdx,brxn->abn', v, np.broadcast_to(member, (nBk, nBr, nj, nbin)) if nj > 1
                                     else np.broadcast_to(member, (nBk, nBr, 1, nbin)).repeat(plan.nx, axis=2))
  return out, out.shape


def _ens_binned_atoms(ctx, dplan, plan, w_buf, w_flags):
  """The interpreter needs no atom tables; a patch never overflows here."""
  return object()


def _run_ens_binned(ctx, dplan, plan, devs, dtype_code, ens_args, w_buf, route):
  """wbx_ens_binned: the five ensemble lanes + the count lane, weights (factored form) and membership applied per point."""
  w_flags, _ = route
  nA, nBk, nBr = plan.n(plan.a_dims), plan.n(plan.bk_dims), plan.n(plan.br_dims)
  nbin = w_buf.shape[-1]
  nj = plan.nj if plan.x_kept else 1
  if engine.S1_EVENT_LOG is not None:
    engine.S1_EVENT_LOG.append({'kind': 'ens_binned', 'nbin': nbin, 'w_flags': w_flags, 'flags': int(plan.flags), 'ms': 0.0})
  with np.errstate(all='ignore'):
    lanes = _ens_lanes(plan, devs, ens_args, plan.flags)
    valid = np.ones(lanes[0].shape, dtype=bool)
    if plan.flags & _hip.FLAG_MASKED:  # any strides (ABI 11): WBX_BINNED_MASK_ON_W only says that A / depth strides are zero
      valid = devs[3].ptr[_offsets(plan, 3)] != 0
      if w_flags & _hip.BINNED_MASK_ON_W:
        assert all(devs[3].layout.stride(d) == 0 for d in tuple(plan.a_dims) + tuple(plan.depth_dims))
    twin_out = bool(w_flags & _hip.BINNED_TWIN_MASK)
    if plan.flags & _hip.FLAG_SKIPNA:
      # [five values | their five counts] (wbx.h): a statistic's NaN points are left out of its sum and its count.  In twin mode
      # the kernel does not form the masked spread / variance nor the unmasked target statistics: NaN, as documented.
      nan = np.full(lanes[0].shape, np.nan)
      def ten(ok_points, dead):
        ok = [ok_points & ~np.isnan(l) for l in lanes]
        vals = [np.where(o, l, 0.0) for o, l in zip(ok, lanes)]
        cnts = [o.astype(np.float64) for o in ok]
        if dead:
          src = 0 if 1 in dead else 1  # (the count lanes of the dead statistics repeat a live one)
          vals = [nan if i in dead else v for i, v in enumerate(vals)]
          cnts = [cnts[src] if i in dead else c for i, c in enumerate(cnts)]
        return vals + cnts
      lanes = ten(valid, (1, 2) if twin_out else ()) + (ten(np.ones_like(valid), (0, 3, 4)) if twin_out else [])
    else:
      every = lanes + [np.ones(lanes[0].shape)]
      lanes = [np.where(valid, l, 0.0) for l in lanes] + [np.broadcast_to(valid, lanes[0].shape).astype(np.float64)]
      if twin_out:  # lanes 6-11: the same statistics over all points
        lanes = lanes + every
    flag, buf = w_buf.factored
    assert flag == (w_flags & (_hip.BINNED_WT_X_ONLY | _hip.BINNED_WT_ROW_ONLY))
    fac = np.asarray(buf.ptr)
    wt = np.broadcast_to(fac.reshape(nBk, 1, nj) if flag == _hip.BINNED_WT_X_ONLY else fac.reshape(nBk, nBr, 1), (nBk, nBr, nj))
    bits = np.asarray(w_buf.bufs[1].ptr).reshape(nBk, nBr, nj)
    member = ((bits[..., None] >> np.arange(nbin, dtype=np.uint64)) & np.uint64(1)).astype(np.float64)
    member = np.broadcast_to(member, (nBk, nBr, nj, nbin)) if nj > 1 else np.broadcast_to(member, (nBk, nBr, 1, nbin)).repeat(plan.nx, axis=2)
    out = np.empty((nA, nBk, len(lanes), 1, nbin))
    for l, v in enumerate(lanes):
      v = v.reshape(nA, nBk, nBr, plan.ndepth, plan.nx) * wt[None, :, :, None, :]
      out[:, :, l, 0, :] = np.einsum('abrdx,brxn->abn', v, member)
  return out, out.shape


def _run_map(ctx, kind, dplan, plan, devs, dtype_code, lane, func=0, ens=None):
  with np.errstate(all='ignore'):
    if kind == 'det':
      nin = {_hip.DET3: 2, _hip.DET6: 3, _hip.PASS1: 1}[func]
      lanes = _det_lanes(func, [devs[i].ptr[_offsets(plan, i)] for i in range(nin)])
    else:
      lanes = _ens_lanes(plan, devs, ens, plan.flags)
  return np.ascontiguousarray(lanes[lane]).reshape(-1)


def _run_s2(ctx, s2, partial, w_buf):
  part = np.asarray(partial).reshape(s2.nA, s2.nBk, s2.nBr, s2.nchunk, s2.nlane, s2.nj)
  if w_buf.kind == 'bits':
    wt = np.asarray(w_buf.bufs[0].ptr).reshape(s2.nBk, s2.nBr, s2.nj)
    bits = np.asarray(w_buf.bufs[1].ptr).reshape(s2.nBk, s2.nBr, s2.nj)
    member = ((bits[..., None] >> np.arange(s2.nbin, dtype=np.uint64)) & np.uint64(1)).astype(np.float64)
    w = wt[..., None] * member
  else:
    w = np.asarray(w_buf.bufs[0].ptr).reshape(s2.nBk, s2.nBr, s2.nj, s2.nbin)
  with np.errstate(all='ignore'):
    if s2.sum_j:
      out = np.einsum('abrclj,brjn->abln', part, w)[:, :, :, None, :]
    else:
      out = np.einsum('abrclj,brjn->abljn', part, w)
  return np.ascontiguousarray(out), out.shape


def install(monkeypatch):
  ctx = FakeCtx()
  monkeypatch.setattr(_hip, 'default_context', lambda device_id=None: ctx)
  monkeypatch.setattr(engine, '_to_device', _to_device)
  monkeypatch.setattr(engine, '_mask_to_device', _mask_to_device)
  monkeypatch.setattr(engine, '_device_plan', lambda c, plan: plan)
  monkeypatch.setattr(engine, '_swap_gather_table', lambda c, dplan, plan_v: plan_v)
  monkeypatch.setattr(engine, '_run_s1', _run_s1)
  monkeypatch.setattr(engine, '_run_map', _run_map)
  monkeypatch.setattr(engine, '_run_s2', _run_s2)
  monkeypatch.setattr(engine, '_run_binned', _run_binned)
  monkeypatch.setattr(engine, '_run_ens_binned', _run_ens_binned)
  monkeypatch.setattr(engine, '_ens_binned_atoms', _ens_binned_atoms)
  monkeypatch.setattr(engine, '_acc_add', _acc_add)
  monkeypatch.setattr(engine, 'new_context', lambda: FakeCtx())


def _run_spectrum(field, lon_dim, group, scale, ngroup, cache=None):
  """NumPy stand-in for wbx_zonal_spectrum (rows = non-longitude dims in the field's own order)."""
  vals = np.asarray(field.values)
  if vals.dtype != np.float32:
    raise TypeError('zonal spectra take float32 fields (rocFFT single precision); cast the input')
  ax = field.dims.index(lon_dim)
  rows = np.moveaxis(vals.astype(np.float64), ax, -1).reshape(-1, vals.shape[ax])
  n = rows.shape[1]
  F = np.fft.rfft(rows.astype(np.float32).astype(np.float64), axis=1) / n
  power = F.real ** 2 + F.imag ** 2
  power[:, 1:] *= 2
  out = np.zeros((ngroup, n // 2 + 1))
  np.add.at(out, np.asarray(group), power * np.asarray(scale)[:, None])
  return out


class _FakeSlabPool:
  """NumPy stand-in for climatology_cache._HipPool: the "device" pool is a host array, copies land at once."""

  def __init__(self, nslots, slab_shape, np_dtype, swap=False, threads=1):
    self.dtype = np.dtype(np_dtype)
    self.swap = swap
    self.slab_shape = tuple(int(n) for n in slab_shape)
    self.slab_nbytes = int(np.prod(self.slab_shape, dtype=np.int64)) * self.dtype.itemsize
    self.array = np.full((nslots,) + self.slab_shape, np.nan, self.dtype)  # (a slot that was never filled poisons a result)
    self.writes = []  # (slot, key) in the order they landed

  def payload(self, nslots):
    return self.array

  def seed(self, pool_da, dims):
    pass

  def fences_now(self):
    return []

  def submit(self, job):
    self.array[job.slot] = np.swapaxes(np.asarray(job.src), -1, -2) if self.swap else np.asarray(job.src)
    self.writes.append((job.slot, job.key))
    job.event.set()

  def order(self, job):
    pass

  def settle(self, job):
    pass

  def close(self):
    pass


_install_without_spectrum = install


def install(monkeypatch):  # noqa: F811
  _install_without_spectrum(monkeypatch)
  from weatherbenchx_amd import climatology_cache
  from weatherbenchx_amd import spectra
  monkeypatch.setattr(climatology_cache, '_new_pool', _FakeSlabPool)
  monkeypatch.setattr(spectra, '_run_spectrum', _run_spectrum)
