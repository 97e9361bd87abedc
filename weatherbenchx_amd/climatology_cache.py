"""Climatology slab cache: a host-resident climatology behind a small pool of device slabs.

The reference aligns a lazily backed climatology slice by slice -- `climatology.sel(dayofyear=..., hour=...).compute()`
(weatherbenchX/metrics/base.py:396-403) over a dataset that `ClimatologyFromXarray` opened without loading
(weatherbenchX/data_loaders/xarray_loaders.py:266-316): only the slices a chunk's valid times name ever leave the store.  A
[366, 4, 37, 721, 1440] float32 climatology is 225 GB per variable: it fits neither one upload nor (with the chunk buffers) the
288 GB of an MI355X, and a chunk of 1 init x 20 leads touches 20 of its 1464 (dayofyear, hour) slabs.

`SlabCache` keeps the climatology where it is -- a NumPy array, a memory map (`np.load(mmap_mode='r')`, NetCDF-3 through
scipy's mmap) or page-locked memory -- and holds K whole slabs `[level, latitude, longitude]` in ONE device allocation, LRU by
slab key (the positions along the selected dims).  What the kernels see is unchanged: the stage-1 gather table
(`lazy.gather_from_ref`) addresses input 2 as `base + table[init, lead] + inner offsets`; with the cache the base is the pool and
the table holds POOL SLOT offsets instead of source strides, so a plan, a chunk record (`replay.py`: the table is a relocation
already) and every kernel run exactly as with a resident climatology.

Ordering, without blocking the host and without a lock around the device:
  * the slot table is only touched by the thread that runs the chunk loop;
  * misses go to ONE worker thread that owns a copy stream (its own `wbx_ctx`): it waits -- stream-ordered -- for the fences
    that cover the last kernels that read the slot it overwrites, moves the slab (straight DMA from page-locked memory; the
    runtime's staged copy from pageable / mapped memory, 55 GB/s; through a page-locked staging ring when the slab is strided or
    of another dtype) and records a fence;
  * before a chunk's launches every launch stream is told to wait for the fences of the slabs that chunk reads
    (`wbx_ctx_wait_fence`: hipStreamWaitEvent) -- outside the chunk record, so a replayed chunk does the same with ITS fences;
  * `chunk_enqueued()` (called by `pipeline._consume` behind every chunk) stamps the slots that chunk read with fences on the
    launch streams: an eviction waits for the LAST USE of its victim, not for "now", so the upload of chunk k + 1's misses
    overlaps chunk k's kernels (LRU victims were last read many chunks ago);
  * `pipeline._consume` asks for chunk k + 1's slabs right behind chunk k's launches (`prefetch_for`): one chunk ahead on the
    copy stream, like the chunk feeder's uploads of the fields themselves.
A slab is never evicted while a launch that names it may still be enqueued: the slot table of a `ClimatologyRef` is written
when its launch is about to be made (statistics are lazy: a table written at `compute` time could be stale by then), and slabs
named since the last `chunk_enqueued()` are protected; a pool that cannot hold one chunk's slabs raises and names the number.
"""
from __future__ import annotations

import collections
import os
import queue
import threading
import weakref

import numpy as np

from weatherbenchx_amd import _hip
from weatherbenchx_amd import xarray_lite as xr

SLAB_DIM = 'wbx_slab'  # the pool's leading dim: never a dim of a statistic (the gather table selects along it)

# Host climatologies above this size (or memory maps of any size) get a cache by themselves (`cache_for`); below it the whole
# array is uploaded once, as before.  Pool size of such an automatic cache.
AUTO_RESIDENT_BYTES = int(os.environ.get('WBX_CLIM_RESIDENT_BYTES', 8 << 30))
AUTO_POOL_BYTES = int(os.environ.get('WBX_CLIM_POOL_BYTES', 24 << 30))

_LIVE = weakref.WeakSet()  # every cache of this process (chunk_enqueued / prefetch_for walk it)


class _Job:
  """One slab on its way into a slot."""
  __slots__ = ('slot', 'key', 'src', 'after', 'fence', 'event', 'error', 'ordered', 'age')

  def __init__(self, slot, key, src, after):
    self.slot, self.key, self.src, self.after = slot, key, src, after
    self.fence = None
    self.event = threading.Event()
    self.error = None
    self.ordered = set()  # ids of the launch contexts that have been told to wait for `fence`
    self.age = 0


class _HipPool:
  """The device side: one allocation of `nslots` slabs, a copy stream and the worker thread that feeds it."""

  def __init__(self, nslots, slab_shape, np_dtype, threads=4, swap=False):
    from weatherbenchx_amd import engine  # pylint: disable=g-import-not-at-top
    self.dtype = np.dtype(np_dtype)
    self.swap = bool(swap)  # the slabs are stored with their last two dims exchanged (`slab_shape` is the POOL's)
    self.slab_shape = tuple(int(n) for n in slab_shape)
    self.slab_elems = int(np.prod(self.slab_shape, dtype=np.int64))
    self.slab_nbytes = self.slab_elems * self.dtype.itemsize
    self.ctx = _hip.default_context()
    self.buf = self.ctx.alloc(nslots * self.slab_nbytes)
    self.copy_ctx = engine.new_context()  # (used by the worker thread alone from here on)
    self._engine = engine
    self._threads = max(1, int(threads))
    self._pool = None
    self._stage = []  # [array, fence of the DMA that last read it]
    self._queue = queue.Queue()
    self._thread = threading.Thread(target=self._run, name='wbx-clim-slabs', daemon=True)
    self._thread.start()

  # what `engine._to_device` hands to the launches for the pool DataArray
  def payload(self, nslots):
    return np.broadcast_to(np.zeros((), self.dtype), (nslots,) + self.slab_shape)  # (frame only: no host memory behind it)

  def seed(self, pool_da, dims):
    """The pool as the engine addresses it: C-contiguous `[slot, *slab dims]` in ONE device block."""
    from weatherbenchx_amd import planner  # pylint: disable=g-import-not-at-top
    strides, acc = {}, 1
    for d, n in zip(reversed(dims[1:]), reversed(self.slab_shape)):
      strides[d] = acc
      acc *= int(n)
    strides[dims[0]] = self.slab_elems
    code = _hip.F32 if self.dtype == np.float32 else _hip.F64
    lay = planner.InputLayout(strides=strides, itemsize=self.dtype.itemsize, base_alignment=256)
    pool_da.__dict__['_wbx_dev'] = {code: self._engine._Dev(int(self.buf.ptr), lay, code, self.buf, int(self.buf.nbytes))}  # pylint: disable=protected-access

  def launch_contexts(self):
    return self._engine.launch_contexts()

  def fences_now(self):
    """Fences behind everything enqueued so far on every launch stream."""
    return [c.fence() for c in self.launch_contexts()]

  def submit(self, job):
    self._queue.put(job)

  def order(self, job):
    """Every launch stream waits for the slab of `job` (stream-ordered); blocks the host only until the worker has ENQUEUED
    the copy.  Not part of a chunk record: a replayed chunk orders its own slabs."""
    job.event.wait()
    if job.error is not None:
      raise job.error
    rec = _hip.RECORDER
    if rec is not None:
      rec.paused += 1
    try:
      for c in self.launch_contexts():
        if id(c) not in job.ordered:
          c.wait_fence(job.fence)
          job.ordered.add(id(c))
    finally:
      if rec is not None:
        rec.paused -= 1

  def settle(self, job):
    job.event.wait()
    if job.error is not None:
      raise job.error
    job.fence.wait()

  def close(self):
    self._queue.put(None)

  # -- worker thread ------------------------------------------------------------------------------------------------
  def _run(self):
    while True:
      job = self._queue.get()
      if job is None:
        return
      try:
        self._upload(job)
      except BaseException as e:  # pylint: disable=broad-except
        job.error = e
      job.src = None
      job.after = None
      job.event.set()

  def _staging(self):
    if len(self._stage) < 2:
      self._stage.append([self.copy_ctx.pinned_empty(self.slab_shape, self.dtype), None])
      return self._stage[-1]
    entry = self._stage.pop(0)
    if entry[1] is not None:
      entry[1].wait()  # (the DMA that read this block)
    self._stage.append(entry)
    return entry

  def _copy_rows(self, dst, src):
    if self.swap:
      return self._transpose_rows(dst, src)
    n = src.shape[0] if src.ndim else 0
    if self._threads > 1 and n > 1 and dst.nbytes >= (8 << 20):
      if self._pool is None:
        from concurrent.futures import ThreadPoolExecutor  # pylint: disable=g-import-not-at-top
        self._pool = ThreadPoolExecutor(max_workers=self._threads, thread_name_prefix='wbx-clim-copy')
      cuts = np.linspace(0, n, min(self._threads, n) + 1).astype(int)
      list(self._pool.map(lambda ab: np.copyto(dst[ab[0]:ab[1]], src[ab[0]:ab[1]], casting='unsafe'), zip(cuts[:-1], cuts[1:])))
    else:
      np.copyto(dst, src, casting='unsafe')

  def _transpose_rows(self, dst, src):
    """dst[.., c, r] = src[.., r, c] (wbx_host_transpose, planes dealt to the copy threads): the climatology archive is
    [.., longitude, latitude] like the fields (loaders.FileLoader(device_layout=...)), the pool holds what the kernels want."""
    lib = self.copy_ctx.lib
    rows, cols = (int(n) for n in src.shape[-2:])
    nb = int(np.prod(src.shape[:-2], dtype=np.int64))
    if not (src.dtype == self.dtype and src.dtype.isnative and src.flags['C_CONTIGUOUS']):
      src = np.ascontiguousarray(src, dtype=self.dtype)
    s3, d3 = src.reshape(nb, rows, cols), dst.reshape(nb, cols, rows)

    def part(ab):
      _hip.check(lib.wbx_host_transpose(d3[ab[0]:ab[1]].ctypes.data, s3[ab[0]:ab[1]].ctypes.data, ab[1] - ab[0], rows, cols,
                                        self.dtype.itemsize), 'wbx_host_transpose')
    if self._threads > 1 and nb > 1 and dst.nbytes >= (8 << 20):
      if self._pool is None:
        from concurrent.futures import ThreadPoolExecutor  # pylint: disable=g-import-not-at-top
        self._pool = ThreadPoolExecutor(max_workers=self._threads, thread_name_prefix='wbx-clim-copy')
      cuts = np.linspace(0, nb, min(4 * self._threads, nb) + 1).astype(int)
      list(self._pool.map(part, zip(cuts[:-1], cuts[1:])))
    else:
      part((0, nb))

  def _upload(self, job):
    import ctypes as C  # pylint: disable=g-import-not-at-top
    ctx = self.copy_ctx
    for f in job.after or ():
      ctx.wait_fence(f)  # the last kernels that read the slab this one replaces
    src = job.src
    dst = C.c_void_p(int(self.buf.ptr) + job.slot * self.slab_nbytes)
    plain = (not self.swap and isinstance(src, np.ndarray) and src.dtype == self.dtype and src.flags['C_CONTIGUOUS']
             and src.dtype.isnative)
    if plain and _hip.is_pinned(src):
      _hip.check(ctx.lib.wbx_memcpy_h2d_async(ctx.handle, dst, C.c_void_p(src.ctypes.data), self.slab_nbytes), 'wbx_memcpy_h2d_async')
      job.fence = ctx.fence()
      job.fence.keep = src  # (the source block stays with the fence)
    elif plain:
      # pageable or mapped: the runtime's staged copy, on the copy stream (it returns when the source has been read)
      _hip.check(ctx.lib.wbx_memcpy_h2d(ctx.handle, dst, C.c_void_p(src.ctypes.data), self.slab_nbytes), 'wbx_memcpy_h2d')
      job.fence = ctx.fence()
    else:
      entry = self._staging()
      self._copy_rows(entry[0], np.asarray(src))
      _hip.check(ctx.lib.wbx_memcpy_h2d_async(ctx.handle, dst, C.c_void_p(entry[0].ctypes.data), self.slab_nbytes), 'wbx_memcpy_h2d_async')
      job.fence = entry[1] = ctx.fence()


def _new_pool(nslots, slab_shape, np_dtype, swap=False, threads=4):
  """(tests/fake_device.py swaps this for a NumPy pool)"""
  return _HipPool(nslots, slab_shape, np_dtype, threads=threads, swap=swap)


class SlabCache:
  """K device slabs of one climatology variable, LRU by slab key.  `source`: the host DataArray, any dim order; the dims the
  alignment selects along (`dayofyear`, `hour` / `time`) are found at the first `ref` call."""

  def __init__(self, source: xr.DataArray, slots: int | None = None, pool_bytes: int | None = None,
               device_layout: str | None = None, threads: int = 4):
    """`device_layout`: 'lon_fastest' / 'lat_fastest' -- the pool holds the slabs with their last two dims exchanged when the
    host climatology has them the other way round (the staging copy is the transposition: `loaders.FileLoader` does the
    same to the fields of such an archive)."""
    data = source.data
    if xr._is_torch(data):  # pylint: disable=protected-access
      raise TypeError('SlabCache: the climatology is a tensor already (device-resident climatologies are gathered in place)')
    self.source = source
    self._slots_wanted, self._pool_bytes = slots, pool_bytes
    self._device_layout, self._threads = device_layout, threads
    self.sel_dims = None
    self.pool = None
    self.pool_da = None
    self.slot_of = collections.OrderedDict()  # key -> slot, least recently used first
    self.free = []
    self.jobs = {}       # slot -> _Job whose copy may still be in flight
    self.last_use = {}   # slot -> fences behind the last chunk that read it (None: read since the last stamp)
    self.touched = set()  # keys named by launches that may not have been enqueued yet (+ the next chunk's, asked for ahead)
    self.stats = {'uploads': 0, 'upload_bytes': 0, 'hits': 0, 'evictions': 0, 'requests': 0, 'prefetched': 0}
    self._stamps = collections.deque(maxlen=6)
    _LIVE.add(self)

  # -- set-up -----------------------------------------------------------------------------------------------------------
  def _setup(self, sel_dims):
    src = self.source
    self.sel_dims = tuple(sel_dims)
    stored = tuple(d for d in src.dims if d not in self.sel_dims)
    from weatherbenchx_amd import loaders  # pylint: disable=g-import-not-at-top
    swap = loaders._needs_swap(stored, self._device_layout)  # pylint: disable=protected-access
    self.slab_dims = stored[:-2] + (stored[-1], stored[-2]) if swap else stored
    slab_shape = tuple(src.sizes[d] for d in self.slab_dims)
    dt = np.dtype(str(src.dtype))
    self.np_dtype = dt if dt in (np.dtype(np.float32), np.dtype(np.float64)) else np.dtype(np.float64)
    slab_nbytes = int(np.prod(slab_shape, dtype=np.int64)) * self.np_dtype.itemsize
    total = int(np.prod([src.sizes[d] for d in self.sel_dims], dtype=np.int64))
    n = self._slots_wanted
    if n is None:
      n = max(1, int((self._pool_bytes or AUTO_POOL_BYTES) // max(slab_nbytes, 1)))
    self.nslots = int(min(n, total))
    self.pool = _new_pool(self.nslots, slab_shape, self.np_dtype, swap=swap, threads=self._threads)
    weakref.finalize(self, self.pool.close)  # (the worker thread holds the pool, not the cache)
    dims = (SLAB_DIM,) + self.slab_dims
    coords = {k: v for k, v in src._coords.items() if set(v[0]) <= set(self.slab_dims)}  # pylint: disable=protected-access
    self.pool_da = xr.DataArray._assemble(self.pool.payload(self.nslots), dims, coords, name=src.name, attrs=src.attrs)  # pylint: disable=protected-access
    self.pool.seed(self.pool_da, dims)
    self.free = list(range(self.nslots - 1, -1, -1))
    # the source with the selected dims in front: a slab is `front[key]`
    order = [src.dims.index(d) for d in self.sel_dims] + [src.dims.index(d) for d in stored]
    self._front = np.transpose(src.data if isinstance(src.data, np.ndarray) else xr._to_numpy(src.data), order)  # pylint: disable=protected-access

  # -- the slot table (chunk-loop thread only) -------------------------------------------------------------------------
  def ensure(self, keys, *, prefetch=False) -> dict:
    """{key: slot} with every key resident or on its way; the misses are handed to the copy stream.  `prefetch`: best effort
    -- a slab that finds no slot is left out (the launch that needs it asks again)."""
    out, misses = {}, []
    self.stats['requests'] += 1
    for key in keys:
      if key in out:
        continue
      slot = self.slot_of.get(key)
      if slot is None:
        misses.append(key)
        out[key] = None
      else:
        self.slot_of.move_to_end(key)
        out[key] = slot
        self.stats['hits'] += 1
    want = set(out)
    if len(want) > self.nslots and not prefetch:
      raise ValueError(f'climatology slab pool: one request names {len(want)} slabs, the pool has {self.nslots} slots '
                       f'(SlabCache(slots=...) / WBX_CLIM_POOL_BYTES)')
    for key in misses:
      got = self._take_slot(want, prefetch)
      if got is None:
        del out[key]
        want.discard(key)
        continue
      slot, after = got
      self.slot_of[key] = slot
      out[key] = slot
      job = _Job(slot, key, self._front[key], after)
      self.jobs[slot] = job
      self.last_use[slot] = None
      self.pool.submit(job)
      self.stats['uploads'] += 1
      self.stats['upload_bytes'] += self.pool.slab_nbytes
      if prefetch:
        self.stats['prefetched'] += 1
    self.touched |= want
    return out

  def _take_slot(self, want, prefetch):
    if self.free:
      return self.free.pop(), None
    for key, slot in self.slot_of.items():  # least recently used first
      if key in want or key in self.touched:
        continue
      del self.slot_of[key]
      self.stats['evictions'] += 1
      self.jobs.pop(slot, None)  # (a copy of the old slab still in flight: the copy stream runs in order)
      after = self.last_use.get(slot)
      if after is None:
        after = self.pool.fences_now()  # read since the last stamp (or outside a chunk loop): whatever is enqueued by now
      return slot, after
    if prefetch:
      return None
    raise ValueError(f'climatology slab pool: {self.nslots} slots cannot hold the {len(want | self.touched)} slabs named by '
                     'launches in the making (this chunk and what was asked for ahead): raise SlabCache(slots=...) / '
                     'WBX_CLIM_POOL_BYTES')

  def order(self, slots):
    """The launch streams wait for the copies of `slots` that may still be in flight."""
    for slot in slots:
      job = self.jobs.get(slot)
      if job is None:
        continue
      self.pool.order(job)
      job.age += 1
      if job.age > 8:  # long landed: forget the job (the wait returns at once)
        self.pool.settle(job)
        del self.jobs[slot]

  def chunk_enqueued(self):
    """Everything named so far has been launched: stamp the slots read since the last stamp, lift their protection."""
    if not self.touched:
      return
    fences = None
    for key in self.touched:
      slot = self.slot_of.get(key)
      if slot is not None:
        if fences is None:
          fences = self.pool.fences_now()
          # (dropping a fence waits for it -- an event is not destroyed while a stream may be told to wait on it -- and the slots
          #  of consecutive chunks are mostly the same: the stamps of the last few chunks are kept until their kernels are long done,
          #  so that re-stamping a slot never blocks the host on the chunk that is still running)
          self._stamps.append(fences)
        self.last_use[slot] = fences
    self.touched = set()

  # -- what the statistics ask for ---------------------------------------------------------------------------------------
  def ref(self, over_dims, positions: dict, *, prefetch=False):
    """The aligned climatology of one chunk as a `lazy.ClimatologyRef` over the POOL: positions = {SLAB_DIM: slot table}.
    The table is (re)written when a launch is about to read it (`activate`, called by lazy.gather_from_ref): between this
    call and that launch -- statistics are lazy -- other chunks may have taken the slots."""
    from weatherbenchx_amd import lazy  # pylint: disable=g-import-not-at-top
    if self.pool is None:
      self._setup(tuple(positions))
    elif tuple(positions) != self.sel_dims:
      raise ValueError(f'climatology slab cache built for selections along {self.sel_dims}, asked along {tuple(positions)}')
    pos = [np.asarray(positions[d]) for d in self.sel_dims]
    shape = pos[0].shape
    keys = list(zip(*[p.reshape(-1).tolist() for p in pos])) if pos[0].size else []
    if prefetch:
      self.ensure(keys, prefetch=True)
    table = np.full(shape, -1, dtype=np.int64)  # (filled in by `activate`)
    ref = lazy.ClimatologyRef(self.pool_da, over_dims, {SLAB_DIM: table})
    ref.origin = (self.source, dict(positions))
    ref.activate = lambda c=weakref.ref(self), k=keys, t=table: _activate(c, k, t)
    return ref

  def activate(self, keys, table):
    if not _IN_LOOP[0]:
      # outside a chunk loop every launch follows its table at once: what earlier launches named has been enqueued
      self.chunk_enqueued()
    slot_of = self.ensure(keys)
    table.reshape(-1)[:] = [slot_of[k] for k in keys]
    self.order(sorted(set(slot_of.values())))


def _activate(cache_ref, keys, table):
  cache = cache_ref()
  if cache is None:
    raise RuntimeError('the climatology slab cache of this statistic is gone (its climatology was edited or dropped)')
  cache.activate(keys, table)


_IN_LOOP = [0]  # > 0: inside pipeline._consume (launches of a replayed chunk follow their tables later, as one call)


# ---- the hooks of metrics/base.py and pipeline.py ---------------------------------------------------------------------------
def cached(climatology, slots: int | None = None, pool_bytes: int | None = None, device_layout: str | None = None, threads: int = 4):
  """A Dataset / mapping / DataArray of host climatologies with a slab cache on every variable (explicit form of what
  `cache_for` does by itself for memory maps and large arrays).  Returns its argument.  Call it on the object the metrics are
  given (`ACC(cached(ds, slots=48))`): the wish is kept on the DataArray OBJECTS, and a Dataset built from DataArrays afterwards
  holds objects of its own."""
  arrays = [climatology] if isinstance(climatology, xr.DataArray) else [climatology[k] for k in climatology.keys()]
  for da in arrays:
    da = xr.as_dataarray(da)
    if not xr._is_torch(da.data):  # pylint: disable=protected-access
      # (the wish outlives what the engine caches on the object -- an in-place edit `da[...] = x` drops every `_wbx_*` entry, the
      #  pool with its stale slabs among them -- and travels with a pickled metric; the pool itself is rebuilt on first use)
      da.__dict__['_slab_cache_config'] = (slots, pool_bytes, device_layout, threads)
      da.__dict__.pop('_wbx_slab_cache', None)
  return climatology


def _is_memmap(a) -> bool:
  while isinstance(a, np.ndarray):
    if isinstance(a, np.memmap):
      return True
    a = a.base
  import mmap  # pylint: disable=g-import-not-at-top
  return isinstance(a, mmap.mmap)


def cache_for(climatology: xr.DataArray):
  """The slab cache of a climatology variable, or None when it is (to be) resident as a whole: tensors, and host arrays
  below AUTO_RESIDENT_BYTES that are not memory maps."""
  hit = climatology.__dict__.get('_wbx_slab_cache')
  if hit is not None:
    return hit or None
  data = climatology.data
  config = climatology.__dict__.get('_slab_cache_config')
  if config is not None and not xr._is_torch(data):  # pylint: disable=protected-access
    made = SlabCache(climatology, slots=config[0], pool_bytes=config[1], device_layout=config[2], threads=config[3])
  elif isinstance(data, np.ndarray) and '_wbx_dev' not in climatology.__dict__ and (_is_memmap(data) or data.nbytes > AUTO_RESIDENT_BYTES):
    made = SlabCache(climatology)
  else:
    made = False
  climatology.__dict__['_wbx_slab_cache'] = made
  return made or None


def wanted_by(metric_sets) -> bool:
  """Does a climatology of these metrics sit (or will it sit) behind a slab pool?  Decided from the climatology objects alone --
  every rank of a job answers alike."""
  for metrics in metric_sets:
    for metric in metrics.values():
      for stat in metric.statistics.values():
        clim = getattr(stat, '_climatology', None)
        if clim is None:
          continue
        try:
          arrays = [clim] if isinstance(clim, xr.DataArray) else [clim[k] for k in clim.keys()]
        except (AttributeError, TypeError):
          continue
        for da in arrays:
          if cache_for(xr.as_dataarray(da)) is not None:
            return True
  return False


def chunk_enqueued():
  for cache in list(_LIVE):
    if cache.pool is not None:
      cache.chunk_enqueued()


class chunk_loop:  # pylint: disable=invalid-name
  """`with chunk_loop():` -- the launches of a chunk may follow their tables later (a replayed chunk is ONE call behind all
  its tables): protection is lifted by `chunk_enqueued()` alone."""

  def __enter__(self):
    _IN_LOOP[0] += 1

  def __exit__(self, *exc):
    _IN_LOOP[0] -= 1
    chunk_enqueued()
    return False


def prefetch_for(passes, group):
  """The slabs the NEXT chunk (`group`: one (offsets, predictions, targets) per pass) names, asked for one chunk ahead: the
  alignment every climatology statistic of `passes` will ask for is resolved now and left on the predictions
  (`PerVariableStatisticWithClimatology._compute_per_variable` finds it there)."""
  from weatherbenchx_amd.metrics import base as metrics_base  # pylint: disable=g-import-not-at-top
  for (_, metrics, _), (_, predictions, _) in zip(passes, group):
    seen = set()
    for metric in metrics.values():
      for stat in metric.statistics.values():
        clim = getattr(stat, '_climatology', None)
        if clim is None or not isinstance(stat, metrics_base.PerVariableStatisticWithClimatology) or id(clim) in seen:
          continue
        seen.add(id(clim))
        for name in predictions.keys():
          try:
            c = xr.as_dataarray(clim[name])
          except (KeyError, TypeError):
            continue
          if cache_for(c) is not None:
            stat.resolve_climatology(xr.as_dataarray(predictions[name]), c, prefetch=True)


def active() -> bool:
  """Some climatology of this process sits behind a slab pool."""
  return bool(len(_LIVE))
