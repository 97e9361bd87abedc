"""Replayable chunk records: the steady state of the chunk loop without Python in it.

The body of the reference's per-chunk stage (beam_pipeline.py:161-250: the statistics of one chunk -> aggregation states ->
CombinePerKey) is, chunk after chunk, the SAME sequence of calls into libwbx_hip: same plans, same weights / bins / atom tables,
same scratch buffers and accumulator slots.  What follows the chunk is small: the device pointers of its inputs (and of their
`mask` coordinates), the fences of their uploads, and -- for statistics against a climatology -- the plan variant that carries
the chunk's gather table.  `ChunkRecorder` watches ONE chunk go through the ordinary path (every library call of the recording
thread, with its arguments; `_hip._RecordingLib`), `ChunkRecord` is what is left of it: an array of `wbx_call`, the relocations
of the arguments that follow the chunk, and the objects that own every other pointer the calls name.  `ChunkRecord.replay` runs
a later chunk with ONE library call (`wbx_chunk_replay`, include/wbx.h).

Safety comes from refusing, not from guessing: a recording is only turned into a record when
  * every call it saw is either a pure query or one of the enqueue-only entry points a record may hold (`_hip.FN_IDS`): an
    allocation, a synchronous copy, a read-back or a timer in the chunk means "not in steady state" -> no record;
  * every pointer argument is accounted for: inside one of the chunk's inputs (-> relocated), a plan variant of a gather table
    (-> relocated), or inside memory that the record itself keeps alive (plans, weights, atom tables, scratch, accumulator
    blocks, device blocks taken from the pool while recording -- those are pinned to the record).  One stray pointer -> no record;
  * no accumulator slot was created and nothing was summed on the host while recording.
A chunk that cannot be replayed simply takes the ordinary path.  `WBX_CHUNK_REPLAY=0` turns the whole mechanism off (A/B)."""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

from weatherbenchx_amd import _hip
from weatherbenchx_amd import xarray_lite as xr

ENABLED = os.environ.get('WBX_CHUNK_REPLAY', '1') != '0'
_MASK64 = (1 << 64) - 1
TIME_DIMS = frozenset({'init_time', 'lead_time', 'valid_time', 'time'})
STATS = {'recorded': 0, 'refused': 0, 'replayed': 0, 'refusals': []}  # (tests and tools read these)


def reset_stats():
  STATS.update(recorded=0, refused=0, replayed=0, refusals=[])


def active():
  """The recorder of THIS thread's chunk, or None."""
  rec = _hip.RECORDER
  return rec if (rec is not None and rec.thread == threading.get_ident()) else None


def keep(*objs):
  """Called by the launch paths: these objects own memory the calls being made name (plans, weights, tables)."""
  rec = active()
  if rec is not None:
    rec.kept.extend(o for o in objs if o is not None)


def _arg64(arg, keepalive):
  if arg is None:
    return 0
  if isinstance(arg, bool):
    return int(arg)
  if isinstance(arg, (int, np.integer)):
    return int(arg) & _MASK64
  if isinstance(arg, C.c_void_p):
    return int(arg.value or 0)
  obj = getattr(arg, '_obj', None)  # C.byref(struct)
  if obj is not None:
    keepalive.append(obj)
    return C.addressof(obj)
  if isinstance(arg, C._SimpleCData):  # pylint: disable=protected-access
    return int(arg.value or 0) & _MASK64
  if isinstance(arg, (C.Structure, C.Array)):
    keepalive.append(arg)
    return C.addressof(arg)
  raise TypeError(f'cannot record an argument of type {type(arg).__name__}')


def _is_pointer_type(argtype) -> bool:
  return argtype is C.c_void_p or argtype is C.c_char_p or (isinstance(argtype, type) and issubclass(argtype, C._Pointer))  # pylint: disable=protected-access


def _ranges_of(obj, out, seen, depth=0):
  """(address, nbytes) of every device block / host array reachable from `obj` (plans, weights, atom tables, scratch ...)."""
  if obj is None or depth > 6 or id(obj) in seen:
    return
  seen[id(obj)] = obj  # (holds the object: the id of a temporary container must not come round again during the walk)
  if isinstance(obj, _hip.DeviceBuffer):
    out.append((int(obj.ptr), int(obj.nbytes)))
    return
  if isinstance(obj, np.ndarray):
    if obj.size:
      out.append((int(obj.__array_interface__['data'][0]), int(obj.nbytes)))
    return
  if isinstance(obj, (C.Structure, C.Array)):
    out.append((C.addressof(obj), C.sizeof(obj)))
    return
  if isinstance(obj, _hip.PinnedBlock):
    out.append((int(obj.ptr), int(obj._capacity)))  # pylint: disable=protected-access
    return
  if xr._is_torch(obj):  # pylint: disable=protected-access
    if obj.numel():
      out.append((int(obj.data_ptr()), int(obj.numel() * obj.element_size())))
    return
  if isinstance(obj, (str, bytes, int, float, bool, type)):
    return
  if isinstance(obj, dict):
    for v in obj.values():
      _ranges_of(v, out, seen, depth + 1)
    return
  if isinstance(obj, (list, tuple, set, frozenset)):
    for v in obj:
      _ranges_of(v, out, seen, depth + 1)
    return
  ptr, nbytes = getattr(obj, 'ptr', None), getattr(obj, 'nbytes', None)
  if isinstance(ptr, int) and isinstance(nbytes, (int, np.integer)) and nbytes:
    out.append((int(ptr), int(nbytes)))  # (engine._Dev: a cached, constant input such as the climatology)
  d = getattr(obj, '__dict__', None)
  if d:
    for k, v in d.items():
      if k in ('ctx', '_ctx', 'lib', '_lib'):
        continue
      _ranges_of(v, out, seen, depth + 1)


class _Refused(Exception):
  pass


def input_ranges(da) -> list:
  """[(kind, key, address, nbytes, fence handle | 0)] of a chunk array: its payload as the engine will address it (a tensor in
  HBM in place; a host array through the device copies staged on the object) and its `mask` coordinate when that lives in HBM."""
  out = []
  data = da.data if not getattr(da, 'is_lazy', False) else None
  if data is not None and xr._is_torch(data) and data.is_cuda:  # pylint: disable=protected-access
    out.append(('data', 'torch', int(data.data_ptr()), int(data.numel() * data.element_size()), 0))
  else:
    for code, dev in (da.__dict__.get('_wbx_dev') or {}).items():
      fence = getattr(dev, 'fence', None)
      out.append(('data', code, int(dev.ptr), int(dev.nbytes), int(fence._h.value or 0) if fence is not None else 0))  # pylint: disable=protected-access
  mc = da._coords.get('mask') if hasattr(da, '_coords') else None  # pylint: disable=protected-access
  if mc is not None and xr._is_torch(mc[1]) and mc[1].is_cuda:  # pylint: disable=protected-access
    out.append(('mask', 'torch', int(mc[1].data_ptr()), int(mc[1].numel() * mc[1].element_size()), 0))
  return out


class ChunkRecorder:
  """Watches the library calls of one chunk on the calling thread (`with recorder: ...`)."""

  def __init__(self, arrays, acc):
    """arrays: {role: DataArray} -- the chunk's inputs by a name that the next chunk's inputs can be looked up under."""
    self.thread = threading.get_ident()
    self.arrays = dict(arrays)
    self.acc = acc
    self.log = []       # (name, args)
    self.kept = []      # owners of memory the calls name
    self.pinned = []    # device blocks taken from a pool while recording
    self.gathers = []   # {'addr', 'replan', 'source', 'p', 'dtype_code', 'ctx'}: plan variants that carry a chunk's gather table
    self.gather_context = None
    self.bumps = []     # (path, index): constant host terms whose multiplicity a chunk raises
    self.refusal = None
    self.paused = 0
    self.torch_inputs = False

  def __enter__(self):
    if _hip.RECORDER is not None:
      raise RuntimeError('a chunk is already being recorded')
    _hip.RECORDER = self
    self.acc._recorder = self  # pylint: disable=protected-access
    return self

  def __exit__(self, *exc):
    _hip.RECORDER = None
    self.acc._recorder = None  # pylint: disable=protected-access
    return False

  # -- notes left by the paths the chunk takes --------------------------------------------------------------------------------
  def note(self, name, args):
    if not self.paused:
      self.log.append((name, args))

  def refuse(self, why: str):
    if self.refusal is None:
      self.refusal = why

  def note_gather(self, dplan, replan):
    ctx_info = self.gather_context
    if ctx_info is None:
      self.refuse('a gather table without its climatology')
      return
    struct = getattr(dplan, 'struct', None)
    if struct is None:
      self.refuse('a plan without a device struct')
      return
    self.kept.append(dplan)
    self.gathers.append(dict(ctx_info, addr=C.addressof(struct), replan=replan))

  # -- the record -------------------------------------------------------------------------------------------------------------------
  def finish(self):
    """-> ChunkRecord, or None (STATS['refusals'] says why)."""
    try:
      rec = self._build()
      STATS['recorded'] += 1
      return rec
    except _Refused as e:
      STATS['refused'] += 1
      if len(STATS['refusals']) < 32:
        STATS['refusals'].append(str(e))
      return None

  def _build(self):
    if self.refusal is not None:
      raise _Refused(self.refusal)
    keepalive = []
    calls = []
    for name, args in self.log:
      if name in _hip.QUERY_FNS or name in _hip.HOST_FENCE_FNS:
        continue
      if name == 'wbx_malloc':
        # a pool that had no block of this size left (the blocks an earlier record took stay with that record): the new block
        # is pinned to THIS record (`DeviceBuffer.__init__` -> `pinned`), a replay allocates nothing.  What an allocation would
        # be a symptom of -- something uploaded per chunk -- comes with a copy, and that is refused below.
        continue
      fn = _hip.FN_IDS.get(name)
      if fn is None:
        raise _Refused(f'{name} in the chunk (not an enqueue-only entry point: the loop is not in steady state)')
      argtypes = _hip.PROTOS[name]
      if len(args) != len(argtypes) or len(args) > _hip.CALL_MAX_ARGS:
        raise _Refused(f'{name}: {len(args)} arguments')
      try:
        vals = [_arg64(a, keepalive) for a in args]
      except TypeError as e:
        raise _Refused(f'{name}: {e}') from e
      calls.append((name, fn, vals, [_is_pointer_type(t) for t in argtypes]))
    if not any(name not in ('wbx_acc_add', 'wbx_ctx_wait_fence') for name, *_ in calls):
      raise _Refused('no launch was seen (nothing to replay)')

    # what follows the chunk: input payloads / masks (address ranges), upload fences (exact handles), gather plan variants
    slots, slot_of = [], {}
    in_ranges, fence_slots = [], {}
    for role, da in self.arrays.items():
      for kind, key, addr, nbytes, fence in input_ranges(da):
        if nbytes:
          in_ranges.append((addr, nbytes, (role, kind, key)))
        if fence:
          fence_slots[fence] = (role, 'fence', key)
    gather_addr = {g['addr']: i for i, g in enumerate(self.gathers)}
    for g in self.gathers:  # whose time labels does this plan follow?  (statistics re-wrap their inputs: match the payload)
      g['p_role'] = next((role for role, da in self.arrays.items() if da is g['p'] or da.data is g['p'].data), None)
      if g['p_role'] is None:
        raise _Refused('a gather table follows predictions that are not among the chunk\'s inputs')

    def slot_index(tag):
      if tag not in slot_of:
        slot_of[tag] = len(slots)
        slots.append(tag)
      return slot_of[tag]

    # memory the record owns or keeps alive
    owned, seen = [], {}
    from weatherbenchx_amd import engine  # pylint: disable=g-import-not-at-top
    _ranges_of(self.kept, owned, seen)
    _ranges_of(self.pinned, owned, seen)
    _ranges_of(keepalive, owned, seen)
    _ranges_of(list(engine._scratch_bufs.values()), owned, seen)  # pylint: disable=protected-access
    _ranges_of([b.dev for b in self.acc.blocks], owned, seen)
    handles = set()
    ctxs = {}
    for ctx in engine.known_contexts():
      handles.add(int(ctx.handle.value or 0))
      ctxs[int(ctx.handle.value or 0)] = ctx
    owned.sort()
    starts = [a for a, _ in owned]
    import bisect  # pylint: disable=g-import-not-at-top

    def is_owned(v):
      i = bisect.bisect_right(starts, v) - 1
      while i >= 0 and i >= bisect.bisect_right(starts, v) - 8:  # (ranges may nest: look a few entries back)
        a, n = owned[i]
        if a <= v < a + n:
          return True
        i -= 1
      return False

    relocs, used_ctx = [], {}
    for ci, (name, fn, vals, isptr) in enumerate(calls):
      for ai, (v, p) in enumerate(zip(vals, isptr)):
        if not p or v == 0:
          continue
        if ai == 0 and v in handles:
          used_ctx[v] = ctxs[v]
          continue
        if v in gather_addr:
          relocs.append((ci, ai, slot_index(('gather', gather_addr[v])), 0))
          continue
        if name == 'wbx_ctx_wait_fence' and ai == 1:
          tag = fence_slots.get(v)
          if tag is None:
            raise _Refused('the chunk waits on a fence that is not one of its inputs\' uploads')
          relocs.append((ci, ai, slot_index(tag), 0))
          continue
        hit = None
        for addr, nbytes, tag in in_ranges:
          if addr <= v < addr + nbytes:
            hit = (tag, v - addr)
            break
        if hit is not None:
          relocs.append((ci, ai, slot_index(hit[0]), hit[1]))
          continue
        if is_owned(v):
          continue
        raise _Refused(f'{name}: argument {ai} ({v:#x}) points at memory the record does not own')
    if not any(tag[1] in ('data', 'mask') for tag in slots if isinstance(tag, tuple) and len(tag) == 3):
      raise _Refused('no argument of any call points into the chunk\'s inputs')
    for path, *_ in self.bumps:
      if path not in self.acc._host_const:  # pylint: disable=protected-access
        raise _Refused('a constant term vanished')
    return ChunkRecord(calls, relocs, slots, self, list(used_ctx.values()), keepalive)


class ChunkRecord:
  """One recorded chunk: `replay(arrays)` runs the next one."""

  def __init__(self, calls, relocs, slots, recorder: ChunkRecorder, ctxs, keepalive):
    n = len(calls)
    self.calls = (_hip.CallStruct * n)()
    for i, (_, fn, vals, _) in enumerate(calls):
      self.calls[i].fn, self.calls[i].nargs = fn, len(vals)
      for j, v in enumerate(vals):
        self.calls[i].args[j] = v
    self.relocs = (_hip.RelocStruct * max(len(relocs), 1))()
    for i, (c, a, s, off) in enumerate(relocs):
      self.relocs[i].call, self.relocs[i].arg, self.relocs[i].slot, self.relocs[i].offset = c, a, s, off
    self.nrelocs = len(relocs)
    self.slot_tags = list(slots)
    self.slots = (C.c_uint64 * max(len(slots), 1))()
    self.names = [c[0] for c in calls]
    # (what a gather plan needs to be rebuilt for another chunk: NOT the recorded chunk's predictions -- they were only there to
    #  find `p_role`, and with their staged device copies they would pin a chunk of inputs in HBM for as long as the record lives)
    self.gathers = [{k: v for k, v in g.items() if k != 'p'} for g in recorder.gathers]
    recorder.gathers = []
    recorder.arrays = {}
    self.bumps = list(recorder.bumps)
    self.acc = recorder.acc
    self.ctxs = ctxs
    self.keep = (recorder.kept, recorder.pinned, keepalive)
    self.lib = _hip.load_library()
    self.replays = 0
    self.generation = _GENERATION[0]

  def replay(self, arrays) -> list | None:
    """Runs the chunk whose inputs are `arrays` ({role: DataArray}, the recorder's roles).  -> the fences that cover its kernels
    (one per context that got work), or None when this chunk's inputs cannot be addressed the way the recorded ones were (the
    caller then takes the ordinary path)."""
    if self.generation != _GENERATION[0]:
      return None
    from weatherbenchx_amd import engine  # pylint: disable=g-import-not-at-top
    found = {}
    torch_seen = None
    for i, tag in enumerate(self.slot_tags):
      if tag[0] == 'gather':
        continue
      role, kind, key = tag
      da = arrays.get(role)
      if da is None:
        return None
      hit = found.get(role)
      if hit is None:
        hit = found[role] = {(k, c): (addr, fence) for k, c, addr, _, fence in input_ranges(da)}
        if any(c == 'torch' for _, c in hit):
          torch_seen = da
      if kind == 'fence':
        got = hit.get(('data', key))
        if got is None or not got[1]:
          return None
        self.slots[i] = got[1]
      else:
        got = hit.get((kind, key))
        if got is None:
          return None
        self.slots[i] = got[0]
    keep_variants = []
    for i, tag in enumerate(self.slot_tags):
      if tag[0] != 'gather':
        continue
      g = self.gathers[tag[1]]
      p_new = arrays.get(g['p_role'])
      if p_new is None:
        return None
      dplan = g['regather'](p_new, g)
      if dplan is None:
        return None
      keep_variants.append(dplan)
      self.slots[i] = C.addressof(dplan.struct)
    if torch_seen is not None:
      engine._sync_torch_producers([torch_seen.data])  # pylint: disable=protected-access
    _hip.check(self.lib.wbx_chunk_replay(C.addressof(self.calls), len(self.calls), C.addressof(self.relocs), self.nrelocs,
                                         C.addressof(self.slots), len(self.slot_tags)), 'wbx_chunk_replay')
    acc = self.acc
    acc.multi = True
    for path, index, coeff in self.bumps:
      acc._host_const[path][index][1] += coeff  # pylint: disable=protected-access
    self.replays += 1
    STATS['replayed'] += 1
    fences = [ctx.fence() for ctx in self.ctxs]
    return [engine.ResultFence(fences, [arrays, keep_variants])]


_GENERATION = [0]


def invalidate_all():
  """Every record made so far is dead (the caches that own what they point at were cleared)."""
  _GENERATION[0] += 1
