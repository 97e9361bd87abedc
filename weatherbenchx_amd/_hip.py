"""ctypes binding of libwbx_hip.so (the C ABI declared in include/wbx.h).

There is NO CPU fallback: if the shared library is missing, or no gfx950 device is visible, every
compute entry point raises `WbxUnavailableError`.  Host-side planning (planner.py) never needs it.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_LIB_NAME = 'libwbx_hip.so'
_LIB_PATH = os.environ.get('WBX_LIBRARY_PATH') or os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)

MAX_INPUTS = 4
F32, F64 = 0, 1
DET3, DET6, PASS1 = 0, 1, 2
DET_LANES = {DET3: 3, DET6: 6, PASS1: 1}
DET_INPUTS = {DET3: 2, DET6: 3, PASS1: 1}
CAT_EXCEED, CAT_RANK = 0, 1
ENS_LANES = 5
ENS2_LANES = 2  # wbx_ens2_partial: skill over (prediction, target) member pairs, unbiased MSE with both ensembles' counts
ENS_SORT, ENS_PAIRWISE = 0, 1
COMM_ID_BYTES = 128
FLAG_MASKED, FLAG_SKIPNA, FLAG_FAIR, FLAG_SKIPNA_ENS = 1, 2, 4, 8
BINNED_W_ON_X, BINNED_WT_X_ONLY, BINNED_WT_ROW_ONLY, BINNED_MASK_ON_W, BINNED_TWIN_MASK = 1, 2, 4, 8, 16  # `w_on_x` flags (WBX_BINNED_*)
BINNED_ACCUMULATE = 32  # out += result: the launch adds into a chunk loop's accumulator itself (ABI 12)

# every symbol include/wbx.h declares (tests check the built library exports all of them)
EXPORTED_SYMBOLS = (
    'wbx_abi_version', 'wbx_last_error', 'wbx_device_count', 'wbx_ctx_create', 'wbx_ctx_destroy',
    'wbx_ctx_synchronize', 'wbx_ctx_device_name', 'wbx_malloc', 'wbx_free', 'wbx_memcpy_h2d',
    'wbx_memcpy_d2h', 'wbx_memset', 'wbx_timer_start', 'wbx_timer_stop', 'wbx_mark', 'wbx_mark_elapsed', 'wbx_marks_reset', 'wbx_s1_partial_len',
    'wbx_det_partial', 'wbx_ens_partial', 'wbx_contract', 'wbx_contract_bits', 'wbx_det_binned', 'wbx_cat_partial', 'wbx_det_map', 'wbx_ens_map',
    'wbx_zonal_spectrum', 'wbx_zonal_spectrum_slabs', 'wbx_host_alloc', 'wbx_host_free', 'wbx_memcpy_d2h_async', 'wbx_fence_create',
    'wbx_fence_record', 'wbx_fence_wait', 'wbx_fence_destroy', 'wbx_ctx_wait_fence', 'wbx_memcpy_h2d_async',
    'wbx_memcpy_d2d', 'wbx_acc_add', 'wbx_notnan_mask', 'wbx_binned_atoms_size', 'wbx_binned_atoms',
    'wbx_comm_unique_id', 'wbx_comm_create', 'wbx_comm_destroy', 'wbx_comm_info', 'wbx_acc_allreduce', 'wbx_acc_read',
    'wbx_acc_reset', 'wbx_det_spectrum', 'wbx_det_spectrum_slabs', 'wbx_ens_binned', 'wbx_ens_binned_atoms_size', 'wbx_ens_binned_atoms',
    'wbx_ens2_partial', 'wbx_cat_exceed_field', 'wbx_chunk_replay', 'wbx_host_transpose', 'wbx_clock_probe', 'wbx_det_spectrum_folded',
)

# wbx_fn (include/wbx.h): the entry points a chunk record may hold
FN_IDS = {'wbx_det_partial': 1, 'wbx_ens_partial': 2, 'wbx_ens2_partial': 3, 'wbx_cat_partial': 4, 'wbx_cat_exceed_field': 5,
          'wbx_contract': 6, 'wbx_contract_bits': 7, 'wbx_det_binned': 8, 'wbx_ens_binned': 9, 'wbx_zonal_spectrum': 10,
          'wbx_zonal_spectrum_slabs': 11, 'wbx_det_spectrum': 12, 'wbx_det_spectrum_slabs': 13, 'wbx_acc_add': 14,
          'wbx_memset': 15, 'wbx_memcpy_d2d': 16, 'wbx_ctx_wait_fence': 17, 'wbx_fence_record': 18, 'wbx_det_spectrum_folded': 19}
CALL_MAX_ARGS = 20
# pure queries: they touch neither a stream nor memory, a record simply leaves them out
QUERY_FNS = frozenset({'wbx_s1_partial_len', 'wbx_binned_atoms_size', 'wbx_ens_binned_atoms_size', 'wbx_last_error',
                       'wbx_abi_version', 'wbx_device_count', 'wbx_comm_info'})
# the fences a chunk leaves behind for the HOST (when may its inputs be let go of): a replayed chunk records its own
# (replay.ChunkRecord.replay); waiting for / dropping fences of earlier chunks is the loop's business, not the chunk's
HOST_FENCE_FNS = frozenset({'wbx_fence_create', 'wbx_fence_record', 'wbx_fence_wait', 'wbx_fence_destroy'})


class CallStruct(C.Structure):  # wbx_call
  _fields_ = [('fn', C.c_int32), ('nargs', C.c_int32), ('args', C.c_uint64 * CALL_MAX_ARGS)]


class RelocStruct(C.Structure):  # wbx_reloc
  _fields_ = [('call', C.c_int32), ('arg', C.c_int32), ('slot', C.c_int32), ('reserved_', C.c_int32), ('offset', C.c_int64)]


# The chunk recorder of the calling thread's chunk loop (engine.ChunkRecorder), or None.  While one is set, every call the
# recording thread makes into the library is noted with its arguments, and every device block it takes from a context's pool is
# pinned to the recording (the recorded pointer must stay the record's own).
RECORDER = None
PROTOS: dict = {}


class _RecordingLib:
  """The loaded library behind one level of indirection: `lib.wbx_xyz(...)` calls straight through and, while a chunk is being
  recorded on this thread, leaves a note (name, arguments) with the recorder."""

  def __init__(self, lib):
    self.__dict__['_lib'] = lib

  def __getattr__(self, name):
    fn = getattr(self._lib, name)
    if name not in PROTOS:
      return fn

    def call(*args, _fn=fn, _name=name):
      rec = RECORDER
      if rec is not None and rec.thread == threading.get_ident():
        rec.note(_name, args)
      return _fn(*args)
    self.__dict__[name] = call
    return call



class WbxUnavailableError(RuntimeError):
  """libwbx_hip.so missing or no MI355X visible -- the product path has no CPU fallback."""


class WbxError(RuntimeError):
  pass


class S1PlanStruct(C.Structure):
  _fields_ = [
      ('nkey', C.c_int64), ('ndepth', C.c_int64), ('nx', C.c_int64),
      ('x_kept', C.c_int32), ('nchunk', C.c_int32), ('depth_chunk', C.c_int64),
      ('xstride', C.c_int64 * MAX_INPUTS),
      ('key_off', C.c_void_p * MAX_INPUTS),
      ('depth_off', C.c_void_p * MAX_INPUTS),
      ('gather_key', C.c_void_p), ('gather_depth', C.c_void_p), ('gather_tab', C.c_void_p),
      ('n_gather_depth', C.c_int32), ('flags', C.c_uint32),
      ('block_threads', C.c_int32), ('vec', C.c_int32), ('plane_rows', C.c_int32), ('reserved_', C.c_int32), ('x_weights', C.c_void_p),
  ]


class S2PlanStruct(C.Structure):
  _fields_ = [(n, C.c_int64) for n in ('nA', 'nBk', 'nBr', 'nchunk', 'nlane', 'nj', 'nbin')] + [('sum_j', C.c_int32)]


_lib = None
_lib_lock = threading.Lock()


def lib_path() -> str:
  return _LIB_PATH


def load_library():
  """Loads libwbx_hip.so (no device needed) and declares prototypes."""
  global _lib
  with _lib_lock:
    if _lib is not None:
      return _lib
    if not os.path.exists(_LIB_PATH):
      raise WbxUnavailableError(
          f'{_LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
          '(make -C weatherbenchx_amd/csrc). There is no CPU fallback.')
    try:
      lib = C.CDLL(_LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # missing ROCm runtime etc.
      raise WbxUnavailableError(f'cannot load {_LIB_PATH}: {e}') from e
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.wbx_last_error.restype = C.c_char_p
    lib.wbx_abi_version.restype = i32
    protos = {
        'wbx_device_count': [C.POINTER(i32)],
        'wbx_ctx_create': [i32, vp, C.POINTER(vp)],
        'wbx_ctx_destroy': [vp],
        'wbx_ctx_synchronize': [vp],
        'wbx_ctx_device_name': [vp, C.c_char_p, C.c_size_t],
        'wbx_malloc': [vp, C.c_size_t, C.POINTER(vp)],
        'wbx_free': [vp, vp],
        'wbx_memcpy_h2d': [vp, vp, vp, C.c_size_t],
        'wbx_memcpy_d2h': [vp, vp, vp, C.c_size_t],
        'wbx_memset': [vp, vp, i32, C.c_size_t],
        'wbx_host_alloc': [vp, C.c_size_t, C.POINTER(vp)],
        'wbx_host_free': [vp, vp],
        'wbx_memcpy_d2h_async': [vp, vp, vp, C.c_size_t],
        'wbx_fence_create': [vp, C.POINTER(vp)],
        'wbx_fence_record': [vp, vp],
        'wbx_fence_wait': [vp],
        'wbx_fence_destroy': [vp],
        'wbx_ctx_wait_fence': [vp, vp],
        'wbx_memcpy_h2d_async': [vp, vp, vp, C.c_size_t],
        'wbx_memcpy_d2d': [vp, vp, vp, C.c_size_t],
        'wbx_acc_add': [vp, vp, vp, i64, i32],
        'wbx_notnan_mask': [vp, vp, i32, i64, vp],
        'wbx_comm_unique_id': [vp],
        'wbx_comm_create': [vp, vp, C.c_int32, C.c_int32, C.POINTER(vp)],
        'wbx_comm_destroy': [vp],
        'wbx_comm_info': [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(i64)],
        'wbx_acc_allreduce': [vp, vp, vp, i64],
        'wbx_acc_read': [vp, vp, i64, vp],
        'wbx_acc_reset': [vp, vp, i64],
        'wbx_timer_start': [vp],
        'wbx_timer_stop': [vp, C.POINTER(C.c_float)],
        'wbx_mark': [vp, C.POINTER(i32)],
        'wbx_mark_elapsed': [vp, i32, i32, C.POINTER(C.c_float)],
        'wbx_marks_reset': [vp],
        'wbx_s1_partial_len': [C.POINTER(S1PlanStruct), i32, C.POINTER(i64)],
        'wbx_det_partial': [vp, C.POINTER(S1PlanStruct), i32, i32, vp, vp, vp, vp, vp],
        'wbx_ens_partial': [vp, C.POINTER(S1PlanStruct), i32, i32, i64, i32, vp, vp, vp, vp],
        'wbx_contract': [vp, C.POINTER(S2PlanStruct), vp, vp, vp],
        'wbx_contract_bits': [vp, C.POINTER(S2PlanStruct), vp, vp, vp, vp],
        'wbx_det_binned': [vp, C.POINTER(S1PlanStruct), i32, i32, vp, vp, vp, vp, vp, vp, i64, i64, i64, C.c_int32, C.c_int32, vp, vp],
        'wbx_binned_atoms_size': [C.POINTER(S1PlanStruct), i64, i64, i64, C.c_int32, C.POINTER(i64)],
        'wbx_binned_atoms': [vp, C.POINTER(S1PlanStruct), i64, i64, i64, C.c_int32, vp, vp],
        'wbx_ens_binned': [vp, C.POINTER(S1PlanStruct), i32, i32, i64, i32, vp, vp, vp, vp, vp, i64, i64, i64, C.c_int32, C.c_int32, vp, vp],
        'wbx_ens_binned_atoms_size': [C.POINTER(S1PlanStruct), i64, i64, i64, C.c_int32, C.POINTER(i64)],
        'wbx_ens_binned_atoms': [vp, C.POINTER(S1PlanStruct), i64, i64, i64, C.c_int32, vp, vp, C.POINTER(i64)],
        'wbx_ens2_partial': [vp, C.POINTER(S1PlanStruct), i32, i32, i64, i32, i64, vp, vp, vp, vp],
        'wbx_cat_exceed_field': [vp, C.POINTER(S1PlanStruct), i32, i32, i32, i64, vp, vp, vp, i64, vp, vp],
        'wbx_cat_partial': [vp, C.POINTER(S1PlanStruct), i32, i32, i32, i32, i64, vp, vp, vp, vp, vp],
        'wbx_det_map': [vp, C.POINTER(S1PlanStruct), i32, i32, i32, vp, vp, vp, vp],
        'wbx_ens_map': [vp, C.POINTER(S1PlanStruct), i32, i32, i64, i32, i32, vp, vp, vp],
        'wbx_zonal_spectrum': [vp, vp, i64, i64, i64, C.c_int32, vp, vp, C.c_int32, C.c_int32, vp],
        'wbx_det_spectrum': [vp, C.POINTER(S1PlanStruct), i32, i32, vp, vp, vp, vp, vp, i64, vp, vp, vp],
        'wbx_det_spectrum_slabs': [vp, C.POINTER(S1PlanStruct), i32, i32, vp, vp, vp, i64, vp, vp, i64, vp, vp, vp],
        'wbx_det_spectrum_folded': [vp, C.POINTER(S1PlanStruct), i32, i32, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp],
        'wbx_zonal_spectrum_slabs': [vp, vp, i64, i64, i64, i64, vp, C.c_int32, vp, vp, C.c_int32, C.c_int32, vp],
    }
    protos['wbx_chunk_replay'] = [vp, i32, vp, i32, vp, i32]
    protos['wbx_host_transpose'] = [vp, vp, i64, i64, i64, C.c_int32]
    protos['wbx_clock_probe'] = [vp, C.c_int32, C.POINTER(C.c_double)]
    for name, argtypes in protos.items():
      fn = getattr(lib, name)
      fn.argtypes = argtypes
      fn.restype = i32
    PROTOS.update(protos)
    _lib = _RecordingLib(lib)
    return _lib


def check(rc: int, what: str = ''):
  if rc != 0:
    msg = load_library().wbx_last_error().decode(errors='replace')
    if rc == -3:
      raise WbxUnavailableError(f'{what}: {msg}')
    raise WbxError(f'{what} failed (status {rc}): {msg}')


def device_count() -> int:
  lib = load_library()
  n = C.c_int(0)
  rc = lib.wbx_device_count(C.byref(n))
  return int(n.value) if rc == 0 else 0


def is_available() -> bool:
  try:
    return device_count() > 0
  except WbxUnavailableError:
    return False


class DeviceBuffer:
  """Device allocation owned by a Context.  On garbage collection the memory goes back to the context's free list
  (same-size chunks arrive over and over in the chunk loop, and hipFree synchronises the whole device), up to
  DEVICE_POOL_LIMIT_BYTES cached per context; beyond that it is freed."""

  def __init__(self, ctx: 'Context', nbytes: int):
    self.ctx = ctx
    self.nbytes = int(nbytes)
    with ctx._pool_lock:  # pylint: disable=protected-access
      free = ctx._dev_free.get(self.nbytes)  # pylint: disable=protected-access
      ptr = free.pop() if free else 0
      if ptr:
        ctx._dev_cached -= self.nbytes  # pylint: disable=protected-access
    rec = RECORDER
    if rec is not None and rec.thread == threading.get_ident():
      rec.pinned.append(self)  # (a block handed out while a chunk is being recorded stays with the record)
    if not ptr:
      p = C.c_void_p(0)
      rc = ctx.lib.wbx_malloc(ctx.handle, self.nbytes, C.byref(p))
      if rc != 0 and ctx._dev_cached:  # pylint: disable=protected-access
        ctx.release_pool()  # out of memory with idle cached blocks: give them back and retry
        rc = ctx.lib.wbx_malloc(ctx.handle, self.nbytes, C.byref(p))
      check(rc, 'wbx_malloc')
      ptr = p.value or 0
    self.ptr = ptr

  def __del__(self):
    try:
      ctx = self.ctx
      if getattr(self, 'ptr', 0) and ctx.handle:
        with ctx._pool_lock:  # pylint: disable=protected-access
          if ctx._dev_cached + self.nbytes <= DEVICE_POOL_LIMIT_BYTES:  # pylint: disable=protected-access
            ctx._dev_free.setdefault(self.nbytes, []).append(self.ptr)  # pylint: disable=protected-access
            ctx._dev_cached += self.nbytes  # pylint: disable=protected-access
            self.ptr = 0
        if self.ptr:
          ctx.lib.wbx_free(ctx.handle, C.c_void_p(self.ptr))
          self.ptr = 0
    except Exception:  # pylint: disable=broad-except
      pass


# Upper bound on idle device memory a context keeps for reuse (MI355X: 288 GB of HBM3E).
DEVICE_POOL_LIMIT_BYTES = int(os.environ.get('WBX_DEVICE_POOL_BYTES', 32 << 30))


class PinnedBlock:
  """Page-locked host memory from the context's pool, exposed through the array interface: `np.asarray(block)` is a
  view (float64 unless `typestr` says otherwise) that keeps the block alive; when the last view dies the memory goes
  back to the pool."""

  def __init__(self, ctx: 'Context', ptr: int, capacity: int, nbytes: int, typestr: str = '<f8', itemsize: int = 8):
    self._ctx, self._ptr, self._capacity = ctx, ptr, capacity
    self.__array_interface__ = {'shape': (nbytes // itemsize,), 'typestr': typestr, 'data': (ptr, False), 'version': 3}

  @property
  def ptr(self) -> int:
    return self._ptr

  def __del__(self):
    try:
      self._ctx._pinned_free.setdefault(self._capacity, []).append(self._ptr)  # pylint: disable=protected-access
    except Exception:  # pylint: disable=broad-except
      pass


def _pinned_block_of(arr):
  base = arr
  while isinstance(base, np.ndarray):
    base = base.base
  return base if isinstance(base, PinnedBlock) else None


def is_pinned(arr) -> bool:
  """True for numpy arrays (views included) whose memory came from Context.pinned_empty / the read-back pool."""
  return _pinned_block_of(arr) is not None


def is_loader_pinned(arr) -> bool:
  """True only for arrays handed out by Context.pinned_empty (pipeline.pinned_empty) -- the arrays whose contract says
  "filled once, then left alone": those may be uploaded asynchronously.  Views of the read-back pool (results of earlier
  reductions) are page-locked too but are recycled by the pool, so they take the synchronous path."""
  block = _pinned_block_of(arr)
  return block is not None and getattr(block, 'for_loader', False)


class Fence:
  """hipEvent recorded on the context stream: `wait()` blocks the host until everything enqueued before it is done."""

  def __init__(self, ctx: 'Context'):
    self._lib = ctx.lib
    h = C.c_void_p(0)
    check(ctx.lib.wbx_fence_create(ctx.handle, C.byref(h)), 'wbx_fence_create')
    self._h = h
    check(ctx.lib.wbx_fence_record(ctx.handle, h), 'wbx_fence_record')
    self._done = False

  def wait(self):
    if not self._done:
      check(self._lib.wbx_fence_wait(self._h), 'wbx_fence_wait')
      self._done = True

  def __del__(self):
    try:
      if self._h:
        self.wait()  # (an event must not be destroyed while a stream may still be told to wait on it)
        self._lib.wbx_fence_destroy(self._h)
        self._h = None
    except Exception:  # pylint: disable=broad-except
      pass


ALL_CONTEXTS = None  # weakref.WeakSet of every live Context (replay.py maps the handles in a recording back to them)


class Context:
  """One device + one HIP stream (wbx_ctx).  `torch_stream=True` adopts torch's current stream so
  launches order with tensors produced by torch on that stream."""

  def __init__(self, device_id: int = 0, stream_ptr: int | None = None):
    global ALL_CONTEXTS
    if ALL_CONTEXTS is None:
      import weakref  # pylint: disable=g-import-not-at-top
      ALL_CONTEXTS = weakref.WeakSet()
    ALL_CONTEXTS.add(self)
    self.lib = load_library()
    h = C.c_void_p(0)
    check(self.lib.wbx_ctx_create(int(device_id), C.c_void_p(stream_ptr or 0), C.byref(h)), 'wbx_ctx_create')
    self.handle = h
    self.device_id = int(device_id)
    self._pinned_free: dict[int, list[int]] = {}  # capacity -> free page-locked blocks
    self._dev_free: dict[int, list[int]] = {}     # nbytes -> idle device blocks (see DeviceBuffer)
    self._dev_cached = 0
    self._pool_lock = threading.Lock()

  def release_pool(self):
    """hipFree every idle cached device block."""
    with self._pool_lock:
      blocks, self._dev_free, self._dev_cached = self._dev_free, {}, 0
    for ptrs in blocks.values():
      for ptr in ptrs:
        self.lib.wbx_free(self.handle, C.c_void_p(ptr))

  def close(self):
    if getattr(self, 'handle', None):
      self.release_pool()
      for blocks in getattr(self, '_pinned_free', {}).values():
        for ptr in blocks:
          self.lib.wbx_host_free(self.handle, C.c_void_p(ptr))
      self._pinned_free = {}
      self.lib.wbx_ctx_destroy(self.handle)
      self.handle = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  def synchronize(self):
    check(self.lib.wbx_ctx_synchronize(self.handle), 'wbx_ctx_synchronize')

  def device_name(self) -> str:
    buf = C.create_string_buffer(256)
    check(self.lib.wbx_ctx_device_name(self.handle, buf, 256), 'wbx_ctx_device_name')
    return buf.value.decode()

  # memory ------------------------------------------------------------------------------------
  def alloc(self, nbytes: int) -> DeviceBuffer:
    return DeviceBuffer(self, max(int(nbytes), 8))

  def upload(self, arr: np.ndarray) -> DeviceBuffer:
    arr = np.ascontiguousarray(arr)
    buf = self.alloc(arr.nbytes)
    if arr.nbytes:
      check(self.lib.wbx_memcpy_h2d(self.handle, C.c_void_p(buf.ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes),
            'wbx_memcpy_h2d')
    return buf

  def download(self, ptr: int, shape, dtype=np.float64) -> np.ndarray:
    out = np.empty(shape, dtype=dtype)
    if out.nbytes:
      check(self.lib.wbx_memcpy_d2h(self.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes),
            'wbx_memcpy_d2h')
    return out

  def _pinned_block(self, nbytes: int, typestr: str = '<f8', itemsize: int = 8) -> PinnedBlock:
    capacity = max(4096, 1 << (max(nbytes, 1) - 1).bit_length())
    with self._pool_lock:
      free = self._pinned_free.get(capacity)
      hptr = free.pop() if free else None
    if hptr is None:
      p = C.c_void_p(0)
      check(self.lib.wbx_host_alloc(self.handle, capacity, C.byref(p)), 'wbx_host_alloc')
      hptr = p.value
    return PinnedBlock(self, hptr, capacity, nbytes, typestr, itemsize)

  def pinned_empty(self, shape, dtype=np.float32) -> np.ndarray:
    """Uninitialised page-locked array (wbx_host_alloc, pooled): a loader that decodes its chunk straight into it gets
    a pure-DMA upload (`upload_async`) with no bounce copy through the runtime's staging buffers."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64))
    block = self._pinned_block(n * dt.itemsize, dt.str, dt.itemsize)
    block.for_loader = True  # (see is_loader_pinned)
    return np.asarray(block).reshape(shape)

  def upload_async(self, arr: np.ndarray) -> DeviceBuffer:
    """Enqueues the upload of a page-locked, C-contiguous array on the context stream and returns at once; the caller
    orders consumers with a Fence recorded afterwards and keeps `arr` untouched until that fence has been reached."""
    if not (is_pinned(arr) and arr.flags['C_CONTIGUOUS']):
      raise ValueError('upload_async needs a C-contiguous array from Context.pinned_empty')
    buf = self.alloc(arr.nbytes)
    if arr.nbytes:
      check(self.lib.wbx_memcpy_h2d_async(self.handle, C.c_void_p(buf.ptr), C.c_void_p(arr.ctypes.data), arr.nbytes),
            'wbx_memcpy_h2d_async')
    return buf

  def wait_fence(self, fence: 'Fence'):
    """Work enqueued on this context from now on starts after `fence` (recorded on any context): no host blocking."""
    check(self.lib.wbx_ctx_wait_fence(self.handle, fence._h), 'wbx_ctx_wait_fence')  # pylint: disable=protected-access

  def download_async(self, ptr: int, shape) -> np.ndarray:
    """Enqueues the read-back of float64 `shape` into pooled page-locked memory and returns the (not yet valid) view;
    the caller orders its reads with a Fence recorded afterwards."""
    n = int(np.prod(shape, dtype=np.int64))
    nbytes = n * 8
    block = self._pinned_block(nbytes)
    if nbytes:
      check(self.lib.wbx_memcpy_d2h_async(self.handle, C.c_void_p(block.ptr), C.c_void_p(ptr), nbytes), 'wbx_memcpy_d2h_async')
    return np.asarray(block).reshape(shape)

  def pinned_result(self, shape) -> np.ndarray:
    """Pooled page-locked float64 array a kernel writes its result into directly (device-visible: wbx_host_alloc);
    the caller orders its reads with a Fence recorded after that kernel."""
    n = int(np.prod(shape, dtype=np.int64))
    return np.asarray(self._pinned_block(n * 8)).reshape(shape)

  def fence(self) -> Fence:
    return Fence(self)

  def timer_start(self):
    check(self.lib.wbx_timer_start(self.handle), 'wbx_timer_start')

  def timer_stop(self) -> float:
    ms = C.c_float(0)
    check(self.lib.wbx_timer_stop(self.handle, C.byref(ms)), 'wbx_timer_stop')
    return float(ms.value)

  def mark(self) -> int:
    """A timing event on this context's stream; nothing waits (read with mark_elapsed once the work is known to be done)."""
    i = C.c_int(0)
    check(self.lib.wbx_mark(self.handle, C.byref(i)), 'wbx_mark')
    return int(i.value)

  def mark_elapsed(self, i0: int, i1: int) -> float:
    ms = C.c_float(0)
    check(self.lib.wbx_mark_elapsed(self.handle, int(i0), int(i1), C.byref(ms)), 'wbx_mark_elapsed')
    return float(ms.value)

  def clock_probe(self, blocks: int = 2048) -> float:
    """Shader clock in MHz with `blocks` x 256 threads of fp32 FMAs running (wbx_clock_probe)."""
    mhz = C.c_double(0)
    check(self.lib.wbx_clock_probe(self.handle, int(blocks), C.byref(mhz)), 'wbx_clock_probe')
    return float(mhz.value)

  def marks_reset(self):
    check(self.lib.wbx_marks_reset(self.handle), 'wbx_marks_reset')


_default_ctx: dict[int, Context] = {}


def default_context(device_id: int | None = None) -> Context:
  """Process-wide context per device.  Uses LOCAL_RANK as the default device (one process per GPU)."""
  if device_id is None:
    device_id = int(os.environ.get('WBX_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    n = device_count()
    if n == 0:
      raise WbxUnavailableError('no HIP device visible (libwbx_hip has no CPU path)')
    device_id %= n
  if device_id not in _default_ctx:
    _default_ctx[device_id] = Context(device_id)
  return _default_ctx[device_id]
