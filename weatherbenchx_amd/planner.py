"""Host-side planning of the two-stage reduction (pure numpy; no device needed).

A statistic's index space (named dims) is split for stage 1 (csrc/wbx_s1.hpp) into
  x      the innermost dim of the predictions' memory layout (coalescing),
  key    dims that must survive stage 1: dims the caller keeps, plus every dim the weights or
         bin masks depend on (their product W is applied in stage 2),
  depth  the remaining reduced dims: summed inside stage 1.
Key dims are ordered A (kept, W-independent), Bk (kept, W-dependent), Br (reduced, W-dependent) so
stage 2 (csrc/wbx_s2.hip) sees partial[A][Bk][Br][chunk][lane][j] and W[Bk][Br][j][bin].

This replaces the dimension bookkeeping xarray's `xr.dot` does inside
weatherbenchX/aggregation.py:297-335, for arbitrary loader dim orders
(data_loaders/xarray_loaders.py:185-188; real data is latitude-fastest).
"""
from __future__ import annotations

import dataclasses
import os
from typing import Sequence

import numpy as np

MAX_INPUTS = 4
TARGET_BLOCKS = 4096  # >> 256 CUs * resident blocks, so the tail is short
# geometry of the flat one-point-per-lane sweep (s1_xf1_kernel): threads per block, fewest elements per block
FLAT1_THREADS = int(os.environ.get('WBX_FLAT1_THREADS', '256'))
FLAT1_MIN_ELEMENTS = int(os.environ.get('WBX_FLAT1_MIN_ELEMENTS', '2816'))
# ... and of its one-wave flavour (ens_pipe_kernel<.., FLAT>): elements per block, unless that leaves fewer blocks than this
FLAT64_MIN_ELEMENTS = int(os.environ.get('WBX_FLAT64_MIN_ELEMENTS', '2816'))
FLAT64_MIN_BLOCKS = int(os.environ.get('WBX_FLAT64_MIN_BLOCKS', '18432'))


@dataclasses.dataclass
class InputLayout:
  """Element strides of one input along the statistic's dims (0 / missing = broadcast)."""
  strides: dict  # dim -> element stride
  itemsize: int = 4
  base_alignment: int = 16  # bytes the base pointer is aligned to

  def stride(self, dim) -> int:
    return int(self.strides.get(dim, 0))


@dataclasses.dataclass
class GatherSpec:
  """Indirect addressing of input 2 (climatology): element offset table over `dims`.
  metrics/base.py:397-403 `.sel(dayofyear=..., hour=...)` becomes table[init, lead] * slice_stride."""
  dims: tuple
  table: np.ndarray  # int64, shape = sizes of dims


@dataclasses.dataclass
class S1Plan:
  dims: tuple           # statistic dims (ordered)
  sizes: dict
  x_dim: object         # may be None (0-d statistic)
  x_kept: bool
  sum_j: bool           # stage 2 sums over x
  a_dims: tuple
  bk_dims: tuple
  br_dims: tuple
  depth_dims: tuple
  nkey: int
  ndepth: int
  nx: int
  nchunk: int
  depth_chunk: int
  xstride: list
  key_off: list         # per input: int64[nkey] or None
  depth_off: list       # per input: int64[ndepth] or None
  gather_key: np.ndarray | None
  gather_depth: np.ndarray | None
  gather_tab: np.ndarray | None
  n_gather_depth: int
  flags: int
  block_threads: int
  vec: int
  plane_rows: int = 0
  x_weights: np.ndarray | None = None  # float64[nx]: weights on the innermost dim folded into stage 1 (x is summed there)

  @property
  def nj(self) -> int:
    return self.nx if self.x_kept else 1

  @property
  def key_dims(self) -> tuple:
    return self.a_dims + self.bk_dims + self.br_dims

  def n(self, dims) -> int:
    return int(np.prod([self.sizes[d] for d in dims], dtype=np.int64)) if dims else 1

  def partial_shape(self, nlanes_total: int) -> tuple:
    return (self.n(self.a_dims), self.n(self.bk_dims), self.n(self.br_dims), self.nchunk, nlanes_total, self.nj)

  def reduced_count_per_partial(self) -> int:
    """Number of statistic elements folded into one stage-1 partial (all chunks together)."""
    return self.ndepth * (1 if self.x_kept else self.nx)


def _offset_table(dims: Sequence, sizes: dict, layout: InputLayout) -> np.ndarray | None:
  """Flattened (C order over `dims`) element offsets; None when identically zero."""
  if not dims:
    return None
  if all(layout.stride(d) == 0 for d in dims):
    return None
  off = np.zeros((), dtype=np.int64)
  for d in dims:
    off = off[..., None] + np.arange(sizes[d], dtype=np.int64) * layout.stride(d)
  return np.ascontiguousarray(off.reshape(-1))


def _gather_index(dims: Sequence, sizes: dict, gdims: Sequence) -> tuple[np.ndarray | None, int]:
  """Index into the gather-table axes restricted to `gdims` for each flattened position of `dims`."""
  sub = [d for d in dims if d in gdims]
  n = int(np.prod([sizes[d] for d in sub], dtype=np.int64)) if sub else 1
  if not sub:
    return None, 1
  idx = np.zeros((), dtype=np.int64)
  mult = 1
  mults = {}
  for d in reversed(sub):
    mults[d] = mult
    mult *= sizes[d]
  for d in dims:
    step = mults.get(d, 0)
    idx = idx[..., None] + np.arange(sizes[d], dtype=np.int64) * step
  return np.ascontiguousarray(idx.reshape(-1).astype(np.int32)), n


def gather_table(key_dims: Sequence, depth_dims: Sequence, gather: GatherSpec) -> np.ndarray:
  """`gather.table` flattened to [gather-key combos][gather-depth combos], the order stage 1 indexes it in.  It is the
  only table of a plan that follows the chunk's TIME LABELS (climatology alignment, deterministic.py:167-220): the engine
  swaps it into a cached plan instead of rebuilding the plan for every chunk."""
  kd = [d for d in key_dims if d in gather.dims]
  dd = [d for d in depth_dims if d in gather.dims]
  tab = np.asarray(gather.table, dtype=np.int64)
  tab = np.transpose(tab, [gather.dims.index(d) for d in kd + dd])
  return np.ascontiguousarray(tab.reshape(-1))


def choose_x_dim(dims: Sequence, sizes: dict, layout: InputLayout, exclude=()):
  """The contiguous-most dim of the predictions (stride 1 preferred)."""
  cands = [d for d in dims if d not in exclude and sizes[d] > 1 and layout.stride(d) != 0]
  if not cands:
    cands = [d for d in dims if d not in exclude]
    return cands[-1] if cands else None
  return min(cands, key=lambda d: (abs(layout.stride(d)), -list(dims).index(d)))


def build_s1_plan(dims: Sequence, sizes: dict, layouts: Sequence[InputLayout | None], reduce_dims,
                  wdep_dims=(), gather: GatherSpec | None = None, flags: int = 0, allow_vec4: bool = True,
                  target_blocks: int = TARGET_BLOCKS, force_x_dim=None, map_mode: bool = False,
                  fold_x: bool = False) -> S1Plan:
  """Plans stage 1.  `layouts[i]` is None for unused inputs.  `map_mode` keeps every dim (nchunk=1).  `fold_x`: the
  caller folds weights on the innermost dim into stage 1 (S1Plan.x_weights), so x is summed here; when the inner depth
  rows are contiguous the flat float4 sweep is planned (plane_rows = their count, see s1_xf_kernel); fold_x='point' / 'point64' plans
  the one-point-per-lane flavour of it (s1_xf1_kernel, ensemble statistics)."""
  dims = tuple(dims)
  sizes = {d: int(sizes[d]) for d in dims}
  reduce_set = set(reduce_dims) & set(dims)
  wdep = set(wdep_dims) & set(dims)
  lay0 = layouts[0]
  x_dim = force_x_dim if force_x_dim is not None else choose_x_dim(dims, sizes, lay0)
  nx = sizes[x_dim] if x_dim is not None else 1
  if map_mode:
    x_kept, sum_j = True, False
  else:
    x_kept = (x_dim is None) or (x_dim not in reduce_set) or (x_dim in wdep)
    sum_j = x_kept and x_dim is not None and x_dim in reduce_set
  others = [d for d in dims if d != x_dim]
  if map_mode:
    # out[key][depth][x] must be the C-order array over dims with x last: put all other dims in key
    a_dims, bk_dims, br_dims, depth_dims = tuple(others), (), (), ()
  else:
    a_dims = tuple(d for d in others if d not in reduce_set and d not in wdep)
    bk_dims = tuple(d for d in others if d not in reduce_set and d in wdep)
    br_dims = tuple(d for d in others if d in reduce_set and d in wdep)
    depth_dims = tuple(d for d in others if d in reduce_set and d not in wdep)
  key_dims = a_dims + bk_dims + br_dims
  nkey = int(np.prod([sizes[d] for d in key_dims], dtype=np.int64)) if key_dims else 1
  ndepth = int(np.prod([sizes[d] for d in depth_dims], dtype=np.int64)) if depth_dims else 1

  xstride, key_off, depth_off = [], [], []
  for lay in layouts:
    if lay is None:
      xstride.append(0)
      key_off.append(None)
      depth_off.append(None)
    else:
      xstride.append(lay.stride(x_dim) if x_dim is not None else 0)
      key_off.append(_offset_table(key_dims, sizes, lay))
      depth_off.append(_offset_table(depth_dims, sizes, lay))
  while len(xstride) < MAX_INPUTS:
    xstride.append(0)
    key_off.append(None)
    depth_off.append(None)

  gk = gd = gtab = None
  ngd = 1
  if gather is not None:
    if x_dim in gather.dims:
      raise ValueError(f'gather dim {x_dim!r} is the innermost dim; align the climatology on the host instead')
    gk, ngk = _gather_index(key_dims, sizes, gather.dims)
    gd, ngd = _gather_index(depth_dims, sizes, gather.dims)
    gtab = gather_table(key_dims, depth_dims, gather)

  # launch geometry
  block_threads = 256
  if x_kept:
    vec = 1
    nxtile = -(-nx // (block_threads * vec))
    blocks_per_chunk = nkey * nxtile
  else:
    blocks_per_chunk = nkey
  if map_mode:
    nchunk, depth_chunk = 1, max(ndepth, 1)
  else:
    nchunk = int(min(max(ndepth, 1), max(1, -(-target_blocks // max(blocks_per_chunk, 1)))))
    depth_chunk = -(-max(ndepth, 1) // nchunk)
    nchunk = -(-max(ndepth, 1) // depth_chunk)
  if not x_kept:
    rows = depth_chunk
    block_threads = 256 if rows >= 4 else (128 if rows >= 2 else 64)
    if nx <= 64 and rows < 4:
      block_threads = 64
    if nkey * nchunk >= 16384:
      # plenty of blocks already: one wave per block sweeps its rows without any block-level sync
      # (configs[1] with non-temporal loads: 64 / 128 / 256 threads -> 81.1 / 80.6 / 78.5 % of the HBM peak)
      block_threads = 64

  # 16 B per lane per load, only when every row of every streamed input starts 16-B aligned and nx % 4 == 0.
  # (Measured on MI355X: UNALIGNED dwordx4 on 721-long latitude rows runs 6.8 ms vs 4.8 ms for dword loads on
  # configs[1], so ragged / odd rows deliberately stay on the scalar path.)
  vec = 1
  if allow_vec4 and nx >= 4 and nx % 4 == 0 and x_dim is not None:
    ok = True
    for i, lay in enumerate(layouts[:4]):
      if lay is None:
        continue
      want_item = 1 if i == 3 else 4  # input 3 is the uint8 mask: four mask bytes per lane = one dword
      if lay.itemsize != want_item or lay.base_alignment % 16 != 0 or xstride[i] not in (0, 1):
        ok = False
        break
      if xstride[i] == 1:
        for tab in (key_off[i], depth_off[i]):
          if tab is not None and np.any(tab % 4):
            ok = False
        if i == 2 and gtab is not None and np.any(gtab % 4):
          ok = False
    if ok:
      vec = 4

  # "plane mode" for x-kept fp32 reductions whose rows cannot be 16-B aligned (latitude-fastest chunks, nx = 721):
  # groups of R consecutive depth rows are one contiguous span, fetched with aligned loads through LDS.
  plane_rows = 0
  if (allow_vec4 and x_kept and not map_mode and not (flags & 2) and vec == 1 and x_dim is not None
      and depth_dims and 64 <= nx <= 1024):
    inner = depth_dims[-1]
    used = [lay for lay in layouts[:3] if lay is not None]
    ok = all(lay.itemsize == 4 and lay.base_alignment % 16 == 0 and lay.stride(x_dim) == 1
             and lay.stride(inner) == nx for lay in used)
    mask_bytes = 0
    if flags & 1:
      # the validity mask rides along when its spans are contiguous too (same (inner, x) order as the data; it may
      # broadcast over every other dim)
      mlay = layouts[3] if len(layouts) > 3 else None
      ok = ok and mlay is not None and mlay.itemsize == 1 and mlay.base_alignment % 4 == 0 \
          and mlay.stride(x_dim) == 1 and mlay.stride(inner) == nx
    if gather is not None and inner in gather.dims:
      ok = False
    if ok:
      threads = -(-nx // 64) * 64
      for r in (8, 6, 5, 4, 3, 2):
        if flags & 1:
          mask_bytes = r * nx + 16
        lds = len(used) * (r * nx + 8) * 4 + mask_bytes
        if sizes[inner] % r == 0 and lds <= 80 * 1024 and (r * nx + 6) // 4 <= 2 * threads:
          plane_rows = r
          break
    if plane_rows:
      # one block per (key, chunk) covers every x: re-balance the chunking, chunk = multiple of R rows
      nchunk = int(min(max(ndepth // plane_rows, 1), max(1, -(-target_blocks // max(nkey, 1)))))
      depth_chunk = -(-ndepth // nchunk)
      depth_chunk = -(-depth_chunk // plane_rows) * plane_rows
      nchunk = -(-ndepth // depth_chunk)

  if fold_x in ('point', 'point64'):
    # one point per lane (ensemble kernels): the generic chunking stands, rows only have to be one contiguous run
    if not x_kept and not (flags & ~7) and x_dim is not None and depth_dims and nx <= 2048 and gather is None:
      inner = depth_dims[-1]
      ok = all(lay.itemsize == 4 and lay.stride(x_dim) == 1 and lay.stride(inner) == nx
               for lay in layouts[:2] if lay is not None)
      if flags & 1:  # the validity mask is walked flat alongside the data: same (inner, x) order
        mlay = layouts[3] if len(layouts) > 3 else None
        ok = ok and mlay is not None and mlay.itemsize == 1 and mlay.stride(x_dim) == 1 and mlay.stride(inner) == nx
      if ok:
        plane_rows, vec = sizes[inner], 1
        # measured on 8 x 51 x 1440 x 721 (threads, rows per block): (64, 1..2) 0.48 ms, (128, 2..4) 0.41-0.43,
        # (256, 4..8) 0.407, (256, 15) 0.415; the x-kept kernel on the same data 0.452, longitude-fastest data 0.397
        block_threads = FLAT1_THREADS
        depth_chunk = min(max(depth_chunk, -(-FLAT1_MIN_ELEMENTS // nx)), ndepth)
        if fold_x == 'point64':
          # one-wave blocks for the pipelined ensemble sweep (ens_pipe_kernel<.., FLAT>): a chunk's first and last tile
          # are shared with its neighbours (a 64-element piece each, fetched by both), so chunks are a few rows long, but
          # short enough for >= ~6 rounds of the 3072 resident waves
          block_threads = 64
          rows = max(1, -(-FLAT64_MIN_ELEMENTS // nx))
          depth_chunk = int(min(max(1, min(rows, (nkey * ndepth) // FLAT64_MIN_BLOCKS)), ndepth))
        nchunk = -(-ndepth // depth_chunk)
  elif fold_x and not x_kept and not map_mode and not (flags & 2) and x_dim is not None and depth_dims and nx + 3 <= 2048:
    inner = depth_dims[-1]
    r = sizes[inner]
    used = [(i, lay) for i, lay in enumerate(layouts[:3]) if lay is not None]
    ok = r % 4 == 0 and all(lay.itemsize == 4 and lay.base_alignment % 16 == 0 and lay.stride(x_dim) == 1
                            and lay.stride(inner) == nx for _, lay in used)
    if flags & 1:  # the mask is walked flat alongside the data: same (inner, x) order, planes 4-byte aligned
      mlay = layouts[3] if len(layouts) > 3 else None
      ok = ok and mlay is not None and mlay.itemsize == 1 and mlay.base_alignment % 4 == 0 \
          and mlay.stride(x_dim) == 1 and mlay.stride(inner) == nx
      if ok:
        used = used + [(3, mlay)]
    if gather is not None and inner in gather.dims:
      ok = False
    if ok:
      for i, _ in used:
        for tab in (key_off[i], None if depth_off[i] is None else depth_off[i][::r]):
          if tab is not None and np.any(tab % 4):
            ok = False
        if i == 2 and gtab is not None and np.any(gtab % 4):
          ok = False
    if ok:
      # measured on configs[1] (latitude-fastest): 256 threads x 4096 blocks 3.77 ms, 128 x 8192 3.76, 64 x 16384 4.05
      plane_rows, vec, block_threads = r, 1, 256
      rc = int(min(max(ndepth // 4, 1), max(1, -(-target_blocks // max(nkey, 1)))))
      depth_chunk = -(-(-(-ndepth // rc)) // 4) * 4
      nchunk = -(-ndepth // depth_chunk)

  return S1Plan(plane_rows=int(plane_rows), dims=dims, sizes=sizes, x_dim=x_dim, x_kept=bool(x_kept), sum_j=bool(sum_j), a_dims=a_dims,
                bk_dims=bk_dims, br_dims=br_dims, depth_dims=depth_dims, nkey=nkey, ndepth=ndepth, nx=nx,
                nchunk=int(nchunk), depth_chunk=int(depth_chunk), xstride=[int(v) for v in xstride[:MAX_INPUTS]],
                key_off=key_off[:MAX_INPUTS], depth_off=depth_off[:MAX_INPUTS], gather_key=gk, gather_depth=gd,
                gather_tab=gtab, n_gather_depth=int(ngd), flags=int(flags), block_threads=int(block_threads),
                vec=int(vec))


@dataclasses.dataclass
class S2Plan:
  nA: int
  nBk: int
  nBr: int
  nchunk: int
  nlane: int
  nj: int
  nbin: int
  sum_j: bool

  def out_shape(self) -> tuple:
    return (self.nA, self.nBk, self.nlane, 1 if self.sum_j else self.nj, self.nbin)


def build_s2_plan(s1: S1Plan, nlanes_total: int, nbin: int) -> S2Plan:
  return S2Plan(nA=s1.n(s1.a_dims), nBk=s1.n(s1.bk_dims), nBr=s1.n(s1.br_dims), nchunk=s1.nchunk,
                nlane=int(nlanes_total), nj=s1.nj, nbin=int(nbin), sum_j=bool(s1.sum_j))


def layout_of(array, dims_of_array: Sequence) -> InputLayout:
  """InputLayout of a numpy array / torch tensor whose axes are named `dims_of_array`."""
  if hasattr(array, 'stride') and callable(array.stride):  # torch
    st = [int(s) for s in array.stride()]
    itemsize = int(array.element_size())
    ptr = int(array.data_ptr())
  else:
    itemsize = int(array.dtype.itemsize)
    st = []
    for s in array.strides:
      if s % itemsize:
        raise ValueError('array strides are not a multiple of the item size')
      st.append(int(s // itemsize))
    ptr = int(array.__array_interface__['data'][0])
  align = 256
  while align > 1 and ptr % align:
    align //= 2
  return InputLayout(strides=dict(zip(dims_of_array, st)), itemsize=itemsize, base_alignment=align)
