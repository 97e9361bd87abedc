"""Accumulator reduction across GPUs: the counterpart of the Beam combine stage.

The reference sums per-chunk accumulators with `beam.CombinePerKey(CombiningSum())`
(weatherbenchX/beam_pipeline.py:509-510, weatherbenchX/beam_utils.py:30-50).  Here chunks of
(init_time x lead_time) are sharded over one process per GPU and, when `init_time`/`lead_time` is reduced,
every rank's AggregationState is packed into ONE float64 buffer and summed with a single all-reduce
(RCCL over xGMI with backend 'nccl'; 'gloo' on CPU for tests).  The buffer is KBs..MBs, so the
collective is latency bound and is issued once per job / step, never per chunk.
"""
from __future__ import annotations

import numpy as np

from weatherbenchx_amd import xarray_lite as xr
from weatherbenchx_amd.aggregation import AggregationState


def _leaves(tree, prefix=()):
  if isinstance(tree, xr.DataArray):
    yield prefix, tree
  elif isinstance(tree, dict):
    for k in sorted(tree, key=str):
      yield from _leaves(tree[k], prefix + (k,))
  elif tree is not None:
    raise TypeError(f'unsupported leaf type {type(tree)}')


def pack_state(state: AggregationState):
  """-> (flat float64 vector, layout) with a deterministic (sorted-key) order shared by all ranks."""
  items = []
  for which, tree in (('sws', state.sum_weighted_statistics), ('sw', state.sum_weights)):
    for path, da in _leaves(tree):
      items.append((which, path, da))
  flat = np.concatenate([np.asarray(da.values, dtype=np.float64).reshape(-1) for _, _, da in items]) if items \
      else np.zeros(0)
  layout = [(which, path, da.dims, da.shape) for which, path, da in items]
  return flat, layout, items


def unpack_state(flat: np.ndarray, items) -> AggregationState:
  out = {'sws': {}, 'sw': {}}
  pos = 0
  for which, path, da in items:
    n = da.size
    new = da.copy(data=flat[pos:pos + n].reshape(da.shape))
    pos += n
    node = out[which]
    for k in path[:-1]:
      node = node.setdefault(k, {})
    if path:
      node[path[-1]] = new
    else:
      out[which] = new
  return AggregationState(out['sws'], out['sw'])


def all_reduce_state(state: AggregationState, group=None, *, force: bool = False) -> AggregationState:
  """Sum of every rank's AggregationState (all ranks must hold the same statistics / shapes).  `force` runs the
  collectives even in a one-rank group (the RCCL plumbing check of tests/test_gpu_cabi.py)."""
  import torch  # pylint: disable=g-import-not-at-top
  import torch.distributed as dist  # pylint: disable=g-import-not-at-top

  if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
    return state
  state.wait()
  flat, layout, items = pack_state(state)
  backend = dist.get_backend(group)
  dev = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
  # fixed-size fingerprint first (one MIN all-reduce of [n, -n] gives min and max of the packed length): ranks that
  # disagree must fail loudly BEFORE the payload collective, which would otherwise crash or hang on ragged sizes.
  n = float(flat.size + 1000003 * len(layout))
  fp = torch.tensor([n, -n], dtype=torch.float64, device=dev)
  dist.all_reduce(fp, op=dist.ReduceOp.MIN, group=group)
  if float(fp[0]) != -float(fp[1]):
    raise ValueError('AggregationState layouts differ between ranks; cannot all-reduce')
  buf = torch.from_numpy(flat).to(dev)
  dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
  return unpack_state(buf.cpu().numpy(), items)


def shard_chunks(chunks: list, rank: int, world_size: int) -> list:
  """Round-robin assignment of time chunks to ranks (chunk i -> rank i mod n)."""
  return [c for i, c in enumerate(chunks) if i % world_size == rank]
