"""(init_time x lead_time) chunking -- the unit of work that is sharded over GPUs.

Counterpart of weatherbenchX/time_chunks.py:26-202: same constructor arguments, iteration order
(init-major product), `__len__`/`__getitem__`, and `iter_with_chunk_offsets()` whose offsets are
`chunk_size * chunk_index` along each time axis (time_chunks.py:190-202).
"""
from __future__ import annotations

import dataclasses
from collections.abc import Iterable, Iterator
from typing import Optional, Union

import numpy as np

TimeChunk = tuple[np.ndarray, Union[np.ndarray, slice]]


@dataclasses.dataclass(frozen=True)
class TimeChunkOffsets:
  init_time: int
  lead_time: int


def _split(values, size):
  return [values[start:start + size] for start in range(0, len(values), size)]


class TimeChunks(Iterable[TimeChunk]):
  """Iterable of (init_times, lead_times) chunk products."""

  def __init__(self, init_times: np.ndarray, lead_times: Union[np.ndarray, slice],
               init_time_chunk_size: Optional[int] = None, lead_time_chunk_size: Optional[int] = None):
    for label, size in (('init_time_chunk_size', init_time_chunk_size),
                        ('lead_time_chunk_size', lead_time_chunk_size)):
      if size is not None and size < 0:
        raise ValueError(f'{label}={size} but should be non-negative or None')
    init_times = np.asarray(init_times).astype('datetime64[ns]')
    init_size = init_time_chunk_size or len(init_times)  # None / 0 -> one chunk
    self._init_chunks = _split(init_times, init_size) if len(init_times) else []
    if isinstance(lead_times, slice):
      if lead_times.start is None or lead_times.stop is None:
        raise ValueError('Slice start and stop must be specified.')
      if lead_times.step is not None:
        raise ValueError('Slice step must be None.')
      if lead_time_chunk_size:
        raise ValueError('Chunking in lead time not compatible for slice.')
      self._lead_chunks = [lead_times]
      lead_size = lead_time_chunk_size
    elif isinstance(lead_times, np.ndarray):
      lead_times = lead_times.astype('timedelta64[ns]')
      lead_size = lead_time_chunk_size or len(lead_times)
      self._lead_chunks = _split(lead_times, lead_size) if len(lead_times) else []
    else:
      raise ValueError('Lead times must be either np.ndarray or slice.')
    self._init_times, self._lead_times = init_times, lead_times
    self._init_size, self._lead_size = init_size, lead_size

  @property
  def init_times(self) -> np.ndarray:
    return self._init_times

  @property
  def lead_times(self) -> Union[np.ndarray, slice]:
    return self._lead_times

  @property
  def init_time_chunk_size(self) -> int:
    return self._init_size

  @property
  def lead_time_chunk_size(self) -> int:
    return self._lead_size

  def __len__(self) -> int:
    return len(self._init_chunks) * len(self._lead_chunks)

  def __getitem__(self, index: int) -> TimeChunk:
    if index < 0 or index >= len(self):
      raise IndexError(f'TimeChunks index out of range: {index}')
    i, j = divmod(index, len(self._lead_chunks))
    return self._init_chunks[i], self._lead_chunks[j]

  def __iter__(self) -> Iterator[TimeChunk]:
    for index in range(len(self)):
      yield self[index]

  def iter_with_chunk_offsets(self) -> Iterator[tuple[TimeChunkOffsets, TimeChunk]]:
    n_lead = len(self._lead_chunks)
    for index in range(len(self)):
      i, j = divmod(index, n_lead)
      lead_offset = (self._lead_size or 0) * j
      yield TimeChunkOffsets(init_time=self._init_size * i, lead_time=lead_offset), self[index]
