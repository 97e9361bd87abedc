"""Chunk loaders with the reference's module layout (weatherbenchX/data_loaders/): `base`, `xarray_loaders`, `latency_wrappers`,
`sparse_parquet`.  Everything in-memory is implemented in `_memory.py` and re-exported here too, so both
`data_loaders.PredictionsFromXarray` and `data_loaders.xarray_loaders.PredictionsFromXarray` resolve.  The file-backed gridded
loaders that feed the copy stream from page-locked memory live in `weatherbenchx_amd.loaders`."""
from weatherbenchx_amd.data_loaders._memory import (  # noqa: F401
    ClimatologyFromXarray, ConstantLatencyWrapper, ConstantLoader, DataLoader, MultipleConstantLatencyWrapper,
    PersistenceFromXarray, PredictionsFromXarray, ProbabilisticClimatologyFromXarray, TargetsFromXarray, XarrayConstantLatencyWrapper,
    XarrayDataLoader, add_nan_mask_to_data)
