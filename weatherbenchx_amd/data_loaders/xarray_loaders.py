"""weatherbenchX/data_loaders/xarray_loaders.py under its own name."""
from weatherbenchx_amd.data_loaders._memory import (  # noqa: F401
    ClimatologyFromXarray, ConstantLoader, PersistenceFromXarray, PredictionsFromXarray, ProbabilisticClimatologyFromXarray,
    TargetsFromXarray, XarrayDataLoader, _rename_dataset)
