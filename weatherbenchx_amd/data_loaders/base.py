"""weatherbenchX/data_loaders/base.py under its own name."""
from weatherbenchx_amd.data_loaders._memory import DataLoader, add_nan_mask_to_data  # noqa: F401
