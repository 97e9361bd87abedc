"""weatherbenchX/data_loaders/latency_wrappers.py under its own name."""
from weatherbenchx_amd.data_loaders._memory import (  # noqa: F401
    ConstantLatencyWrapper, MultipleConstantLatencyWrapper, XarrayConstantLatencyWrapper)
