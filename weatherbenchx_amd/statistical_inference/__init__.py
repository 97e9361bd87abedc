"""Confidence intervals and significance tests for metrics computed from per-unit aggregation states (counterpart of
weatherbenchX/statistical_inference/): what the accumulators this package produces on the device are handed to afterwards.  Host
side, small arrays (one value per experimental unit and output point); outside the path SURVEY section 8 names."""
