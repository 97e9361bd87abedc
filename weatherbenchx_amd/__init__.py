"""weatherbenchx_amd: MI355X-native engine for WeatherBench-X's scoring hot path
(Statistic.compute -> Aggregator reduce -> AggregationState), behind the reference's plugin API.

Layout: csrc/ (HIP kernels + C ABI, built to libwbx_hip.so), _hip.py (ctypes binding), planner.py / engine.py
(two-stage reduction), lazy.py (fusion of the unfused plugin API), xarray_lite.py (labeled arrays),
metrics/, aggregation.py, weighting.py, binning.py, time_chunks.py, xarray_tree.py (mirror of the reference
surface), pipeline.py / distributed.py (chunk loop + RCCL accumulator all-reduce), spectra.py.
Around the path, host side and without kernels of their own (round 5): data_loaders/ (in-memory, latency, Parquet station loaders),
loaders.py (file-backed, page-locked), interpolations.py, metrics/{categorical,spatial,multivariate}.py, statistical_inference/,
beam_pipeline.py (`define_pipeline`, run on the spot), test_utils.py.
"""
__version__ = '0.1.0'
