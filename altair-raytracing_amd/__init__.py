"""altair-raytracing_amd — MI355X-native integrating-sphere ray tracer (hot path only).

The product is libisx.so (csrc/, C ABI in include/isx.h) plus the C++ host driver that
keeps the reference's ROOT-macro entry signatures (host/).  This Python package is a thin
ctypes view of the C ABI used by tests/ and bench.py; it contains no compute of its own and
raises if the HIP library is missing.

Because the directory name carries a hyphen, import it with
    importlib.import_module("altair-raytracing_amd")
or through the `altair_raytracing_amd` shim at the repo root.
"""
from . import _abi as abi  # noqa: F401
from . import sharding  # noqa: F401
from .sharding import shard, step_slice, fluxmap_sharded, disc_sweep_sharded  # noqa: F401
from ._abi import (  # noqa: F401
    Config, Stats, IsxError, default_config, init, shutdown, device_info, set_option, fluxmap, fluxmap_device, sync,
    take_stats, last_kernel_ms, trace_endstates, disc_sweep, disc_sweep_per_position, exit_dz_hist, fluxmap_per_position, trace_rays_detector, exit_directions, fluxmap_series, detector_table, mathprobe, load, LIB_PATH, EXPORTS,
    SOURCE_PENCIL, SOURCE_BRDF, RAY_EXITED, RAY_ABSORBED, RAY_SUSPENDED,
)

__all__ = ["abi", "sharding", "shard", "step_slice", "fluxmap_sharded", "disc_sweep_sharded", "Config", "Stats", "IsxError", "default_config", "init", "shutdown", "device_info", "set_option",
           "fluxmap", "fluxmap_device", "fluxmap_per_position", "trace_rays_detector", "exit_directions", "fluxmap_series", "sync", "take_stats", "last_kernel_ms", "trace_endstates", "disc_sweep", "disc_sweep_per_position", "exit_dz_hist",
           "detector_table", "mathprobe", "load", "LIB_PATH", "EXPORTS",
           "SOURCE_PENCIL", "SOURCE_BRDF", "RAY_EXITED", "RAY_ABSORBED", "RAY_SUSPENDED"]
