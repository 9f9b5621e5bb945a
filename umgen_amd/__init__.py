"""umgen_amd -- MI355X-native next-scene rollout engine behind UMGen's model/registry surface.

Only the hot path ``UMGen.inference`` lives here: ``csrc/`` (HIP kernels + the C ABI of include/umgen.h),
``engine.py`` (ctypes handle), ``model.py`` (drop-in ``UMGen`` nn.Module + registry), plus the resolved config,
state-dict key set and synthetic inputs used by tests and the bench.
"""
from .config import RolloutConfig, large_config, tiny_config, wide2x_config  # noqa: F401

__all__ = ["RolloutConfig", "large_config", "tiny_config", "wide2x_config"]
