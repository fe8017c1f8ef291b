"""magcache_amd -- MI355X-native MagCache denoising engine (Wan2.1 DiT forward + MagCache skip /
residual cache), HIP kernels behind a C ABI (include/magcache_hip.h).  See DESIGN.md."""
from .mag_ratios import TABLES, SOURCES  # noqa: F401

__all__ = ["TABLES", "SOURCES"]


def __getattr__(name):
    # heavy members (they need torch / the HIP library) are imported lazily
    if name in ("Engine", "WAN_T2V_1_3B", "WAN_T2V_14B", "WAN_I2V_14B", "WAN_VACE_1_3B", "WAN_VACE_14B",
                "synthetic_weights", "weight_names"):
        from . import engine
        return getattr(engine, name)
    if name in ("WanModelHIP", "magcache_forward", "magcache_calibration", "init_magcache",
                "init_magcache_calibration", "disable_magcache", "nearest_interp", "resample_cfg_table",
                "select_table", "plain_forward", "magcache_vace_forward", "magcache_vace_calibration",
                "vace_plain_forward"):
        from . import model
        return getattr(model, name)
    if name in ("sample", "flow_timesteps", "cfg_euler_"):
        from . import sampler
        return getattr(sampler, name)
    if name in ("mmdit", "wan22", "metrics", "parallel", "generate"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
