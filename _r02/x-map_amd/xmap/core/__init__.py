"""Tool classes of the hot path + re-export of its pipeline functions.

The reference's xmap/core/__init__.py is empty; BASELINE.json names the API "xmap.core.*", so the
pipeline functions are importable from here as well as from xmap.utils.assist."""
from xmap.core.baselinerSim import BaselinerSim  # noqa: F401
from xmap.core.extender import ExtendSim  # noqa: F401
from xmap.core.generator import Generator  # noqa: F401


def __getattr__(name):
    if name in ("baseliner_calculate_sim_pipeline", "extender_pipeline", "generator_pipeline",
                "extract_siminfo", "map_to_dict"):
        from xmap.utils import assist
        return getattr(assist, name)
    raise AttributeError(name)
