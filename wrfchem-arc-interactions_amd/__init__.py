"""MI355X-native column-radiation engine for WRF-Chem's aerosol + clean-sky
RRTMG double call (drop-in for RRTMG_SWRAD / RRTMG_LWRAD of the
douglowe/WRFChem-ARC-Interactions patch set)."""
from . import driver_steps, engine, engine_lw, engine_optics, forcing, synth, synth_clouds, synth_mosaic, tiles  # noqa: F401

__all__ = ["synth", "synth_mosaic", "synth_clouds", "engine", "engine_lw", "engine_optics", "tiles", "forcing", "driver_steps"]
