"""machisplin_amd -- MI355X (gfx950) backend for MACHISPLIN's data-parallel hot path.

The package holds only what the path needs: ``csrc/`` (hand-written HIP kernels and the
C ABI of include/machisplin_hip.h, built into ``libmachisplin_hip.so``) and the host-side
mirror of the reference's interface for the path (``Tps``/``interpolate``/``predict`` as
``machisplin.mltps`` calls them, V73:442-930).  Importing the package needs neither a GPU
nor the built library; every compute entry point does, and raises without them.
"""
from . import _lib
from ._lib import MhsError, init
from .raster import Geometry, RasterStack
from . import models
from .models import predict, ensemble_predict
from . import tps
from .tps import Tps, fit_many, interpolate, eval_mode, EVAL_AUTO, EVAL_DIRECT, EVAL_FAR_FIELD
from . import tiles, mltps, cv
from .mltps import mltps as mltps_layers, mltps_predict, tps_residual_surface

__all__ = ["MhsError", "init", "Geometry", "RasterStack", "Tps", "interpolate", "eval_mode", "EVAL_AUTO", "EVAL_DIRECT", "EVAL_FAR_FIELD", "predict",
           "ensemble_predict", "models", "tiles", "mltps", "mltps_predict",
           "tps_residual_surface", "_lib"]
