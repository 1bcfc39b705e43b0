"""xgcm_amd -- MI355X-native backend for the xgcm staggered-grid stencil hot path.

Public surface mirrors the reference package (`xgcm/__init__.py:6-7`): `Grid`, `as_grid_ufunc`,
`apply_as_grid_ufunc`; plus the labelled-array duck types used when xarray is unavailable.
"""

from .axis import Axis
from .grid import Grid
from .grid_ufunc import GridUFunc, apply_as_grid_ufunc, as_grid_ufunc
from .labeled import DataArray, Dataset

# the reference's module names for the metadata functions (`xgcm.metadata_parsers.parse_sgrid(ds)` in its user guide,
# `from xgcm.metadata_parsers import parse_comodo` in its tests): one module here, reachable under all three names
import sys as _sys

from . import metadata as metadata_parsers  # noqa: E402

sgrid = comodo = metadata_parsers
for _name in ("metadata_parsers", "sgrid", "comodo"):
    _sys.modules.setdefault(f"{__name__}.{_name}", metadata_parsers)

__version__ = "0.1.0"
__all__ = ["Grid", "Axis", "GridUFunc", "as_grid_ufunc", "apply_as_grid_ufunc", "DataArray", "Dataset"]
