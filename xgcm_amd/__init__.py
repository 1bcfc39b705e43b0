"""xgcm_amd -- MI355X-native backend for the xgcm staggered-grid stencil hot path.

Public surface mirrors the reference package (`xgcm/__init__.py:6-7`): `Grid`, `as_grid_ufunc`,
`apply_as_grid_ufunc`; plus the labelled-array duck types used when xarray is unavailable.
"""

from .axis import Axis
from .grid import Grid
from .grid_ufunc import GridUFunc, apply_as_grid_ufunc, as_grid_ufunc
from .labeled import DataArray, Dataset

__version__ = "0.1.0"
__all__ = ["Grid", "Axis", "GridUFunc", "as_grid_ufunc", "apply_as_grid_ufunc", "DataArray", "Dataset"]
