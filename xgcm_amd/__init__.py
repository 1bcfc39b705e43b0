"""xgcm_amd -- MI355X-native backend for the xgcm staggered-grid stencil hot path."""

__version__ = "0.1.0"
