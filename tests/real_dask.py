"""Real dask for the chunked-input tests: the locator lives with the rest of the test infrastructure (oracle/real_dask.py), where
`bench.py`'s `cpu_baseline` leg finds it too."""
from oracle.real_dask import dask_array  # noqa: F401
