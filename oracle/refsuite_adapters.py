"""Private names of the reference that its TESTS import, bound to what `xgcm_amd` has instead.

TEST INFRASTRUCTURE for `oracle/run_reference_suite.py` (build container only).  The reference's tests reach into a few
underscore helpers of `xgcm/padding.py`; the product has no reason to carry those names, so the scratch `xgcm.padding`
module gets them from here:

* `_resolve_pivot` (xgcm/padding.py:156-180; tests: test_fold.py:94-113): the product's `padding.pole_on_edges`
  answers the same question as a pair of booleans; this adapter words the answer as the reference does.
* `_maybe_swap_dimension_names`, `_strip_all_coords` (xgcm/padding.py:201-227): used by test_padding.py ONLY to build
  its expected arrays from the test's own xarray objects; they never see a product object.
"""


def extras(product_padding):
    def _resolve_pivot(pivot, fold_axis, seam_axis):
        seam_on_edge, fold_on_edge = product_padding.pole_on_edges(pivot, fold_axis, seam_axis)
        return {"seam": "edge" if seam_on_edge else "center", "fold": "edge" if fold_on_edge else "center"}

    def _maybe_swap_dimension_names(da, from_name, to_name):
        mapping = {from_name: to_name} if from_name in da.dims else {}
        if to_name in da.dims:
            mapping[to_name] = from_name
        return da.rename(mapping)  # one simultaneous rename: a swap when both dims are present

    def _strip_all_coords(obj):
        if isinstance(obj, dict):
            return {k: _strip_all_coords(v) for k, v in obj.items()}
        if hasattr(obj, "_replace"):  # a product array
            return product_padding._strip_all_coords(obj)
        return obj.reset_coords(drop=True).reset_index([d for d in obj.dims if d in obj.coords], drop=True)

    return {"_resolve_pivot": _resolve_pivot, "_maybe_swap_dimension_names": _maybe_swap_dimension_names,
            "_strip_all_coords": _strip_all_coords}
