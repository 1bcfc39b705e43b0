"""Enumeration of axis subsets whose metrics can be multiplied into a composite metric.

Behaviour of reference xgcm/metrics.py:4-30 (`iterate_axis_combinations`): first the full
set, then every split of the axes into one group of `k` axes (k = N-1 .. 1) plus groups of
equal smaller size drawn from the remaining axes.
"""

from itertools import combinations


def iterate_axis_combinations(items):
    whole = frozenset(items)
    yield (whole,)
    n = len(items)
    for k in range(n - 1, 0, -1):
        rest_size = n - k
        for group_size in range(min(rest_size, k), 0, -1):
            for head in combinations(whole, k):
                head = frozenset(head)
                remainder = whole - head
                yield (head,) + tuple(frozenset(c) for c in combinations(remainder, group_size))
