"""Bookkeeping of conditional assertions in the GPU parity tests.

Several parity assertions are exact only when the oracle saw no margin within 1e-5 of the gate
(core/ml/SparseSVM.scala:27-28): an fp32 engine may then legitimately gate a row differently and the test falls
back to a looser statement.  A green run must be able to say WHICH branch ran: every such site calls `strict()`
when it asserted the tight statement and `waived()` when it took the loose one.  tests/test_zz_waivers.py fails if a
family never took its strict branch in the session; conftest.py prints the table and writes gpurun_out/waivers.json.
"""

from __future__ import annotations

from collections import OrderedDict

COUNTS: "OrderedDict[str, list]" = OrderedDict()   # family -> [strict, waived, [reasons]]


def _slot(family):
    return COUNTS.setdefault(family, [0, 0, []])


def strict(family):
    _slot(family)[0] += 1
    return True


def waived(family, why=""):
    s = _slot(family)
    s[1] += 1
    if why and len(s[2]) < 8:
        s[2].append(str(why))
    return False


def check(family, is_strict, why=""):
    """Record the branch; returns is_strict so that `if waivers.check(...)` reads like the condition it wraps."""
    return strict(family) if is_strict else waived(family, why)


def tight(family, ok_tight, exposed, why=""):
    """The standard shape of a conditional parity assertion: the TIGHT statement is evaluated first and recorded as
    strict when it holds; when it does not, that is acceptable only if the oracle reported a row within 1e-5 of the
    gate (`exposed`) -- recorded as waived with the reason -- and an assertion failure otherwise.  Returns ok_tight."""
    if ok_tight:
        strict(family)
        return True
    assert exposed, "%s: tight statement failed without a near-gate row to explain it (%s)" % (family, why)
    waived(family, why)
    return False


def table():
    return {k: {"strict": v[0], "waived": v[1], "reasons": v[2]} for k, v in COUNTS.items()}
