"""Runs last (-m gpu): every family of conditional parity assertions must have taken its STRICT branch at least once
in this session -- a green run in which a tight bound was silently waived everywhere is a failure."""

import pytest

import waivers
from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="no gfx950 device")]


def test_every_family_took_its_strict_branch():
    t = waivers.table()
    print("\n".join("%-58s strict %4d waived %4d" % (k, v["strict"], v["waived"]) for k, v in t.items()))
    never = [k for k, v in t.items() if v["strict"] == 0]
    assert not never, "tight assertions never ran for: %s (table: %s)" % (never, t)
