"""-m gpu: the DISPATCHER, not only the families.  A row range goes to one of four gradient-kernel
families by its size (column lists 512 .. 98,303 rows, row chunks from 65,536 rows / beyond the column lists, the three
streaming launches where row chunks decline, the row-wise kernel below), every threshold measured on ONE synthetic shape
(Zipf(1.1) columns, ~75 non-zeros per row, D = 47,236).  Here: shapes the thresholds were NOT tuned on -- Zipf 0.9 with 40
non-zeros per row, Zipf 1.3 with 150, a narrow model (D = 20,000) and one wider than the cold LDS tile (D = 70,000) -- at
2 K .. 800 K rows: every family the range could take is forced (DSGD_* knobs) and timed, and the library's own choice must
be within 15 % of the best.  The table goes to gpurun_out/ (copied to profiles/)."""

import os
import sys

import pytest

from conftest import ROOT, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="no gfx950 device")]


def test_the_default_dispatch_is_within_15_percent_of_the_best_family():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dispatch_table

    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "dispatch_table.jsonl"), "w") as f:
        cells, worst = dispatch_table.table(tol=0.15, out=f)
    bad = [(c["shape"], c["rows"], c["default_over_best"], c["families"]["default"]["kernel"], c["best_kernel"]) for c in cells if not c["ok"]]
    for c in cells:
        print("%-46s rows %7d: default %8.1f us (%s), best %8.1f us (%s)" % (
            c["shape"], c["rows"], c["families"]["default"]["us_per_step"], c["families"]["default"]["kernel"], c["best_us"], c["best_kernel"]))
    assert not bad, bad
