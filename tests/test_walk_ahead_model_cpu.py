"""The claim bl_astar2_ahead.h rests on, held on the CPU: the reference's search replayed over an explicit array heap with libstdc++'s index
operations (tests/tools/walk_ahead_model.py) pops and pushes exactly what the oracle does, and a pop's walk taken BEFORE the previous
expansion's pushes are in reads a position those pushes wrote in a few per cent of the iterations at most (searches of >= 1e4 pops) --
and whenever the early walk really differs from the true one, it did read such a position (the test a kernel makes is conservative)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))


@pytest.mark.parametrize("name,case,bound", [("maze", 0, 0.30), ("maze", 2, 0.04), ("convex", 0, 0.06)])
def test_early_walk_is_rarely_invalidated(name, case, bound):
    import walk_ahead_model as wam
    st = wam.run(name, case)
    assert (st["pops"], st["pushes"]) == st["oracle"]                       # the model IS the reference's search, pop for pop
    assert st["early_walk_reads_a_written_position"] <= bound
    assert st["early_walk_differs"] <= st["early_walk_reads_a_written_position"]
    assert st["pushes_that_do_not_rise"] >= 0.5
