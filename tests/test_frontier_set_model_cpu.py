"""The LDS set of k_frontier_grow2 (bl_frontiers.hip), held on the CPU: every frontier-class cell of the grid sits in one table, a word
per cell (cell | visited << 31), linear probing from a mixed multiplicative home WITHOUT wrap-around (FG_PAD spare slots behind the
last home), nothing ever removed.  The growth decides a neighbour from the home slot and the next with two compares:

  * a word EQUAL to the cell is the cell, not yet grown (a grown one carries bit 31);
  * the cell one slot behind its home implies an occupied home -- so `v0 == nc || v1 == nc` needs no test of v0;
  * only when BOTH slots are taken by other cells does the chain go on (d = 2 .. FG_DMAX - 1), ending at the first empty slot;
  * an off-grid neighbour is looked up as -2, which no slot ever holds (empty slots hold -1).

Checked here against a Python set over random and adversarial cell sets (columns of grids whose width makes a multiplicative home
alone crowd: 10 946, 50 549), up to the half load the kernel allows, with cells marked grown at random."""
import random

FG_SLOTS = 32768
FG_PAD = 128
FG_DMAX = 64
FG_CELL_MAX = 16384
M32 = 0xFFFFFFFF
EMPTY = M32                      # -1


def fg_home(c):
    h = (c * 2654435761) & M32
    h ^= h >> 15
    return ((h * 0x85EBCA6B) & M32) >> 17


def build(cells):
    """the kernel's insert: CAS on home + d, d < FG_DMAX; None when a cell finds no room (the kernel then declines the sweep)"""
    tab = [EMPTY] * (FG_SLOTS + FG_PAD)
    for c in cells:
        h = fg_home(c)
        for d in range(FG_DMAX):
            if tab[h + d] == EMPTY:
                tab[h + d] = c
                break
        else:
            return None
    return tab


def lookup(tab, nc):
    """(slot, fresh) as the growth loop of k_frontier_grow2 forms them; nc = -2 (as a 32-bit word) for an off-grid neighbour"""
    w = nc & M32
    h = fg_home(w)
    v0, v1 = tab[h], tab[h + 1]
    slot = h + (0 if v0 == w else 1)
    fresh = v0 == w or v1 == w
    if max(v0, v1) != EMPTY:                                               # both slots taken
        if not fresh and (v0 & 0x7FFFFFFF) != w and (v1 & 0x7FFFFFFF) != w and nc >= 0:
            for d in range(2, FG_DMAX):
                v = tab[h + d]
                if v == EMPTY:
                    break
                if (v & 0x7FFFFFFF) == w:
                    slot, fresh = h + d, v < 0x80000000
                    break
    return slot, fresh


def check(cells, rng, probes):
    tab = build(cells)
    assert tab is not None, "no room within FG_DMAX slots of a home"
    member = set(cells)
    grown = set()
    longest = 0
    for c in cells:                                                        # how far from home the cells sit
        h = fg_home(c)
        longest = max(longest, next(d for d in range(FG_DMAX) if tab[h + d] == c))
    order = list(cells)
    rng.shuffle(order)
    for step, c in enumerate(order):                                       # grow them one by one, looking cells up in between
        for q in probes(c):
            slot, fresh = lookup(tab, q)
            assert fresh == (q in member and q not in grown), (q, slot)
            if fresh:
                assert tab[slot] == q
        slot, fresh = lookup(tab, c)
        assert fresh and tab[slot] == c
        tab[slot] = c | 0x80000000                                         # the finding lane's store
        grown.add(c)
        assert lookup(tab, c)[1] is False
    return longest


def test_random_cell_sets_up_to_half_load():
    rng = random.Random(2026)
    for n in (1, 50, 3000, 9000, FG_CELL_MAX):
        W = rng.choice([1000, 4096, 65535])
        cells = rng.sample(range(W * 3000), n)
        longest = check(cells, rng, lambda c: [c - 1, c + 1, c + W, c - W, -2, rng.randrange(W * 3000)])
        assert longest < FG_DMAX // 2


def test_columns_and_rows_of_the_widths_that_crowd_a_multiplicative_home():
    """a frontier is a curve: long runs of cells a constant stride apart.  Under index * 2654435761 >> 17 alone a stride of 10 946
    steps the home by 0.63 slots and one of 50 549 by 0.09: 600 cells of such a column found no room within 64 slots"""
    rng = random.Random(7)
    for W in (10946, 50549, 4096, 21892):
        for stride in (W, W + 1, W - 1, 1):
            base = 5 * W + 17
            cells = [base + j * stride for j in range(4000)]
            longest = check(cells, rng, lambda c: [c - stride, c + stride, c + 1, -2])
            assert longest < FG_DMAX // 2, (W, stride, longest)


def test_the_home_region_never_runs_past_the_padding():
    assert max(fg_home(c) for c in list(range(0, 1 << 22, 7)) + [M32 - 1, M32]) < FG_SLOTS
    assert FG_PAD >= FG_DMAX + 1                                            # slot home + 1 is read without a bounds test
