"""The two rules the push wave of the three-wave search loop rests on beyond LDS (bl_astar2_ahead.h, A2P_PATCH / A2P_TEST), held on the CPU
over random heaps with libstdc++'s index operations:

  * patches: the ancestor line of slot s_j (lane a = its (a + 1)-th ancestor) read BEFORE the earlier pushes of the same expansion,
    patched for each earlier push i in order -- if it dropped d_i ancestors and d_i >= c = bit length of (s_i xor s_j): lanes
    c - 1 .. d_i - 2 take their upper neighbour's value, lane d_i - 1 the entry of push i -- equals the line read from the heap AFTER
    those pushes, as long as the two slots lie on one level of the heap (across a power of two the lines share the root at different
    heights: such a line is read again -- this test found that case; the fixtures never met it);
  * the pop's writes: a pop_heap writes exactly the nodes from the root to where its value lands (an ancestor-closed set), so a line
    read BEFORE the pop agrees with the heap after it on every lane up to the one that stops the entry whenever the node n levels above
    that stopper (n = the number of patches applied: a patch moves values one lane down) is not on that path.
"""
import random


def push_heap(h, key):
    """std::push_heap after push_back (__push_heap: the hole rises while the parent is LESS in the comparator's order; the open list is
    a min-heap on the key, so: while parent key > key).  Returns (slot, drops): 1-based slot of the new leaf, ancestors that dropped."""
    h.append(key)
    hole = len(h) - 1
    slot = hole + 1
    drops = 0
    while hole > 0:
        parent = (hole - 1) // 2
        if h[parent][0] > key[0]:
            h[hole] = h[parent]; hole = parent; drops += 1
        else:
            break
    h[hole] = key
    return slot, drops


def pop_heap(h):
    """std::pop_heap + pop_back (__adjust_heap: the hole goes down along the smaller children to a leaf, then the value from the back
    rises).  Returns the 1-based node the value landed on."""
    top = h[0]
    value = h.pop()
    n = len(h)
    if n == 0:
        return top, 0
    hole = 0
    child = 0
    while child < (n - 1) // 2:
        child = 2 * (child + 1)
        if h[child][0] > h[child - 1][0] or (h[child][0] == h[child - 1][0] and False):
            child -= 1
        h[hole] = h[child]; hole = child
    if (n & 1) == 0 and child == (n - 2) // 2:
        child = 2 * (child + 1)
        h[hole] = h[child - 1]; hole = child - 1
    # __push_heap of the value from the hole
    while hole > 0:
        parent = (hole - 1) // 2
        if h[parent][0] > value[0]:
            h[hole] = h[parent]; hole = parent
        else:
            break
    h[hole] = value
    return top, hole + 1


def line(h, slot, lanes=26):
    """lane a: the (a + 1)-th ancestor of the 1-based slot (None beyond the root)"""
    out = []
    for a in range(lanes):
        anc = slot >> (a + 1)
        out.append(h[anc - 1] if anc >= 1 else None)
    return out


def patch(ln, d_i, s_i, key_i, s_j):
    """None: the two slots lie on different levels of the heap (s_j is, or follows, a power of two): their lines share the root at
    different heights -- the kernel reads such a line again"""
    if (s_i ^ s_j) > s_i:
        return None, 0
    c = (s_i ^ s_j).bit_length()
    if d_i < c:
        return ln, 0
    out = list(ln)
    for lane in range(c - 1, d_i - 1):
        out[lane] = ln[lane + 1]
    out[d_i - 1] = key_i
    return out, 1


def on_path(node, landing):
    """is the 1-based node an ancestor-or-self of the 1-based landing node?"""
    return node >= 1 and landing >> (landing.bit_length() - node.bit_length()) == node if node.bit_length() <= landing.bit_length() else False


def test_patched_lines_equal_lines_read_after_the_pushes():
    rng = random.Random(20261004)
    applied = reread = 0
    for trial in range(300):
        n0 = rng.choice([3, 7, 20, 100, 1000, 5000])
        spread = rng.choice([3, 10, 1000])                 # few distinct keys: ties and long rises
        h = []
        for k in range(n0):
            push_heap(h, (rng.randrange(spread), ("old", k)))
        for step in range(30):
            npush = rng.randrange(1, 4)
            base = len(h)
            lines = [line(h, base + 1 + j) for j in range(npush)]            # read before any push of this expansion
            info = []
            for j in range(npush):
                key = (rng.randrange(spread) if rng.random() < 0.7 else min(x[0] for x in h) - (1 if rng.random() < 0.3 else 0), ("new", trial, step, j))
                ln = lines[j]
                for (d_i, s_i, key_i) in info:
                    ln, did = patch(ln, d_i, s_i, key_i, base + 1 + j)
                    if ln is None:
                        reread += 1
                        ln = line(h, base + 1 + j)
                        break
                    applied += did
                assert ln == line(h, base + 1 + j), (trial, step, j)         # what memory holds now
                slot, drops = push_heap(h, key)
                assert slot == base + 1 + j
                info.append((drops, slot, key))
            if rng.random() < 0.5 and len(h) > 2:
                pop_heap(h)
    assert applied > 200 and reread > 0                                       # the patches were exercised, and the level boundary met


def test_a_line_read_before_the_pop_is_good_where_the_rule_says_so():
    rng = random.Random(4096)
    fresh = stale = 0
    for trial in range(400):
        n0 = rng.choice([5, 30, 300, 3000])
        spread = rng.choice([4, 50, 100000])
        h = []
        for k in range(n0):
            push_heap(h, (rng.randrange(spread), ("old", k)))
        for step in range(20):
            if len(h) < 3:
                break
            s1 = len(h)                                     # the slot the first push takes once the pop has removed an entry
            before = line(h, s1)                            # (the ancestors of that slot are the same nodes before and after)
            top, landing = pop_heap(h)
            after = line(h, s1)
            key = (rng.randrange(spread), ("new", trial, step))
            # how far the entry rises on the line as read BEFORE the pop: it passes the ancestors with a larger key
            d = 0
            while d < len(before) and before[d] is not None and before[d][0] > key[0]:
                d += 1
            stopper = s1 >> (d + 1)                         # 0: it rose to the root
            if stopper >= 1 and not on_path(stopper, landing):
                fresh += 1
                assert before[:d + 1] == after[:d + 1], (trial, step)        # every lane the push looks at is what memory holds
            else:
                stale += 1                                  # the kernel reads the line again
            push_heap(h, key)
    assert fresh > 1000 and stale > 10


def test_the_push_wave_as_the_kernel_runs_it():
    """The whole sequence of an iteration beyond LDS: the three lines are read BEFORE the pop (the slots follow from the length), the
    pop goes in, then each push decides from its line -- patched for the earlier pushes, read again where a rule says so -- and must
    drop exactly the ancestors, with exactly the values, that std::push_heap drops on the heap as it stands."""
    rng = random.Random(77)
    decided = reread = 0
    for trial in range(500):
        n0 = rng.choice([4, 9, 33, 130, 1030, 4100])
        spread = rng.choice([2, 5, 40, 100000])
        h = []
        for k in range(n0):
            push_heap(h, (rng.randrange(spread), ("old", k)))
        for step in range(25):
            if len(h) < 3:
                break
            npush = rng.randrange(0, 4)
            base = len(h) - 1                               # the length behind the pop
            lines = [line(h, base + 1 + j) for j in range(3)]               # asked for at B1, beside the pop
            top, landing = pop_heap(h)
            info = []
            for j in range(npush):
                s_j = base + 1 + j
                lo = min(x[0] for x in h)
                key = (rng.choice([lo - 1, lo, lo, rng.randrange(spread), rng.randrange(spread)]), ("new", trial, step, j))
                ln, patches, again = lines[j], 0, False
                for (d_i, s_i, key_i) in info:
                    ln2, did = patch(ln, d_i, s_i, key_i, s_j)
                    if ln2 is None:
                        again = True
                        break
                    ln, patches = ln2, patches + did
                if not again:
                    d = 0
                    while d < len(ln) and ln[d] is not None and ln[d][0] > key[0]:
                        d += 1
                    node = s_j >> (d + 1 + patches)         # the stopper, or the node `patches` levels above it
                    again = node < 1 or on_path(node, landing)
                if again:
                    reread += 1
                    ln = line(h, s_j)
                    d = 0
                    while d < len(ln) and ln[d] is not None and ln[d][0] > key[0]:
                        d += 1
                else:
                    decided += 1
                dropped = ln[:d]
                true_line = line(h, s_j)
                slot, drops = push_heap(h, key)
                assert slot == s_j and drops == d, (trial, step, j, again)
                assert dropped == true_line[:d], (trial, step, j, again)
                info.append((drops, slot, key))
    assert decided > 3000 and reread > 100
