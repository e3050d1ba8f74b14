"""Closed form of the reference's Bresenham variant (src/slam/mapping.cpp:101-127), checked against the loop itself:
cell k (k = 0 .. max(dx,dy)-1, start included, end excluded) advances the major axis by k and the minor axis by
floor((2*k*dmin + dmaj) / (2*dmaj)).  k_map_update's segment walk relies on it."""


def walk(x0, y0, x1, y1):
    dx, dy = abs(x1 - x0), abs(y1 - y0)
    sx, sy = (1 if x0 < x1 else -1), (1 if y0 < y1 else -1)
    err, x, y, out = dx - dy, x0, y0, []
    while x != x1 or y != y1:
        out.append((x, y))
        e2 = 2 * err
        if e2 >= -dy:
            err -= dy; x += sx
        if e2 <= dx:
            err += dx; y += sy
    return out


bad = 0
for dx in range(0, 230):
    for dy in range(0, 230, 3):
        for sx, sy in ((1, 1), (-1, -1), (1, -1), (-1, 1)):
            cells = walk(5, 7, 5 + sx * dx, 7 + sy * dy)
            assert len(cells) == max(dx, dy)
            for k, (x, y) in enumerate(cells):
                if dx >= dy:
                    e = (5 + sx * k, 7 + sy * ((2 * k * dy + dx) // (2 * dx) if dx else 0))
                else:
                    e = (5 + sx * ((2 * k * dx + dy) // (2 * dy)), 7 + sy * k)
                bad += (x, y) != e
print("mismatches:", bad)
