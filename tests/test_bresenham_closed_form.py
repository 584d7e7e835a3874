"""The device replaces the reference's sequential Bresenham walk (src/rasterizer.rs:1777-1821) by an O(1)
per-pixel membership test (rusterix_amd/csrc/rxr_kernels.hip, bresenham_hits).  This pins the closed
form against the walk itself -- restated here exactly as the reference writes it -- over every segment
in a window and a random sample of long ones; the GPU tests then compare rendered line batches with the
oracle (tests/test_gpu_parity.py::test_bresenham_lines_bit_exact, test_2d_many_rectangles_bit_exact)."""
import itertools
import random


def walk(x0, y0, x1, y1):
    pts = []
    dx, dy = abs(x1 - x0), abs(y1 - y0)
    sx = 1 if x0 < x1 else -1
    sy = 1 if y0 < y1 else -1
    err = dx - dy
    x, y = x0, y0
    while x != x1 or y != y1:
        pts.append((x, y))
        e2 = err * 2
        if e2 > -dy:
            err -= dy
            x += sx
        if e2 < dx:
            err += dx
            y += sy
    return pts


def closed(x0, y0, x1, y1, px, py):
    dx, dy = abs(x1 - x0), abs(y1 - y0)
    sx = 1 if x0 < x1 else -1
    sy = 1 if y0 < y1 else -1
    if dx >= dy:
        k = (px - x0) * sx
        if k < 0 or k >= dx:
            return False
        a = 2 * k * dy - dx
        j = 0 if a <= 0 else (a + 2 * dx - 1) // (2 * dx)
        return py == y0 + sy * j
    k = (py - y0) * sy
    if k < 0 or k >= dy:
        return False
    a = 2 * k * dx - dy
    i = 0 if a <= 0 else (a + 2 * dy - 1) // (2 * dy)
    return px == x0 + sx * i


def test_every_segment_in_a_window():
    r = range(-8, 9)
    for x1, y1 in itertools.product(r, r):
        for x0, y0 in ((0, 0), (3, -2)):
            pts = walk(x0, y0, x1, y1)
            assert len(pts) == len(set(pts)) == max(abs(x1 - x0), abs(y1 - y0))
            s = set(pts)
            for px in range(-11, 12):
                for py in range(-11, 12):
                    assert closed(x0, y0, x1, y1, px, py) == ((px, py) in s), (x0, y0, x1, y1, px, py)


def test_random_long_segments():
    rnd = random.Random(7)
    for _ in range(200):
        x0, y0, x1, y1 = (rnd.randint(-3000, 3000) for _ in range(4))
        s = set(walk(x0, y0, x1, y1))
        probes = list(s)[:60] + [(rnd.randint(-3100, 3100), rnd.randint(-3100, 3100)) for _ in range(100)]
        probes += [(x + dx_, y + dy_) for (x, y) in list(s)[:20] for dx_, dy_ in ((1, 0), (0, 1), (-1, 0), (0, -1))]
        for px, py in probes:
            assert closed(x0, y0, x1, y1, px, py) == ((px, py) in s)
