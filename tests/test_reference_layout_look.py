"""CPU: the arithmetic of the reference layout's FIRST LOOK (csrc/device_math.hpp, ref_fast_look_constants on the host and ref_first_look_fast
on the device), restated in Python integers.

A wave tile of the vector kernels asks "can a scalar head or tail of a reference partition (src/piquant.cpp:145-157 for the split,
kernels_specialized.inl:52-56 / 178-182 for the head and the tail) reach into me?" with two preloaded constants and two scalar multiplies.  The look may
say yes too often -- a tile that passes it takes the exact second look -- but never no for a tile that holds a scalar position.  Checked here, for the
geometry of every kernel that takes the look: (a) the look passes every tile that really holds a scalar position of the partition rule, (b) it passes
every tile the 64-bit form of rounds 6's first sessions passed, and hardly any other.  The GPU suite checks the instructions
(tests/test_gpu_01_default_is_reference_exact.py); this file checks the mathematics, on every machine."""
import random

import pytest

M64 = (1 << 64) - 1
M32 = (1 << 32) - 1


def margins(pack, blk):
    """RefMargins (device_math.hpp)"""
    return 16 + pack + 2, blk + pack + 2


def look_constants(n, T, tile, pack, blk):
    """ref_fast_look_constants: (m, w) or None when the launch takes the look that reads the RefSplit"""
    below, above = margins(pack, blk)
    width = tile + below + above
    if width * T >= n or n > 1 << 31:       # ref_prepare_first_look's `always`; the fast look's size limit
        return None
    m = (T << 64) // n
    if m == 0 or 2 * m < n or (m * width) >> 64:
        return None
    w = (m * width + M32) >> 32
    if w >> 32:
        return None
    return m, w


def look32(m, w, wave_tile, tile, below):
    """ref_first_look_fast as the device computes it: 32-bit element index, high half of the 64-bit fraction, carry of the add"""
    if wave_tile == 0:
        return True
    a = (wave_tile * tile - below) & M32
    f = (a * (m >> 32) + ((a * (m & M32)) >> 32)) & M32
    return f + w > M32


def look64(m, wave_tile, tile, below, width):
    """the 64-bit form the kernels used before"""
    f = ((wave_tile * tile - below) * m) & M64
    return f + width * m > M64


def boundary(n, T, t, pack):
    """src/piquant.cpp:145-157: partition t starts at n t / T, aligned down to a whole packed byte; the last one keeps the ragged end"""
    if t >= T:
        return n
    return (n * t // T) & ~(pack - 1)


def scalar_windows(n, T, pack, blk, out_align, ts):
    """[lo, hi) of scalar positions around the boundaries in ts: the tail of partition t - 1 and the head of partition t"""
    for t in ts:
        b = boundary(n, T, t, pack)
        lo = hi = b
        if t > 0:
            begin = boundary(n, T, t - 1, pack)
            length = b - begin
            head = min((16 - ((out_align + begin) & 15)) & 15, length) if out_align >= 0 else 0
            lo = begin + head + ((length - head) & ~(blk - 1))
        if t < T:
            end = boundary(n, T, t + 1, pack)
            head = min((16 - ((out_align + b) & 15)) & 15, end - b) if out_align >= 0 else 0
            hi = b + head
        if hi > lo:
            yield lo, hi


# (elements per wave tile, elements per packed byte, SIMD block of the reference kernel, has a scalar head): quantize f32 -> u8 (128 threads, U = 2),
# bf16 -> u8 / u4 / u2, dequantize -> bf16 from u8 / u4 / u2 (SET and ADD tiles), u2 -> f32 ADD
GEOMETRIES = [(512, 1, 64, True), (1024, 1, 64, False), (1024, 2, 16, False), (1024, 4, 16, False), (1024, 1, 64, False), (2048, 2, 128, False),
              (2048, 4, 256, False), (512, 4, 4, False)]


@pytest.mark.parametrize("seed", range(4))
def test_first_look_passes_every_tile_that_holds_a_scalar_position(seed):
    rng = random.Random(1000 + seed)
    checked = 0
    for _ in range(600):
        tile, pack, blk, has_head = rng.choice(GEOMETRIES)
        n = rng.choice([27264000, rng.randint(tile * 4, 40_000_000), rng.randint(tile * 4, 1 << 31), (1 << 31) - rng.randint(0, 4096), 1 << 31])
        T = rng.choice([1, 2, 3, 7, 64, 127, 255, 256, 1000, 65536, rng.randint(1, 65536)])
        c = look_constants(n, T, tile, pack, blk)
        if c is None:
            continue
        m, w = c
        below, _ = margins(pack, blk)
        out_align = rng.randrange(16) if has_head else -1
        ts = set(rng.sample(range(T + 1), min(T + 1, 12))) | {0, T}
        for lo, hi in scalar_windows(n, T, pack, blk, out_align, ts):
            for wave_tile in range(lo // tile, (hi - 1) // tile + 1):
                if wave_tile * tile < n:
                    assert look32(m, w, wave_tile, tile, below), (n, T, tile, pack, blk, out_align, wave_tile, lo, hi)
                    checked += 1
    assert checked > 2000


def test_first_look_is_a_superset_of_the_64_bit_look_and_hardly_larger():
    rng = random.Random(7)
    total = missed = extra = 0
    for _ in range(1500):
        tile, pack, blk, _ = rng.choice(GEOMETRIES)
        n = rng.choice([27264000, rng.randint(tile * 4, 40_000_000), rng.randint(tile * 4, 1 << 31), 1 << 31])
        T = rng.choice([1, 3, 7, 255, 256, 65536, rng.randint(1, 65536)])
        c = look_constants(n, T, tile, pack, blk)
        if c is None:
            continue
        m, w = c
        below, above = margins(pack, blk)
        tiles = {rng.randrange(1, max(2, n // tile)) for _ in range(200)}
        for t in rng.sample(range(T + 1), min(T + 1, 16)):
            tiles |= {x for x in (n * t // T // tile - 1, n * t // T // tile, n * t // T // tile + 1) if 1 <= x < n // tile}
        for wave_tile in tiles:
            old, new = look64(m, wave_tile, tile, below, tile + below + above), look32(m, w, wave_tile, tile, below)
            total += 1
            missed += old and not new
            extra += new and not old
    assert missed == 0
    assert total > 100_000 and extra * 10_000 < total
