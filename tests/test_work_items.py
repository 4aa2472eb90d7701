"""The kernel's work decomposition, replayed on the host (pt_test_work_items: no GPU): a launch's work items and their 64 lanes
through the same indexing code the render kernel runs (pt_shade.h: pt_item_lane, pt_slot_to_pixel). Every sample of every pixel
of the rank's 8x8 tiles inside the slice exactly once, nothing outside, the ranks' shares disjoint and complete, chunk lengths
adding up to SAMPLES - for the sample counts, slices and rank counts the GPU tests and the configs use, and awkward ones."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def H():
    from portrayer_amd import _hip
    return _hip


def replay(H, w, h, rect, samples, rank, ranks):
    p = H.PtRenderParams(w, h, H.PtRect(*rect), samples, 0, H.SAMPLE_RNG, 1, rank, ranks, 0)
    count = np.zeros((h, w), dtype=np.uint32)
    index_sum = np.zeros((h, w), dtype=np.uint64)
    chunk_sum = np.zeros((h, w), dtype=np.uint32)
    n_items = C.c_uint64(0)
    shape = (C.c_uint32 * 3)()
    rc = H.lib().pt_test_work_items(C.byref(p), count.ctypes.data_as(C.POINTER(C.c_uint32)), index_sum.ctypes.data_as(C.POINTER(C.c_uint64)),
                                    chunk_sum.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(n_items), shape)
    assert rc == 0
    return count, index_sum, chunk_sum, n_items.value, tuple(shape)


def own_tiles(w, h, rect, rank, ranks):
    """Mask of the pixels of the slice that lie in 8x8 tiles (over the slice rectangle, row-major) t with t % ranks == rank."""
    x0, y0, x1, y1 = rect
    m = np.zeros((h, w), dtype=bool)
    tiles_x = (x1 - x0 + 1 + 7) // 8
    ys, xs = np.mgrid[y0:y1 + 1, x0:x1 + 1]
    t = ((ys - y0) // 8) * tiles_x + (xs - x0) // 8
    m[y0:y1 + 1, x0:x1 + 1] = (t % ranks) == rank
    return m


@pytest.mark.parametrize("samples", [1, 2, 3, 5, 7, 8, 9, 16, 19, 20, 32, 63, 64, 65, 100, 256])
@pytest.mark.parametrize("w,h,rect", [(40, 24, (0, 0, 39, 23)), (37, 21, (3, 2, 33, 19)), (9, 5, (1, 1, 1, 1)), (64, 8, (0, 0, 63, 7))])
def test_every_sample_of_every_pixel_once(H, samples, w, h, rect):
    count, index_sum, chunk_sum, n_items, (P, Cc, K) = replay(H, w, h, rect, samples, 0, 1)
    inside = own_tiles(w, h, rect, 0, 1)
    assert P * Cc * K == 64 and K == (8 if samples >= 8 else 1 << (samples - 1).bit_length())
    assert np.array_equal(count, np.where(inside, samples, 0))
    assert np.array_equal(index_sum, np.where(inside, samples * (samples - 1) // 2, 0).astype(np.uint64))
    assert np.array_equal(chunk_sum, np.where(inside, samples, 0))
    tiles = ((rect[2] - rect[0] + 8) // 8) * ((rect[3] - rect[1] + 8) // 8)
    groups = -(-(-(-samples // 8)) // Cc)
    assert n_items == tiles * groups * (64 // P)


@pytest.mark.parametrize("ranks", [2, 3, 8])
@pytest.mark.parametrize("samples", [4, 16, 64])
def test_ranks_share_the_slice_without_gap_or_overlap(H, ranks, samples):
    w, h, rect = 157, 93, (11, 5, 149, 90)
    total = np.zeros((h, w), dtype=np.uint32)
    for rank in range(ranks):
        count, index_sum, chunk_sum, _, _ = replay(H, w, h, rect, samples, rank, ranks)
        mine = own_tiles(w, h, rect, rank, ranks)
        assert np.array_equal(count, np.where(mine, samples, 0))
        assert np.array_equal(chunk_sum, np.where(mine, samples, 0))
        total += count
    full = own_tiles(w, h, rect, 0, 1)
    assert np.array_equal(total, np.where(full, samples, 0))


@pytest.mark.parametrize("w,h,rect,samples,ranks", [(8191, 9, (0, 0, 8190, 8), 100, 1),     # 1024 tiles a row (a power of two), 13 chunk groups
                                                  (8185, 17, (3, 0, 8183, 16), 999, 3),   # 1023 tiles a row, 125 chunk groups, three ranks
                                                  (24007, 8, (0, 0, 24006, 7), 24, 7),    # 3001 tiles a row (a prime), seven ranks
                                                  (1, 4099, (0, 0, 0, 4098), 9, 1)])      # one tile a row, 513 rows of tiles
def test_divisions_without_a_divider(H, w, h, rect, samples, ranks):
    """pt_item_lane_fast (what the kernels run: shifts, masks and two multiply-high divisions with constants from the host, PtFastDiv) against
    pt_item_lane (plain / and %): pt_test_work_items fails if a single lane of a single item differs. Divisors that are powers of two, primes,
    one; thousands of tiles a row; hundreds of chunk groups."""
    total = np.zeros((h, w), dtype=np.uint32)
    for rank in range(ranks):
        count, _, chunk_sum, _, _ = replay(H, w, h, rect, samples, rank, ranks)
        assert np.array_equal(count, chunk_sum)
        total += count
    assert np.array_equal(total, np.where(own_tiles(w, h, rect, 0, 1), samples, 0))


@pytest.mark.parametrize("chunks", ["1", "2", "4", "8"])
def test_forced_chunk_layouts(H, monkeypatch, chunks):
    """PORTRAYER_LANE_CHUNKS: every layout of a wavefront the switch allows covers the same samples."""
    monkeypatch.setenv("PORTRAYER_LANE_CHUNKS", chunks)
    for samples in (8, 24, 64, 100):
        count, index_sum, chunk_sum, _, (P, Cc, K) = replay(H, 33, 17, (0, 0, 32, 16), samples, 0, 1)
        assert Cc == int(chunks) and np.all(count == samples) and np.all(index_sum == samples * (samples - 1) // 2) and np.all(chunk_sum == samples)


def test_configuration_sizes(H):
    """BASELINE.json's sizes: one pixel per wavefront at 64 samples, 2.07 M items; the 8-way split of the 4K frame."""
    count, _, _, n_items, shape = replay(H, 1920, 1080, (0, 0, 1919, 1079), 64, 0, 1)
    assert shape == (1, 8, 8) and n_items == 240 * 135 * 64 and np.all(count == 64)
    count, _, _, n_items, shape = replay(H, 1280, 720, (0, 0, 1279, 719), 16, 0, 1)
    assert shape == (4, 2, 8) and np.all(count == 16)
    count, _, _, n_items, shape = replay(H, 3840, 2160, (0, 0, 3839, 2159), 256, 3, 8)
    assert shape == (1, 8, 8) and np.array_equal(count > 0, own_tiles(3840, 2160, (0, 0, 3839, 2159), 3, 8))
