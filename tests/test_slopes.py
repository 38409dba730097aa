"""Pile::FindChimericRegions as the kernel runs it (raven_amd/csrc/slopes.h through rvn_test_find_chimeric_regions, host
side of the same __host__ __device__ code: windowed scans instead of deques) against the oracle's restatement of
RavenLib/src/pile.cc:176-187, :373-400, :403-600 with the reference's own deques / std::sort / vectors, on synthetic
coverage profiles: plateaus with pits (chimeric junctions), spikes (repeats), ramps, noise, zeroed ends."""
import numpy as np

from oracle import oracle
from raven_amd import hip


def _profile(rng, cells):
    base = int(rng.integers(8, 60))
    d = np.full(cells, base, dtype=np.int64)
    d += rng.integers(-2, 3, size=cells)
    for _ in range(int(rng.integers(0, 5))):  # pits: coverage drops (chimeric junctions), various widths / depths
        c, wdt = int(rng.integers(60, cells - 60)), int(rng.integers(1, 40))
        depth = rng.choice([0.05, 0.2, 0.45, 0.6])
        lo, hi = max(0, c - wdt), min(cells, c + wdt)
        d[lo:hi] = (d[lo:hi] * depth).astype(np.int64)
    for _ in range(int(rng.integers(0, 4))):  # spikes (repeats)
        c, wdt = int(rng.integers(60, cells - 60)), int(rng.integers(3, 80))
        d[max(0, c - wdt):min(cells, c + wdt)] *= int(rng.integers(2, 5))
    if rng.random() < 0.5:  # ramps at the ends, as real piles have
        r = int(rng.integers(10, 60))
        d[:r] = (d[:r] * np.linspace(0.1, 1, r)).astype(np.int64)
        d[-r:] = (d[-r:] * np.linspace(1, 0.1, r)).astype(np.int64)
    if rng.random() < 0.5:  # zeroed outside the valid region (UpdateValidRegion)
        a, b = int(rng.integers(0, 30)), int(rng.integers(0, 30))
        d[:a] = 0
        if b:
            d[-b:] = 0
    if rng.random() < 0.1:
        d[rng.integers(0, cells, size=5)] = 65535  # saturated cells: the clamp matters
    return np.clip(d, 0, 65535).astype(np.uint16)


def test_find_chimeric_regions_matches_the_restatement_of_pile_cc():
    rng = np.random.default_rng(2026)
    n_regions = 0
    n_with = 0
    for trial in range(3000):
        cells = int(rng.integers(130, 1500))
        d = _profile(rng, cells)
        got = hip.test_find_chimeric_regions(d)
        want = oracle.find_chimeric_regions(d)
        assert got.shape == want.shape and np.array_equal(got, want), (trial, got, want)
        n_regions += got.shape[0]
        n_with += got.shape[0] > 0
    assert n_with > 500 and n_regions > 700  # the generator does produce pits the rule accepts


def test_flat_and_degenerate_profiles():
    for d in (np.full(100, 30, np.uint16), np.zeros(200, np.uint16), np.arange(300, dtype=np.uint16),
              np.arange(300, dtype=np.uint16)[::-1].copy(), np.full(90, 65535, np.uint16)):
        assert np.array_equal(hip.test_find_chimeric_regions(d), oracle.find_chimeric_regions(d))
    d = np.full(400, 40, np.uint16)
    d[200:203] = 3  # one clean pit
    got = hip.test_find_chimeric_regions(d)
    assert got.shape[0] == 1 and got[0, 0] <= 200 and got[0, 1] >= 202
    assert np.array_equal(got, oracle.find_chimeric_regions(d))
