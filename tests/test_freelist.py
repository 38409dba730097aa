"""The offset bookkeeping of the device arena (raven_amd/csrc/freelist.h through rvn_test_freelist of libraven_hip_test.so;
no GPU): random allocate / give-back sequences against a plain Python model — blocks never overlap, stay inside the
arena, are multiples of the grain, the lowest hole that fits is taken, neighbours coalesce (everything given back = one
hole of the whole size), giving back twice or giving back what never was a block is refused."""
import ctypes as C

import numpy as np

from raven_amd import hip


def _run(size, grain, ops):
    T = hip.test_lib()
    ops = np.asarray(ops, dtype=np.int64)
    out = np.zeros(len(ops), dtype=np.int64)
    state = np.zeros(3, dtype=np.uint64)
    rc = T.rvn_test_freelist(size, grain, ops.ctypes.data_as(C.c_void_p), len(ops), out.ctypes.data_as(C.c_void_p),
                             state.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out, [int(x) for x in state]


def _model(size, grain, ops):
    """first fit over a sorted hole list, coalescing on release"""
    size = size // grain * grain
    holes = [(0, size)] if size else []
    live, out = {}, []
    for i, op in enumerate(ops):
        if op > 0:
            need = max(grain, (op + grain - 1) // grain * grain)
            got = -1
            for h, (o, ln) in enumerate(holes):
                if ln >= need:
                    got = o
                    holes[h:h + 1] = [(o + need, ln - need)] if ln > need else []
                    live[i] = (o, need)
                    break
            out.append(got)
        else:
            j = -op
            if j in live:
                o, ln = live.pop(j)
                holes.append((o, ln))
                holes.sort()
                merged = []
                for a, b in holes:
                    if merged and merged[-1][0] + merged[-1][1] == a:
                        merged[-1] = (merged[-1][0], merged[-1][1] + b)
                    else:
                        merged.append((a, b))
                holes = merged
                out.append(1)
            else:
                out.append(0)
    return out, [sum(b for _, b in holes), max([b for _, b in holes] or [0]), len(live)]


def test_random_sequences_match_the_model():
    rng = np.random.default_rng(3)
    for trial in range(30):
        grain = int(rng.choice([1, 64, 4096, 65536]))
        size = int(rng.integers(50, 4000)) * grain + int(rng.integers(0, grain))
        ops, allocs = [], []
        for i in range(int(rng.integers(50, 600))):
            if allocs and rng.random() < 0.45:
                j = int(rng.choice(allocs))
                ops.append(-j)
                if rng.random() < 0.9:
                    allocs.remove(j)  # (otherwise it is given back twice later: must be refused)
            else:
                if i == 0:
                    ops.append(int(rng.integers(1, 20 * grain)))  # operation 0 cannot be named by a give-back
                    continue
                ops.append(int(rng.integers(1, max(2, size // 6))))
                allocs.append(i)
        out, state = _run(size, grain, ops)
        want, wstate = _model(size, grain, ops)
        assert list(out) == want, trial
        assert state == wstate, trial
        # blocks in use are disjoint, aligned and inside
        live = {}
        for i, op in enumerate(ops):
            if op > 0 and out[i] >= 0:
                live[i] = (int(out[i]), max(grain, (op + grain - 1) // grain * grain))
            elif op <= 0 and out[i] == 1:
                live.pop(-op)
            spans = sorted(live.values())
            for (a, la), (b, _) in zip(spans, spans[1:]):
                assert a + la <= b
            assert all(o % grain == 0 and o + ln <= size // grain * grain for o, ln in spans)


def test_everything_given_back_is_one_hole_again():
    grain = 65536
    size = 1000 * grain
    ops = [3 * grain, 5 * grain + 1, grain, 100 * grain, 7]           # operations 0 .. 4
    ops += [-1, -3, -2, -4]                                            # in an order that needs both-side coalescing
    out, state = _run(size, grain, ops)
    assert list(out[:5]) == [0, 3 * grain, 9 * grain, 10 * grain, 110 * grain]
    assert list(out[5:]) == [1, 1, 1, 1]
    assert state == [size - 3 * grain, size - 3 * grain, 1]
    out, state = _run(size, grain, ops + [3 * grain])                  # a hole of exactly three grains at offset 3 grains? no: first fit
    assert out[-1] == 3 * grain
    out, state = _run(size, grain, [size + 1, size, -1, -1, 0])
    assert list(out) == [-1, 0, 1, 0, 0] and state == [size, size, 0]
