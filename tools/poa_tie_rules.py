#!/usr/bin/env python3
"""Which tie decides the consensus differences between the device's row order and spoa's DFS rank, and what
a device rule independent of the row order would give.  CPU only: the POA oracle in both orders (oracle.poa_window(...,
device_order=True) = the device's incremental order, equal to the device on 20 000 windows: tools/poa_parity.py) on the
windows of tools/poa_parity.py's seeded set.
    python tools/poa_tie_rules.py [n_windows] [threads]
1. windows whose consensus differs between the two orders;
2. of those: is it the order during the ALIGNMENTS (end node among equal scores, Subgraph rows) or in the CONSENSUS
   (start of the heaviest bundle, branch completion) that decides;
3. the same comparison with the end node of an alignment chosen by SMALLEST NODE ID among equal scores instead of by
   rank (oracle.poa_window(..., end_tie=1): since round 5 the device kernels' rule) in the device-order run."""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import poa_parity as pp  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    rng = np.random.default_rng(20260927)
    wins = [pp.make_window(rng)[0] for _ in range(n)]

    def cons(i, device_order, end_tie=0, order_where=0):
        w = wins[i]
        return oracle.poa_window(w["layers"], begins=w["begins"], ends=w["ends"], device_order=device_order, end_tie=end_tie,
                                 order_where=order_where)[0]

    def run(idx, device_order, end_tie=0, order_where=0):
        with ThreadPoolExecutor(max_workers=threads) as ex:  # (the oracle releases the GIL inside its C++ call)
            return list(ex.map(lambda i: cons(i, device_order, end_tie, order_where), idx))

    t0 = time.time()
    every = list(range(n))
    spoa = run(every, False)
    dev = run(every, True)
    differ = [i for i in every if not np.array_equal(spoa[i], dev[i])]
    out = {"windows": n, "differ_between_the_orders": len(differ), "which": differ}
    by_alignment = by_consensus = 0
    for where, name in ((1, "alignments"), (2, "consensus")):
        part = run(differ, True, 0, where)
        same = sum(bool(np.array_equal(part[k], dev[i])) for k, i in enumerate(differ))
        out["device_order_in_the_%s_alone_gives_the_device_consensus" % name] = same
    dev_id = run(every, True, 1)
    left = [i for i in every if not np.array_equal(spoa[i], dev_id[i])]
    out["differ_with_end_node_by_smallest_id"] = len(left)
    out["which_with_end_node_by_smallest_id"] = left
    out["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
