"""Input path alone on the GPU box: rvn_reads_load of a synthetic FASTQ (N(15 kb, 3 kb) reads, Phred-10 qualities) as
blocked gzip, as one gzip member and as plain text; every file twice (the second load finds the page-locked slabs of
the first).  python tools/bench_io.py [Mbase] [io_threads=N] [io_slab_mb=N] [io_ring=N] [io_zlib=1]  (engine options, include/raven_hip.h)"""
import gzip
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raven_amd import hip, seqio  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if "=" not in a]
    opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
    mbase = int(args[0]) if args else 150
    rng = np.random.default_rng(1)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    recs, tot = [], 0
    while tot < mbase * 1_000_000:
        ln = max(500, int(rng.normal(15000, 3000)))
        recs.append(b"@r%d\n" % len(recs) + lut[rng.integers(0, 4, ln)].tobytes() + b"\n+\n" + b"+" * ln + b"\n")
        tot += ln
    text = b"".join(recs)
    n_reads = len(recs)
    del recs
    tmpd = tempfile.mkdtemp(prefix="rvn_io_")
    eng = hip.Engine(15, 5)
    for name, value in opts.items():
        eng.set_option(name, int(value))
    out = {"cpu_count": os.cpu_count(), "bases": tot, "reads": n_reads, "text_bytes": len(text), "options": opts}
    for tag, make in (("bgzf", lambda: seqio.bgzf_compress(text, 1)), ("single_member", lambda: gzip.compress(text, 1)),
                      ("plain", lambda: text)):
        path = os.path.join(tmpd, "reads_%s.fastq%s" % (tag, "" if tag == "plain" else ".gz"))
        blob = make()
        with open(path, "wb") as f:
            f.write(blob)
        runs = []
        for rep in range(2):
            t0 = time.perf_counter()
            rd = eng.load(path)
            dt = time.perf_counter() - t0
            st = rd.load_stats
            assert st["n_bases"] == tot and rd.n == n_reads
            runs.append({"load_s": round(dt, 4), "gbase_per_s": round(tot / dt / 1e9, 3), "scan_s": round(st["parse_s"], 4),
                         "device_s": round(st["device_s"], 4), "threads": st["inflate_threads"], "members": st["members"],
                         "streaming": st["streaming"]})
            rd.close()
        out[tag] = {"file_bytes": len(blob), "first": runs[0], "second": runs[1]}
        os.remove(path)
    os.rmdir(tmpd)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
