import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

from oracle import seqio_oracle  # noqa: E402  (the independent FASTA / FASTQ parser: test infrastructure)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Build the HIP library and the oracle when a fresh checkout has no binaries yet (hipcc cross-compiles
    gfx950 without a GPU; the binaries are git-ignored but travel with gpurun snapshots)."""
    need = [os.path.join(ROOT, "raven_amd", "lib", "libraven_hip.so"), os.path.join(ROOT, "oracle", "libraven_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()
    # The device-resident sharded pass hands torch CUDA tensors to libraven_hip in the same process.  torch ships its
    # own HIP runtime; it has to initialise BEFORE libraven_hip touches the GPU (the order bench.py has anyway),
    # otherwise torch reports "No HIP GPUs are available".
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # no torch / no GPU: the CPU suite does not need it
        pass


@pytest.fixture(scope="session")
def lambda_reads():
    from raven_amd import seqio
    return seqio_oracle.load_reads(os.path.join(GOLDEN, "ERA476754.fastq.gz"))


@pytest.fixture(scope="session")
def lambda_genome():
    from raven_amd import seqio
    return seqio_oracle.load_reads(os.path.join(GOLDEN, "NC_001416.fasta.gz"))


@pytest.fixture(scope="session")
def synth_small():
    """200 kb genome, 20x, 8 kb ONT-like reads (seeded) + truth."""
    from raven_amd import synth
    g = synth.make_genome(200_000, seed=11)
    rs, truth = synth.make_reads(g, 20, 8000, seed=12)
    return g, rs, truth
