"""Shared builder for the polishing tests: a genome, a noisy draft (the 'unitig'), ONT-like reads with optional
per-base qualities."""
import numpy as np

from raven_amd import seqio, synth


mutate = synth.mutate


def make_case(genome_len=30_000, coverage=25, read_len=3000, draft_err=(0.01, 0.008, 0.008), seed=5, with_qual=False,
              n_targets=1):
    rng = np.random.default_rng(seed)
    g = synth.make_genome(genome_len, seed=seed + 100)
    bounds = np.linspace(0, genome_len, n_targets + 1).astype(int)
    truths = [g[bounds[i]:bounds[i + 1]] for i in range(n_targets)]
    drafts = [mutate(rng, t, *draft_err) for t in truths]
    targets = seqio.pack_reads(drafts)
    reads, _ = synth.make_reads(g, coverage, read_len, seed=seed + 200)
    quals = None
    if with_qual:
        quals = [np.full(int(n), 33 + 12, dtype=np.uint8) for n in reads.lengths]
    return truths, drafts, targets, reads, quals
