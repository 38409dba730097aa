import sys, time, torch
sys.path.insert(0, '.')
from raven_amd import hip, synth
dev = torch.device('cuda', 0)
g = synth.make_genome_torch(5_000_000, seed=11, device=dev)
rs, truth = synth.make_reads_torch(g, 30, 10000, length_model='fixed', sub=0.04, ins=0.03, dele=0.03, seed=12)
eng = hip.Engine(15, 5)
rd = eng.upload(rs)
for timing in (False, True):
    eng.set_kernel_timing(timing)
    for i in range(6):
        eng.reset_stats()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p = eng.find_overlaps_and_create_piles(rd)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        p.close()
        km = sorted(eng.kernel_ms().items(), key=lambda x: -x[1][0])
        print(timing, i, round(dt * 1e3, 2), 'ms', sum(v[0] for k, v in km) if timing else '', [(k, round(v[0], 2), v[1]) for k, v in km[:6]] if timing else '')
