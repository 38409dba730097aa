"""Test helpers for the sharded pass: an in-process communicator (G threads = G virtual ranks, each with its own
engine) and the comparison with the single-GPU result."""
import threading

import numpy as np


class LocalGroup:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world

    def comm(self, rank):
        return LocalComm(self, rank)


class LocalComm:
    """Same interface as raven_amd.sharded.Comm, exchanging through shared memory between threads."""

    def __init__(self, group, rank):
        self.g, self.rank, self.world = group, rank, group.world
        self.bytes_sent = 0

    def all_to_all_v(self, parts):
        self.g.slots[self.rank] = parts
        self.g.barrier.wait()
        res = [np.array(self.g.slots[s][self.rank], copy=True) for s in range(self.world)]
        self.bytes_sent += sum(p.nbytes for i, p in enumerate(parts) if i != self.rank)
        self.g.barrier.wait()
        return res

    def all_to_all_t(self, parts):
        """torch tensors (same device) between threads."""
        self.g.slots[self.rank] = parts
        self.g.barrier.wait()
        res = [self.g.slots[s][self.rank].clone() for s in range(self.world)]
        import torch
        torch.cuda.synchronize()  # the copies must have happened before the senders may recycle their buffers
        self.bytes_sent += sum(8 * int(p.shape[0]) for i, p in enumerate(parts) if i != self.rank)
        self.g.barrier.wait()
        return res

    def all_to_all_flat_t(self, flat, send_lens):
        """DeviceComm.all_to_all_flat_t: parts back to back in one tensor; returns (flat receive tensor, lengths)."""
        import torch
        send_lens = [int(x) for x in send_lens]
        parts = list(torch.split(flat[:sum(send_lens)], send_lens))
        res = self.all_to_all_t(parts)
        lens = [int(x.shape[0]) for x in res]
        return (torch.cat(res) if res else flat[:0]), lens

    def all_reduce_sum(self, a):
        parts = self.all_to_all_v([a] * self.world)
        return np.sum(parts, axis=0)

    def all_gather_v(self, a):
        return np.concatenate(self.all_to_all_v([a] * self.world))


def run_ranks(world, fn):
    """fn(rank, comm) in `world` threads; returns the list of results, re-raising the first exception."""
    group = LocalGroup(world)
    out, err = [None] * world, [None] * world

    def work(r):
        try:
            out[r] = fn(r, group.comm(r))
        except BaseException as ex:  # noqa: BLE001
            err[r] = ex
            group.barrier.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for ex in err:
        if ex is not None and not isinstance(ex, threading.BrokenBarrierError):
            raise ex
    for ex in err:
        if ex is not None:
            raise ex
    return out


def check_against_single(res, data, poff, kept, koff):
    """A rank's slice (raven_amd.sharded result dict) vs the single-GPU pass arrays."""
    lo, hi = res["lo"], res["hi"]
    assert np.array_equal(res["pile_off"], poff[lo:hi + 1] - poff[lo])
    assert np.array_equal(res["pile_data"], data[int(poff[lo]):int(poff[hi])])
    assert np.array_equal(res["overlap_off"], koff[lo:hi + 1] - koff[lo])
    assert np.array_equal(res["overlaps"], kept[int(koff[lo]):int(koff[hi])])


def free_port():
    """A TCP port nobody listens on right now (rendezvous of the multi-process tests)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
