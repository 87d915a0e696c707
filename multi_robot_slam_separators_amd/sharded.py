"""One find-and-verify step of ONE robot pair sharded over the G GPUs of a node, exactly as SURVEY.md section 8(e)
writes it (the reference runs this step on one CPU core: find_separators.py:59-133):

  NN stage       the LOCAL rows of the query (data_handler.py:166-189) are cut into G contiguous blocks; rank r
                 searches block r against the replicated received database and emits per-row (distance, column)
                 minima; ONE all-gather of those; every rank then runs the identical sequential walk
                 (data_handler.py:191-205) on the gathered minima, so all ranks hold the same candidate list.
  verification   candidate p goes to rank p mod G (`dist.shard_pairs`) over a REPLICATED keyframe store
                 (stereoCamGeometricTools.cpp:122-178 is stateless per call).
  exchange       one all-gather of the per-candidate success flags (1 byte each) + ONE all-gather of the accepted
                 separator records (`dist.RecordExchange`); `interleave` puts the accepted records back into the
                 candidate order of the walk, so the node's output is byte-identical to a single GPU's.

The compute is behind a small backend interface so that the SAME orchestration runs on the GPUs (bench.py, the
library) and, in the CPU tests, on the oracle with gloo ranks:

  backend.row_minima(lo, hi) -> (float64[hi-lo], int32[hi-lo])   minima of local rows [lo, hi) (+inf: none < thr)
  backend.walk(row_min, row_arg) -> structured MATCH_DTYPE array  the replicated walk
  backend.verify(matches) -> uint8 tensor [len, RESULT bytes]     results of these candidates, on `device`

N robot pairs (BASELINE configs[4]: 5 robots = 10 robot pairs) are flattened into one candidate list by
`flatten_candidates` before the round-robin.
"""
import numpy as np

from . import _abi, dist


def row_blocks(n_local, world):
    """Contiguous block [lo, hi) of local rows per rank."""
    return [(r * n_local // world, (r + 1) * n_local // world) for r in range(world)]


def flatten_candidates(per_robot_pair):
    """[(robot_pair_id, matches)] -> (pair_id int32[n], idx_local int32[n], idx_other int32[n]): the candidates of
    all robot pairs in robot-pair order, each list in its walk order -- the list that is sharded p mod G."""
    ids, il, io = [], [], []
    for rp, m in per_robot_pair:
        ids.append(np.full(len(m), rp, dtype=np.int32))
        il.append(np.asarray(m["idx_local"], dtype=np.int32))
        io.append(np.asarray(m["idx_other"], dtype=np.int32))
    if not ids:
        z = np.zeros(0, np.int32)
        return z, z, z
    return np.concatenate(ids), np.concatenate(il), np.concatenate(io)


def interleave(flags_by_rank, recs_by_rank, world):
    """Accepted records of every rank (each in its own candidate order) -> the global candidate order.
    flags_by_rank[r]: success flags of rank r's candidates p = r, r + G, ...; recs_by_rank[r]: its accepted records.
    Returns (flags [n] in candidate order, accepted records in candidate order)."""
    import torch
    n = sum(int(f.numel()) for f in flags_by_rank)
    flags = torch.zeros(n, dtype=torch.bool)
    for r in range(world):
        flags[r::world] = flags_by_rank[r].cpu().bool()
    rec_bytes = recs_by_rank[0].shape[1]
    out = torch.empty((int(flags.sum()), rec_bytes), dtype=torch.uint8, device=recs_by_rank[0].device)
    pos = torch.cumsum(flags.to(torch.int64), 0) - 1          # rank of every accepted candidate among the accepted
    for r in range(world):
        mine = flags[r::world]
        dst = pos[r::world][mine]
        if dst.numel():
            out[dst.to(out.device)] = recs_by_rank[r][: dst.numel()]
    return flags, out


class ShardedStep:
    """Orchestration of one sharded step (see the module docstring).  `coll_device`: where the collectives run
    (the GPU for RCCL, "cpu" for gloo)."""

    def __init__(self, backend, rank, world, n_local, coll_device, group=None, accept_cap=None):
        import torch
        self.b, self.rank, self.world, self.n_local = backend, rank, world, int(n_local)
        self.group = group
        self.dev = torch.device(coll_device)
        self.blocks = row_blocks(self.n_local, world)
        self.rec_bytes = _abi.RESULT_DTYPE.itemsize
        self.max_block = max(hi - lo for lo, hi in self.blocks)
        cap = accept_cap if accept_cap is not None else self.n_local // world + 256
        self.exch = dist.RecordExchange(self.rec_bytes, max(self.n_local, 1), cap, self.dev, group) if world > 1 else None
        self.off_success = _abi.RESULT_DTYPE.fields["success"][1]

    def find_matches(self):
        """Row-sharded NN stage -> the candidate list (identical on every rank)."""
        import torch
        lo, hi = self.blocks[self.rank]
        d, a = self.b.row_minima(lo, hi)
        if self.world > 1:
            td = dist._dist()
            # fixed-size blocks (padded to the largest) so that ONE all_gather_into_tensor carries the minima
            send = torch.zeros((self.max_block, 12), dtype=torch.uint8)
            send[: hi - lo, :8] = torch.from_numpy(np.ascontiguousarray(d)).view(torch.uint8).reshape(-1, 8)
            send[: hi - lo, 8:] = torch.from_numpy(np.ascontiguousarray(a)).view(torch.uint8).reshape(-1, 4)
            send = send.to(self.dev)
            recv = torch.empty((self.world * self.max_block, 12), dtype=torch.uint8, device=self.dev)
            td.all_gather_into_tensor(recv, send, group=self.group)
            recv = recv.cpu().view(self.world, self.max_block, 12)
            ds, as_ = [], []
            for r, (l, h) in enumerate(self.blocks):
                ds.append(recv[r, : h - l, :8].contiguous().view(torch.float64).reshape(-1).numpy())
                as_.append(recv[r, : h - l, 8:].contiguous().view(torch.int32).reshape(-1).numpy())
            d, a = np.concatenate(ds), np.concatenate(as_)
        return self.b.walk(d, a)

    def verify(self, matches):
        """Candidates p mod G -> (flags [n] bool in candidate order, accepted records in candidate order on the
        collective device).  `matches`: the full candidate list (identical on every rank)."""
        import torch
        n = len(matches)
        mine = dist.shard_pairs(n, self.rank, self.world)
        res = self.b.verify(matches[mine])                           # uint8 [len(mine), rec_bytes]
        ok = res[:, self.off_success] != 0 if len(mine) else torch.zeros(0, dtype=torch.bool, device=res.device)
        acc = res[ok]
        if self.world == 1:
            return ok.cpu(), acc
        td = dist._dist()
        per = (n + self.world - 1) // self.world
        fs = torch.zeros(per, dtype=torch.uint8, device=self.dev)
        fs[: len(mine)] = ok.to(self.dev).to(torch.uint8)
        fr = torch.empty(self.world * per, dtype=torch.uint8, device=self.dev)
        td.all_gather_into_tensor(fr, fs, group=self.group)
        n_acc = int(acc.shape[0])
        self.exch.payload[:n_acc].copy_(acc.to(self.dev))
        self.exch.exchange(n_acc)
        if self.dev.type == "cuda":
            torch.cuda.synchronize()
        allrec, counts = self.exch.all_gathered()
        fr = fr.view(self.world, per)
        flags_by_rank = [fr[r, : len(dist.shard_pairs(n, r, self.world))] for r in range(self.world)]
        recs_by_rank, off = [], 0
        for r in range(self.world):
            recs_by_rank.append(allrec[off: off + counts[r]])
            off += counts[r]
        return interleave(flags_by_rank, recs_by_rank, self.world)

    def step(self):
        m = self.find_matches()
        flags, acc = self.verify(m)
        return m, flags, acc
