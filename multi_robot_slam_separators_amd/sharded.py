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

The step keeps its data on the device between the stages, like the single-GPU step (sf_step_issue / sf_step_retire),
and since round 4 a rank's step has ONE host wait, at its end:

  * the rank's minima are written by the NN kernels straight into the all-gather's send block (GPU backend:
    sf_nn_row_minima_device -- no walk that is thrown away, no candidate list on the host);
  * the gathered blocks are unpacked on the device and the walk runs THERE (sf_nn_walk_device, data_handler.py:191-205:
    every rank the identical walk on identical minima); the rank's candidates p = rank, rank + G, ... are taken from the
    device match list, slots past the device-side match count are void pairs;
  * the verification's accepted records are compacted straight into the record exchange's payload, their count is
    stamped into its header on the device, and the per-candidate success flags ride in the same block (the `extra`
    region of dist.RecordExchange): ONE collective, mirrored into pinned host memory together with the match list;
  * the wait is the caller's synchronisation behind `finish`; the interleave is then host arithmetic on the pinned
    mirror.  What the device cannot decide -- a rank's candidate set too dense for the filter level (status word) --
    shows in the mirrored status after that wait; the step is then run again in round 3's form (`find_matches` +
    `verify`: the ranks concerned recompute synchronously, the minima are gathered once more, host walk).

The compute is behind a small backend interface so that the SAME orchestration runs on the GPUs (bench.py, the
library) and, in the CPU tests, on the oracle with gloo ranks.  All tensors are on the collective device:

  backend.row_minima_into(row_min f64[n], row_arg i32[n], status i32[1])   this rank's block, asynchronous;
                                           status 1 = "too dense for the device path" (every rank then sees it)
  backend.row_minima_sync() -> (float64[n], int32[n])   the synchronous form that cannot fail (taken on status 1)
  backend.walk(row_min, row_arg) -> structured MATCH_DTYPE array  the replicated walk (host; the fallback path)
  backend.verify_into(matches, payload u8[rows, RESULT bytes], count i32[1] view, flags u8[len])
                                           results of these candidates: accepted records compacted into `payload` in
                                           candidate order, their number into `count`, success flags into `flags`
  backend.walk_into(row_min f64[n_local], row_arg i32[n_local], status i32[1], matches u8[n_local, 16], n i32[2])
                                           the replicated walk on tensors of the collective device, asynchronous:
                                           MATCH_DTYPE rows + n[0] = their number, n[1] = status (non-zero voids it)
  backend.verify_mine_into(matches u8[n_local, 16], n i32[2], rank, world, max_mine, payload, count, flags u8[max_mine])
                                           the same for the candidates p = rank, rank + world, ... of the DEVICE list
  backend.sync()                           wait for the work queued so far (the step's final synchronisation)

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
    flags_by_rank[r]: uint8/bool numpy, success flags of rank r's candidates p = r, r + G, ...; recs_by_rank[r]: numpy
    uint8 [n_accepted_r, record bytes].  Returns (bool[n] in candidate order, accepted records in candidate order)."""
    n = sum(int(f.size) for f in flags_by_rank)
    flags = np.zeros(n, dtype=bool)
    for r in range(world):
        flags[r::world] = flags_by_rank[r].astype(bool)
    rec_bytes = recs_by_rank[0].shape[1]
    out = np.empty((int(flags.sum()), rec_bytes), dtype=np.uint8)
    pos = np.cumsum(flags) - 1                          # rank of every accepted candidate among the accepted
    for r in range(world):
        dst = pos[r::world][flags[r::world]]
        if dst.size:
            out[dst] = recs_by_rank[r][: dst.size]
    return flags, out


class ShardedStep:
    """Orchestration of one sharded step (see the module docstring).  `coll_device`: where the collectives run
    (the GPU for RCCL, "cpu" for gloo).  Without a process group (one rank) the collectives are copies."""

    def __init__(self, backend, rank, world, n_local, coll_device, group=None, accept_cap=None):
        import torch
        self.b, self.rank, self.world, self.n_local = backend, rank, world, int(n_local)
        self.group = group
        self.dev = torch.device(coll_device)
        self.td = dist.collective(group)
        assert self.td.get_world_size(group) == world
        self.blocks = row_blocks(self.n_local, world)
        self.rec_bytes = _abi.RESULT_DTYPE.itemsize
        mb = self.max_block = max(1, max(hi - lo for lo, hi in self.blocks))
        # the minima block of one rank: float64[mb] | int32[mb] | int32 status (+ padding to 16 bytes)
        self.min_bytes = (mb * 12 + 4 + 15) // 16 * 16
        self.min_send = torch.zeros(self.min_bytes, dtype=torch.uint8, device=self.dev)
        self.min_recv = torch.empty(world * self.min_bytes, dtype=torch.uint8, device=self.dev)
        self.h_min = torch.zeros(world * self.min_bytes, dtype=torch.uint8)
        self.ev_min = None
        if self.dev.type == "cuda":
            self.h_min = self.h_min.pin_memory()
            self.ev_min = torch.cuda.Event()
        self.v_min = self.min_send[: mb * 8].view(torch.float64)
        self.v_arg = self.min_send[mb * 8: mb * 12].view(torch.int32)
        self.v_status = self.min_send[mb * 12: mb * 12 + 4].view(torch.int32)
        self.max_mine = (self.n_local + world - 1) // world          # candidates of one rank, at most
        cap = accept_cap if accept_cap is not None else self.n_local // world + 256
        self.exch = dist.RecordExchange(self.rec_bytes, max(self.max_mine, 1), cap, self.dev, group,
                                        extra_bytes=max(self.max_mine, 1), host_mirror=True)
        self.v_count = self.exch.send[0, :4].view(torch.int32)
        self.waits = 0                                               # host waits of the last step (2 when all is well)

    def _gather_minima(self):
        """all-gather of the minima blocks -> pinned host copy; ONE wait.  Returns (d, a, statuses)."""
        import torch
        mb = self.max_block
        self.td.all_gather_into_tensor(self.min_recv, self.min_send, group=self.group)
        self.h_min.copy_(self.min_recv, non_blocking=True)
        if self.ev_min is not None:
            self.ev_min.record()
            self.ev_min.synchronize()
        self.waits += 1
        h = self.h_min.numpy().reshape(self.world, self.min_bytes)
        ds, as_, st = [], [], []
        for r, (lo, hi) in enumerate(self.blocks):
            ds.append(h[r, : mb * 8].view(np.float64)[: hi - lo])
            as_.append(h[r, mb * 8: mb * 12].view(np.int32)[: hi - lo])
            st.append(int(h[r, mb * 12: mb * 12 + 4].view(np.int32)[0]))
        return np.concatenate(ds), np.concatenate(as_), st

    def find_matches(self):
        """Row-sharded NN stage -> the candidate list (identical on every rank)."""
        import torch
        lo, hi = self.blocks[self.rank]
        n = hi - lo
        self.waits = 0
        if n:
            self.b.row_minima_into(self.v_min[:n], self.v_arg[:n], self.v_status)
        else:
            self.v_status.zero_()
        d, a, st = self._gather_minima()
        if any(st):
            # some rank's candidate set was too dense for its device path: EVERY rank sees that in the gathered
            # block, the ranks concerned recompute synchronously, and all of them gather once more
            if st[self.rank]:
                dd, aa = self.b.row_minima_sync()
                self.v_min[:n].copy_(torch.from_numpy(np.ascontiguousarray(dd)))
                self.v_arg[:n].copy_(torch.from_numpy(np.ascontiguousarray(aa)))
                self.v_status.zero_()
            d, a, st = self._gather_minima()
            assert not any(st)
        return self.b.walk(d, a)

    def verify(self, matches):
        """Candidates p mod G -> (flags bool[n] in candidate order, accepted records uint8[n_acc, RESULT bytes] in
        candidate order), host arrays.  `matches`: the full candidate list (identical on every rank)."""
        n = len(matches)
        assert n <= self.max_mine * self.world
        mine = dist.shard_pairs(n, self.rank, self.world)
        self.b.verify_into(matches[mine], self.exch.payload, self.v_count, self.exch.extra[: len(mine)])
        self.exch.exchange(None, finish=True)
        self.b.sync()                                   # the step's final wait
        self.waits += 1
        return self._collect(n)

    def _collect(self, n):
        """Behind the step's synchronisation: flags + accepted records of every rank from the pinned mirror, interleaved
        into the candidate order."""
        counts = self.exch.counts(limit=self.max_mine)
        if max(counts) > self.exch.cap:                 # (rare) some rank accepted more than the block holds
            allrec, counts = self.exch.all_gathered()
            allrec = allrec.cpu().numpy()
            recs, off = [], 0
            for r in range(self.world):
                recs.append(allrec[off: off + counts[r]])
                off += counts[r]
        else:
            recs = [self.exch.host_gathered(r, counts) for r in range(self.world)]
        flags_by_rank = [self.exch.host_gathered_extra(r)[: len(dist.shard_pairs(n, r, self.world))]
                         for r in range(self.world)]
        return interleave(flags_by_rank, recs, self.world)

    def _unpack_minima(self):
        """The gathered per-rank blocks -> contiguous minima of all local rows + the worst status, on the device."""
        import torch
        mb = self.max_block
        g = self.min_recv.view(self.world, self.min_bytes)
        if self.world == 1:
            lo, hi = self.blocks[0]
            return g[0, : mb * 8].view(torch.float64)[: hi - lo], g[0, mb * 8: mb * 12].view(torch.int32)[: hi - lo], \
                g[0, mb * 12: mb * 12 + 4].view(torch.int32)
        d = torch.cat([g[r, : mb * 8].view(torch.float64)[: hi - lo] for r, (lo, hi) in enumerate(self.blocks)])
        a = torch.cat([g[r, mb * 8: mb * 12].view(torch.int32)[: hi - lo] for r, (lo, hi) in enumerate(self.blocks)])
        st = g[:, mb * 12: mb * 12 + 4].contiguous().view(torch.int32).max().reshape(1)
        return d, a, st

    def step(self):
        """One step with ONE host wait (see the module docstring); falls back to find_matches() + verify() when some
        rank's candidate set was too dense for its device path."""
        import torch
        if not hasattr(self.b, "walk_into"):            # (a backend without the device walk: round 3's form)
            m = self.find_matches()
            flags, acc = self.verify(m)
            return m, flags, acc
        lo, hi = self.blocks[self.rank]
        n = hi - lo
        self.waits = 0
        if n:
            self.b.row_minima_into(self.v_min[:n], self.v_arg[:n], self.v_status)
        else:
            self.v_status.zero_()
        self.td.all_gather_into_tensor(self.min_recv, self.min_send, group=self.group)
        d, a, st = self._unpack_minima()
        if not hasattr(self, "d_matches"):
            self.d_matches = torch.zeros((max(self.n_local, 1), _abi.MATCH_DTYPE.itemsize), dtype=torch.uint8, device=self.dev)
            self.d_n = torch.zeros(2, dtype=torch.int32, device=self.dev)
            self.h_matches = torch.zeros_like(self.d_matches, device="cpu")
            self.h_n = torch.zeros(2, dtype=torch.int32)
            if self.dev.type == "cuda":
                self.h_matches, self.h_n = self.h_matches.pin_memory(), self.h_n.pin_memory()
        self.b.walk_into(d, a, st, self.d_matches, self.d_n)
        self.b.verify_mine_into(self.d_matches, self.d_n, self.rank, self.world, self.max_mine, self.exch.payload,
                                self.v_count, self.exch.extra[: self.max_mine])
        self.exch.exchange(None, finish=True)
        self.h_matches.copy_(self.d_matches, non_blocking=True)
        self.h_n.copy_(self.d_n, non_blocking=True)
        self.b.sync()                                   # the step's ONE wait
        self.waits += 1
        if int(self.h_n[1]) != 0:
            w = self.waits
            m = self.find_matches()                     # (resets and counts its own waits)
            flags, acc = self.verify(m)
            self.waits += w
            return m, flags, acc
        n_m = int(self.h_n[0])
        m = np.frombuffer(self.h_matches.numpy()[:n_m].tobytes(), dtype=_abi.MATCH_DTYPE).copy()
        flags, acc = self._collect(n_m)
        return m, flags, acc

class GpuShardBackend:
    """multi_robot_slam_separators_amd.sharded backend on one GPU: the handle holds THIS rank's block of local
    NetVLAD rows, the whole received database and the whole (replicated) keyframe store of one robot pair.  Nothing
    is staged through the host: the NN kernels write the minima into the all-gather's send block, the compaction
    writes the accepted records, their count and the flags into the record exchange's block."""

    def __init__(self, f, lo, n_received, slot_a, slot_b, n_local_total, dev, world=1):
        import torch
        self.f, self.lo, self.n_r, self.slot_a, self.slot_b, self.dev = f, lo, n_received, slot_a, slot_b, dev
        self.RB = _abi.RESULT_DTYPE.itemsize
        self.d_res = torch.empty((max((n_local_total + world - 1) // world, 1), self.RB), dtype=torch.uint8, device=dev)
        self.n_total = n_local_total

    def row_minima_into(self, row_min, row_arg, status):
        self.f.nn_row_minima_device(row_min.data_ptr(), row_arg.data_ptr(), status.data_ptr())

    def row_minima_sync(self):
        n_l, _ = self.f.nn_sizes()
        self.f.nn_find_matches(cap=n_l)                 # walks the filter ladder; its own walk is not used
        return self.f.nn_last_row_minima()

    def walk(self, d, a):
        return self.f.nn_walk(d, a, self.n_r, cap=self.n_total)

    def verify_into(self, matches, payload, count, flags):
        n = len(matches)
        if n == 0:
            count.zero_()
            return
        self.f.verify_matches_device(np.ascontiguousarray(matches), self.slot_a, self.slot_b, self.d_res.data_ptr())
        self.f.compact_accepted_device_async(self.d_res.data_ptr(), n, payload.data_ptr(), flags.data_ptr(), count.data_ptr())

    def walk_into(self, row_min, row_arg, status, matches, n):
        self.f.nn_walk_device(row_min.data_ptr(), row_arg.data_ptr(), status.data_ptr(), int(row_min.numel()), self.n_r,
                              matches.data_ptr(), int(matches.shape[0]), n.data_ptr())
        n[1:2].copy_(status)                             # (the walk is void when it is non-zero: the caller looks here)

    def verify_mine_into(self, matches, n, rank, world, max_mine, payload, count, flags):
        import torch
        if max_mine <= 0:
            count.zero_()
            return
        rc = matches.view(torch.int32).view(-1, 4)[rank::world][:max_mine]       # (idx_local, idx_other, distance lo, hi)
        k = int(rc.shape[0])
        if not hasattr(self, "d_from") or self.d_from.numel() < max_mine:
            self.d_from = torch.empty(max_mine, dtype=torch.int32, device=self.dev)
            self.d_to = torch.empty(max_mine, dtype=torch.int32, device=self.dev)
            self.p_idx = torch.arange(max_mine, dtype=torch.int32, device=self.dev) * world + rank
        valid = self.p_idx[:k] < n[0]                    # candidate p exists: p < the device-side match count
        self.d_from.fill_(-1)
        self.d_to.fill_(-1)
        self.d_from[:k] = torch.where(valid, rc[:, 1] + self.slot_a, -1)     # "from" = the querying robot's frame
        self.d_to[:k] = torch.where(valid, rc[:, 0] + self.slot_b, -1)       # "to"   = the computing robot's frame
        self.f.verify_pairs_device(self.d_from.data_ptr(), self.d_to.data_ptr(), max_mine, self.d_res.data_ptr())
        self.f.compact_accepted_device_async(self.d_res.data_ptr(), max_mine, payload.data_ptr(), flags.data_ptr(),
                                             count.data_ptr())

    def sync(self):
        import torch
        torch.cuda.synchronize()
