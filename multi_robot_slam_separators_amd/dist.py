"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm;
"gloo" in the CPU tests).

The path shards naturally: candidate pairs are independent
(stereoCamGeometricTools.cpp:122-178 is stateless per call) and NN rows are independent until
the final sequential walk (data_handler.py:191-205).  So there is exactly ONE data-path exchange:
an all-gather of fixed-size separator records (one row of ReceiveSeparators.srv each), plus, for
the row-sharded NN stage, an all-gather of per-row (distance, index) minima followed by the walk
replicated on every rank.
"""
import numpy as np


def shard_pairs(n_pairs, rank, world):
    """Round-robin partition: pair p belongs to rank p mod world."""
    return np.arange(rank, n_pairs, world, dtype=np.int64)


def _dist():
    import torch.distributed as td
    return td


def allgather_records(local, group=None):
    """All-gather a ragged set of fixed-size records.

    local: torch.uint8 tensor [n_local, record_bytes] on the device of the process group's backend.
    Returns (records [sum n, record_bytes], counts list).  Two-phase: counts first, then one
    all_gather_into_tensor of the payload padded to the largest shard (a single direct exchange
    over xGMI for the few-MB shards this path produces)."""
    import torch
    td = collective(group)
    world = td.get_world_size(group)
    n_local, rec = int(local.shape[0]), int(local.shape[1])
    cnt = torch.tensor([n_local], dtype=torch.int64, device=local.device)
    counts = torch.zeros(world, dtype=torch.int64, device=local.device)
    td.all_gather_into_tensor(counts, cnt, group=group)
    counts = [int(c) for c in counts.tolist()]
    cap = max(max(counts), 1)
    send = torch.zeros((cap, rec), dtype=torch.uint8, device=local.device)
    if n_local:
        send[:n_local] = local
    recv = torch.empty((world * cap, rec), dtype=torch.uint8, device=local.device)
    td.all_gather_into_tensor(recv, send, group=group)
    parts = [recv[r * cap: r * cap + counts[r]] for r in range(world)]
    return torch.cat(parts, dim=0), counts


def allgather_records_fixed(local, cap, group=None):
    """All-gather a ragged set of fixed-size records with ONE collective.

    Every rank sends `cap` record slots preceded by a header slot that carries its true count (int64 in
    the first 8 bytes), so no count exchange precedes the payload: one all_gather_into_tensor, one
    device-to-host read of the `world` headers.  If some rank holds more than `cap` records (the caller
    sized `cap` from the expected acceptance rate) the overflow is exchanged with the two-phase
    allgather_records -- correct for any input, one collective in the common case.
    Returns (records [sum n, record_bytes], counts list), rank-major like allgather_records."""
    import torch
    td = _dist()
    world = td.get_world_size(group)
    n_local, rec = int(local.shape[0]), int(local.shape[1])
    if rec < 8:
        return allgather_records(local, group)
    cap = max(int(cap), 1)
    send = torch.zeros((cap + 1, rec), dtype=torch.uint8, device=local.device)
    send[0, :8] = torch.tensor([n_local], dtype=torch.int64).view(torch.uint8).to(local.device)
    k = min(n_local, cap)
    if k:
        send[1: 1 + k] = local[:k]
    recv = torch.empty((world * (cap + 1), rec), dtype=torch.uint8, device=local.device)
    td.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, cap + 1, rec)
    counts = [int(c) for c in recv[:, 0, :8].contiguous().view(torch.int64).reshape(-1).tolist()]
    parts = [recv[r, 1: 1 + min(counts[r], cap)] for r in range(world)]
    if max(counts) > cap:   # rare: exchange what did not fit
        extra, ecounts = allgather_records(local[cap:] if n_local > cap else local[:0], group)
        off = 0
        for r in range(world):
            if ecounts[r]:
                parts[r] = torch.cat([parts[r], extra[off: off + ecounts[r]]], dim=0)
            off += ecounts[r]
    return torch.cat(parts, dim=0), counts


class SelfCollective:
    """torch.distributed's slice of interface the exchanges use, for ONE rank without a process group: the
    all-gather is a copy (on the caller's current stream), so that the same step code runs with and without a node."""

    class _Done:
        def wait(self):
            return None

    def get_world_size(self, group=None):
        return 1

    def get_rank(self, group=None):
        return 0

    def all_gather_into_tensor(self, recv, send, group=None, async_op=False):
        recv.view(-1).copy_(send.reshape(-1), non_blocking=True)
        return SelfCollective._Done() if async_op else None


def collective(group=None):
    """torch.distributed when a process group is up, SelfCollective otherwise."""
    td = _dist()
    return td if td.is_available() and td.is_initialized() else SelfCollective()


class RecordExchange:
    """The per-step separator exchange with persistent buffers: ONE all_gather_into_tensor per call, no
    allocation, no host-to-device copy and no synchronisation of its own.

    The producer writes its records straight into `payload` (rows 1.. of the send buffer: hand
    `payload.data_ptr()` to sf_compact_accepted_device); `exchange(n)` stamps the count into the header
    row ON THE DEVICE (a fill, not an upload), issues the collective over the first `cap + 1` rows and
    queues a copy of the `world` header rows into pinned host memory.  After the caller's own stream
    synchronisation `counts()` is a plain host read, and `gathered(r)` / `all_gathered()` are views of the
    receive buffer (rank-major).  If some rank held more than `cap` records, `counts()` reports it and
    `all_gathered()` fetches the overflow with the two-phase allgather_records (correct for any input, one
    collective in the common case).  Works on CPU tensors too (gloo rehearsals and tests)."""

    def __init__(self, record_bytes, max_rows, cap, device, group=None, extra_bytes=0, host_mirror=False):
        """extra_bytes: an opaque per-rank byte region that travels in the same collective (rows 1..E of the block,
        in front of the records; `extra` is this rank's view of it, `gathered_extra(r)` rank r's) -- the per-candidate
        success flags of the section 8(e) step ride here.  host_mirror: finish() queues a copy of the WHOLE receive
        buffer into pinned host memory (`host_gathered*`), so that the step's results are host-readable behind the
        caller's one synchronisation."""
        import torch
        self.td = collective(group)
        self.group = group
        self.world = self.td.get_world_size(group)
        self.rec = int(record_bytes)
        self.cap = max(1, min(int(cap), int(max_rows)))
        self.device = torch.device(device)
        assert self.rec >= 8
        self.extra_bytes = int(extra_bytes)
        E = self.extra_rows = (self.extra_bytes + self.rec - 1) // self.rec
        self.blk = self.cap + 1 + E                                 # rows per rank in the collective
        self.send = torch.zeros((int(max_rows) + 1 + E, self.rec), dtype=torch.uint8, device=self.device)
        self.extra = self.send[1: 1 + E].reshape(-1)[: self.extra_bytes]
        self.payload = self.send[1 + E:]
        self._hdr = self.send[0, :8].view(torch.int64)           # this rank's record count
        self.recv = torch.empty((self.world, self.blk, self.rec), dtype=torch.uint8, device=self.device)
        self._hdr_all = self.recv[:, 0, :8]                        # [world, 8] bytes, strided view
        self.h_recv = None
        if host_mirror:
            self.h_recv = torch.zeros((self.world, self.blk, self.rec), dtype=torch.uint8)
            if self.device.type == "cuda":
                self.h_recv = self.h_recv.pin_memory()
        self._h_counts = torch.zeros((self.world, 8), dtype=torch.uint8)
        if self.device.type == "cuda":
            self._h_counts = self._h_counts.pin_memory()
        self._n_local = 0
        self._work = None

    @property
    def count_ptr(self):
        """Device address of this rank's count (int32 view of the header row), for a producer that stamps it."""
        return self.send.data_ptr()

    def exchange(self, n_local, finish=True):
        """Stamp the count (n_local = None: the producer already did, on the device) and start the all-gather.  With finish=False the collective is left in flight (RCCL
        runs it on its own stream) so that the caller can queue independent work -- e.g. the copy of its own
        records to the host -- before calling finish()."""
        if n_local is None:
            # the producer stamped the count itself (sf_compact_accepted_device_async writes its int32 count into
            # the first 4 bytes of the header row; the other 4 stay zero): no host knowledge of it is needed
            self._n_local = None
        else:
            self._n_local = int(n_local)
            # low word: the count; high word: this rank's number + 1 -- evidence, in the gathered headers themselves, of
            # which ranks took part in the collective (ranks_seen)
            self._hdr.fill_(self._n_local | ((self.td.get_rank(self.group) + 1) << 32))
        self._work = self.td.all_gather_into_tensor(self.recv.view(self.world * self.blk, self.rec),
                                                    self.send[: self.blk], group=self.group, async_op=True)
        if finish:
            self.finish()

    def finish(self):
        """Order the caller's stream behind the collective and queue the copy of the header rows to the host."""
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self.h_recv is not None:
            self.h_recv.copy_(self.recv, non_blocking=True)
        else:
            self._h_counts.copy_(self._hdr_all, non_blocking=True)

    def counts(self, limit=None):
        """Per-rank record counts; valid once the stream the exchange ran on has been synchronised.  The header row
        carries the count in its first FOUR bytes as an int32 when the producer stamped it on the device
        (sf_compact_accepted_device_async; -1 = the compaction's look-back timed out) and as an int64 when exchange(n)
        did: the low word is read as int32 either way, a negative count -- or one above `limit`, the number of records a
        rank can hold at all -- raises on every rank alike (they all read the same gathered headers)."""
        import torch
        h = self._h_counts if self.h_recv is None else self.h_recv[:, 0, :8].contiguous()
        out = [int(c) for c in h.view(torch.int32).reshape(-1, 2)[:, 0].tolist()]
        for r, c in enumerate(out):
            if c < 0 or (limit is not None and c > limit):
                raise RuntimeError("record exchange: rank %d reports %d records (limit %s): the producer's count is invalid"
                                   % (r, c, limit))
        return out

    def ranks_seen(self):
        """Ranks whose header row arrived in the last finished all-gather, read from the high words of the gathered
        headers (each rank stamps rank + 1 there when it stamps its count on the host): what a bench line reports as
        `ranks_in_allgather` -- counted from the data, not from WORLD_SIZE.  Empty when the producer stamped the count on
        the device (the high word stays zero)."""
        import torch
        h = self._h_counts if self.h_recv is None else self.h_recv[:, 0, :8].contiguous()
        hi = h.view(torch.int32).reshape(-1, 2)[:, 1].tolist()
        return sorted(int(x) - 1 for x in hi if int(x) > 0)

    def bytes_per_exchange(self):
        """Bytes one all-gather moves into every rank's receive buffer (world x block rows x record bytes)."""
        return int(self.world * self.blk * self.rec)

    def gathered(self, r, counts=None):
        c = (counts or self.counts())[r]
        return self.recv[r, 1 + self.extra_rows: 1 + self.extra_rows + min(c, self.cap)]

    def gathered_extra(self, r):
        return self.recv[r, 1: 1 + self.extra_rows].reshape(-1)[: self.extra_bytes]

    def host_gathered(self, r, counts=None):
        """numpy view of rank r's records in the pinned mirror (host_mirror=True; valid behind the synchronisation)."""
        c = (counts or self.counts())[r]
        return self.h_recv[r, 1 + self.extra_rows: 1 + self.extra_rows + min(c, self.cap)].numpy()

    def host_gathered_extra(self, r):
        return self.h_recv[r, 1: 1 + self.extra_rows].reshape(-1)[: self.extra_bytes].numpy()

    def all_gathered(self):
        """(records [sum n, record_bytes] rank-major, counts) -- materialises a copy; the overflow of a rank
        that held more than `cap` records is exchanged here."""
        import torch
        counts = self.counts()
        parts = [self.gathered(r, counts) for r in range(self.world)]
        if max(counts) > self.cap:
            n_local = counts[self.td.get_rank(self.group)] if self._n_local is None else self._n_local
            extra, ecounts = allgather_records(self.payload[self.cap: n_local] if n_local > self.cap
                                               else self.payload[:0], self.group)
            off = 0
            for r in range(self.world):
                if ecounts[r]:
                    parts[r] = torch.cat([parts[r], extra[off: off + ecounts[r]]], dim=0)
                off += ecounts[r]
        return torch.cat(parts, dim=0), counts


def torch_int64():
    import torch
    return torch.int64


def interleave_round_robin(records, counts):
    """Undo shard_pairs: records gathered rank-major -> global pair order (p = i*world + rank)."""
    import torch
    world = len(counts)
    total = sum(counts)
    out = torch.empty_like(records)
    off = 0
    for r in range(world):
        idx = torch.arange(r, total, world, device=records.device)[: counts[r]]
        out[idx] = records[off: off + counts[r]]
        off += counts[r]
    return out


def walk_matches(row_min, row_arg, netvlad_distance, max_matches_nb):
    """The sequential tail of DataHandler.find_matches (data_handler.py:191-205) on gathered row
    minima; ties in the sort resolve to the lowest index.  Returns [(idx_local, idx_other)]."""
    row_min = np.asarray(row_min, dtype=np.float64)
    order = np.lexsort((np.arange(row_min.size), row_min))
    matches, taken = [], set()
    for s in range(min(row_min.size, int(max_matches_nb))):
        il = int(order[s])
        io = int(row_arg[il])
        if io in taken:
            continue
        if row_min[il] < netvlad_distance:
            matches.append((il, io))
            taken.add(io)
        else:
            break
    return matches


def allgather_row_minima(row_min, row_arg, group=None):
    """NN stage sharded over LOCAL rows: every rank searched a contiguous block of local rows
    against the replicated received database; gather the (float64 distance, int32 index) pairs so
    that every rank can run the identical walk.  Inputs: torch tensors on the backend's device."""
    import torch
    rec = torch.empty((row_min.shape[0], 12), dtype=torch.uint8, device=row_min.device)
    rec[:, :8] = row_min.contiguous().view(torch.uint8).reshape(-1, 8)
    rec[:, 8:] = row_arg.contiguous().view(torch.uint8).reshape(-1, 4)
    allrec, counts = allgather_records(rec, group)
    d = allrec[:, :8].contiguous().view(torch.float64).reshape(-1)
    i = allrec[:, 8:].contiguous().view(torch.int32).reshape(-1)
    return d, i, counts
