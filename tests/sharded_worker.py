"""Worker of tests/test_sharded_step.py: one gloo rank running multi_robot_slam_separators_amd.sharded.ShardedStep
(the section 8(e) partition the multi-GPU bench uses) with the CPU ORACLE as the compute backend -- the tests may use
the oracle; the orchestration (row blocks, all-gather of minima, replicated walk, p mod G, flag + record exchange,
interleave) is the product's.  usage: sharded_worker.py RANK WORLD PORT OUT_PREFIX [dense]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleShardBackend:
    """The backend interface of sharded.ShardedStep on CPU tensors.  `dense_once`: report "too dense for the device
    path" on the first query, so that the synchronous fallback (and its second all-gather) runs too."""

    def __init__(self, params, local, received, feats_other, feats_local, lo, hi, dense_once=False):
        self.p, self.local, self.received = params, local, received
        self.fo, self.fl = feats_other, feats_local
        # (two reports: the step's device attempt, then the first gather of the fallback -- which recomputes synchronously)
        self.lo, self.hi, self.dense = lo, hi, 2 if dense_once else 0
        self.sync_calls = 0

    def row_minima_sync(self):
        from oracle import pyoracle
        _, d, a = pyoracle.find_matches(self.local[self.lo:self.hi], self.received, (), (), (), self.p.netvlad_distance,
                                        self.p.netvlad_max_matches_nb)
        self.sync_calls += 1
        return np.asarray(d, dtype=np.float64), np.asarray(a, dtype=np.int32)

    def row_minima_into(self, row_min, row_arg, status):
        import torch
        if self.dense:
            self.dense -= 1
            row_min.fill_(float("nan"))
            status.fill_(1)
            return
        d, a = self.row_minima_sync()
        self.sync_calls -= 1
        row_min.copy_(torch.from_numpy(d))
        row_arg.copy_(torch.from_numpy(a))
        status.zero_()

    def walk(self, d, a):
        from multi_robot_slam_separators_amd import _abi, dist
        pairs = dist.walk_matches(d, a, self.p.netvlad_distance, self.p.netvlad_max_matches_nb)
        m = np.zeros(len(pairs), dtype=_abi.MATCH_DTYPE)
        for k, (il, io) in enumerate(pairs):
            m[k] = (il, io, d[il])
        return m

    def verify_into(self, matches, payload, count, flags):
        import torch
        from oracle import pyoracle
        count.zero_()
        if len(matches) == 0:
            return
        A = [self.fo[int(r["idx_other"])] for r in matches]      # "from" = the querying robot's frame
        B = [self.fl[int(r["idx_local"])] for r in matches]      # "to"   = the computing robot's frame
        res = pyoracle.estimate_transform_batch(self.p, A, B, 1)
        ok = torch.from_numpy(np.asarray(res["success"] != 0))
        raw = torch.from_numpy(np.frombuffer(res.tobytes(), dtype=np.uint8).reshape(len(matches), -1).copy())
        acc = raw[ok]
        payload[: acc.shape[0]].copy_(acc)
        count.fill_(int(acc.shape[0]))
        flags.copy_(ok.to(torch.uint8))

    def walk_into(self, row_min, row_arg, status, matches, n):
        """The replicated walk on the collective device's tensors (CPU here: synchronous)."""
        import torch
        st = int(status.reshape(-1)[0])
        n[1] = st
        n[0] = 0
        if st != 0:
            return
        m = self.walk(row_min.numpy().copy(), row_arg.numpy().copy())
        matches[: len(m)].copy_(torch.from_numpy(np.frombuffer(m.tobytes(), dtype=np.uint8).reshape(len(m), -1).copy()))
        n[0] = len(m)

    def verify_mine_into(self, matches, n, rank, world, max_mine, payload, count, flags):
        from multi_robot_slam_separators_amd import _abi
        flags.zero_()
        count.zero_()
        n_m = int(n[0])
        m = np.frombuffer(matches.numpy()[:n_m].tobytes(), dtype=_abi.MATCH_DTYPE)
        mine = m[rank::world]
        if len(mine):
            self.verify_into(mine, payload, count, flags[: len(mine)])

    def sync(self):
        pass


def make_problem(seed=4711, n=36):
    from multi_robot_slam_separators_amd import synth
    p = synth.camera_params()
    p.iterations = 100
    p.netvlad_dimensions = 64
    p.netvlad_max_matches_nb = n
    A, B, is_true, _ = synth.make_pairs(seed, n, k=120, cols=32, true_frac=0.5)
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n, 64)); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = a + rng.normal(size=(n, 64)) * (0.04 / 8.0)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    b[5] = rng.normal(size=64) / 8.0           # a local row without a neighbour under the threshold
    b[7] = b[6]                                  # two local rows nearest to the same received row (slot consumed)
    return p, b.astype(np.float32).astype(np.float64), a.astype(np.float32).astype(np.float64), A, B


def main():
    rank, world, port, prefix = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import torch
    import torch.distributed as td
    from multi_robot_slam_separators_amd import sharded
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = port
        td.init_process_group("gloo", rank=rank, world_size=world)
    p, local, received, A, B = make_problem()
    dense = len(sys.argv) > 5 and sys.argv[5] == "dense" and rank == world - 1
    lo, hi = sharded.row_blocks(len(local), world)[rank]
    be = OracleShardBackend(p, local, received, A, B, lo, hi, dense_once=dense)
    st = sharded.ShardedStep(be, rank, world, len(local), "cpu",
                             accept_cap=3)      # small capacity: the overflow path of the record exchange runs too
    m, flags, acc = st.step()
    # one wait per step; the dense fallback adds round 3's form behind it (gather, re-gather, verification)
    assert st.waits == (4 if len(sys.argv) > 5 and sys.argv[5] == "dense" else 1), st.waits
    assert be.sync_calls == (1 if dense else 0)
    np.save(prefix + "_m_%d.npy" % rank, m)
    np.save(prefix + "_flags_%d.npy" % rank, flags)
    np.save(prefix + "_acc_%d.npy" % rank, acc)
    if world > 1:
        td.barrier()
        td.destroy_process_group()
    print("rank %d ok: %d candidates, %d accepted" % (rank, len(m), int(flags.sum())))


if __name__ == "__main__":
    main()
