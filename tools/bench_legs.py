"""The N > 1 legs of bench.py, one at a time, without leaving a rank behind (round-5 advisor: a one-rank exception inside a
broad try/except used to leave the other ranks blocked in the next collective until the RCCL timeout).

Every rank calls `Legs.run(key, fn)` in the same order.  A leg runs in its own try-block; afterwards the ranks agree on its
outcome (all-reduce MIN of an ok flag): a leg that failed on ANY rank is recorded as failed on rank 0 ("failed on another
rank" when rank 0's own call succeeded) and poisons the remaining COLLECTIVE legs - they are skipped on all ranks, because a
rank that raised in the middle of a collective sequence can no longer be trusted to issue the same sequence as the others.
Pure torch.distributed logic: tests/test_distributed_cpu.py runs it over gloo with two ranks."""
import torch
import torch.distributed as dist


class Legs:
    def __init__(self, rank, world, red_dev, line):
        self.rank, self.world, self.red_dev, self.line = rank, world, red_dev, line
        self.healthy = True

    def agree(self, ok):
        """all ranks: did the leg succeed everywhere?"""
        if self.world == 1:
            return bool(ok)
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=self.red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def run(self, key, fn, keep=None):
        """-> the leg's object (rank 0 also stores it under line[key], reduced to the keys in `keep` when given)."""
        if not self.healthy:
            out = {"error": "skipped: an earlier collective leg failed on some rank"}
        else:
            try:
                out = fn()
                ok = True
            except Exception as e:                                # noqa: BLE001
                out = {"error": f"{type(e).__name__}: {e}"[:300]}
                ok = False
            try:
                if not self.agree(ok):
                    self.healthy = False
                    if ok:
                        out = {"error": "failed on another rank"}
            except Exception as e:                                # noqa: BLE001 - the agreement itself failed: stop collectives
                self.healthy = False
                out = {"error": f"ranks could not agree after the leg: {type(e).__name__}: {e}"[:300]}
        if self.rank == 0 and self.line is not None:
            self.line[key] = out if keep is None or "error" in out else {k: out[k] for k in keep if k in out}
        return out
