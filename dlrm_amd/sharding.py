"""Embedding sharding beyond the reference's contiguous table-wise blocks.

`extend_distributed.py` / `DLRM_Net.distributed_forward` place tables in contiguous blocks of equal COUNT (26 tables over 8 ranks
-> 4,4,3,3,3,3,3,3) and replicate the inputs on every rank.  That is fine for one-hot Criteo; it cannot balance the MLPerf-v2
multi-hot benchmark (BASELINE.json configs[4]), where ONE table (40 M rows, 100 lookups per sample) carries 47 % of all lookups
and a quarter of the bytes.  The torchrec trainer the reference uses for that benchmark (`torchrec_dlrm/dlrm_main.py:654-673`)
hands placement to torchrec's `EmbeddingShardingPlanner` (third-party, not in the tree); this module is its MI355X-sized
counterpart, with the costs that matter on this part:

  * cost of a table per step  = HBM bytes its lookups move: B * h_t * (fwd: R + idx, bwd+update: 3R + idx), R = 4*D bytes;
  * capacity of a rank        = HBM bytes for tables + row-wise optimizer state (288 GB per MI355X, a reservation kept free);
  * placement                 = every table whose cost exceeds `row_wise_threshold` x (total cost / world) — it could not be
                                balanced by moving it around — or whose bytes exceed a rank's capacity is sharded ROW-WISE over
                                all ranks (contiguous row ranges, every rank sees the whole batch's ids for it and pools the
                                rows it owns; partial sums meet in a reduce-scatter); the others are placed TABLE-WISE by
                                longest-processing-time-first on the remaining per-rank load (cost, ties by bytes).

All of it is integer bookkeeping, identical on every rank (no communication), tested on CPU.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple


@dataclass(frozen=True)
class TableShard:
    table: int                      # global table number
    kind: str                       # "table" (whole table on `rank`) or "row" (rows split over all ranks)
    rank: int = -1                  # owner for kind == "table"
    row_ranges: Tuple[Tuple[int, int], ...] = ()    # kind == "row": [lo, hi) per rank


@dataclass
class ShardingPlan:
    world: int
    shards: List[TableShard]
    cost: List[int]                                  # per table, bytes per step
    rank_cost: List[int] = field(default_factory=list)
    rank_bytes: List[int] = field(default_factory=list)

    def table_wise(self, rank: Optional[int] = None) -> List[int]:
        return [s.table for s in self.shards if s.kind == "table" and (rank is None or s.rank == rank)]

    def row_wise(self) -> List[int]:
        return [s.table for s in self.shards if s.kind == "row"]

    def tables_per_rank(self) -> List[int]:
        return [len(self.table_wise(r)) for r in range(self.world)]

    def imbalance(self) -> float:
        m = sum(self.rank_cost) / max(len(self.rank_cost), 1)
        return max(self.rank_cost) / m if m > 0 else 1.0


def split_rows(rows: int, world: int) -> Tuple[Tuple[int, int], ...]:
    """contiguous row ranges, the first rows % world ranks get one extra (the reference's block arithmetic,
    extend_distributed.py:47-51, applied to rows)"""
    base, extra = divmod(rows, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return tuple(out)


def table_cost(batch: int, hot: int, D: int, idx_bytes: int = 4) -> int:
    R = 4 * D
    return batch * hot * ((R + idx_bytes) + (3 * R + idx_bytes)) + batch * (R + idx_bytes)     # lookups fwd + update, pooled row write


def plan(rows: Sequence[int], hot: Sequence[int], D: int, world: int, batch: int, capacity_bytes: int = 250 * 10 ** 9,
         state_bytes_per_row: int = 4, row_wise_threshold: float = 0.5, idx_bytes: int = 4) -> ShardingPlan:
    T = len(rows)
    if len(hot) != T or world < 1:
        raise ValueError("plan: rows / hot length mismatch or bad world size")
    cost = [table_cost(batch, int(h), D, idx_bytes) for h in hot]
    nbytes = [int(n) * (4 * D + state_bytes_per_row) for n in rows]
    if world == 1:
        return ShardingPlan(1, [TableShard(t, "table", 0) for t in range(T)], cost, [sum(cost)], [sum(nbytes)])
    fair = sum(cost) / world
    rw = [t for t in range(T) if (cost[t] > row_wise_threshold * fair or nbytes[t] > capacity_bytes) and rows[t] >= world]
    rank_cost = [0] * world
    rank_bytes = [0] * world
    shards: List[Optional[TableShard]] = [None] * T
    for t in rw:
        rr = split_rows(int(rows[t]), world)
        shards[t] = TableShard(t, "row", -1, rr)
        for r, (lo, hi) in enumerate(rr):
            # every rank streams the whole batch's ids of a row-wise table and touches its share of the rows
            rank_cost[r] += cost[t] // world + batch * int(hot[t]) * idx_bytes
            rank_bytes[r] += (hi - lo) * (4 * D + state_bytes_per_row)
    for t in sorted((t for t in range(T) if shards[t] is None), key=lambda t: (-cost[t], -nbytes[t], t)):
        fits = [r for r in range(world) if rank_bytes[r] + nbytes[t] <= capacity_bytes]
        if not fits:
            raise ValueError("plan: table %d (%d bytes) fits no rank; lower row_wise_threshold or raise capacity" % (t, nbytes[t]))
        r = min(fits, key=lambda r: (rank_cost[r], rank_bytes[r], r))
        shards[t] = TableShard(t, "table", r)
        rank_cost[r] += cost[t]
        rank_bytes[r] += nbytes[t]
    return ShardingPlan(world, [s for s in shards if s is not None], cost, rank_cost, rank_bytes)


def reference_plan(T: int, world: int) -> List[int]:
    """owner of every table under the reference's contiguous block partition (extend_distributed.py:47-62): for comparison"""
    base, extra = divmod(T, world)
    out = []
    for r in range(world):
        out += [r] * (base + (1 if r < extra else 0))
    return out
