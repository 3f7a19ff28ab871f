"""Test-only stand-ins for the per-rank GPU engine and the HIP partitioner, built on the oracle, so that the exchange /
staged-Q3 logic of datafusion-comet_amd/parallel.py can run in CPU processes (gloo).  Never imported by the product."""
import pyarrow as pa

from datafusion_comet_amd import native, serde as S
from oracle import oracle as O


def _arrow(t):
    return t.to_arrow() if isinstance(t, native.DeviceTable) else t


class OracleEngine:
    def run_host(self, plan, tables, ncols):
        tabs = [_arrow(t) for t in tables]
        out = O.run_plan_to_arrow(S, plan, tabs if len(tabs) > 1 else tabs[0])
        assert out.num_columns == ncols
        return out

    def run_device(self, plan, tables, ncols):
        return native.DeviceTable.from_arrow(self.run_host(plan, tables, ncols), device="cpu")


class OraclePartitioner:
    def __call__(self, table, key_cols, num_partitions):
        t = _arrow(table)
        pids = O.hash_partition_ids(S, t, key_cols, num_partitions)
        starts, idx = O.partition_starts_and_indices(pids, num_partitions)
        return native.DeviceTable.from_arrow(t.take(pa.array(idx)), device="cpu"), [int(x) for x in starts]
