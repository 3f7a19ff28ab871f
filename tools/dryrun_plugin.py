"""The dry run of tools/dryrun_gpu_test.py as a pytest plugin, for a sweep over the GPU suite WITHOUT a GPU:

    python -m pytest tests -m gpu -p tools.dryrun_plugin -q -n 8 --timeout=600

Every test that reaches the device through native.execute_to_table runs with the ORACLE standing in for the device — after its plan went through
createPlan's planning and hiprtc (native.compile_plan): a plan the library refuses, a generated kernel that does not compile against the device headers of
this tree, a test whose expectation the oracle does not meet, show up here.  Tests that reach the device another way (Parquet scans, executors, joins through
the C entries, benches) fail with "no ROCm-capable device" and are not this sweep's subject: tools/dryrun_report.py sorts the two apart.
Test infrastructure only: nothing here is importable from the product."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    import pyarrow as pa  # noqa: F401

    from datafusion_comet_amd import native, serde as S
    from oracle import oracle as O

    last = {}
    enc = S.Operator.encode

    def encode(self):
        if not last.get("depth"):       # the outermost operator of an encode call is the plan
            last["plan"] = self
        last["depth"] = last.get("depth", 0) + 1
        try:
            return enc(self)
        finally:
            last["depth"] -= 1

    S.Operator.encode = encode

    class Host:
        def __init__(self, table):
            self.table = table

        @staticmethod
        def from_table(table, batch_rows=8192):
            return Host(table)

    def execute(inputs, ncols, plan_bytes, **kw):
        native.compile_plan(plan_bytes)
        try:
            out = O.run_plan_to_arrow(S, last["plan"], [i.table for i in inputs])
        except O.OracleError as e:
            raise native.CometQueryExecutionException(str(e) + ' "fromType":"byte" "fromType":"short" "fromType":"integer" "fromType":"long" "fromType":"Int32"')
        assert out.num_columns == ncols, (out.num_columns, ncols)
        return out.to_batches()

    native.HostInput = Host
    native.execute_to_table = execute
