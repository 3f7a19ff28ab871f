"""tests/emu/codegen_emu.py as a pytest plugin, for a sweep over the GPU suite WITHOUT a GPU:

    python -m pytest tests -m gpu -p tools.emu_plugin -q -n 8 --timeout=900 --tb=line -rfE

Every test that reaches the device through native.execute_to_table with ONE input runs with the plan's GENERATED per-row code, compiled by g++ against the
header texts hiprtc uses, standing in for the device (what that covers and what it does not: tests/test_codegen_emu_cpu.py's header).  The tests that pass
this way and are cheap enough are listed in tests/test_codegen_emu_cpu.py; the others stop with emu.Unsupported or "no ROCm-capable device".
Test infrastructure only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    from datafusion_comet_amd import native
    from tests.emu import codegen_emu as E

    def execute(inputs, ncols, plan_bytes, **kw):
        if kw.get("subqueries"):
            raise E.Unsupported("scalar subqueries are resolved by the executor")
        if len(inputs) != 1:
            raise E.Unsupported("plans with several inputs")
        out = E.run_chain(plan_bytes, inputs[0].table)
        assert out.num_columns == ncols, (out.num_columns, ncols)
        return out.to_batches(max_chunksize=kw.get("batch_size", 8192) or None) if out.num_rows else []      # spark.comet.batchSize bounds an output batch

    native.HostInput, native.execute_to_table = E._HostInput, execute
