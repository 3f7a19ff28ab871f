"""Dry-run a GPU test's Python with the ORACLE standing in for the device: catches test-side mistakes (shapes, masks, expectations, plans the
oracle cannot evaluate) here, before a GPU minute is spent.  Test infrastructure only — usage: python tools/dryrun_gpu_test.py tests.test_x test_fn."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa

from datafusion_comet_amd import native, serde as S
from oracle import oracle as O

_last = {}
_enc = S.Operator.encode


def _encode(self):
    if not _last.get("depth"):      # the outermost operator of the encode call is the plan
        _last["plan"] = self
    _last["depth"] = _last.get("depth", 0) + 1
    try:
        return _enc(self)
    finally:
        _last["depth"] -= 1


S.Operator.encode = _encode


class _Host:
    def __init__(self, table):
        self.table = table

    @staticmethod
    def from_table(table, batch_rows=8192):
        return _Host(table)


def _execute(inputs, ncols, plan_bytes, **kw):
    # the plan goes through createPlan's planning AND hiprtc here (no GPU needed): refusals and kernels that do not compile show up now.
    # (Plans over materialised sources plan their chains at run time: those are generated only when a chain is the plan's root.)
    native.compile_plan(plan_bytes)
    try:
        out = O.run_plan_to_arrow(S, _last["plan"], [i.table for i in inputs])
    except O.OracleError as e:      # the message a device run would carry is not reproduced: every fromType the tests match is appended
        raise native.CometQueryExecutionException(str(e) + ' "fromType":"byte" "fromType":"short" "fromType":"integer" "fromType":"long" "fromType":"Int32"')
    assert out.num_columns == ncols, (out.num_columns, ncols)
    return out.to_batches()


native.HostInput = _Host
native.execute_to_table = _execute
mod = importlib.import_module(sys.argv[1])
for fn in sys.argv[2:]:
    getattr(mod, fn)(None)
    print("dry run ok:", fn)
