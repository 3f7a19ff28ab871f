"""Where one Q1 task's wall time goes outside the k_gagg launch: createPlan / first executePlan / second executePlan (end of stream) /
releasePlan / harness, over the SF100 resident shard.  Usage: python tools/task_breakdown.py [--rows N] [--steps K]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=600_037_902)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    import datafusion_comet_amd  # noqa: F401 — before torch: the JIT then compiles with the installed ROCm's compiler (see that module)
    import torch
    from datafusion_comet_amd import native, tpch
    plan = tpch.q1_plan().encode()
    dtab, _ = tpch.lineitem_q1_device(a.rows, device="cuda:0", seed=1)
    dtab = dtab.with_string_hints()
    torch.cuda.synchronize()
    ncols = tpch.Q1_NUM_OUTPUT_COLS
    acc = {"input": 0.0, "createPlan": 0.0, "executePlan_1": 0.0, "executePlan_2": 0.0, "kernel_stats": 0.0, "releasePlan": 0.0, "kernel_ms": 0.0}
    for i in range(a.steps + 2):
        t = [time.perf_counter()]
        inp = native.DeviceInput(dtab)
        t.append(time.perf_counter())
        it = native.CometExecIterator([inp], ncols, plan)
        t.append(time.perf_counter())
        b = native.Native.executePlan(it.handle, ncols)
        t.append(time.perf_counter())
        e = native.Native.executePlan(it.handle, ncols)
        t.append(time.perf_counter())
        st = it.kernel_stats()
        t.append(time.perf_counter())
        it.close()
        t.append(time.perf_counter())
        assert b is not None and e is None
        if i >= 2:
            for k, name in enumerate(["input", "createPlan", "executePlan_1", "executePlan_2", "kernel_stats", "releasePlan"]):
                acc[name] += (t[k + 1] - t[k]) * 1e3 / a.steps
            acc["kernel_ms"] += st[0] / max(st[1], 1) / a.steps
    acc["task_ms"] = sum(v for k, v in acc.items() if k != "kernel_ms")
    print(json.dumps({k: round(v, 4) for k, v in acc.items()}))


if __name__ == "__main__":
    main()
