import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import datafusion_comet_amd  # noqa: E402,F401 — BEFORE anything imports torch: the JIT compiles with the installed ROCm's compiler, as under Spark (see that module)


# the time-zone database: the system's when it has one, else the tzdata wheel's (csrc/tz.cpp reads $TZDIR first; Python's zoneinfo, the tests'
# referee, searches the same places in the same order)
if "TZDIR" not in os.environ and not os.path.isdir("/usr/share/zoneinfo"):
    try:
        import tzdata
        os.environ["TZDIR"] = os.path.join(os.path.dirname(tzdata.__file__), "zoneinfo")
    except ImportError:
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """libcomet.so + the C oracle, built once per session (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    return True


# COMET_COMPILE_SWEEP=1 python -m pytest tests -m gpu   (on a box WITHOUT a GPU): every plan a GPU test would create is decoded, planned, generated and
# hiprtc-compiled for gfx950 instead (comet_compile_plan needs no device), then the test is skipped — a compiler change (a ROCm upgrade) is checked
# against the whole suite's plan shapes in minutes, before any GPU time is spent.  Compilation errors fail the test.
if os.environ.get("COMET_COMPILE_SWEEP") == "1":
    @pytest.fixture(autouse=True)
    def _compile_instead_of_run(monkeypatch):
        from datafusion_comet_amd import native

        def create(inputs, plan, config=b"", *a, **k):
            try:
                native.compile_plan(bytes(plan))
            except native.CometNativeException as e:
                if "hiprtc" in str(e):
                    raise
            pytest.skip("compile sweep: plan compiled")
        monkeypatch.setattr(native.Native, "createPlan", staticmethod(create))
        yield
