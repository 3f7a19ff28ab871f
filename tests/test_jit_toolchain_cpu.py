"""Which compiler the JIT uses is decided by what the process loaded first (datafusion_comet_amd/__init__.py, csrc/jit.cpp compiler_identity): the
installed ROCm's libamd_comgr when this package is imported before torch — the deployment's compiler, a JVM holds no other ROCm —, the torch wheel's
bundled one otherwise.  Each case in its own interpreter (the choice is made once per process)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INSTALLED = "/opt/rocm/lib/libamd_comgr.so.3"


def _toolchain(code, env=None):
    e = dict(os.environ)
    e.pop("COMET_SYSTEM_COMGR", None)
    e.update(env or {})
    p = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r})\n" + code + "\nfrom datafusion_comet_amd import native\nprint('TC=' + native.jit_toolchain())"],
                       capture_output=True, text=True, env=e, cwd=ROOT, timeout=300)
    assert p.returncode == 0, p.stderr[-800:]
    return [l for l in p.stdout.splitlines() if l.startswith("TC=")][0][3:], p.stderr


@pytest.mark.skipif(not os.path.exists(INSTALLED), reason="no installed ROCm code object manager")
def test_package_first_gives_the_installed_compiler(built):
    tc, err = _toolchain("import datafusion_comet_amd\nimport torch")
    assert os.path.realpath(INSTALLED) in os.path.realpath(tc.split()[-1]) or "/opt/rocm" in tc, tc
    assert "torch was imported first" not in err


@pytest.mark.skipif(not os.path.exists(INSTALLED), reason="no installed ROCm code object manager")
def test_torch_first_keeps_the_wheels_compiler_and_says_so(built):
    tc, err = _toolchain("import torch")
    assert "/opt/rocm" not in tc, tc
    assert "torch was imported first" in err


def test_switch_keeps_the_wheels_compiler_silently(built):
    tc, err = _toolchain("import datafusion_comet_amd\nimport torch", {"COMET_SYSTEM_COMGR": "0"})
    assert "/opt/rocm" not in tc, tc
    assert "torch was imported first" not in err


def test_this_suite_runs_with_the_installed_compiler(built):
    """conftest.py imports the package before anything imports torch"""
    import datafusion_comet_amd
    from datafusion_comet_amd import native
    if os.path.exists(INSTALLED) and os.environ.get("COMET_SYSTEM_COMGR", "1") != "0":
        assert datafusion_comet_amd.SYSTEM_COMGR is not None
        assert "/opt/rocm" in native.jit_toolchain(), native.jit_toolchain()
