"""Import shim: the real package lives in ``datafusion-comet_amd/`` (hyphenated like the reference repo).

It also decides which COMPILER the engine's JIT uses in this Python harness.  Under Spark libcomet.so is the only ROCm client of the JVM: its hiprtc
and the code object manager behind it (libamd_comgr, which holds the whole clang / LLVM) are the installed ROCm's (libcomet.so's RUNPATH, /opt/rocm/lib).
This harness keeps device buffers in torch tensors, and torch's wheel bundles an OLDER ROCm (hiprtc + comgr 7.0.2 = clang 20 against the installed
7.2.0 = clang 22): with torch imported first, every generated kernel was compiled by the wheel's compiler — 10–15 % slower join kernels than the ones
the deployment compiles (profiles/r6_jit_compiler.md).  Loading the installed libamd_comgr into the GLOBAL symbol scope BEFORE torch is imported makes
its amd_comgr_* symbols the ones every later hiprtc binds to — what rocprofv3's preloaded tool library does to any program it profiles.  So: import
this package before torch (tests/conftest.py, bench.py, __graft_entry__.py and the tools do).  COMET_SYSTEM_COMGR=0 keeps the wheel's compiler;
COMET_COMGR_LIBRARY names another library.  native.jit_toolchain() tells which compiler a process ended up with; it is part of the JIT cache key."""
import ctypes as _ctypes
import os as _os
import sys as _sys

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "datafusion-comet_amd"))

SYSTEM_COMGR = None      # the library preloaded below (None: not done — switched off, not found, or torch was imported first)


def _prefer_installed_compiler():
    global SYSTEM_COMGR
    if _os.environ.get("COMET_SYSTEM_COMGR", "1") == "0" or "torch" in _sys.modules:
        return
    for p in (_os.environ.get("COMET_COMGR_LIBRARY"), "/opt/rocm/lib/libamd_comgr.so.3", "/opt/rocm/lib/libamd_comgr.so"):
        if p and _os.path.exists(p):
            try:
                _ctypes.CDLL(p, mode=_ctypes.RTLD_GLOBAL)
                SYSTEM_COMGR = _os.path.realpath(p)
                return
            except OSError:
                continue


_prefer_installed_compiler()
