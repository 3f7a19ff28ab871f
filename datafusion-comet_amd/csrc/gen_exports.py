"""The linker version script of libcomet.so: every comet_* entry include/comet_amd.h declares (the TEST ABI section included — the repo's own tests bind it) and the
JNI names the JVM resolves (Java_org_apache_comet_*, jni_shim.cpp); everything else — kernel launchers, kernel host stubs, C++ symbols — stays local."""
import re
import sys

text = open(sys.argv[1]).read()
text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
names = sorted(set(re.findall(r"\b(comet_[a-z0-9_]+)\s*\(", text)))
print("{\n  global:\n    Java_org_apache_comet_*;\n    JNI_OnLoad;\n    JNI_OnUnload;")
for n in names:
    print(f"    {n};")
print("  local:\n    *;\n};")
