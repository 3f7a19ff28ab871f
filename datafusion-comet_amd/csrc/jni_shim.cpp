// placeholder until the JNI shim lands (see jni_shim.cpp in a later commit)
