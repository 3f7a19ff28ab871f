// JNI entry points of libcomet.so — the symbols org.apache.comet.Native / NativeBase bind
// (spark/src/main/scala/org/apache/comet/Native.scala:60-111, spark/src/main/java/org/apache/comet/NativeBase.java).
// Each export is a thin shim over the JVM-free C ABI (include/comet_amd.h): unpack Java arrays/objects,
// call comet_*, map failures to the Java exception classes the reference throws
// (native/jni-bridge/src/errors.rs:473-560) and return the type's zero value (errors.rs:390-470).
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/comet_amd.h"
#include "third_party/jni_min.h"

namespace {

JavaVM* g_vm = nullptr;

struct BlockIterator;
struct JvmSubqueries { jlong plan_id; };      // createPlan's plan id: what CometScalarSubquery's static methods are asked with
struct JavaSide {                 // global refs held for the lifetime of a plan (jni_api.rs:423-436,517-527)
  std::vector<jobject> iterators;
  jobject metrics_node = nullptr;
  long long metrics_interval_ms = 0;                           // createPlan's metricsUpdateInterval (jni_api.rs:906-909)
  struct MemoryManager* memory = nullptr;                      // createPlan's taskMemoryManager (owned; freed by releasePlan)
  JvmSubqueries* subqueries = nullptr;                    // createPlan's plan id, for CometScalarSubquery's static methods (owned)
  std::chrono::steady_clock::time_point last_metrics_push;
};

// The JNIEnv of the Native.executePlan call running on this thread: input callbacks (CometShuffleBlockIterator.hasNext /
// getBuffer) are made on the calling task thread with that call's env, like the reference does before polling
// (shuffle_scan.rs:112-135 "JNI calls cannot happen from within poll_next on tokio threads").
thread_local JNIEnv* t_env = nullptr;

// org.apache.spark.CometTaskMemoryManager: acquireMemory(J)J / releaseMemory(J)V (native/jni-bridge/src/comet_task_memory_manager.rs:32-60),
// called the way CometUnifiedMemoryPool calls them (unified_pool.rs:64-80) — on the task thread, with the env of the JNI call in progress
struct MemoryManager {
  jobject obj = nullptr;       // global ref
  jmethodID acquire = nullptr, release = nullptr;
};
int64_t mm_acquire(void* ctx, int64_t bytes) {
  auto* m = (MemoryManager*)ctx;
  JNIEnv* env = t_env;
  if (!env || !m->obj) return bytes;          // outside a JNI call nothing can be asked: grant (the bytes are still counted)
  const jlong got = jni_CallLongMethodJ(env, m->obj, m->acquire, (jlong)bytes);
  if (jni_ExceptionCheck(env)) return 0;      // the pending throwable surfaces when the native call returns
  return (int64_t)got;
}
void mm_release(void* ctx, int64_t bytes) {
  auto* m = (MemoryManager*)ctx;
  JNIEnv* env = t_env;
  if (!env || !m->obj) return;
  jni_CallVoidMethodJ(env, m->obj, m->release, (jlong)bytes);
}

// org.apache.comet.CometShuffleBlockIterator seen as a CometShuffleBlockStream (native/jni-bridge/src/shuffle_block_iterator.rs:40-66)
struct BlockIterator {
  CometShuffleBlockStream c;
  jobject iter;                // global ref, owned by JavaSide::iterators
  jmethodID has_next, get_buffer;
  std::string error;
};
int64_t bi_next(CometShuffleBlockStream* self, const uint8_t** data) {
  auto* b = (BlockIterator*)self->private_data;
  JNIEnv* env = t_env;
  if (!env) { b->error = "shuffle block iterator polled outside Native.executePlan"; return -2; }
  const jint len = jni_CallIntMethod0(env, b->iter, b->has_next);   // reads the next block, returns its length or -1
  if (jni_ExceptionCheck(env)) { b->error = "CometShuffleBlockIterator.hasNext threw"; return -2; }
  if (len == -1) return -1;
  jobject buf = jni_CallObjectMethod0(env, b->iter, b->get_buffer);
  if (jni_ExceptionCheck(env) || !buf) { b->error = "CometShuffleBlockIterator.getBuffer failed"; return -2; }
  void* p = jni_GetDirectBufferAddress(env, buf);
  jni_DeleteLocalRef(env, buf);
  if (!p) { b->error = "CometShuffleBlockIterator.getBuffer did not return a direct ByteBuffer"; return -2; }
  *data = (const uint8_t*)p;
  return len;
}
const char* bi_error(CometShuffleBlockStream* self) { return ((BlockIterator*)self->private_data)->error.c_str(); }
void bi_release(CometShuffleBlockStream* self) { delete (BlockIterator*)self->private_data; }
std::mutex g_mu;
std::map<jlong, JavaSide> g_java;

void throw_java(JNIEnv* env, int kind, const char* msg) {
  if (jni_ExceptionCheck(env)) return;  // a pending Java throwable is re-thrown as is (errors.rs:481-506)
  const char* cls = kind == COMET_ERR_QUERY_EXECUTION ? "org/apache/comet/exceptions/CometQueryExecutionException"
                                                       : "org/apache/comet/CometNativeException";
  jclass c = jni_FindClass(env, cls);
  if (c) jni_ThrowNew(env, c, msg ? msg : "native error");
}

std::vector<uint8_t> byte_array(JNIEnv* env, jbyteArray a) {
  std::vector<uint8_t> v;
  if (!a) return v;
  jsize n = jni_GetArrayLength(env, a);
  v.resize((size_t)n);
  if (n) jni_GetByteArrayRegion(env, a, 0, n, (jbyte*)v.data());
  return v;
}

int pick_device(jlong task_attempt_id) {
  // COMET_GPU_DEVICES="0,1,2,3": tasks are spread round-robin over the listed HIP devices
  // (one native plan = one Spark partition = one GPU).
  const char* e = getenv("COMET_GPU_DEVICES");
  std::vector<int> devs;
  if (e && *e) {
    const char* p = e;
    while (*p) {
      devs.push_back(atoi(p));
      while (*p && *p != ',') p++;
      if (*p == ',') p++;
    }
  }
  if (devs.empty()) devs.push_back(0);
  return devs[(size_t)((uint64_t)task_attempt_id % devs.size())];
}

void push_metrics(JNIEnv* env, jlong handle, jobject node) {
  // CometMetricNode.set_all_from_bytes([B)V (native/jni-bridge/src/comet_metric_node.rs:62, metrics/utils.rs:30-45)
  if (!node) return;
  int64_t n = comet_plan_metrics(handle, nullptr, 0);
  if (n <= 0) return;
  std::vector<uint8_t> buf((size_t)n);
  comet_plan_metrics(handle, buf.data(), buf.size());
  jclass cls = jni_GetObjectClass(env, node);
  if (!cls) return;
  jmethodID mid = jni_GetMethodID(env, cls, "set_all_from_bytes", "([B)V");
  if (!mid) return;
  jbyteArray arr = jni_NewByteArray(env, (jsize)n);
  if (!arr) return;
  jni_SetByteArrayRegion(env, arr, 0, (jsize)n, (const jbyte*)buf.data());
  jni_CallVoidMethod1(env, node, mid, arr);
  jni_DeleteLocalRef(env, arr);
}

}  // namespace

extern "C" {

JNIEXPORT void JNICALL Java_org_apache_comet_NativeBase_init(JNIEnv* env, jclass, jstring /*logConfPath*/, jstring /*logLevel*/) {
  // native/core/src/lib.rs:91-126: remember the JavaVM for worker-thread attachment; logging goes to stderr
  jni_GetJavaVM(env, &g_vm);
}
JNIEXPORT void JNICALL Java_org_apache_comet_NativeBase_release(JNIEnv*, jclass) {}
JNIEXPORT jboolean JNICALL Java_org_apache_comet_NativeBase_isFeatureEnabled(JNIEnv*, jclass, jstring) { return 0; }
JNIEXPORT jboolean JNICALL Java_org_apache_comet_NativeBase_isObjectStoreSchemeSupported(JNIEnv* env, jclass, jstring url) {
  if (!url) return 0;
  const char* s = jni_GetStringUTFChars(env, url);
  jboolean ok = s && (strncmp(s, "file:", 5) == 0 || s[0] == '/');
  if (s) jni_ReleaseStringUTFChars(env, url, s);
  return ok;
}

// Native.createPlan (jni_api.rs:371-562)
// Scalar subqueries: org.apache.spark.sql.comet.CometScalarSubquery's static methods (jni-bridge/src/comet_exec.rs:54-126), asked with the plan id createPlan was
// given and the subquery's id — on the task thread, with the env of the executePlan call in progress (expressions/subquery.rs:72-180).
// modified UTF-8 (what GetStringUTFChars hands out) → UTF-8: C0 80 is NUL, a supplementary character comes as two three-byte surrogates
static std::string from_modified_utf8(const char* s) {
  std::string o;
  const unsigned char* p = (const unsigned char*)s;
  while (*p) {
    if (p[0] == 0xC0 && p[1] == 0x80) { o.push_back('\0'); p += 2; continue; }
    if (p[0] == 0xED && (p[1] & 0xF0) == 0xA0 && p[2] && p[3] == 0xED && (p[4] & 0xF0) == 0xB0 && p[5]) {
      const unsigned hi = 0xD000u | ((p[1] & 0x3Fu) << 6) | (p[2] & 0x3Fu), lo = 0xD000u | ((p[4] & 0x3Fu) << 6) | (p[5] & 0x3Fu);
      const unsigned cp = 0x10000u + ((hi - 0xD800u) << 10) + (lo - 0xDC00u);
      o.push_back((char)(0xF0 | (cp >> 18)));
      o.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
      o.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
      o.push_back((char)(0x80 | (cp & 0x3F)));
      p += 6;
      continue;
    }
    o.push_back((char)*p++);
  }
  return o;
}
static int32_t subquery_from_jvm(void* vctx, int64_t id, int32_t type_id, int32_t* is_null, uint8_t* out, int64_t cap, int64_t* len) {
  JNIEnv* env = t_env;
  JvmSubqueries* c = (JvmSubqueries*)vctx;
  if (!env || !c) return -1;
  jclass cls = jni_FindClass(env, "org/apache/spark/sql/comet/CometScalarSubquery");
  if (!cls || jni_ExceptionCheck(env)) return -1;
  struct LocalRef {      // the class reference goes on EVERY way out: the calls run inside one long-lived executePlan frame, where leaked locals pile up
    JNIEnv* env;
    jobject o;
    ~LocalRef() { if (o) jni_DeleteLocalRef(env, o); }
  } cls_guard{env, (jobject)cls};
  auto method = [&](const char* name, const char* sig) { return jni_GetStaticMethodID(env, cls, name, sig); };
  jmethodID m_null = method("isNull", "(JJ)Z");
  if (!m_null || jni_ExceptionCheck(env)) return -1;
  const bool nul = jni_CallStaticBooleanMethodJJ(env, cls, m_null, c->plan_id, (jlong)id) != 0;
  if (jni_ExceptionCheck(env)) return -1;
  *is_null = nul ? 1 : 0;
  *len = 0;
  if (nul) return 1;
  auto put64 = [&](int64_t v) { if (cap >= 8) memcpy(out, &v, 8); *len = 8; };
  auto putf = [&](double v) { if (cap >= 8) memcpy(out, &v, 8); *len = 8; };
  auto put_bytes = [&](const void* p, size_t n) { if ((int64_t)n <= cap) memcpy(out, p, n); *len = (int64_t)n; };
  jmethodID m = nullptr;
  switch (type_id) {      // types.proto:43-66
    case 0: m = method("getBoolean", "(JJ)Z"); if (m) { const uint8_t b = jni_CallStaticBooleanMethodJJ(env, cls, m, c->plan_id, (jlong)id) ? 1 : 0; put_bytes(&b, 1); } break;
    case 1: m = method("getByte", "(JJ)B"); if (m) put64((int64_t)jni_CallStaticByteMethodJJ(env, cls, m, c->plan_id, (jlong)id)); break;
    case 2: m = method("getShort", "(JJ)S"); if (m) put64((int64_t)jni_CallStaticShortMethodJJ(env, cls, m, c->plan_id, (jlong)id)); break;
    case 3: case 12: m = method("getInt", "(JJ)I"); if (m) put64((int64_t)jni_CallStaticIntMethodJJ(env, cls, m, c->plan_id, (jlong)id)); break;            // int, date
    case 4: case 9: case 11: m = method("getLong", "(JJ)J"); if (m) put64((int64_t)jni_CallStaticLongMethodJJ(env, cls, m, c->plan_id, (jlong)id)); break;   // long, timestamp, timestamp_ntz
    case 5: m = method("getFloat", "(JJ)F"); if (m) putf((double)jni_CallStaticFloatMethodJJ(env, cls, m, c->plan_id, (jlong)id)); break;
    case 6: m = method("getDouble", "(JJ)D"); if (m) putf(jni_CallStaticDoubleMethodJJ(env, cls, m, c->plan_id, (jlong)id)); break;
    case 10: case 8: {      // decimal: BigInteger.toByteArray; binary: the bytes
      m = method(type_id == 10 ? "getDecimal" : "getBinary", "(JJ)[B");
      if (!m) break;
      jbyteArray a = (jbyteArray)jni_CallStaticObjectMethodJJ(env, cls, m, c->plan_id, (jlong)id);
      LocalRef a_guard{env, (jobject)a};
      if (!a || jni_ExceptionCheck(env)) return -1;
      const jsize n = jni_GetArrayLength(env, a);
      std::vector<uint8_t> tmp((size_t)n + 1);
      jni_GetByteArrayRegion(env, a, 0, n, (jbyte*)tmp.data());
      put_bytes(tmp.data(), (size_t)n);
      break;
    }
    case 7: {      // string
      m = method("getString", "(JJ)Ljava/lang/String;");
      if (!m) break;
      jstring js = (jstring)jni_CallStaticObjectMethodJJ(env, cls, m, c->plan_id, (jlong)id);
      LocalRef js_guard{env, (jobject)js};
      if (!js || jni_ExceptionCheck(env)) return -1;
      const char* chars = jni_GetStringUTFChars(env, js);
      const std::string u = from_modified_utf8(chars ? chars : "");
      if (chars) jni_ReleaseStringUTFChars(env, js, chars);
      put_bytes(u.data(), u.size());
      break;
    }
    default: return -1;
  }
  const bool failed = !m || jni_ExceptionCheck(env);      // (one class lookup per subquery and task: the values are asked for once)
  return failed ? -1 : 1;
}

JNIEXPORT jlong JNICALL Java_org_apache_comet_Native_createPlan(
    JNIEnv* env, jclass, jlong plan_id, jobjectArray iterators, jbyteArray plan, jbyteArray configMap, jint partitionCount,
    jobject metricsNode, jlong metricsUpdateInterval, jobject taskMemoryManager, jobjectArray /*localDirs*/, jint batchSize,
    jboolean /*offHeapMode*/, jstring /*memoryPoolType*/, jlong /*memoryLimit*/, jlong /*memoryLimitPerTask*/, jlong taskAttemptId,
    jlong /*taskCPUs*/, jobject /*keyUnwrapper*/, jobject /*taskContext*/, jobject /*classLoader*/) {
  std::vector<uint8_t> plan_b = byte_array(env, plan), cfg_b = byte_array(env, configMap);
  JavaSide js;
  std::vector<void*> inputs;
  std::vector<int32_t> kinds;
  const jsize n_it = iterators ? jni_GetArrayLength(env, iterators) : 0;
  // an error before comet_create_plan is reached: nothing has been handed to the library yet — drop what this call created (block
  // iterator wrappers, global refs); the Arrow streams stay with the JVM, which closes them when createPlan throws
  auto abandon = [&]() {
    for (size_t k = 0; k < inputs.size(); k++)
      if (kinds[k] == COMET_INPUT_SHUFFLE_BLOCKS) bi_release((CometShuffleBlockStream*)inputs[k]);
    for (jobject g : js.iterators) jni_DeleteGlobalRef(env, g);
    js.iterators.clear();
  };
  for (jsize i = 0; i < n_it; i++) {
    jobject it = jni_GetObjectArrayElement(env, iterators, i);
    if (!it) { abandon(); throw_java(env, COMET_ERR_NATIVE, "null input iterator"); return 0; }
    // org.apache.arrow.c.ArrowArrayStream.memoryAddress()J (native/jni-bridge/src/arrow_array_stream.rs:45); the
    // native side takes ownership of the C struct at that address (scan.rs:98-106)
    jclass cls = jni_GetObjectClass(env, it);
    jmethodID mid = cls ? jni_GetMethodID(env, cls, "memoryAddress", "()J") : nullptr;
    if (!mid) {
      // not an ArrowArrayStream: a CometShuffleBlockIterator feeding a ShuffleScan leaf (jni_api.rs:455-470)?
      if (jni_ExceptionCheck(env)) jni_ExceptionClear(env);   // NoSuchMethodError from the probe above
      jmethodID has_next = cls ? jni_GetMethodID(env, cls, "hasNext", "()I") : nullptr;
      jmethodID get_buffer = has_next ? jni_GetMethodID(env, cls, "getBuffer", "()Ljava/nio/ByteBuffer;") : nullptr;
      if (!has_next || !get_buffer) {
        if (jni_ExceptionCheck(env)) jni_ExceptionClear(env);
        abandon();
        throw_java(env, COMET_ERR_NATIVE, "input iterator is neither an org.apache.arrow.c.ArrowArrayStream nor an org.apache.comet.CometShuffleBlockIterator");
        return 0;
      }
      auto* b = new BlockIterator();
      b->iter = jni_NewGlobalRef(env, it);
      b->has_next = has_next;
      b->get_buffer = get_buffer;
      b->c.next_block = bi_next;
      b->c.get_last_error = bi_error;
      b->c.release = bi_release;
      b->c.private_data = b;
      inputs.push_back(&b->c);
      kinds.push_back(COMET_INPUT_SHUFFLE_BLOCKS);
      js.iterators.push_back(b->iter);
      jni_DeleteLocalRef(env, it);
      continue;
    }
    jlong addr = jni_CallLongMethod0(env, it, mid);
    if (jni_ExceptionCheck(env)) { abandon(); return 0; }   // the pending Java throwable is re-thrown as is
    inputs.push_back((void*)(intptr_t)addr);
    kinds.push_back(COMET_INPUT_HOST_STREAM);
    js.iterators.push_back(jni_NewGlobalRef(env, it));
    jni_DeleteLocalRef(env, it);
  }
  int64_t h = comet_create_plan(plan_b.data(), plan_b.size(), cfg_b.empty() ? nullptr : cfg_b.data(), cfg_b.size(), inputs.data(),
                                kinds.data(), (int32_t)inputs.size(), partitionCount, batchSize, pick_device(taskAttemptId));
  if (h == 0) {
    // comet_create_plan released every stream it was handed (the block-iterator wrappers included); only the global refs are ours
    for (jobject g : js.iterators) jni_DeleteGlobalRef(env, g);
    throw_java(env, comet_last_error_kind(0), comet_last_error(0));
    return 0;
  }
  if (taskMemoryManager) {
    jclass cls = jni_GetObjectClass(env, taskMemoryManager);
    jmethodID acq = cls ? jni_GetMethodID(env, cls, "acquireMemory", "(J)J") : nullptr;
    jmethodID rel = acq ? jni_GetMethodID(env, cls, "releaseMemory", "(J)V") : nullptr;
    if (acq && rel) {
      js.memory = new MemoryManager();
      js.memory->obj = jni_NewGlobalRef(env, taskMemoryManager);
      js.memory->acquire = acq;
      js.memory->release = rel;
      comet_plan_set_memory_manager(h, mm_acquire, mm_release, js.memory, (int64_t)taskAttemptId);
    } else if (jni_ExceptionCheck(env)) {
      jni_ExceptionClear(env);      // not a CometTaskMemoryManager: run unaccounted rather than fail the task
    }
  }
  js.subqueries = new JvmSubqueries{plan_id};
  comet_plan_set_subquery_provider(h, subquery_from_jvm, js.subqueries);
  if (metricsNode) js.metrics_node = jni_NewGlobalRef(env, metricsNode);
  js.metrics_interval_ms = (long long)metricsUpdateInterval;
  js.last_metrics_push = std::chrono::steady_clock::now();
  std::lock_guard<std::mutex> lk(g_mu);
  g_java[h] = js;
  return (jlong)h;
}

// Native.executePlan (jni_api.rs:767-957): returns rows, or -1 at end of stream
JNIEXPORT jlong JNICALL Java_org_apache_comet_Native_executePlan(JNIEnv* env, jclass, jint /*stage*/, jint /*partition*/, jlong handle,
                                                                 jlongArray arrayAddrs, jlongArray schemaAddrs) {
  const jsize n = arrayAddrs ? jni_GetArrayLength(env, arrayAddrs) : 0;
  const jsize ns = schemaAddrs ? jni_GetArrayLength(env, schemaAddrs) : 0;
  if (n != ns) { throw_java(env, COMET_ERR_NATIVE, "arrayAddrs and schemaAddrs differ in length"); return 0; }
  std::vector<jlong> aa((size_t)n), sa((size_t)n);
  if (n) {
    jni_GetLongArrayRegion(env, arrayAddrs, 0, n, aa.data());
    jni_GetLongArrayRegion(env, schemaAddrs, 0, n, sa.data());
  }
  std::vector<struct ArrowArray*> arrays((size_t)n);
  std::vector<struct ArrowSchema*> schemas((size_t)n);
  for (jsize i = 0; i < n; i++) {
    arrays[(size_t)i] = (struct ArrowArray*)(intptr_t)aa[(size_t)i];
    schemas[(size_t)i] = (struct ArrowSchema*)(intptr_t)sa[(size_t)i];
  }
  t_env = env;
  int64_t rows = comet_execute_plan(handle, arrays.data(), schemas.data(), (int32_t)n);
  t_env = nullptr;
  if (rows == -2) {
    throw_java(env, comet_last_error_kind(handle), comet_last_error(handle));
    return 0;
  }
  // metrics reach the JVM at end of stream and, while batches flow, at most every metricsUpdateInterval ms (jni_api.rs:897-909)
  jobject node = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_java.find(handle);
    if (it != g_java.end()) {
      JavaSide& js = it->second;
      const auto now = std::chrono::steady_clock::now();
      const bool due = js.metrics_interval_ms > 0 &&
                       std::chrono::duration_cast<std::chrono::milliseconds>(now - js.last_metrics_push).count() >= js.metrics_interval_ms;
      if (rows == -1 || due) {
        node = js.metrics_node;
        js.last_metrics_push = now;
      }
    }
  }
  if (node) push_metrics(env, handle, node);
  return (jlong)rows;
}

// Native.releasePlan (jni_api.rs:961-990): final metrics push, then drop (fires ArrowArrayStream.release)
JNIEXPORT void JNICALL Java_org_apache_comet_Native_releasePlan(JNIEnv* env, jclass, jlong handle) {
  JavaSide js;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_java.find(handle);
    if (it != g_java.end()) {
      js = it->second;
      g_java.erase(it);
    }
  }
  push_metrics(env, handle, js.metrics_node);
  t_env = env;                   // the plan's buffers go back to the task memory manager from this call
  comet_release_plan(handle);
  t_env = nullptr;
  if (js.memory) {
    jni_DeleteGlobalRef(env, js.memory->obj);
    delete js.memory;
  }
  delete js.subqueries;
  for (jobject g : js.iterators) jni_DeleteGlobalRef(env, g);
  if (js.metrics_node) jni_DeleteGlobalRef(env, js.metrics_node);
}

// tracing / misc entry points the JVM may call on any plan (jni_api.rs:1186-1232): accepted, no-ops here
JNIEXPORT void JNICALL Java_org_apache_comet_Native_traceBegin(JNIEnv*, jclass, jstring) {}
JNIEXPORT void JNICALL Java_org_apache_comet_Native_traceEnd(JNIEnv*, jclass, jstring) {}
JNIEXPORT void JNICALL Java_org_apache_comet_Native_logMemoryUsage(JNIEnv*, jclass, jstring, jlong) {}
JNIEXPORT jlong JNICALL Java_org_apache_comet_Native_getRustThreadId(JNIEnv*, jclass) { return 0; }

// JVM-shuffle (row-based) entry points (jni_api.rs:1047,1130): host-memory work on Spark's UnsafeRow pages (row_shuffle.cpp)
// Native.writeSortedFileNative (Native.scala:146-158, jni_api.rs:1043-1127): the JVM shuffle sorter's rows → shuffle blocks appended to a file.
// preferDictionaryRatio only chooses between a dictionary and a plain encoding of string columns in the reference (row.rs:1440-1486);
// readers unpack dictionaries (shuffle_scan.rs:175-183), so the plain encoding written here decodes to the same batches.
JNIEXPORT jlongArray JNICALL Java_org_apache_comet_Native_writeSortedFileNative(JNIEnv* env, jclass, jlongArray addresses, jintArray rowSizes,
                                                                                jobjectArray datatypes, jstring file, jdouble /*preferDictionaryRatio*/,
                                                                                jint batchSize, jboolean checksumEnabled, jint checksumAlgo,
                                                                                jlong currentChecksum, jstring compressionCodec, jint compressionLevel,
                                                                                jboolean /*tracingEnabled*/) {
  const jsize n = addresses ? jni_GetArrayLength(env, addresses) : 0;
  const jsize ns = rowSizes ? jni_GetArrayLength(env, rowSizes) : 0;
  if (n != ns) { throw_java(env, COMET_ERR_NATIVE, "writeSortedFileNative: addresses and rowSizes differ in length"); return nullptr; }
  std::vector<jlong> addrs((size_t)n);
  std::vector<jint> sizes((size_t)n);
  if (n) {
    jni_GetLongArrayRegion(env, addresses, 0, n, addrs.data());
    jni_GetIntArrayRegion(env, rowSizes, 0, n, sizes.data());
  }
  const jsize nc = datatypes ? jni_GetArrayLength(env, datatypes) : 0;
  std::vector<std::vector<uint8_t>> types((size_t)nc);
  std::vector<const uint8_t*> tptr((size_t)nc);
  std::vector<int32_t> tlen((size_t)nc);
  for (jsize i = 0; i < nc; i++) {
    jbyteArray b = (jbyteArray)jni_GetObjectArrayElement(env, datatypes, i);
    const jsize len = b ? jni_GetArrayLength(env, b) : 0;
    types[(size_t)i].resize((size_t)len + 1);
    if (len) jni_GetByteArrayRegion(env, b, 0, len, (jbyte*)types[(size_t)i].data());
    if (b) jni_DeleteLocalRef(env, b);
    tptr[(size_t)i] = types[(size_t)i].data();
    tlen[(size_t)i] = (int32_t)len;
  }
  const char* path = file ? jni_GetStringUTFChars(env, file) : nullptr;
  const char* codec = compressionCodec ? jni_GetStringUTFChars(env, compressionCodec) : nullptr;
  int64_t res[3] = {0, 0, 0};
  const int32_t rc = comet_write_sorted_rows((const int64_t*)addrs.data(), (const int32_t*)sizes.data(), n, tptr.data(), tlen.data(), (int32_t)nc, path, batchSize,
                                             checksumEnabled ? 1 : 0, checksumAlgo, currentChecksum, codec, compressionLevel, res);
  if (path) jni_ReleaseStringUTFChars(env, file, path);
  if (codec) jni_ReleaseStringUTFChars(env, compressionCodec, codec);
  if (rc != 0) { throw_java(env, comet_last_error_kind(0), comet_last_error(0)); return nullptr; }
  jlongArray out = (jlongArray)jni_NewLongArray(env, 3);
  if (!out) { throw_java(env, COMET_ERR_NATIVE, "writeSortedFileNative: cannot allocate the result array"); return nullptr; }
  jni_SetLongArrayRegion(env, out, 0, 3, (const jlong*)res);
  return out;
}
// Native.sortRowPartitionsNative (jni_api.rs:1130-1160)
JNIEXPORT void JNICALL Java_org_apache_comet_Native_sortRowPartitionsNative(JNIEnv* env, jclass, jlong address, jlong size, jboolean /*tracingEnabled*/) {
  if (address == 0 || size < 0) { throw_java(env, COMET_ERR_NATIVE, "sortRowPartitionsNative: null address or negative size"); return; }
  comet_sort_row_partitions((int64_t*)(intptr_t)address, size);
}
// Native.decodeShuffleBlock (jni_api.rs:1163-1181): one block in a direct ByteBuffer → Arrow C Data structs at the given addresses
JNIEXPORT jlong JNICALL Java_org_apache_comet_Native_decodeShuffleBlock(JNIEnv* env, jclass, jobject byteBuffer, jint length, jlongArray arrayAddrs,
                                                                        jlongArray schemaAddrs, jboolean /*tracingEnabled*/) {
  void* p = byteBuffer ? jni_GetDirectBufferAddress(env, byteBuffer) : nullptr;
  if (!p) { throw_java(env, COMET_ERR_NATIVE, "decodeShuffleBlock: not a direct ByteBuffer"); return 0; }
  const jsize n = arrayAddrs ? jni_GetArrayLength(env, arrayAddrs) : 0;
  const jsize ns = schemaAddrs ? jni_GetArrayLength(env, schemaAddrs) : 0;
  if (n != ns) { throw_java(env, COMET_ERR_NATIVE, "arrayAddrs and schemaAddrs differ in length"); return 0; }
  std::vector<jlong> aa((size_t)n), sa((size_t)n);
  if (n) {
    jni_GetLongArrayRegion(env, arrayAddrs, 0, n, aa.data());
    jni_GetLongArrayRegion(env, schemaAddrs, 0, n, sa.data());
  }
  std::vector<struct ArrowArray*> arrays((size_t)n);
  std::vector<struct ArrowSchema*> schemas((size_t)n);
  for (jsize i = 0; i < n; i++) {
    arrays[(size_t)i] = (struct ArrowArray*)(intptr_t)aa[(size_t)i];
    schemas[(size_t)i] = (struct ArrowSchema*)(intptr_t)sa[(size_t)i];
  }
  const int64_t rows = comet_decode_shuffle_block((const uint8_t*)p, length, arrays.data(), schemas.data(), (int32_t)n);
  if (rows < 0) {
    throw_java(env, comet_last_error_kind(0), comet_last_error(0));
    return 0;
  }
  return (jlong)rows;
}
// Native.columnarToRowInit / Convert / Close (jni_api.rs:1253-1377).  The serialized schema is not needed: the types come with the
// Arrow C Data structs of every batch.
JNIEXPORT jlong JNICALL Java_org_apache_comet_Native_columnarToRowInit(JNIEnv* env, jclass, jobjectArray /*serializedSchema*/, jint batchSize) {
  const int64_t h = comet_columnar_to_row_init(batchSize, pick_device(0));
  if (h == 0) throw_java(env, COMET_ERR_NATIVE, comet_columnar_to_row_error(0));
  return (jlong)h;
}
JNIEXPORT jobject JNICALL Java_org_apache_comet_Native_columnarToRowConvert(JNIEnv* env, jclass, jlong handle, jlongArray arrayAddrs, jlongArray schemaAddrs, jint numRows) {
  const jsize n = arrayAddrs ? jni_GetArrayLength(env, arrayAddrs) : 0;
  const jsize ns = schemaAddrs ? jni_GetArrayLength(env, schemaAddrs) : 0;
  if (n != ns) { throw_java(env, COMET_ERR_NATIVE, "arrayAddrs and schemaAddrs differ in length"); return nullptr; }
  std::vector<jlong> aa((size_t)n), sa((size_t)n);
  if (n) {
    jni_GetLongArrayRegion(env, arrayAddrs, 0, n, aa.data());
    jni_GetLongArrayRegion(env, schemaAddrs, 0, n, sa.data());
  }
  std::vector<struct ArrowArray*> arrays((size_t)n);
  std::vector<struct ArrowSchema*> schemas((size_t)n);
  for (jsize i = 0; i < n; i++) {
    arrays[(size_t)i] = (struct ArrowArray*)(intptr_t)aa[(size_t)i];
    schemas[(size_t)i] = (struct ArrowSchema*)(intptr_t)sa[(size_t)i];
  }
  const uint8_t* buf = nullptr;
  const int32_t *offs = nullptr, *lens = nullptr;
  if (comet_columnar_to_row_convert(handle, arrays.data(), schemas.data(), (int32_t)n, numRows, &buf, &offs, &lens) != 0) {
    throw_java(env, COMET_ERR_NATIVE, comet_columnar_to_row_error(handle));
    return nullptr;
  }
  // new NativeColumnarToRowInfo(long memoryAddress, int[] offsets, int[] lengths)
  jobject jo = jni_NewIntArray(env, numRows), jl = jni_NewIntArray(env, numRows);
  if (!jo || !jl) return nullptr;
  if (numRows) {
    jni_SetIntArrayRegion(env, jo, 0, numRows, (const jint*)offs);
    jni_SetIntArrayRegion(env, jl, 0, numRows, (const jint*)lens);
  }
  jclass cls = jni_FindClass(env, "org/apache/comet/NativeColumnarToRowInfo");
  jmethodID ctor = cls ? jni_GetMethodID(env, cls, "<init>", "(J[I[I)V") : nullptr;
  if (!ctor) { throw_java(env, COMET_ERR_NATIVE, "org.apache.comet.NativeColumnarToRowInfo(long, int[], int[]) not found"); return nullptr; }
  return jni_NewObject3(env, cls, ctor, (jlong)(intptr_t)buf, jo, jl);
}
JNIEXPORT void JNICALL Java_org_apache_comet_Native_columnarToRowClose(JNIEnv*, jclass, jlong handle) { comet_columnar_to_row_close(handle); }


// ---- org.apache.comet.parquet.Native (native/core/src/parquet/mod.rs:135-330): the record-batch reader of the iceberg-compat scan ----
// errors of the file format itself (footer, Thrift, page walk, codecs: messages that start "parquet:") are the reference's CometError::Parquet →
// org/apache/comet/ParquetRuntimeException (jni-bridge/src/errors.rs:333-336); everything else keeps its class
static void throw_reader_error(JNIEnv* env) {
  const char* msg = comet_last_error(0);
  if (msg && (strncmp(msg, "parquet:", 8) == 0 || strncmp(msg, "snappy:", 7) == 0) && !jni_ExceptionCheck(env)) {
    jclass c = jni_FindClass(env, "org/apache/comet/ParquetRuntimeException");
    if (c) { jni_ThrowNew(env, c, msg); return; }
  }
  throw_java(env, comet_last_error_kind(0), msg);
}
JNIEXPORT jlong JNICALL Java_org_apache_comet_parquet_Native_initRecordBatchReader(
    JNIEnv* env, jclass, jstring filePath, jlong fileSize, jlongArray starts, jlongArray lengths, jbyteArray filter, jbyteArray requiredSchema,
    jbyteArray dataSchema, jstring sessionTimezone, jint batchSize, jboolean caseSensitive, jboolean /*returnNullStructIfAllFieldsMissing*/,
    jobject /*objectStoreOptions*/, jobject keyUnwrapper, jobject /*metricsNode*/) {
  if (keyUnwrapper) { throw_java(env, COMET_ERR_NATIVE, "Parquet modular encryption is not supported by the MI355X native engine"); return 0; }
  if (!filePath) { throw_java(env, COMET_ERR_NATIVE, "initRecordBatchReader: null file path"); return 0; }
  const char* p = jni_GetStringUTFChars(env, filePath);
  std::string path = p ? p : "";
  if (p) jni_ReleaseStringUTFChars(env, filePath, p);
  std::string tz = "UTC";
  if (sessionTimezone) {
    const char* t = jni_GetStringUTFChars(env, sessionTimezone);
    if (t) { tz = t; jni_ReleaseStringUTFChars(env, sessionTimezone, t); }
  }
  const jsize ns = starts ? jni_GetArrayLength(env, starts) : 0, nl = lengths ? jni_GetArrayLength(env, lengths) : 0;
  if (ns != nl) { throw_java(env, COMET_ERR_NATIVE, "initRecordBatchReader: starts and lengths differ in length"); return 0; }
  std::vector<jlong> st((size_t)ns), ln((size_t)ns);
  if (ns) {
    jni_GetLongArrayRegion(env, starts, 0, ns, st.data());
    jni_GetLongArrayRegion(env, lengths, 0, ns, ln.data());
  }
  std::vector<uint8_t> f = byte_array(env, filter), rs = byte_array(env, requiredSchema), ds = byte_array(env, dataSchema);
  static_assert(sizeof(jlong) == sizeof(int64_t), "jlong is 64 bits");
  int64_t h = comet_parquet_reader_init(path.c_str(), (int64_t)fileSize, (const int64_t*)st.data(), (const int64_t*)ln.data(), (int32_t)ns,
                                        f.empty() ? nullptr : f.data(), f.size(), rs.empty() ? nullptr : rs.data(), rs.size(),
                                        ds.empty() ? nullptr : ds.data(), ds.size(), tz.c_str(), (int32_t)batchSize, caseSensitive ? 1 : 0, pick_device(0));
  if (h == 0) { throw_reader_error(env); return 0; }
  return (jlong)h;
}
JNIEXPORT jint JNICALL Java_org_apache_comet_parquet_Native_readNextRecordBatch(JNIEnv* env, jclass, jlong handle) {
  const int32_t rows = comet_parquet_reader_next((int64_t)handle);
  if (rows == -2) { throw_reader_error(env); return 0; }
  return (jint)rows;
}
JNIEXPORT void JNICALL Java_org_apache_comet_parquet_Native_currentColumnBatch(JNIEnv* env, jclass, jlong handle, jint columnIdx, jlong arrayAddr,
                                                                               jlong schemaAddr) {
  if (comet_parquet_reader_column((int64_t)handle, (int32_t)columnIdx, (struct ArrowArray*)(intptr_t)arrayAddr, (struct ArrowSchema*)(intptr_t)schemaAddr) != 0)
    throw_reader_error(env);
}
JNIEXPORT void JNICALL Java_org_apache_comet_parquet_Native_closeRecordBatchReader(JNIEnv*, jclass, jlong handle) {
  comet_parquet_reader_close((int64_t)handle);
}
}  // extern "C"
