/* A JVM-less JNIEnv for exercising libcomet's JNI shim from ctypes (TEST INFRASTRUCTURE).
 * Implements exactly the JNI functions jni_shim.cpp calls, at their specified function-table indices, over
 * tiny C "objects".  It proves the shim's marshalling, ownership and exception mapping without a JVM. */
#include <stdarg.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../datafusion-comet_amd/csrc/third_party/jni_min.h"

enum { K_BYTES = 1, K_LONGS, K_OBJS, K_STREAM, K_METRICS, K_STRING, K_CLASS, K_BLOCKITER, K_INTS, K_INFO, K_MEMMGR };
typedef struct MObj {
  int kind;
  int64_t len;
  void* data;          /* bytes / longs / MObj** / char* */
  int64_t addr;        /* K_STREAM: ArrowArrayStream address */
  int global_refs;
  const char* cname;   /* K_CLASS */
} MObj;

static struct JNINativeInterface_min g_tab;
static const struct JNINativeInterface_min* g_env = &g_tab;
static struct JNIInvokeInterface_min g_vmtab;
static const struct JNIInvokeInterface_min* g_vm = &g_vmtab;
static char g_exc_class[256], g_exc_msg[8192];
static int g_exc_pending;
static int g_live_global_refs;
static MObj g_cls_stream = {K_CLASS, 0, 0, 0, 0, "org/apache/arrow/c/ArrowArrayStream"};
static MObj g_cls_metrics = {K_CLASS, 0, 0, 0, 0, "org/apache/spark/sql/comet/CometMetricNode"};
static MObj g_cls_other = {K_CLASS, 0, 0, 0, 0, "java/lang/Object"};
static MObj g_cls_blockiter = {K_CLASS, 0, 0, 0, 0, "org/apache/comet/CometShuffleBlockIterator"};
static MObj g_cls_memmgr = {K_CLASS, 0, 0, 0, 0, "org/apache/spark/CometTaskMemoryManager"};
static int g_mid_memaddr, g_mid_setall, g_mid_hasnext, g_mid_getbuffer, g_mid_info_ctor, g_mid_acquire, g_mid_release;

static jclass f_FindClass(JNIEnv* e, const char* n) { (void)e; MObj* c = calloc(1, sizeof *c); c->kind = K_CLASS; c->cname = strdup(n); return c; }
static jint f_ThrowNew(JNIEnv* e, jclass c, const char* m) {
  (void)e; g_exc_pending = 1;
  strncpy(g_exc_class, ((MObj*)c)->cname, sizeof g_exc_class - 1);
  strncpy(g_exc_msg, m ? m : "", sizeof g_exc_msg - 1);
  return 0;
}
static jboolean f_ExceptionCheck(JNIEnv* e) { (void)e; return (jboolean)g_exc_pending; }
static jobject f_NewGlobalRef(JNIEnv* e, jobject o) { (void)e; if (o) { ((MObj*)o)->global_refs++; g_live_global_refs++; } return o; }
static void f_DeleteGlobalRef(JNIEnv* e, jobject o) { (void)e; if (o) { ((MObj*)o)->global_refs--; g_live_global_refs--; } }
static void f_DeleteLocalRef(JNIEnv* e, jobject o) { (void)e; (void)o; }
static jclass f_GetObjectClass(JNIEnv* e, jobject o) {
  (void)e; MObj* m = o;
  return m->kind == K_STREAM ? (jclass)&g_cls_stream : m->kind == K_METRICS ? (jclass)&g_cls_metrics :
         m->kind == K_BLOCKITER ? (jclass)&g_cls_blockiter : m->kind == K_MEMMGR ? (jclass)&g_cls_memmgr : (jclass)&g_cls_other;
}
static jmethodID f_GetMethodID(JNIEnv* e, jclass c, const char* n, const char* sig) {
  (void)e;
  if (c == &g_cls_stream && !strcmp(n, "memoryAddress") && !strcmp(sig, "()J")) return &g_mid_memaddr;
  if (c == &g_cls_metrics && !strcmp(n, "set_all_from_bytes") && !strcmp(sig, "([B)V")) return &g_mid_setall;
  if (c && ((MObj*)c)->kind == K_CLASS && ((MObj*)c)->cname && !strcmp(((MObj*)c)->cname, "org/apache/comet/NativeColumnarToRowInfo") && !strcmp(n, "<init>") && !strcmp(sig, "(J[I[I)V"))
    return &g_mid_info_ctor;
  if (c == &g_cls_blockiter && !strcmp(n, "hasNext") && !strcmp(sig, "()I")) return &g_mid_hasnext;
  if (c == &g_cls_blockiter && !strcmp(n, "getBuffer") && !strcmp(sig, "()Ljava/nio/ByteBuffer;")) return &g_mid_getbuffer;
  if (c == &g_cls_memmgr && !strcmp(n, "acquireMemory") && !strcmp(sig, "(J)J")) return &g_mid_acquire;
  if (c == &g_cls_memmgr && !strcmp(n, "releaseMemory") && !strcmp(sig, "(J)V")) return &g_mid_release;
  return NULL;   /* a real JVM would also raise NoSuchMethodError */
}
/* K_MEMMGR (CometTaskMemoryManager): data = int64[6] {limit, used, peak, acquire calls, release calls, bytes refused} */
static jlong f_CallLongMethod(JNIEnv* e, jobject o, jmethodID m, ...) {
  (void)e;
  if (m == &g_mid_memaddr) return ((MObj*)o)->addr;
  if (m == &g_mid_acquire) {
    va_list ap; va_start(ap, m); jlong want = va_arg(ap, jlong); va_end(ap);
    int64_t* st = ((MObj*)o)->data;
    int64_t room = st[0] - st[1];
    int64_t got = want <= room ? want : (room > 0 ? room : 0);   /* Spark grants what is left, possibly less than asked */
    st[1] += got; if (st[1] > st[2]) st[2] = st[1];
    st[3]++; st[5] += want - got;
    return got;
  }
  return 0;
}
/* K_BLOCKITER: data = MObj** blocks (K_BYTES standing in for direct ByteBuffers), len = count, addr = cursor */
static jint f_CallIntMethod(JNIEnv* e, jobject o, jmethodID m, ...) {
  (void)e; MObj* it = o;
  if (m != &g_mid_hasnext) return 0;
  if (it->addr >= it->len) return -1;
  it->addr++;
  return (jint)((MObj**)it->data)[it->addr - 1]->len;
}
static jobject f_CallObjectMethod(JNIEnv* e, jobject o, jmethodID m, ...) {
  (void)e; MObj* it = o;
  if (m != &g_mid_getbuffer || it->addr < 1) return NULL;
  return ((MObj**)it->data)[it->addr - 1];
}
/* ---- org.apache.spark.sql.comet.CometScalarSubquery: static isNull / get<Type>(planId, id) over a table the test fills ---- */
typedef struct { int64_t plan_id, id; int is_null; int64_t i; double d; char* bytes; int64_t nbytes; } MSub;
static MSub g_subs[32];
static int g_nsubs;
static char g_static_calls[4096];      /* "isNull(7,3);getInt(7,3);" … in call order */
static int g_mids_static[12];
static const char* k_static_names[12] = {"isNull", "getBoolean", "getByte", "getShort", "getInt", "getLong", "getFloat", "getDouble", "getDecimal", "getString", "getBinary", 0};
static jmethodID f_GetStaticMethodID(JNIEnv* e, jclass c, const char* n, const char* sig) {
  (void)e; (void)sig;
  if (!c || ((MObj*)c)->kind != K_CLASS || strcmp(((MObj*)c)->cname, "org/apache/spark/sql/comet/CometScalarSubquery")) return NULL;
  for (int k = 0; k_static_names[k]; k++)
    if (!strcmp(n, k_static_names[k])) return &g_mids_static[k];
  return NULL;
}
static MSub* sub_of(jmethodID m, va_list ap) {
  const int64_t plan = va_arg(ap, int64_t), id = va_arg(ap, int64_t);
  const int k = (int)((int*)m - g_mids_static);
  char note[96];
  snprintf(note, sizeof note, "%s(%lld,%lld);", k_static_names[k], (long long)plan, (long long)id);
  strncat(g_static_calls, note, sizeof g_static_calls - strlen(g_static_calls) - 1);
  for (int s = 0; s < g_nsubs; s++)
    if (g_subs[s].plan_id == plan && g_subs[s].id == id) return &g_subs[s];
  return NULL;
}
#define STATIC_CALL(NAME, T, EXPR) static T NAME(JNIEnv* e, jclass c, jmethodID m, ...) { (void)e; (void)c; va_list ap; va_start(ap, m); MSub* s = sub_of(m, ap); va_end(ap); return (T)(EXPR); }
STATIC_CALL(f_CallStaticBooleanMethod, jboolean, (int*)m == &g_mids_static[0] ? (!s || s->is_null) : (s && s->i != 0))
STATIC_CALL(f_CallStaticByteMethod, jbyte, s ? s->i : 0)
STATIC_CALL(f_CallStaticShortMethod, jshort, s ? s->i : 0)
STATIC_CALL(f_CallStaticIntMethod, jint, s ? s->i : 0)
STATIC_CALL(f_CallStaticLongMethod, jlong, s ? s->i : 0)
STATIC_CALL(f_CallStaticFloatMethod, jfloat, s ? s->d : 0)
STATIC_CALL(f_CallStaticDoubleMethod, jdouble, s ? s->d : 0)
static jobject f_CallStaticObjectMethod(JNIEnv* e, jclass c, jmethodID m, ...) {
  (void)e; (void)c;
  va_list ap; va_start(ap, m); MSub* s = sub_of(m, ap); va_end(ap);
  if (!s) return NULL;
  MObj* o = calloc(1, sizeof *o);
  if ((int*)m == &g_mids_static[9]) { o->kind = K_STRING; o->data = strdup(s->bytes ? s->bytes : ""); return o; }
  o->kind = K_BYTES; o->len = s->nbytes; o->data = malloc((size_t)s->nbytes + 1); memcpy(o->data, s->bytes, (size_t)s->nbytes);
  return o;
}
static void* f_GetDirectBufferAddress(JNIEnv* e, jobject b) { (void)e; return b && ((MObj*)b)->kind == K_BYTES ? ((MObj*)b)->data : NULL; }
static void f_ExceptionClear(JNIEnv* e) { (void)e; g_exc_pending = 0; }
/* K_INTS: data = int32[len]; K_INFO (NativeColumnarToRowInfo): addr = memory address, data = MObj*[2] {offsets, lengths} */
static jobject f_NewIntArray(JNIEnv* e, jsize n) { (void)e; MObj* o = calloc(1, sizeof *o); o->kind = K_INTS; o->len = n; o->data = calloc((size_t)n + 1, 4); return o; }
static void f_SetIntArrayRegion(JNIEnv* e, jobject a, jsize s, jsize l, const jint* b) { (void)e; memcpy((jint*)((MObj*)a)->data + s, b, (size_t)l * 4); }
static jobject f_NewObject(JNIEnv* e, jclass c, jmethodID m, ...) {
  (void)e; (void)c;
  if (m != &g_mid_info_ctor) return NULL;
  va_list ap; va_start(ap, m);
  MObj* o = calloc(1, sizeof *o); o->kind = K_INFO;
  o->addr = va_arg(ap, jlong);
  MObj** parts = malloc(2 * sizeof(MObj*));
  parts[0] = va_arg(ap, MObj*); parts[1] = va_arg(ap, MObj*);
  va_end(ap);
  o->data = parts;
  return o;
}
static void f_CallVoidMethod(JNIEnv* e, jobject o, jmethodID m, ...) {
  (void)e;
  if (m == &g_mid_release) {
    va_list ap; va_start(ap, m); jlong n = va_arg(ap, jlong); va_end(ap);
    int64_t* st = ((MObj*)o)->data;
    st[1] -= n; st[4]++;
    return;
  }
  if (m != &g_mid_setall) return;
  va_list ap; va_start(ap, m); MObj* arr = va_arg(ap, MObj*); va_end(ap);
  MObj* node = o;
  free(node->data);
  node->data = malloc((size_t)arr->len ? (size_t)arr->len : 1);
  memcpy(node->data, arr->data, (size_t)arr->len);
  node->len = arr->len;
  node->addr++;   /* number of set_all_from_bytes up-calls on this node */
}
static const char* f_GetStringUTFChars(JNIEnv* e, jstring s, jboolean* c) { (void)e; if (c) *c = 0; return ((MObj*)s)->data; }
static void f_ReleaseStringUTFChars(JNIEnv* e, jstring s, const char* c) { (void)e; (void)s; (void)c; }
static jsize f_GetArrayLength(JNIEnv* e, jarray a) { (void)e; return (jsize)((MObj*)a)->len; }
static jobject f_GetObjectArrayElement(JNIEnv* e, jobjectArray a, jsize i) { (void)e; return ((MObj**)((MObj*)a)->data)[i]; }
static jbyteArray f_NewByteArray(JNIEnv* e, jsize n) { (void)e; MObj* o = calloc(1, sizeof *o); o->kind = K_BYTES; o->len = n; o->data = calloc(1, (size_t)n + 1); return o; }
static void f_GetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize s, jsize l, jbyte* b) { (void)e; memcpy(b, (char*)((MObj*)a)->data + s, (size_t)l); }
static void f_SetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize s, jsize l, const jbyte* b) { (void)e; memcpy((char*)((MObj*)a)->data + s, b, (size_t)l); }
static void f_GetIntArrayRegion(JNIEnv* e, jintArray a, jsize s, jsize l, jint* b) { (void)e; memcpy(b, (jint*)((MObj*)a)->data + s, (size_t)l * 4); }
static jobject f_NewLongArray(JNIEnv* e, jsize n) { (void)e; MObj* o = calloc(1, sizeof *o); o->kind = K_LONGS; o->len = n; o->data = calloc((size_t)n + 1, 8); return o; }
static void f_SetLongArrayRegion(JNIEnv* e, jobject a, jsize s, jsize l, const jlong* b) { (void)e; memcpy((jlong*)((MObj*)a)->data + s, b, (size_t)l * 8); }
static void f_GetLongArrayRegion(JNIEnv* e, jlongArray a, jsize s, jsize l, jlong* b) { (void)e; memcpy(b, (jlong*)((MObj*)a)->data + s, (size_t)l * 8); }
static jint f_GetJavaVM(JNIEnv* e, JavaVM** vm) { (void)e; *vm = (JavaVM*)&g_vm; return 0; }

JNIEnv* mock_env(void) {
  g_tab.fn[JNI_GetStaticMethodID] = (void*)f_GetStaticMethodID; g_tab.fn[JNI_CallStaticObjectMethod] = (void*)f_CallStaticObjectMethod;
  g_tab.fn[JNI_CallStaticBooleanMethod] = (void*)f_CallStaticBooleanMethod; g_tab.fn[JNI_CallStaticByteMethod] = (void*)f_CallStaticByteMethod;
  g_tab.fn[JNI_CallStaticShortMethod] = (void*)f_CallStaticShortMethod; g_tab.fn[JNI_CallStaticIntMethod] = (void*)f_CallStaticIntMethod;
  g_tab.fn[JNI_CallStaticLongMethod] = (void*)f_CallStaticLongMethod; g_tab.fn[JNI_CallStaticFloatMethod] = (void*)f_CallStaticFloatMethod;
  g_tab.fn[JNI_CallStaticDoubleMethod] = (void*)f_CallStaticDoubleMethod;
  g_tab.fn[JNI_FindClass] = (void*)f_FindClass; g_tab.fn[JNI_ThrowNew] = (void*)f_ThrowNew; g_tab.fn[JNI_ExceptionCheck] = (void*)f_ExceptionCheck;
  g_tab.fn[JNI_NewGlobalRef] = (void*)f_NewGlobalRef; g_tab.fn[JNI_DeleteGlobalRef] = (void*)f_DeleteGlobalRef;
  g_tab.fn[JNI_DeleteLocalRef] = (void*)f_DeleteLocalRef; g_tab.fn[JNI_GetObjectClass] = (void*)f_GetObjectClass;
  g_tab.fn[JNI_GetMethodID] = (void*)f_GetMethodID; g_tab.fn[JNI_CallLongMethod] = (void*)f_CallLongMethod;
  g_tab.fn[JNI_CallVoidMethod] = (void*)f_CallVoidMethod; g_tab.fn[JNI_GetStringUTFChars] = (void*)f_GetStringUTFChars;
  g_tab.fn[JNI_ReleaseStringUTFChars] = (void*)f_ReleaseStringUTFChars; g_tab.fn[JNI_GetArrayLength] = (void*)f_GetArrayLength;
  g_tab.fn[JNI_GetObjectArrayElement] = (void*)f_GetObjectArrayElement; g_tab.fn[JNI_NewByteArray] = (void*)f_NewByteArray;
  g_tab.fn[JNI_GetByteArrayRegion] = (void*)f_GetByteArrayRegion; g_tab.fn[JNI_SetByteArrayRegion] = (void*)f_SetByteArrayRegion;
  g_tab.fn[JNI_GetLongArrayRegion] = (void*)f_GetLongArrayRegion; g_tab.fn[JNI_GetJavaVM] = (void*)f_GetJavaVM;
  g_tab.fn[JNI_CallIntMethod] = (void*)f_CallIntMethod; g_tab.fn[JNI_CallObjectMethod] = (void*)f_CallObjectMethod;
  g_tab.fn[JNI_GetDirectBufferAddress] = (void*)f_GetDirectBufferAddress; g_tab.fn[JNI_ExceptionClear] = (void*)f_ExceptionClear;
  g_tab.fn[JNI_GetIntArrayRegion] = (void*)f_GetIntArrayRegion; g_tab.fn[JNI_NewLongArray] = (void*)f_NewLongArray; g_tab.fn[JNI_SetLongArrayRegion] = (void*)f_SetLongArrayRegion;
  g_tab.fn[JNI_NewIntArray] = (void*)f_NewIntArray; g_tab.fn[JNI_SetIntArrayRegion] = (void*)f_SetIntArrayRegion; g_tab.fn[JNI_NewObject] = (void*)f_NewObject;
  return (JNIEnv*)&g_env;
}
void* mock_bytes(const void* p, int64_t n) { MObj* o = calloc(1, sizeof *o); o->kind = K_BYTES; o->len = n; o->data = malloc((size_t)n + 1); memcpy(o->data, p, (size_t)n); return o; }
void* mock_longs(const int64_t* p, int64_t n) { MObj* o = calloc(1, sizeof *o); o->kind = K_LONGS; o->len = n; o->data = malloc((size_t)n * 8 + 8); memcpy(o->data, p, (size_t)n * 8); return o; }
void* mock_ints(const int32_t* p, int64_t n) { MObj* o = calloc(1, sizeof *o); o->kind = K_INTS; o->len = n; o->data = malloc((size_t)n * 4 + 8); memcpy(o->data, p, (size_t)n * 4); return o; }
const void* mock_array_data(void* o) { return ((MObj*)o)->data; }
int64_t mock_array_len(void* o) { return ((MObj*)o)->len; }
void* mock_objs(void** p, int64_t n) { MObj* o = calloc(1, sizeof *o); o->kind = K_OBJS; o->len = n; o->data = malloc((size_t)n * 8 + 8); memcpy(o->data, p, (size_t)n * 8); return o; }
void* mock_stream(int64_t addr) { MObj* o = calloc(1, sizeof *o); o->kind = K_STREAM; o->addr = addr; return o; }
void* mock_block_iterator(void** blocks, int64_t n) {
  MObj* o = calloc(1, sizeof *o); o->kind = K_BLOCKITER; o->len = n; o->data = malloc((size_t)n * 8 + 8); memcpy(o->data, blocks, (size_t)n * 8); return o;
}
int64_t mock_info_address(void* o) { return ((MObj*)o)->addr; }
int64_t mock_info_rows(void* o) { return ((MObj**)((MObj*)o)->data)[0]->len; }
const void* mock_info_offsets(void* o) { return ((MObj**)((MObj*)o)->data)[0]->data; }
const void* mock_info_lengths(void* o) { return ((MObj**)((MObj*)o)->data)[1]->data; }
void* mock_memory_manager(int64_t limit) { MObj* o = calloc(1, sizeof *o); o->kind = K_MEMMGR; int64_t* st = calloc(6, 8); st[0] = limit; o->data = st; return o; }
const int64_t* mock_memory_stats(void* o) { return ((MObj*)o)->data; }
void* mock_plain_object(void) { MObj* o = calloc(1, sizeof *o); o->kind = K_OBJS; return o; }
void* mock_metrics_node(void) { MObj* o = calloc(1, sizeof *o); o->kind = K_METRICS; return o; }
void* mock_string(const char* s) { MObj* o = calloc(1, sizeof *o); o->kind = K_STRING; o->data = strdup(s); return o; }
int64_t mock_metrics_len(void* n) { return ((MObj*)n)->len; }
long long mock_metrics_pushes(void* node) { return ((MObj*)node)->addr; }
const void* mock_metrics_bytes(void* n) { return ((MObj*)n)->data; }
/* CometScalarSubquery.setSubquery(planId, …): is_null / integer / double / bytes (decimal: BigInteger.toByteArray, string: modified UTF-8) */
void mock_set_subquery(int64_t plan_id, int64_t id, int is_null, int64_t i, double d, const void* bytes, int64_t nbytes) {
  MSub* s = &g_subs[g_nsubs++ % 32];
  s->plan_id = plan_id; s->id = id; s->is_null = is_null; s->i = i; s->d = d; s->nbytes = nbytes;
  s->bytes = malloc((size_t)nbytes + 1);
  if (nbytes) memcpy(s->bytes, bytes, (size_t)nbytes);
  s->bytes[nbytes] = 0;
}
const char* mock_static_calls(void) { return g_static_calls; }
void mock_static_calls_clear(void) { g_static_calls[0] = 0; g_nsubs = 0; }
int mock_exception_pending(void) { return g_exc_pending; }
const char* mock_exception_class(void) { return g_exc_class; }
const char* mock_exception_msg(void) { return g_exc_msg; }
void mock_exception_clear(void) { g_exc_pending = 0; g_exc_class[0] = 0; g_exc_msg[0] = 0; }
int mock_live_global_refs(void) { return g_live_global_refs; }
