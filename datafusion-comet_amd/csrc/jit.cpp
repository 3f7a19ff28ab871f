#include "jit.hpp"

#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <link.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <condition_variable>
#include <mutex>
#include <set>
#include <thread>

#include "plan.hpp"

namespace comet {

// device/comet_device.hpp and kparams.h embedded at build time (see Makefile: embedded_headers.inc)
extern const char* const kEmbeddedDeviceHeader;
extern const char* const kEmbeddedKParamsHeader;
extern const char* const kEmbeddedRyuHeader;
extern const char* const kEmbeddedStrtodHeader;
extern const char* const kEmbeddedStrtsHeader;
extern const char* const kEmbeddedRegexVmHeader;

void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) {
    throw CometError(std::string("HIP error ") + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ") in " + what);
  }
}

namespace {

uint64_t fnv1a(const std::string& s, uint64_t h = 0xcbf29ce484222325ull) {
  for (unsigned char c : s) {
    h ^= c;
    h *= 0x100000001b3ull;
  }
  return h;
}

std::string cache_dir() {
  if (const char* e = getenv("COMET_JIT_CACHE_DIR")) return e;
  // default: next to libcomet.so so that a cache warmed at build time travels with the library
  Dl_info info;
  if (dladdr((void*)&hip_check, &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    size_t k = p.rfind('/');
    if (k != std::string::npos) return p.substr(0, k) + "/jit_cache";
  }
  return "/tmp/comet_jit_cache";
}

std::mutex g_mu;
std::map<std::string, std::shared_ptr<CodeObject>> g_mem_cache;
// single flight: tasks of one stage start together and ask for the same plan shape at the same moment — one of them compiles,
// the others wait for its result instead of compiling (and writing the cache file) as well
std::condition_variable g_cv;
std::set<std::string> g_in_flight;

// The compiler that produced a cached code object is part of its identity: a ROCm upgrade must not load stale objects.  hiprtc is a thin layer; the
// compiler itself (clang + LLVM) lives in the code object manager, libamd_comgr — and a process can hold a hiprtc of one ROCm release bound to the comgr
// of another (a Python process that imported a torch wheel with its own bundled ROCm: its comgr, or — preloaded into the global scope — the installed
// one, answers every hiprtc).  amd_comgr_get_version, resolved the way hiprtc's own calls are, names the one that will compile; the first loaded
// libamd_comgr's path is added for the reader.
struct ComgrSeen { std::string path; };
int comgr_seen_cb(struct dl_phdr_info* info, size_t, void* data) {
  const char* n = info->dlpi_name;
  if (n && strstr(n, "libamd_comgr")) {
    auto* s = (ComgrSeen*)data;
    if (s->path.empty()) s->path = n;
  }
  return 0;
}
std::string compiler_identity() {
  std::string s;
  typedef void (*GetVersion)(size_t*, size_t*);
  if (auto gv = (GetVersion)dlsym(RTLD_DEFAULT, "amd_comgr_get_version")) {
    size_t a = 0, b = 0;
    gv(&a, &b);
    s = "comgr " + std::to_string(a) + "." + std::to_string(b);
    Dl_info info;
    if (dladdr((void*)gv, &info) && info.dli_fname) s += std::string(" ") + info.dli_fname;
  } else if (dlopen("libamd_comgr.so.3", RTLD_LAZY | RTLD_GLOBAL) && dlsym(RTLD_DEFAULT, "amd_comgr_get_version")) {
    return compiler_identity();      // (hiprtc loads it on demand by this very name: now it is the one it will find)
  } else {
    ComgrSeen seen;
    dl_iterate_phdr(comgr_seen_cb, &seen);
    s = "comgr " + (seen.path.empty() ? std::string("(loaded on demand by hiprtc)") : seen.path);
  }
  return s;
}
std::string toolchain_tag() {
  int major = 0, minor = 0;
  hiprtcVersion(&major, &minor);
  return "hiprtc " + std::to_string(major) + "." + std::to_string(minor) + "; " + compiler_identity();
}
// a code object is an ELF image; anything else in the cache (a truncated or foreign file) is ignored and recompiled
bool looks_like_code_object(const std::vector<char>& b) { return b.size() > 64 && b[0] == 0x7f && b[1] == 'E' && b[2] == 'L' && b[3] == 'F'; }

}  // namespace

std::string jit_toolchain() { return toolchain_tag(); }

std::shared_ptr<CodeObject> jit_compile(const std::string& source_in) {
  // COMET_LD_NT=0/1 (an experiment switch): column loads of EVERY generated kernel as ordinary / non-temporal loads, whatever the generator chose
  static const char* nt = getenv("COMET_LD_NT");
  std::string source = nt ? std::string("#define COMET_LD_NT ") + (atoi(nt) ? "1" : "0") + "\n" + source_in : source_in;
  // COMET_JIT_DEFINES="NAME=value;NAME2=value2" (an experiment switch): #define lines in front of every generated source — the tuning constants of
  // comet_device.hpp that are written as #ifndef defaults (join tile shape, …) can be A/B-ed on one box without rebuilding the library; part of the cache key
  static const char* defs = getenv("COMET_JIT_DEFINES");
  if (defs && *defs) {
    std::string head, d = defs;
    size_t pos = 0;
    while (pos <= d.size()) {
      size_t e = d.find(';', pos);
      std::string item = d.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
      if (!item.empty()) {
        size_t eq = item.find('=');
        head += "#define " + (eq == std::string::npos ? item + " 1" : item.substr(0, eq) + " " + item.substr(eq + 1)) + "\n";
      }
      if (e == std::string::npos) break;
      pos = e + 1;
    }
    source = head + source;
  }
  // the key covers the generated source AND the hand-written headers it instantiates
  static const uint64_t h2 = fnv1a(toolchain_tag(), fnv1a(kEmbeddedDeviceHeader, fnv1a(kEmbeddedKParamsHeader, fnv1a(kEmbeddedRyuHeader, fnv1a(kEmbeddedStrtodHeader, fnv1a(kEmbeddedStrtsHeader, fnv1a(kEmbeddedRegexVmHeader)))))));
  uint64_t h1 = fnv1a(source);
  char keybuf[64];
  snprintf(keybuf, sizeof keybuf, "%016llx_%016llx", (unsigned long long)h1, (unsigned long long)h2);
  std::string key = keybuf;
  {
    std::unique_lock<std::mutex> lk(g_mu);
    for (;;) {
      auto it = g_mem_cache.find(key);
      if (it != g_mem_cache.end()) return it->second;
      if (!g_in_flight.count(key)) break;
      g_cv.wait(lk);                       // another thread is compiling this very key
    }
    g_in_flight.insert(key);
  }
  struct Flight {                          // whatever happens below, waiters are released
    std::string key;
    ~Flight() {
      { std::lock_guard<std::mutex> lk(g_mu); g_in_flight.erase(key); }
      g_cv.notify_all();
    }
  } flight{key};
  if (const char* dd = getenv("COMET_JIT_DUMP_DIR")) {
    mkdir(dd, 0755);
    std::ofstream f(std::string(dd) + "/" + key + ".hip");
    f << source;
  }
  const bool disk = !(getenv("COMET_JIT_NO_DISK_CACHE"));
  std::string dir = cache_dir(), path = dir + "/" + key + ".hsaco";
  if (disk) {
    std::ifstream f(path, std::ios::binary);
    if (f) {
      auto co = std::make_shared<CodeObject>();
      co->bytes.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
      if (looks_like_code_object(co->bytes)) {
        std::lock_guard<std::mutex> lk(g_mu);
        g_mem_cache[key] = co;
        return co;
      }
    }
  }
  hiprtcProgram prog;
  const char* headers[] = {kEmbeddedDeviceHeader, kEmbeddedKParamsHeader, kEmbeddedRyuHeader, kEmbeddedStrtodHeader, kEmbeddedStrtsHeader, kEmbeddedRegexVmHeader};
  const char* names[] = {"comet_device.hpp", "kparams.h", "comet_ryu.hpp", "comet_strtod.hpp", "comet_strts.hpp", "comet_regex_vm.hpp"};
  if (hiprtcCreateProgram(&prog, source.c_str(), "comet_pipeline.hip", 6, headers, names) != HIPRTC_SUCCESS)
    throw CometError("hiprtcCreateProgram failed");
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value"};
  hiprtcResult rc = hiprtcCompileProgram(prog, 5, opts);
  auto co = std::make_shared<CodeObject>();
  size_t logsz = 0;
  hiprtcGetProgramLogSize(prog, &logsz);
  if (logsz > 1) {
    co->log.resize(logsz);
    hiprtcGetProgramLog(prog, &co->log[0]);
  }
  if (rc != HIPRTC_SUCCESS) {
    hiprtcDestroyProgram(&prog);
    if (getenv("COMET_JIT_DUMP")) fprintf(stderr, "---- failing source ----\n%s\n", source.c_str());
    throw CometError(std::string("hiprtc compilation failed: ") + hiprtcGetErrorString(rc) + "\n" + co->log);
  }
  size_t sz = 0;
  hiprtcGetCodeSize(prog, &sz);
  co->bytes.resize(sz);
  hiprtcGetCode(prog, co->bytes.data());
  hiprtcDestroyProgram(&prog);
  if (disk) {
    mkdir(dir.c_str(), 0755);
    // unique per process AND thread, renamed into place only when complete: a reader never sees a partial file
    std::string tmp = path + ".tmp" + std::to_string((long)getpid()) + "_" + std::to_string((unsigned long long)std::hash<std::thread::id>()(std::this_thread::get_id()));
    std::ofstream f(tmp, std::ios::binary);
    if (f) {
      f.write(co->bytes.data(), (std::streamsize)co->bytes.size());
      f.close();
      rename(tmp.c_str(), path.c_str());
    }
  }
  std::lock_guard<std::mutex> lk(g_mu);
  g_mem_cache[key] = co;
  return co;
}

hipFunction_t LoadedModule::fn(const std::string& name) {
  std::lock_guard<std::mutex> lk(mu);
  auto it = fns.find(name);
  if (it != fns.end()) return it->second;
  hipFunction_t f;
  hip_check(hipModuleGetFunction(&f, mod, name.c_str()), ("hipModuleGetFunction(" + name + ")").c_str());
  fns[name] = f;
  return f;
}

LoadedModule::~LoadedModule() {
  if (mod) (void)hipModuleUnload(mod);
}

std::shared_ptr<LoadedModule> jit_load(const std::shared_ptr<CodeObject>& co) {
  // one loaded module per (device, code object), shared by every plan handle of the process: Spark runs
  // the same stage plan once per task, only the first task on a device pays hipModuleLoadData.
  static std::mutex mu;
  static std::map<std::pair<int, const CodeObject*>, std::shared_ptr<LoadedModule>> loaded;
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  auto key = std::make_pair(dev, co.get());
  auto it = loaded.find(key);
  if (it != loaded.end()) return it->second;
  auto m = std::make_shared<LoadedModule>();
  HIP_CHECK(hipModuleLoadData(&m->mod, co->bytes.data()));
  m->keepalive = co;
  loaded[key] = m;
  return m;
}

}  // namespace comet
