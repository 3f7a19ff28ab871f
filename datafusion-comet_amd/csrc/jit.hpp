// hiprtc front-end: HIP source (generated functor + hand-written templates) → gfx950 code object →
// loaded module.  Code objects are cached in memory and on disk keyed by the source hash, so each
// distinct plan shape is compiled once per machine, not once per Spark task.
#pragma once
#include <hip/hip_runtime_api.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace comet {

struct CodeObject {
  std::vector<char> bytes;
  std::string log;
};

// compile (or fetch from cache); throws CometError with the compiler log on failure. No GPU needed.
std::shared_ptr<CodeObject> jit_compile(const std::string& source);
// "hiprtc <version>; comgr <version> <path of the library that compiles>" — part of the cache key, reported by comet_jit_toolchain
std::string jit_toolchain();

struct LoadedModule {
  hipModule_t mod = nullptr;
  std::map<std::string, hipFunction_t> fns;
  std::mutex mu;
  std::shared_ptr<CodeObject> keepalive;
  hipFunction_t fn(const std::string& name);
  ~LoadedModule();
};

// load on the CURRENT device
std::shared_ptr<LoadedModule> jit_load(const std::shared_ptr<CodeObject>& co);

void hip_check(hipError_t e, const char* what);
#define HIP_CHECK(x) ::comet::hip_check((x), #x)

}  // namespace comet
