// Library identification and process-level switches of libssbev_hip.so.
#include "common.h"

#include <cstdlib>
#include <mutex>
#include <string>
#include <unordered_map>

namespace {
struct EnvTable {
  std::mutex mu;
  std::unordered_map<std::string, std::pair<bool, std::string>> seen;     // name -> (set?, value); nodes never move
};
EnvTable& env_table() {
  static EnvTable* t = new EnvTable();       // leaked on purpose: kernels may be launched from static destructors of the host
  return *t;
}
}  // namespace

const char* ssbev_env(const char* name) {
  EnvTable& t = env_table();
  std::lock_guard<std::mutex> lock(t.mu);
  auto it = t.seen.find(name);
  if (it == t.seen.end()) {
    const char* v = std::getenv(name);
    it = t.seen.emplace(name, std::make_pair(v != nullptr, std::string(v ? v : ""))).first;
  }
  return it->second.first ? it->second.second.c_str() : nullptr;
}

extern "C" {
int ssbev_version(void) { return 100; /* 0.1.0 */ }
const char* ssbev_build_arch(void) { return "gfx950"; }

// Forget every switch read so far: the next use of a name reads the environment again.  (Switches that a launch helper folded
// into a function-local constant at its first call -- kernel A/B hooks -- stay as they were read.)  Not safe against launches in
// flight on other threads: strings handed out earlier are freed.
void ssbev_env_refresh(void) {
  EnvTable& t = env_table();
  std::lock_guard<std::mutex> lock(t.mu);
  t.seen.clear();
}
}
