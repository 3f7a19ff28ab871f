// RLIKE pattern compiler (regex.cpp): the subset of the Rust `regex` syntax that can be matched exactly over UTF-8 bytes → a search DFA.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace comet {

struct RegexDfa {
  int nstates = 0;                 // ≤ 200; state 0 = before the first byte
  std::vector<uint8_t> trans;      // [nstates][256] next state
  std::vector<uint8_t> flags;      // bit 0: a match has been found (absorbing); bit 1: a match if the text ends here
};
// throws CometError naming the construct for anything outside the subset
RegexDfa compile_rlike(const std::string& pattern);
// host-side walk of the same tables the device walks (tests)
bool regex_dfa_match(const RegexDfa& d, const uint8_t* s, size_t n);

}  // namespace comet
