// RLIKE pattern compiler (regex.cpp): the subset of the Rust `regex` syntax that can be matched exactly over UTF-8 bytes → a search DFA.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace comet {

struct RegexDfa {
  int nstates = 0;                 // ≤ 4096; state 0 = before the first byte
  int nclasses = 0;                // bytes that no state tells apart share a class
  std::vector<uint8_t> classes;    // [256] class of a byte
  std::vector<uint8_t> trans;      // [nstates][nclasses] next state, 16 bits little-endian each
  std::vector<uint8_t> flags;      // bit 0: a match has been found (absorbing); bit 1: a match if the text ends here
  int next(int state, uint8_t byte) const {
    const size_t at = ((size_t)state * (size_t)nclasses + classes[byte]) * 2;
    return trans[at] | (trans[at + 1] << 8);
  }
};
// throws CometError naming the construct for anything outside the subset
RegexDfa compile_rlike(const std::string& pattern);
// host-side walk of the same tables the device walks (tests)
bool regex_dfa_match(const RegexDfa& d, const uint8_t* s, size_t n);

// regexp_extract (string_funcs/regexp_extract.rs): the pattern as a program of device/regex_vm.hpp's matcher that reports group `group`;
// `fn` names the function in refusals.  Throws the reference's message for a group index out of range (regexp_extract_common.rs:85-92).
struct RegexProg {
  std::vector<uint32_t> words;
  int ngroups = 0;
};
RegexProg compile_regex_captures(const std::string& pattern, int group, const char* fn);
// the device's matcher run on the host (tests): → matched?; [*start, *start + *len) = the group's bytes (empty when unset / no match)
bool regex_prog_extract(const RegexProg& prog, const uint8_t* s, size_t n, int32_t* start, int32_t* len);

// split(str, pattern, limit) (string_funcs/split.rs) with a group-0 program: the pieces as (start, length) inside `s`, by the device's own two passes
std::vector<std::pair<int32_t, int32_t>> regex_prog_split(const RegexProg& prog, const uint8_t* s, size_t n, int32_t limit);

// regexp_extract_all: group spans of every match, by the device's two passes (`whole` = the group-0 program, `group` = the wanted group's or `whole` itself)
std::vector<std::pair<int32_t, int32_t>> regex_prog_find_all(const RegexProg& whole, const RegexProg& group, const uint8_t* s, size_t n);

}  // namespace comet
