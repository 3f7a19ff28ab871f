// RLIKE patterns → a byte-level DFA the generated kernels walk (device/comet_device.hpp utf8_rlike).
//
// The reference evaluates RLike with the Rust `regex` crate: Regex::new(pattern)?.is_match(value) — an unanchored search over the value's
// Unicode scalar values (native/spark-expr/src/predicate_funcs/rlike.rs:47-90).  This compiler accepts the part of that syntax whose meaning
// can be reproduced EXACTLY over UTF-8 bytes and refuses the rest by name, so an unsupported pattern fails at createPlan (the JVM side then
// keeps the expression on Spark) instead of matching differently:
//   literals (any UTF-8), `.` (any scalar value but \n), classes [a-z0-9_] / [^…] whose members are any scalar values (beyond ASCII: the UTF-8 byte-range sequences of utf8-ranges), escapes of punctuation and \t \n \r,
//   hexadecimal escapes of scalar values (\x41 \x{1F600} \u00e9 \U0001F600),
//   grouping ( ) and (?: ), alternation |, * + ? {m} {m,} {m,n} (lazy forms mean the same for a yes/no answer), ^ and $ (start / end of
//   the text, as in the crate without the m flag).
//   \A and \z (start / end of the text), leading (?s) (`.` matches \n too) and (?m) (^ / $ also at line boundaries), alone or combined
//   with each other and (?i).
//   A leading (?i) makes the whole pattern case insensitive the way the crate does it — Unicode SIMPLE case folding — which for ASCII
//   letters means: the other ASCII case, and for k / s also U+212A KELVIN SIGN / U+017F LATIN SMALL LETTER LONG S (the only non-ASCII
//   scalar values that fold to an ASCII letter); non-ASCII literals under (?i), and negated classes that would have to exclude those two,
//   are refused.
//   [[:alpha:]] and the other POSIX bracket classes (ASCII-only in the crate), \s / \S (the White_Space property: the same 25 scalar values in every Unicode version).
// Refused: \d \w \b and the other Perl / Unicode classes (Unicode-aware in the crate and growing with every Unicode release: neither an
// ASCII rendering nor a table of another version would match it), other flags and scoped flag groups, look-around and back-references (the crate refuses those too), non-ASCII class
// members under (?i), counted repetitions above 64.
//
// Construction: parse → Thompson NFA over byte sets (a `.` or a negated class becomes the UTF-8 sequence alternation) → subset
// construction of the SEARCH automaton (the start state is re-injected after every byte; ^ is passable only before the first byte).
// A state that contains MATCH is absorbing — the kernel returns true there; `$` is decided by a per-state "accepts at end of text" flag.
#include "regex.hpp"

#include <algorithm>
#include <array>
#include <functional>
#include <map>
#include <set>

#include "plan.hpp"
#include "regex_unicode_tables.hpp"
#include "device/regex_vm.hpp"

namespace comet {
namespace {

typedef std::array<uint64_t, 4> ByteSet;
void bs_add(ByteSet& s, int b) { s[(size_t)b >> 6] |= (uint64_t)1 << (b & 63); }
bool bs_has(const ByteSet& s, int b) { return (s[(size_t)b >> 6] >> (b & 63)) & 1; }
ByteSet bs_range(int lo, int hi) {
  ByteSet s{};
  for (int b = lo; b <= hi; b++) bs_add(s, b);
  return s;
}

struct Node;
typedef std::shared_ptr<Node> NodeP;
// a big class (\w: 796 ranges of scalar values) as a MINIMAL acyclic byte automaton instead of an alternation of ~1200 UTF-8 range sequences:
// node 0 is the entry; an edge leads to another node or (−1) to the class's end
struct TrieNode { std::vector<std::pair<ByteSet, int>> edges; };
struct Node {
  // OneChar: ONE scalar value — kids[0] is its byte-level automaton (what the search DFA is built from), `set` / `wide` its members as scalar
  // values (what the capture program tests); Group: a capturing group, `min` = its number
  enum Kind { Bytes, Cat, Alt, Repeat, Bol, Eol, Empty, Trie, WordB, NotWordB, OneChar, Group } kind = Empty;
  ByteSet set{};
  std::vector<NodeP> kids;
  int min = 0, max = -1;   // Repeat: max −1 = unbounded
  bool lazy = false;       // Repeat: x*? prefers to stop (the same language; another capture)
  std::vector<TrieNode> trie;
  std::vector<std::pair<int, int>> wide;   // OneChar: members beyond ASCII, sorted ranges
};
NodeP mk(Node::Kind k) {
  auto n = std::make_shared<Node>();
  n->kind = k;
  return n;
}
NodeP mk_bytes(const ByteSet& s) {
  auto n = mk(Node::Bytes);
  n->set = s;
  return n;
}
NodeP mk_char(NodeP bytes, const ByteSet& ascii, std::vector<std::pair<int, int>> wide) {
  auto n = mk(Node::OneChar);
  n->kids.push_back(std::move(bytes));
  n->set = ascii;
  n->wide = std::move(wide);
  return n;
}
NodeP mk_cat(std::vector<NodeP> kids) {
  if (kids.empty()) return mk(Node::Empty);
  if (kids.size() == 1) return kids[0];
  auto n = mk(Node::Cat);
  n->kids = std::move(kids);
  return n;
}
NodeP mk_alt(std::vector<NodeP> kids) {
  if (kids.size() == 1) return kids[0];
  auto n = mk(Node::Alt);
  n->kids = std::move(kids);
  return n;
}
std::string utf8_bytes(int cp) {
  std::string o;
  if (cp < 0x80) o += (char)cp;
  else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
  else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
  else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
  return o;
}
// the scalar values of `wide` (sorted ranges beyond ASCII, surrogates skipped) as the minimal automaton over their UTF-8 bytes
NodeP class_trie(const std::vector<std::pair<int, int>>& wide) {
  struct Raw { std::map<int, int> kid; };      // byte → raw node, −1 = end
  std::vector<Raw> raw(1);
  for (auto& r : wide)
    for (int cp = std::max(r.first, 0x80); cp <= r.second; cp++) {
      if (cp >= 0xD800 && cp <= 0xDFFF) { cp = 0xDFFF; continue; }
      const std::string b = utf8_bytes(cp);
      int n = 0;
      for (size_t k = 0; k + 1 < b.size(); k++) {
        auto it = raw[(size_t)n].kid.find((unsigned char)b[k]);
        if (it == raw[(size_t)n].kid.end()) {
          raw.emplace_back();
          const int fresh = (int)raw.size() - 1;
          raw[(size_t)n].kid[(unsigned char)b[k]] = fresh;
          n = fresh;
        } else n = it->second;
      }
      raw[(size_t)n].kid[(unsigned char)b.back()] = -1;
    }
  // equal subtrees become one node (children first: a raw node's children have larger indices)
  std::vector<int> canon(raw.size(), -1);
  std::map<std::vector<std::pair<int, int>>, int> seen;
  std::vector<std::vector<std::pair<int, int>>> sigs;
  for (size_t k = raw.size(); k-- > 0;) {
    std::vector<std::pair<int, int>> sig;
    for (auto& e : raw[k].kid) sig.emplace_back(e.first, e.second < 0 ? -1 : canon[(size_t)e.second]);
    auto it = seen.find(sig);
    if (it == seen.end()) {
      seen[sig] = (int)sigs.size();
      canon[k] = (int)sigs.size();
      sigs.push_back(sig);
    } else canon[k] = it->second;
  }
  // the root was numbered last: make it node 0
  auto n = mk(Node::Trie);
  const int root = canon[0], last = (int)sigs.size() - 1;
  auto renum = [&](int c) { return c < 0 ? -1 : c == root ? 0 : c == 0 ? root : c; };
  (void)last;
  n->trie.resize(sigs.size());
  for (size_t c = 0; c < sigs.size(); c++) {
    std::map<int, ByteSet> by_target;
    for (auto& e : sigs[c]) bs_add(by_target[renum(e.second)], e.first);
    TrieNode& t = n->trie[(size_t)renum((int)c)];
    for (auto& e : by_target) t.edges.emplace_back(e.second, e.first);
  }
  return n;
}
// every UTF-8 encoded scalar value of two or more bytes (well-formed sequences, surrogates excluded like the crate's utf8 ranges)
NodeP multibyte() {
  const ByteSet cont = bs_range(0x80, 0xBF);
  auto seq = [&](std::vector<ByteSet> parts) {
    std::vector<NodeP> k;
    for (auto& p : parts) k.push_back(mk_bytes(p));
    return mk_cat(k);
  };
  return mk_alt({seq({bs_range(0xC2, 0xDF), cont}),
                 seq({bs_range(0xE0, 0xE0), bs_range(0xA0, 0xBF), cont}), seq({bs_range(0xE1, 0xEC), cont, cont}), seq({bs_range(0xED, 0xED), bs_range(0x80, 0x9F), cont}),
                 seq({bs_range(0xEE, 0xEF), cont, cont}),
                 seq({bs_range(0xF0, 0xF0), bs_range(0x90, 0xBF), cont, cont}), seq({bs_range(0xF1, 0xF3), cont, cont, cont}), seq({bs_range(0xF4, 0xF4), bs_range(0x80, 0x8F), cont, cont})});
}
// one scalar value: the ASCII members of `ascii`, plus (if `and_multibyte`) everything beyond ASCII
NodeP one_char(const ByteSet& ascii, bool and_multibyte) {
  std::vector<NodeP> alts;
  bool any = false;
  for (int b = 0; b < 128; b++) any |= bs_has(ascii, b);
  if (any) alts.push_back(mk_bytes(ascii));
  if (and_multibyte) alts.push_back(multibyte());
  if (alts.empty()) {        // a class nothing can match: a byte set with no member
    return mk_bytes(ByteSet{});
  }
  return mk_alt(alts);
}

struct Parser {
  const std::string& p;
  size_t i = 0;
  int depth = 0;          // open groups (the descent is recursive)
  bool icase = false;     // a leading (?i)
  bool dotall = false;    // a leading (?s): `.` matches \n too
  bool multiline = false; // a leading (?m): ^ also matches after a \n, $ also before one
  bool has_wordb = false; // the pattern holds \b / \B
  int ngroups = 0;        // capturing groups seen so far
  bool bytes_too = true;  // build the byte-level automaton of every class (the search DFA's input; the capture program tests scalar values instead)
  explicit Parser(const std::string& s) : p(s) {
    // leading flags (?i) (?s) (?m), alone or combined: they hold for the whole pattern
    if (p.compare(0, 2, "(?") == 0) {
      size_t j = 2;
      bool fi = false, fs = false, fm = false;
      while (j < p.size() && (p[j] == 'i' || p[j] == 's' || p[j] == 'm')) { (p[j] == 'i' ? fi : p[j] == 's' ? fs : fm) = true; j++; }
      if (j > 2 && j < p.size() && p[j] == ')') { icase = fi; dotall = fs; multiline = fm; i = j + 1; }
    }
  }
  static bool is_letter(int b) { return (b >= 'a' && b <= 'z') || (b >= 'A' && b <= 'Z'); }
  // the non-ASCII scalar values whose simple case folding is the ASCII letter b (CaseFolding.txt: 212A → k, 017F → s), as UTF-8 sequences
  static NodeP fold_partner(int b) {
    auto seq = [](std::initializer_list<int> bytes) {
      std::vector<NodeP> k;
      for (int x : bytes) { ByteSet s{}; bs_add(s, x); k.push_back(mk_bytes(s)); }
      return mk_cat(k);
    };
    if (b == 'k' || b == 'K') return seq({0xE2, 0x84, 0xAA});
    if (b == 's' || b == 'S') return seq({0xC5, 0xBF});
    return nullptr;
  }
  [[noreturn]] void fail(const std::string& why) const { throw CometError("RLIKE pattern '" + p + "' is not supported by the MI355X native engine: " + why); }
  bool more() const { return i < p.size(); }
  NodeP parse_alt() {
    std::vector<NodeP> alts;
    alts.push_back(parse_cat());
    while (more() && p[i] == '|') {
      i++;
      alts.push_back(parse_cat());
    }
    return mk_alt(alts);
  }
  NodeP parse_cat() {
    std::vector<NodeP> items;
    while (more() && p[i] != '|' && p[i] != ')') items.push_back(parse_repeat());
    return mk_cat(items);
  }
  NodeP parse_repeat() {
    NodeP a = parse_atom();
    int stacked = 0;      // a{1}{1}{1}… nests one Repeat per quantifier without adding NFA states on the way down; build() and the node
                          // destructor recurse once per level, and regex-syntax refuses what nests deeper than its nest_limit as well
    while (more()) {
      int mn, mx;
      const char ch = p[i];
      if (ch == '*') { mn = 0; mx = -1; i++; }
      else if (ch == '+') { mn = 1; mx = -1; i++; }
      else if (ch == '?') { mn = 0; mx = 1; i++; }
      else if (ch == '{') {
        size_t j = i + 1;
        auto num = [&](int& v) {
          if (j >= p.size() || p[j] < '0' || p[j] > '9') return false;
          long long x = 0;
          while (j < p.size() && p[j] >= '0' && p[j] <= '9') { x = x * 10 + (p[j] - '0'); if (x > 1000) fail("a repetition count above 64"); j++; }
          v = (int)x;
          return true;
        };
        if (!num(mn)) fail("'{' that does not start a counted repetition");
        mx = mn;
        if (j < p.size() && p[j] == ',') {
          j++;
          if (j < p.size() && p[j] == '}') mx = -1;
          else if (!num(mx)) fail("a malformed counted repetition");
        }
        if (j >= p.size() || p[j] != '}') fail("a malformed counted repetition");
        if (mn > 64 || mx > 64 || (mx >= 0 && mx < mn)) fail("a repetition count above 64 (or max below min)");
        i = j + 1;
      } else break;
      bool lazy = false;
      if (more() && (p[i] == '?')) { i++; lazy = true; }   // lazy: the same language, another preference (captures only)
      else if (more() && p[i] == '+') fail("possessive quantifiers");
      if (a->kind == Node::Bol || a->kind == Node::Eol) fail("a quantifier on an anchor");
      if (++stacked + depth > 100) fail("quantifiers stacked more than 100 deep");
      auto r = mk(Node::Repeat);
      r->kids.push_back(a);
      r->min = mn;
      r->max = mx;
      r->lazy = lazy;
      a = r;
    }
    return a;
  }
  // \xHH \x{H…} \uHHHH \u{H…} \UHHHHHHHH \U{H…} at p[at] == '\\': the scalar value and the escape's length, or -1 (not such an escape).
  // Malformed digits, surrogates and values beyond U+10FFFF are refused like the crate refuses them.
  int hex_escape(size_t at, size_t& len) const {
    if (at + 1 >= p.size()) return -1;
    const char k = p[at + 1];
    if (k != 'x' && k != 'u' && k != 'U') return -1;
    auto hexval = [](char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; };
    size_t j = at + 2;
    long cp = 0;
    if (j < p.size() && p[j] == '{') {
      j++;
      size_t digits = 0;
      while (j < p.size() && p[j] != '}') {
        const int h = hexval(p[j]);
        if (h < 0 || ++digits > 8) fail("a malformed hexadecimal escape");
        cp = cp * 16 + h;
        j++;
      }
      if (j >= p.size() || digits == 0) fail("a malformed hexadecimal escape");
      j++;
    } else {
      const size_t want = k == 'x' ? 2 : k == 'u' ? 4 : 8;
      for (size_t d = 0; d < want; d++, j++) {
        const int h = j < p.size() ? hexval(p[j]) : -1;
        if (h < 0) fail("a malformed hexadecimal escape");
        cp = cp * 16 + h;
      }
    }
    if (cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) fail("a hexadecimal escape that is not a Unicode scalar value");
    len = j - at;
    return (int)cp;
  }
  static std::string utf8_of(int cp) {
    std::string o;
    if (cp < 0x80) o += (char)cp;
    else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    return o;
  }
  // one literal scalar value as an item: its UTF-8 bytes in sequence (under (?i): an ASCII letter in both cases plus its fold partner)
  NodeP literal_item(int cp) {
    if (icase && cp >= 0x80) fail("case-insensitive matching of a non-ASCII literal");
    if (icase && is_letter(cp)) return folded_letter(cp);
    std::vector<NodeP> seq;
    for (unsigned char b : utf8_of(cp)) {
      ByteSet s{};
      bs_add(s, b);
      seq.push_back(mk_bytes(s));
    }
    ByteSet as{};
    if (cp < 128) bs_add(as, cp);
    return mk_char(mk_cat(seq), as, cp < 128 ? std::vector<std::pair<int, int>>{} : std::vector<std::pair<int, int>>{{cp, cp}});
  }
  // an ASCII letter under (?i): both cases, and the scalar value beyond ASCII that folds to it (k, s)
  static NodeP folded_letter(int b) {
    ByteSet s{};
    bs_add(s, b | 0x20);
    bs_add(s, b & ~0x20);
    NodeP extra = fold_partner(b);
    std::vector<std::pair<int, int>> wide;
    if ((b | 0x20) == 'k') wide.emplace_back(0x212A, 0x212A);
    if ((b | 0x20) == 's') wide.emplace_back(0x17F, 0x17F);
    return mk_char(extra ? mk_alt({mk_bytes(s), extra}) : mk_bytes(s), s, wide);
  }
  // the byte of an escaped punctuation / control character, or -1
  int simple_escape(char c) const {
    switch (c) {
      case 't': return '\t';
      case 'n': return '\n';
      case 'r': return '\r';
      case 'f': return '\f';
      case 'v': return '\v';
      default: break;
    }
    if ((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z')) return -1;   // \d \w \b \1 \p{..} \x.. …: not reproduced
    if ((unsigned char)c >= 0x80) return -1;
    if (c == '<' || c == '>') return -1;      // \< / \> are word-boundary assertions in regex-syntax 0.8 (and an error inside a class), not literals
    return (unsigned char)c;
  }
  NodeP parse_atom() {
    const unsigned char ch = (unsigned char)p[i];
    if (ch == '(') {
      if (++depth > 100) fail("groups nested more than 100 deep");
      struct Leave { int& d; ~Leave() { d--; } } leave{depth};
      i++;
      int number = 0;
      if (more() && p[i] == '?') {
        if (i + 1 < p.size() && p[i + 1] == ':') i += 2;
        else fail("group flags, named groups and look-around ((?…) other than (?:…) and a leading (?i))");
      } else number = ++ngroups;
      NodeP inner = parse_alt();
      if (!more() || p[i] != ')') fail("an unclosed group");
      i++;
      if (!number) return inner;
      auto g = mk(Node::Group);
      g->kids.push_back(inner);
      g->min = number;
      return g;
    }
    if (ch == '[') return parse_class();
    if (ch == '.') {
      i++;
      ByteSet s = bs_range(0, 127);
      if (!dotall) s[0] &= ~((uint64_t)1 << '\n');
      return mk_char(one_char(s, true), s, {{128, 0x10FFFF}});
    }
    if (ch == '^') { i++; return mk(Node::Bol); }
    if (ch == '$') { i++; return mk(Node::Eol); }
    if (ch == '*' || ch == '+' || ch == '?') fail("a quantifier with nothing to repeat");
    if (ch == '{') fail("'{' that does not follow an item");
    if (ch == '\\') {
      if (i + 1 >= p.size()) fail("a trailing backslash");
      size_t hlen = 0;
      const int cp = hex_escape(i, hlen);
      if (cp >= 0) {
        i += hlen;
        return literal_item(cp);
      }
      if ((p[i + 1] == 'A' || p[i + 1] == 'z') && multiline) fail("\\A / \\z under (?m)");
      if (p[i + 1] == 'A') { i += 2; return mk(Node::Bol); }      // \A / \z: start / end of the text — what ^ / $ mean without (?m)
      if (p[i + 1] == 'z') { i += 2; return mk(Node::Eol); }
      if (p[i + 1] == 's' || p[i + 1] == 'S') {
        ByteSet ws{};
        std::vector<std::pair<int, int>> wide;
        add_white_space(ws, wide);
        const bool negate = p[i + 1] == 'S';
        i += 2;
        return finish_class(ws, wide, negate, true);
      }
      if (p[i + 1] == 'b' || p[i + 1] == 'B') {                       // Unicode word boundary / not a word boundary
        if (multiline) fail("\\b under (?m)");
        has_wordb = true;
        const bool nb = p[i + 1] == 'B';
        i += 2;
        return mk(nb ? Node::NotWordB : Node::WordB);
      }
      if (p[i + 1] == 'd' || p[i + 1] == 'D' || p[i + 1] == 'w' || p[i + 1] == 'W') {
        ByteSet as{};
        std::vector<std::pair<int, int>> wide;
        add_perl_class(p[i + 1] | 0x20, as, wide);
        const bool negate = p[i + 1] == 'D' || p[i + 1] == 'W';
        i += 2;
        return finish_class(as, wide, negate, true);
      }
      const int b = simple_escape(p[i + 1]);
      if (b < 0) fail(std::string("the escape \\") + p[i + 1] + " (the \\p{…} classes are not reproduced; back-references do not exist in the reference)");
      i += 2;
      ByteSet s{};
      bs_add(s, b);
      return mk_char(mk_bytes(s), s, {});
    }
    if (icase && ch >= 0x80) fail("case-insensitive matching of a non-ASCII literal");
    if (icase && is_letter(ch)) {
      i++;
      return folded_letter(ch);
    }
    // a literal character: its UTF-8 bytes in sequence form ONE item (a quantifier after "é" repeats both bytes)
    const size_t first = i;
    std::vector<NodeP> seq;
    do {
      ByteSet s{};
      bs_add(s, (unsigned char)p[i]);
      seq.push_back(mk_bytes(s));
      i++;
    } while (ch >= 0x80 && more() && ((unsigned char)p[i] & 0xC0) == 0x80);
    size_t at = first;
    const int cp = ch < 0x80 ? ch : decode_utf8_at(at);
    ByteSet as{};
    if (cp < 128) bs_add(as, cp);
    return mk_char(mk_cat(seq), as, cp < 128 ? std::vector<std::pair<int, int>>{} : std::vector<std::pair<int, int>>{{cp, cp}});
  }
  // one scalar value of the pattern at p[i] (UTF-8 decoded; the pattern is valid UTF-8 — it arrived as a Java String)
  int decode_utf8_at(size_t& at) const {
    const unsigned char c = (unsigned char)p[at];
    if (c < 0x80) { at++; return c; }
    const int n = c >= 0xF0 ? 4 : c >= 0xE0 ? 3 : 2;
    if (at + (size_t)n > p.size()) fail("a truncated UTF-8 sequence in the pattern");
    int cp = c & (0x7F >> n);
    for (int k = 1; k < n; k++) cp = (cp << 6) | ((unsigned char)p[at + (size_t)k] & 0x3F);
    at += (size_t)n;
    return cp;
  }
  // a class member at p[i]: a literal scalar value or an escape of one
  int class_member() {
    if (!more()) fail("an unclosed character class");
    if (p[i] != '\\') return decode_utf8_at(i);
    if (i + 1 >= p.size()) fail("a trailing backslash");
    size_t hlen = 0;
    const int cp = hex_escape(i, hlen);
    if (cp >= 0) { i += hlen; return cp; }
    const int b = simple_escape(p[i + 1]);
    if (b < 0) fail(std::string("the escape \\") + p[i + 1] + " inside a class");
    i += 2;
    return b;
  }
  NodeP parse_class() {
    i++;   // [
    bool neg = false;
    if (more() && p[i] == '^') { neg = true; i++; }
    ByteSet s{};                                  // ASCII members
    std::vector<std::pair<int, int>> wide;        // members beyond ASCII, as ranges of scalar values
    bool first = true;
    while (true) {
      if (!more()) fail("an unclosed character class");
      const unsigned char c = (unsigned char)p[i];
      if (c == ']' && !first) { i++; break; }
      first = false;
      if (c == '[' && i + 1 < p.size() && p[i + 1] == ':') {
        // [:name:] — the POSIX classes are ASCII-only in the crate, so they mean the same bytes here
        const size_t close = p.find(":]", i + 2);
        if (close == std::string::npos) fail("an unclosed [:posix:] class");
        const std::string name = p.substr(i + 2, close - (i + 2));
        auto add = [&](int lo, int hi) { for (int b = lo; b <= hi; b++) bs_add(s, b); };
        if (name == "alnum") { add('0', '9'); add('A', 'Z'); add('a', 'z'); }
        else if (name == "alpha") { add('A', 'Z'); add('a', 'z'); }
        else if (name == "ascii") add(0, 127);
        else if (name == "blank") { add(' ', ' '); add('\t', '\t'); }
        else if (name == "cntrl") { add(0, 31); add(127, 127); }
        else if (name == "digit") add('0', '9');
        else if (name == "graph") add('!', '~');
        else if (name == "lower") add('a', 'z');
        else if (name == "print") add(' ', '~');
        else if (name == "punct") { add('!', '/'); add(':', '@'); add('[', '`'); add('{', '~'); }
        else if (name == "space") { add('\t', '\r'); add(' ', ' '); }
        else if (name == "upper") add('A', 'Z');
        else if (name == "word") { add('0', '9'); add('A', 'Z'); add('a', 'z'); add('_', '_'); }
        else if (name == "xdigit") { add('0', '9'); add('A', 'F'); add('a', 'f'); }
        else fail("the class [:" + name + ":] (negated and unknown POSIX classes)");
        i = close + 2;
        continue;
      }
      if (c == '[') fail("nested classes");
      if (c == '&' && i + 1 < p.size() && p[i + 1] == '&') fail("class intersections");
      if (c == '\\' && i + 1 < p.size() && p[i + 1] == 's') {      // [\s,;]: the White_Space members join the class
        if (icase) fail("\\s inside a class under (?i)");
        add_white_space(s, wide);
        i += 2;
        continue;
      }
      if (c == '\\' && i + 1 < p.size() && (p[i + 1] == 'd' || p[i + 1] == 'w')) {      // [\w.-]: the class's members join
        if (icase) fail("\\d / \\w inside a class under (?i)");
        add_perl_class(p[i + 1], s, wide);
        i += 2;
        continue;
      }
      const int lo = class_member();
      int hi = lo;
      if (i + 1 < p.size() && p[i] == '-' && p[i + 1] != ']') {
        if (p[i + 1] == '[') fail("nested classes");
        i++;
        hi = class_member();
        if (hi < lo) fail("a reversed range in a character class");
      }
      for (int b = lo; b <= hi && b < 128; b++) bs_add(s, b);
      if (hi >= 128) wide.emplace_back(std::max(lo, 128), hi);
    }
    return finish_class(s, wide, neg, false);
  }
  // White_Space (\\s): stable across Unicode versions, so it can be reproduced exactly (the other Perl classes cannot: \\d and \\w grow
  // with every Unicode release)
  static void add_white_space(ByteSet& s, std::vector<std::pair<int, int>>& wide) {
    for (int b = 9; b <= 13; b++) bs_add(s, b);
    bs_add(s, 32);
    for (auto r : {std::pair<int, int>{0x85, 0x85}, {0xA0, 0xA0}, {0x1680, 0x1680}, {0x2000, 0x200A}, {0x2028, 0x2029}, {0x202F, 0x202F}, {0x205F, 0x205F}, {0x3000, 0x3000}})
      wide.push_back(r);
  }
  // \d (General_Category = Nd) and \w (Alphabetic, M, Nd, Pc, Join_Control) as the crate's Unicode 16.0 tables have them
  // (regex_unicode_tables.hpp: probed from the crate itself, tools/gen_regex_tables.py)
  static void add_perl_class(char which, ByteSet& s, std::vector<std::pair<int, int>>& wide) {
    const int (*t)[2] = which == 'd' ? kPerlDigit : kPerlWord;
    const size_t n = which == 'd' ? sizeof kPerlDigit / sizeof kPerlDigit[0] : sizeof kPerlWord / sizeof kPerlWord[0];
    for (size_t k = 0; k < n; k++) {
      for (int b = t[k][0]; b <= t[k][1] && b < 128; b++) bs_add(s, b);
      if (t[k][1] >= 128) wide.emplace_back(std::max(t[k][0], 128), t[k][1]);
    }
  }
  // members collected → the class's node: case folding, negation, ASCII byte set + UTF-8 range sequences
  NodeP finish_class(ByteSet s, std::vector<std::pair<int, int>> wide, bool neg, bool perl) {
    bool has_k = false, has_s = false;
    if (icase) {
      if (!wide.empty() && !perl) fail("non-ASCII members of a character class under (?i)");
      // the class is folded first, then (if asked) negated — [^a] under (?i) is [^aA]
      for (int b = 'a'; b <= 'z'; b++)
        if (bs_has(s, b) || bs_has(s, b & ~0x20)) { bs_add(s, b); bs_add(s, b & ~0x20); }
      has_k = bs_has(s, 'k');
      has_s = bs_has(s, 's');
      // U+017F LONG S and U+212A KELVIN SIGN fold to s / k: members of the folded class
      if (has_s) wide.emplace_back(0x17F, 0x17F);
      if (has_k) wide.emplace_back(0x212A, 0x212A);
    }
    std::sort(wide.begin(), wide.end());
    if (neg) {
      ByteSet inv{};
      for (int b = 0; b < 128; b++)
        if (!bs_has(s, b)) bs_add(inv, b);
      s = inv;
      std::vector<std::pair<int, int>> rest;
      int next = 128;
      for (auto& r : wide) {
        if (r.first > next) rest.emplace_back(next, r.first - 1);
        next = std::max(next, r.second + 1);
      }
      if (next <= 0x10FFFF) rest.emplace_back(next, 0x10FFFF);
      wide = rest;
    }
    if (!bytes_too) return mk_char(mk(Node::Empty), s, wide);
    std::vector<NodeP> alts;
    bool any = false;
    for (int b = 0; b < 128; b++) any |= bs_has(s, b);
    if (any) alts.push_back(mk_bytes(s));
    if (wide.size() == 1 && wide[0].first == 128 && wide[0].second == 0x10FFFF) alts.push_back(multibyte());
    else if (wide.size() > 24) alts.push_back(class_trie(wide));      // \w, \W, \D, [^…] of those: one shared automaton instead of a thousand sequences
    else for (auto& r : wide) utf8_range_items(r.first, r.second, alts);
    if (alts.empty()) return mk_char(mk_bytes(ByteSet{}), s, wide);     // a class nothing can match
    return mk_char(mk_alt(alts), s, wide);
  }
  // scalar values [lo, hi] (beyond ASCII) as alternatives of byte-range sequences: split at the surrogate gap and at the encoded-length
  // boundaries, then until the continuation bytes of every piece span full ranges (the construction of the crate's utf8-ranges)
  static void utf8_range_items(int lo, int hi, std::vector<NodeP>& out) {
    if (lo > hi) return;
    if (lo <= 0xDFFF && hi >= 0xD800) {          // surrogates are not scalar values
      utf8_range_items(lo, 0xD7FF, out);
      utf8_range_items(0xE000, hi, out);
      return;
    }
    for (int mx : {0x7F, 0x7FF, 0xFFFF})
      if (lo <= mx && hi > mx) {
        utf8_range_items(lo, mx, out);
        utf8_range_items(mx + 1, hi, out);
        return;
      }
    const int n = hi < 0x80 ? 1 : hi < 0x800 ? 2 : hi < 0x10000 ? 3 : 4;
    for (int k = 1; k < n; k++) {
      const int m = (1 << (6 * k)) - 1;
      if ((lo & ~m) != (hi & ~m)) {
        if ((lo & m) != 0) {
          utf8_range_items(lo, lo | m, out);
          utf8_range_items((lo | m) + 1, hi, out);
          return;
        }
        if ((hi & m) != m) {
          utf8_range_items(lo, (hi & ~m) - 1, out);
          utf8_range_items(hi & ~m, hi, out);
          return;
        }
      }
    }
    const std::string a = utf8_of(lo), b = utf8_of(hi);
    std::vector<NodeP> seq;
    for (size_t k = 0; k < a.size(); k++) seq.push_back(mk_bytes(bs_range((unsigned char)a[k], (unsigned char)b[k])));
    out.push_back(mk_cat(seq));
  }
};

// ---- Thompson NFA ----
struct NState {
  enum Kind { Byte, Split, Bol, Eol, Match, WordB, NotWordB } kind = Match;
  ByteSet set{};
  int out = -1, out2 = -1;
};
struct Nfa {
  std::vector<NState> st;
  int add(NState::Kind k) {
    if (st.size() > 40000) throw CometError("RLIKE pattern is too large for the MI355X native engine (more than 40000 automaton states)");
    NState s;
    s.kind = k;
    st.push_back(s);
    return (int)st.size() - 1;
  }
};
// builds the fragment for `n` ending in state `next`; returns its entry state
int build(Nfa& nfa, const NodeP& n, int next) {
  switch (n->kind) {
    case Node::Empty: return next;
    case Node::OneChar: case Node::Group: return build(nfa, n->kids[0], next);
    case Node::Bytes: {
      const int s = nfa.add(NState::Byte);
      nfa.st[(size_t)s].set = n->set;
      nfa.st[(size_t)s].out = next;
      return s;
    }
    case Node::Bol: case Node::Eol: {
      const int s = nfa.add(n->kind == Node::Bol ? NState::Bol : NState::Eol);
      nfa.st[(size_t)s].out = next;
      return s;
    }
    case Node::WordB: case Node::NotWordB: {
      const int s = nfa.add(n->kind == Node::WordB ? NState::WordB : NState::NotWordB);
      nfa.st[(size_t)s].out = next;
      return s;
    }
    case Node::Trie: {
      std::vector<int> entry(n->trie.size(), -1);
      std::function<int(int)> make = [&](int k) -> int {
        if (entry[(size_t)k] >= 0) return entry[(size_t)k];
        int cur = -1;
        for (auto& e : n->trie[(size_t)k].edges) {
          const int b = nfa.add(NState::Byte);
          nfa.st[(size_t)b].set = e.first;
          const int to = e.second < 0 ? next : make(e.second);
          nfa.st[(size_t)b].out = to;
          if (cur < 0) cur = b;
          else {
            const int sp = nfa.add(NState::Split);
            nfa.st[(size_t)sp].out = b;
            nfa.st[(size_t)sp].out2 = cur;
            cur = sp;
          }
        }
        return entry[(size_t)k] = cur;
      };
      return make(0);
    }
    case Node::Cat: {
      int cur = next;
      for (size_t k = n->kids.size(); k-- > 0;) cur = build(nfa, n->kids[k], cur);
      return cur;
    }
    case Node::Alt: {
      int cur = build(nfa, n->kids.back(), next);
      for (size_t k = n->kids.size() - 1; k-- > 0;) {
        const int a = build(nfa, n->kids[k], next);
        const int s = nfa.add(NState::Split);
        nfa.st[(size_t)s].out = a;
        nfa.st[(size_t)s].out2 = cur;
        cur = s;
      }
      return cur;
    }
    case Node::Repeat: {
      const NodeP& kid = n->kids[0];
      int cur = next;
      if (n->max < 0) {
        // kid* : loop state
        const int loop = nfa.add(NState::Split);
        const int body = build(nfa, kid, loop);
        nfa.st[(size_t)loop].out = body;
        nfa.st[(size_t)loop].out2 = next;
        cur = loop;
      } else {
        for (int k = 0; k < n->max - n->min; k++) {      // optional copies, innermost last
          const int body = build(nfa, kid, cur);
          const int s = nfa.add(NState::Split);
          nfa.st[(size_t)s].out = body;
          nfa.st[(size_t)s].out2 = next;
          cur = s;
        }
      }
      for (int k = 0; k < n->min; k++) cur = build(nfa, kid, cur);
      return cur;
    }
  }
  return next;
}

// (sorted vectors and a visited array: the sets of a \\w pattern hold hundreds of states, and std::set made its construction take seconds)
typedef std::vector<int> StateSet;
struct Closer {
  const Nfa& nfa;
  std::vector<char> mark;
  std::vector<int> stack;
  explicit Closer(const Nfa& n) : nfa(n), mark(n.st.size(), 0) {}
  StateSet run(const StateSet& core, int extra, bool at_start, bool at_end) {
    StateSet out;
    stack.assign(core.begin(), core.end());
    if (extra >= 0) stack.push_back(extra);
    while (!stack.empty()) {
      const int s = stack.back();
      stack.pop_back();
      if (s < 0 || mark[(size_t)s]) continue;
      mark[(size_t)s] = 1;
      out.push_back(s);
      const NState& st = nfa.st[(size_t)s];
      switch (st.kind) {
        case NState::Split: stack.push_back(st.out); stack.push_back(st.out2); break;
        case NState::Bol: if (at_start) stack.push_back(st.out); break;
        case NState::Eol: if (at_end) stack.push_back(st.out); break;
        default: break;
      }
    }
    for (int s2 : out) mark[(size_t)s2] = 0;
    std::sort(out.begin(), out.end());
    return out;
  }
};
bool has_state(const StateSet& s, int x) { return std::binary_search(s.begin(), s.end(), x); }

// ---- patterns with \b / \B ----
// A word boundary looks at the character BEFORE the position and the one BEHIND it (Unicode \w, the text's ends count as non-word).  Over
// bytes: the automaton carries, beside its threads, a classifier of the character it is inside of (the \w automaton: which node, or "not a
// word character, k bytes to go") and whether the last complete character was a word character.  A thread that passes \b at a character
// boundary knows the left side and is TAGGED with what the right side has to be (1: a word character, 2: anything else or the end); when the
// classifier completes the next character the tags are checked and dropped.  Threads are numbered state·3 + tag.
struct WordClassifier {
  std::vector<TrieNode> trie;      // multi-byte \w
  std::vector<int> rem;            // bytes still to come below each node
  ByteSet ascii{};
  int nonword_base = 0;            // cs ≥ nonword_base: not a word character, cs − nonword_base + 1 continuation bytes to go
  WordClassifier() {
    std::vector<std::pair<int, int>> wide;
    Parser::add_perl_class('w', ascii, wide);
    trie = class_trie(wide)->trie;
    rem.assign(trie.size(), 0);
    std::function<int(int)> depth = [&](int k) -> int {
      if (rem[(size_t)k]) return rem[(size_t)k];
      const int to = trie[(size_t)k].edges[0].second;
      return rem[(size_t)k] = 1 + (to < 0 ? 0 : depth(to));
    };
    for (size_t k = 0; k < trie.size(); k++) depth((int)k);
    nonword_base = (int)trie.size() + 1;
  }
  // cs: 0 = between characters; 1 + node = inside a character that may still be a word character.  → the next cs; `done` / `word` when the byte ends a character
  int step(int cs, int b, bool& done, bool& word) const {
    done = false;
    word = false;
    if (cs >= nonword_base) {
      const int left = cs - nonword_base;      // continuation bytes behind this one
      if (left == 0) { done = true; return 0; }
      return cs - 1;
    }
    if (cs == 0 && b < 0x80) { done = true; word = bs_has(ascii, b); return 0; }
    if (cs == 0 && (b < 0xC2 || b > 0xF4)) { done = true; return 0; }      // (never in valid UTF-8: a character of its own, not a word character)
    const int node = cs == 0 ? 0 : cs - 1;
    for (auto& e : trie[(size_t)node].edges)
      if (bs_has(e.first, b)) {
        if (e.second < 0) { done = true; word = true; return 0; }
        return e.second + 1;
      }
    const int left = cs == 0 ? (b >= 0xF0 ? 3 : b >= 0xE0 ? 2 : 1) : rem[(size_t)node] - 1;
    if (left == 0) { done = true; return 0; }
    return nonword_base + left - 1;
  }
};

RegexDfa compile_with_word_boundaries(const std::string& pattern, const Nfa& nfa, int start, int match) {
  static const WordClassifier wc;
  RegexDfa dfa;
  {
    std::set<ByteSet> uniq;
    for (const NState& st : nfa.st)
      if (st.kind == NState::Byte) uniq.insert(st.set);
    for (const TrieNode& t : wc.trie)
      for (auto& e : t.edges) uniq.insert(e.first);
    uniq.insert(wc.ascii);
    for (int lim : {0x80, 0xC2, 0xE0, 0xF0, 0xF5}) uniq.insert(bs_range(0, lim - 1));
    std::map<std::vector<bool>, int> sig_id;
    dfa.classes.assign(256, 0);
    for (int b = 0; b < 256; b++) {
      std::vector<bool> sig;
      for (const ByteSet& u : uniq) sig.push_back(bs_has(u, b));
      auto it = sig_id.find(sig);
      if (it == sig_id.end()) it = sig_id.emplace(sig, (int)sig_id.size()).first;
      dfa.classes[(size_t)b] = (uint8_t)it->second;
    }
    dfa.nclasses = (int)sig_id.size();
  }
  const size_t NC = (size_t)dfa.nclasses;
  std::vector<int> class_byte(NC, 0);
  for (int b = 255; b >= 0; b--) class_byte[dfa.classes[(size_t)b]] = b;

  std::vector<char> mark(nfa.st.size() * 3, 0);
  std::vector<int> stack;
  // closure of tagged threads.  boundary: between characters (only there \b can be passed); prev_word: the character before; at_end: the text ends here
  auto closed = [&](const StateSet& core, bool inject, bool boundary, bool prev_word, bool at_start, bool at_end) {
    StateSet out;
    stack.assign(core.begin(), core.end());
    if (inject) stack.push_back(start * 3);
    while (!stack.empty()) {
      const int t = stack.back();
      stack.pop_back();
      if (t < 0 || mark[(size_t)t]) continue;
      mark[(size_t)t] = 1;
      out.push_back(t);
      const int sidx = t / 3, tag = t % 3;
      const NState& st = nfa.st[(size_t)sidx];
      switch (st.kind) {
        case NState::Split: stack.push_back(st.out * 3 + tag); stack.push_back(st.out2 * 3 + tag); break;
        case NState::Bol: if (at_start) stack.push_back(st.out * 3 + tag); break;
        case NState::Eol: if (at_end && tag != 1) stack.push_back(st.out * 3); break;        // the end is "not a word character"
        case NState::WordB: case NState::NotWordB: {
          if (!boundary) break;
          const bool want_differ = st.kind == NState::WordB;
          if (at_end) {
            if (tag == 1) break;
            if ((prev_word != false) == want_differ) stack.push_back(st.out * 3);
            break;
          }
          const int need = (want_differ ? !prev_word : prev_word) ? 1 : 2;
          if (tag == 0 || tag == need) stack.push_back(st.out * 3 + need);
          break;
        }
        default: break;
      }
    }
    for (int t : out) mark[(size_t)t] = 0;
    std::sort(out.begin(), out.end());
    return out;
  };
  struct Key {
    StateSet set; int cs; bool prev_word;
    bool operator<(const Key& o) const { return cs != o.cs ? cs < o.cs : prev_word != o.prev_word ? prev_word < o.prev_word : set < o.set; }
  };
  std::map<Key, int> ids;
  std::vector<Key> keys;
  auto intern = [&](Key k, bool is_initial) {
    if (!is_initial) {
      auto it = ids.find(k);
      if (it != ids.end()) return it->second;
    }
    if (keys.size() >= 4096) throw CometError("RLIKE pattern '" + pattern + "' needs more than 4096 automaton states: not supported by the MI355X native engine");
    const int id = (int)keys.size();
    if (!is_initial) ids[k] = id;
    keys.push_back(std::move(k));
    return id;
  };
  intern(Key{closed(StateSet(), true, true, false, true, false), 0, false}, true);      // state 0: before the first byte
  for (size_t cur = 0; cur < keys.size(); cur++) {
    const Key K = keys[cur];
    const bool initial = cur == 0;
    uint8_t flags = 0;
    if (has_state(K.set, match * 3)) flags |= 1;        // (an untagged MATCH: a tagged one still waits for the character behind it)
    if (K.cs == 0) {
      // the text ends here: tags asking for "no word character" are met, the others die; $ and \b read the end
      StateSet core;
      for (int t : K.set)
        if (t % 3 != 1) core.push_back(t - t % 3);
      if (has_state(closed(core, true, true, K.prev_word, initial, true), match * 3)) flags |= 2;
    }
    dfa.flags.push_back(flags);
    dfa.trans.resize((cur + 1) * NC * 2, 0);
    if (flags & 1) continue;
    for (size_t bc = 0; bc < NC; bc++) {
      const int b = class_byte[bc];
      bool done = false, word = false;
      const int cs2 = wc.step(K.cs, b, done, word);
      StateSet core;
      for (int t : K.set) {
        const NState& st = nfa.st[(size_t)(t / 3)];
        if (st.kind != NState::Byte || !bs_has(st.set, b)) continue;
        int tag = t % 3;
        if (done && tag != 0) {
          if ((tag == 1) != word) continue;       // the character behind the boundary is not what the thread needed
          tag = 0;
        }
        core.push_back(st.out * 3 + tag);
      }
      if (done) {
        // a MATCH that waited for this character
        for (int t : K.set)
          if (t / 3 == match && t % 3 != 0 && ((t % 3 == 1) == word)) core.push_back(match * 3);
      } else {
        for (int t : K.set)
          if (t / 3 == match && t % 3 != 0) core.push_back(t);
      }
      std::sort(core.begin(), core.end());
      core.erase(std::unique(core.begin(), core.end()), core.end());
      const bool pw = done ? word : K.prev_word;
      const int to = intern(Key{closed(core, true, done, pw, false, false), cs2, pw}, false);
      dfa.trans[(cur * NC + bc) * 2] = (uint8_t)(to & 0xff);
      dfa.trans[(cur * NC + bc) * 2 + 1] = (uint8_t)(to >> 8);
    }
  }
  dfa.nstates = (int)keys.size();
  dfa.trans.resize((size_t)dfa.nstates * NC * 2, 0);
  return dfa;
}

}  // namespace

RegexDfa compile_rlike(const std::string& pattern) {
  Parser ps(pattern);
  NodeP ast = ps.parse_alt();
  if (ps.more()) ps.fail("an unmatched ')'");
  Nfa nfa;
  const int match = nfa.add(NState::Match);
  const int start = build(nfa, ast, match);
  if (ps.has_wordb) return compile_with_word_boundaries(pattern, nfa, start, match);

  Closer closer(nfa);
  auto closed = [&](const StateSet& core, bool inject_start, bool at_start, bool at_end) { return closer.run(core, inject_start ? start : -1, at_start, at_end); };
  RegexDfa dfa;
  // bytes that every Byte state treats alike step alike: the construction (and the table) work on those classes
  {
    std::map<std::vector<bool>, int> sig_id;
    std::vector<const ByteSet*> sets_seen;
    std::set<ByteSet> uniq;
    for (const NState& st : nfa.st)
      if (st.kind == NState::Byte) uniq.insert(st.set);
    dfa.classes.assign(256, 0);
    for (int b = 0; b < 256; b++) {
      std::vector<bool> sig;
      sig.reserve(uniq.size() + 1);
      for (const ByteSet& u : uniq) sig.push_back(bs_has(u, b));
      sig.push_back(ps.multiline && b == '\n');
      auto it = sig_id.find(sig);
      if (it == sig_id.end()) it = sig_id.emplace(sig, (int)sig_id.size()).first;
      dfa.classes[(size_t)b] = (uint8_t)it->second;
    }
    dfa.nclasses = (int)sig_id.size();
  }
  std::vector<int> class_byte((size_t)dfa.nclasses, 0);
  for (int b = 255; b >= 0; b--) class_byte[dfa.classes[(size_t)b]] = b;
  const size_t NC = (size_t)dfa.nclasses;
  std::map<std::pair<StateSet, bool>, int> ids;
  std::vector<StateSet> sets;
  std::vector<bool> initial;                         // ^ is passable in this state: before the first byte, or (?m) right behind a \n
  auto intern = [&](const StateSet& s, bool is_initial, bool line_start = false) {
    auto it = ids.find({s, line_start});
    if (it != ids.end() && !is_initial) return it->second;
    if (sets.size() >= 4096) throw CometError("RLIKE pattern '" + pattern + "' needs more than 4096 automaton states: not supported by the MI355X native engine");
    const int id = (int)sets.size();
    if (!is_initial) ids[{s, line_start}] = id;
    sets.push_back(s);
    initial.push_back(is_initial || line_start);
    return id;
  };
  intern(closed(StateSet(), true, true, false), true);      // state 0: before the first byte (^ passable)
  for (size_t cur = 0; cur < sets.size(); cur++) {
    const StateSet S = sets[cur];                   // (copy: `sets` grows below)
    uint8_t flags = 0;
    if (has_state(S, match)) flags |= 1;
    // end of text here: $ becomes passable; the start state may still be injected (an empty remainder can match, e.g. "x*$")
    if (has_state(closed(S, true, initial[cur], true), match)) flags |= 2;
    dfa.flags.push_back(flags);
    dfa.trans.resize((cur + 1) * NC * 2, 0);
    if (flags & 1) continue;                        // absorbing: the kernel has already answered true
    // (?m): in front of a \n byte `$` is passable (so that byte steps from the set closed that way — a set that already holds MATCH answers
    // true at once), and behind it `^` is
    const StateSet S_eol = ps.multiline ? closed(S, true, initial[cur], true) : StateSet();
    // the Byte states of the set, once (not once per class)
    std::vector<const NState*> byte_states, byte_states_eol;
    for (int s2 : S) if (nfa.st[(size_t)s2].kind == NState::Byte) byte_states.push_back(&nfa.st[(size_t)s2]);
    for (int s2 : S_eol) if (nfa.st[(size_t)s2].kind == NState::Byte) byte_states_eol.push_back(&nfa.st[(size_t)s2]);
    for (size_t bc = 0; bc < NC; bc++) {
      const int b = class_byte[bc];
      const bool nl = ps.multiline && b == '\n';
      int to;
      if (nl && has_state(S_eol, match)) {
        to = intern(StateSet{match}, false);
      } else {
        StateSet core;
        for (const NState* st : (nl ? byte_states_eol : byte_states))
          if (bs_has(st->set, b)) core.push_back(st->out);
        to = intern(closed(core, true, nl, false), false, nl);
      }
      dfa.trans[(cur * NC + bc) * 2] = (uint8_t)(to & 0xff);
      dfa.trans[(cur * NC + bc) * 2 + 1] = (uint8_t)(to >> 8);
    }
  }
  dfa.nstates = (int)sets.size();
  dfa.trans.resize((size_t)dfa.nstates * NC * 2, 0);
  return dfa;
}

// ---- regexp_extract: the capture program of device/regex_vm.hpp ----
namespace {
bool nullable(const NodeP& n) {
  switch (n->kind) {
    case Node::Empty: case Node::Bol: case Node::Eol: case Node::WordB: case Node::NotWordB: return true;
    case Node::Bytes: case Node::Trie: case Node::OneChar: return false;
    case Node::Cat: for (auto& k : n->kids) if (!nullable(k)) return false; return true;
    case Node::Alt: for (auto& k : n->kids) if (nullable(k)) return true; return false;
    case Node::Repeat: return n->min == 0 || nullable(n->kids[0]);
    case Node::Group: return nullable(n->kids[0]);
  }
  return false;
}
struct ProgBuilder {
  const std::string& pattern;
  const char* fn;
  int wanted;
  std::vector<std::array<uint32_t, 2>> ins;
  struct Cls { ByteSet ascii; std::vector<std::pair<int, int>> wide; };
  std::vector<Cls> classes;
  [[noreturn]] void fail(const std::string& why) const {
    throw CometError(std::string(fn) + " pattern '" + pattern + "' is not supported by the MI355X native engine: " + why);
  }
  int add(uint32_t kind, int out, uint32_t arg) {
    if ((int)ins.size() >= kRxMaxInstr) fail("it needs more than " + std::to_string(kRxMaxInstr) + " matcher instructions (long counted repetitions)");
    ins.push_back({kind | ((uint32_t)out << 8), arg});
    return (int)ins.size() - 1;
  }
  int add_class(const ByteSet& ascii, std::vector<std::pair<int, int>> wide) {
    // sorted, merged ranges (the matcher's binary search wants them disjoint)
    std::sort(wide.begin(), wide.end());
    std::vector<std::pair<int, int>> m;
    for (auto& r : wide) {
      if (!m.empty() && r.first <= m.back().second + 1) m.back().second = std::max(m.back().second, r.second);
      else m.push_back(r);
    }
    for (size_t c = 0; c < classes.size(); c++)
      if (classes[c].ascii == ascii && classes[c].wide == m) return (int)c;
    classes.push_back({ascii, m});
    return (int)classes.size() - 1;
  }
  // the fragment for `n` that continues at instruction `next`; → its entry
  int build(const NodeP& n, int next) {
    switch (n->kind) {
      case Node::Empty: return next;
      case Node::Bytes: case Node::Trie: fail("internal: a byte-level item outside a character");
      case Node::OneChar: {
        int members = 0, only = -1;
        for (int b = 0; b < 128; b++) if (bs_has(n->set, b)) { members++; only = b; }
        if (n->wide.empty() && members == 1) return add(1, next, (uint32_t)only);
        if (members == 0 && n->wide.size() == 1 && n->wide[0].first == n->wide[0].second) return add(1, next, (uint32_t)n->wide[0].first);
        return add(2, next, (uint32_t)add_class(n->set, n->wide));
      }
      case Node::Bol: return add(5, next, 0);
      case Node::Eol: return add(6, next, 0);
      case Node::WordB: return add(7, next, 0);
      case Node::NotWordB: return add(8, next, 0);
      case Node::Group: {
        if (n->min != wanted) return build(n->kids[0], next);
        const int close = add(4, next, 1);
        const int body = build(n->kids[0], close);
        return add(4, body, 0);
      }
      case Node::Cat: {
        int cur = next;
        for (size_t k = n->kids.size(); k-- > 0;) cur = build(n->kids[k], cur);
        return cur;
      }
      case Node::Alt: {
        int cur = build(n->kids.back(), next);
        for (size_t k = n->kids.size() - 1; k-- > 0;) {
          const int a = build(n->kids[k], next);
          cur = add(3, a, (uint32_t)cur);
        }
        return cur;
      }
      case Node::Repeat: {
        const NodeP& kid = n->kids[0];
        // (the crate compiles x* over an x that can match nothing as (x+)? to keep its preference order; such patterns are refused instead)
        if (n->max < 0 && nullable(kid)) fail("an unbounded repetition of something that can match the empty string");
        auto split = [&](int body, int exit) { return n->lazy ? add(3, exit, (uint32_t)body) : add(3, body, (uint32_t)exit); };
        int cur = next;
        if (n->max < 0) {
          const int loop = add(3, 0, 0);
          const int body = build(kid, loop);
          ins[(size_t)loop] = n->lazy ? std::array<uint32_t, 2>{3u | ((uint32_t)next << 8), (uint32_t)body} : std::array<uint32_t, 2>{3u | ((uint32_t)body << 8), (uint32_t)next};
          cur = loop;
        } else {
          for (int k = 0; k < n->max - n->min; k++) {
            const int body = build(kid, cur);
            cur = split(body, next);
          }
        }
        for (int k = 0; k < n->min; k++) cur = build(kid, cur);
        return cur;
      }
    }
    return next;
  }
};
}  // namespace

RegexProg compile_regex_captures(const std::string& pattern, int group, const char* fn) {
  Parser ps(pattern);
  ps.bytes_too = false;
  NodeP ast;
  try {
    ast = ps.parse_alt();
    if (ps.more()) ps.fail("an unmatched ')'");
  } catch (const CometError& e) {
    // (the parser names RLIKE: the same syntax, another function)
    std::string m = e.what();
    const size_t at = m.find("RLIKE pattern");
    if (at != std::string::npos) m.replace(at, 5, fn);
    throw CometError(m);
  }
  // regexp_extract_common.rs:85-92
  if (group < 0 || group > ps.ngroups)
    throw CometError("The value of parameter `idx` in `" + std::string(fn) + "` is invalid: Expects group index between 0 and " + std::to_string(ps.ngroups) + ", but got " + std::to_string(group) + ".");
  if (ps.has_wordb && ps.multiline) throw CometError(std::string(fn) + " pattern '" + pattern + "': \\b under (?m) is not supported by the MI355X native engine");
  ProgBuilder b{pattern, fn, group, {}, {}};
  const int match = b.add(0, 0, 0);
  int entry;
  if (group == 0) {
    const int close = b.add(4, match, 1);
    const int body = b.build(ast, close);
    entry = b.add(4, body, 0);
  } else entry = b.build(ast, match);
  uint32_t word_class = 0;
  if (ps.has_wordb) {
    ByteSet as{};
    std::vector<std::pair<int, int>> wide;
    Parser::add_perl_class('w', as, wide);
    word_class = (uint32_t)b.add_class(as, wide);
  }
  RegexProg prog;
  prog.ngroups = ps.ngroups;
  std::vector<uint32_t>& w = prog.words;
  w.assign(6, 0);
  w[0] = (uint32_t)b.ins.size();
  w[1] = (uint32_t)entry;
  w[2] = (ps.multiline ? 1u : 0u) | (ps.has_wordb ? 2u : 0u);
  w[4] = word_class;
  for (auto& i : b.ins) { w.push_back(i[0]); w.push_back(i[1]); }
  w[3] = (uint32_t)w.size();
  const size_t table = w.size();
  w.resize(table + 6 * b.classes.size(), 0);
  for (size_t c = 0; c < b.classes.size(); c++) {
    for (int k = 0; k < 2; k++) {
      w[table + 6 * c + 2 * (size_t)k] = (uint32_t)b.classes[c].ascii[(size_t)k];
      w[table + 6 * c + 2 * (size_t)k + 1] = (uint32_t)(b.classes[c].ascii[(size_t)k] >> 32);
    }
    w[table + 6 * c + 4] = (uint32_t)w.size();
    w[table + 6 * c + 5] = (uint32_t)b.classes[c].wide.size();
    for (auto& r : b.classes[c].wide) { w.push_back((uint32_t)r.first); w.push_back((uint32_t)r.second); }
  }
  if (w.size() > 16384) throw CometError(std::string(fn) + " pattern '" + pattern + "' is too large for the MI355X native engine");
  return prog;
}

bool regex_prog_extract(const RegexProg& prog, const uint8_t* s, size_t n, int32_t* start, int32_t* len) {
  int32_t m0 = -1, m1 = -1;
  const bool hit = rx_extract(prog.words.data(), s, (int32_t)n, m0, m1);
  *start = m0 < 0 ? 0 : m0;
  *len = m0 < 0 ? 0 : m1 - m0;
  return hit;
}

std::vector<std::pair<int32_t, int32_t>> regex_prog_split(const RegexProg& prog, const uint8_t* s, size_t n, int32_t limit) {
  // the device's two passes: count, then write that many pieces
  const int32_t count = rx_split(prog.words.data(), s, (int32_t)n, limit, 0, [](int32_t, int32_t, int32_t) {});
  std::vector<std::pair<int32_t, int32_t>> out((size_t)count, {-1, -1});
  rx_split(prog.words.data(), s, (int32_t)n, limit, count, [&](int32_t k, int32_t a, int32_t len) { out[(size_t)k] = {a, len}; });
  return out;
}

std::vector<std::pair<int32_t, int32_t>> regex_prog_find_all(const RegexProg& whole, const RegexProg& group, const uint8_t* s, size_t n) {
  const uint32_t* w0 = whole.words.data();
  const uint32_t* wg = &group == &whole ? w0 : group.words.data();
  const int32_t count = rx_find_all(w0, wg, s, (int32_t)n, 0, [](int32_t, int32_t, int32_t) {});
  std::vector<std::pair<int32_t, int32_t>> out((size_t)count, {-1, -1});
  rx_find_all(w0, wg, s, (int32_t)n, count, [&](int32_t k, int32_t a, int32_t len) { out[(size_t)k] = {a, len}; });
  return out;
}

bool regex_dfa_match(const RegexDfa& d, const uint8_t* s, size_t n) {
  int st = 0;
  if (d.flags[0] & 1) return true;
  for (size_t k = 0; k < n; k++) {
    st = d.next(st, s[k]);
    if (d.flags[(size_t)st] & 1) return true;
  }
  return (d.flags[(size_t)st] & 2) != 0;
}

}  // namespace comet
