// regexp_extract's matcher: a Pike VM over the SCALAR VALUES of one Utf8 value (string_funcs/regexp_extract.rs:79-108: the crate's
// Regex::captures_read = the LEFTMOST match, alternatives and repetitions preferred in pattern order, and the span of ONE group in it).
//
// The program (csrc/regex.cpp compile_regex_captures) is an array of 32-bit words:
//   [0] instructions (≤ kRxMaxInstr)  [1] entry  [2] flags (1: (?m), 2: the pattern holds \b / \B)  [3] word index of the class table
//   [4] the class that is \w (word boundaries)  [5] 0
//   instruction k at [6 + 2k]: kind | out << 8, then one argument:
//     0 Match   1 Char (arg = the scalar value)   2 Class (arg = class index)   3 Split (out preferred over arg)   4 Save (arg = slot 0 / 1)
//     5 Bol   6 Eol   7 WordB   8 NotWordB
//   class c at [table + 6c]: four words of ASCII members, word index of its ranges beyond ASCII, their count; a range = lo, hi.
// Threads live in two lists ordered by preference; a state enters a list once per position (the first, i.e. preferred, arrival wins), a
// thread that reaches Match ends every thread behind it, and no new search starts once something has matched: what is left when the lists
// run dry is the crate's answer.  Positions are character boundaries only (an empty match never splits a character).
//
// Plain C++ over a byte pointer: the generated kernels include it (comet_regex_vm.hpp), the host runs the same source for the CPU tests
// (capi.cpp comet_regexp_extract_host).
#pragma once
typedef unsigned char rx_u8;
typedef unsigned int rx_u32;
typedef int rx_i32;
typedef unsigned long long rx_u64;
#ifndef RXVM_FN
#define RXVM_FN inline
#endif
#ifndef RXVM_ENTRY
#define RXVM_ENTRY inline      // (the device build keeps the matcher out of line: one stack frame for its lists however many expressions call it)
#endif

constexpr int kRxMaxInstr = 64;

RXVM_FN bool rx_class_has(const rx_u32* w, rx_u32 cls, rx_u32 cp) {
  const rx_u32* c = w + w[3] + 6u * cls;
  if (cp < 128u) return (c[cp >> 5] >> (cp & 31u)) & 1u;
  const rx_u32* r = w + c[4];
  rx_i32 lo = 0, hi = (rx_i32)c[5];
  while (lo < hi) {
    const rx_i32 mid = (lo + hi) >> 1;
    if (r[2 * mid + 1] < cp) lo = mid + 1; else hi = mid;
  }
  return lo < (rx_i32)c[5] && r[2 * lo] <= cp;
}

// one scalar value of (valid) UTF-8 at text[pos]; something that is not a character is a character of its own that nothing matches
template <class P>
RXVM_FN rx_u32 rx_decode(P text, rx_i32 pos, rx_i32 n, rx_i32& len) {
  const rx_u32 c = text[pos];
  len = 1;
  if (c < 0x80u) return c;
  if (c >= 0xC2u && c < 0xE0u && pos + 1 < n) { len = 2; return ((c & 0x1Fu) << 6) | (text[pos + 1] & 0x3Fu); }
  if (c >= 0xE0u && c < 0xF0u && pos + 2 < n) { len = 3; return ((c & 0x0Fu) << 12) | ((rx_u32)(text[pos + 1] & 0x3Fu) << 6) | (text[pos + 2] & 0x3Fu); }
  if (c >= 0xF0u && c < 0xF5u && pos + 3 < n) {
    len = 4;
    return ((c & 0x07u) << 18) | ((rx_u32)(text[pos + 1] & 0x3Fu) << 12) | ((rx_u32)(text[pos + 2] & 0x3Fu) << 6) | (text[pos + 3] & 0x3Fu);
  }
  return 0x110000u + c;
}

struct RxList {
  rx_u8 pc[kRxMaxInstr];
  rx_i32 s0[kRxMaxInstr], s1[kRxMaxInstr];
  rx_i32 n;
  rx_u64 seen;
};
struct RxStack {
  rx_u8 pc[2 * kRxMaxInstr + 2];
  rx_i32 s0[2 * kRxMaxInstr + 2], s1[2 * kRxMaxInstr + 2];
};

// what a position looks like to the assertions
struct RxAt { bool start, end, after_nl, before_nl, prev_word, next_word; };

// `pc` and everything reachable from it without consuming a character joins the list, preferred branches first
RXVM_FN void rx_add(const rx_u32* w, RxList& l, RxStack& st, rx_u32 pc, rx_i32 pos, const RxAt& at, rx_i32 s0, rx_i32 s1) {
  const bool multiline = (w[2] & 1u) != 0;
  rx_i32 top = 0;
  st.pc[0] = (rx_u8)pc;
  st.s0[0] = s0;
  st.s1[0] = s1;
  top = 1;
  while (top > 0) {
    top--;
    const rx_u32 p = st.pc[top];
    const rx_i32 a = st.s0[top], b = st.s1[top];
    if ((l.seen >> p) & 1ull) continue;
    l.seen |= 1ull << p;
    const rx_u32 i0 = w[6 + 2 * p], arg = w[7 + 2 * p];
    const rx_u32 kind = i0 & 0xffu, out = i0 >> 8;
    bool follow = false;
    rx_i32 na = a, nb = b;
    switch (kind) {
      case 3:      // Split: `out` first — it is pushed last
        st.pc[top] = (rx_u8)arg; st.s0[top] = a; st.s1[top] = b; top++;
        follow = true;
        break;
      case 4:
        if (arg == 0u) na = pos; else nb = pos;
        follow = true;
        break;
      case 5: follow = at.start || (multiline && at.after_nl); break;
      case 6: follow = at.end || (multiline && at.before_nl); break;
      case 7: follow = at.prev_word != at.next_word; break;
      case 8: follow = at.prev_word == at.next_word; break;
      default: {   // Match, Char, Class: a thread that waits for the next character (or reports)
        const rx_i32 k = l.n++;
        l.pc[k] = (rx_u8)p;
        l.s0[k] = a;
        l.s1[k] = b;
        break;
      }
    }
    if (follow) { st.pc[top] = (rx_u8)out; st.s0[top] = na; st.s1[top] = nb; top++; }
  }
}

// The leftmost match that starts at or behind byte `from` (a character boundary; what lies before it is still seen by ^, \b and (?m)).
// → matched?; [m0, m1) = the wanted group's bytes, m0 < 0 when the group took no part in the match
template <class P>
RXVM_ENTRY bool rx_search(const rx_u32* w, P text, rx_i32 n, rx_i32 from, rx_i32& m0, rx_i32& m1) {
  RxList la, lb;
  RxStack st;
  RxList* cl = &la;
  RxList* nl = &lb;
  const bool wordb = (w[2] & 2u) != 0;
  const rx_u32 wcls = w[4], entry = w[1];
  bool matched = false;
  m0 = m1 = -1;
  if (from > n) return false;
  rx_i32 pos = from, len0 = 0, len1 = 0;
  rx_u32 cp0 = 0, cp1 = 0;
  if (pos < n) cp0 = rx_decode(text, pos, n, len0);
  RxAt at;
  at.start = pos == 0;
  at.end = pos >= n;
  at.after_nl = pos > 0 && text[pos - 1] == (rx_u8)'\n';
  at.before_nl = pos < n && cp0 == (rx_u32)'\n';
  at.prev_word = false;
  if (wordb && pos > 0) {
    rx_i32 k = pos - 1, lk = 0;
    while (k > 0 && (text[k] & 0xC0u) == 0x80u) k--;
    const rx_u32 pc = rx_decode(text, k, n, lk);
    at.prev_word = pc < 0x110000u && rx_class_has(w, wcls, pc);
  }
  at.next_word = wordb && pos < n && cp0 < 0x110000u && rx_class_has(w, wcls, cp0);
  cl->n = 0;
  cl->seen = 0;
  rx_add(w, *cl, st, entry, pos, at, -1, -1);
  while (true) {
    const bool at_end = pos >= n;
    const rx_i32 q = at_end ? pos : pos + len0;
    if (!at_end) {
      // the position behind this character, as the threads that consume it will find it
      const bool q_end = q >= n;
      if (!q_end) cp1 = rx_decode(text, q, n, len1);
      at.start = false;
      at.end = q_end;
      at.after_nl = cp0 == (rx_u32)'\n';
      at.before_nl = !q_end && cp1 == (rx_u32)'\n';
      if (wordb) {
        at.prev_word = cp0 < 0x110000u && rx_class_has(w, wcls, cp0);
        at.next_word = !q_end && cp1 < 0x110000u && rx_class_has(w, wcls, cp1);
      }
    }
    nl->n = 0;
    nl->seen = 0;
    for (rx_i32 k = 0; k < cl->n; k++) {
      const rx_u32 p = cl->pc[k];
      const rx_u32 i0 = w[6 + 2 * p], arg = w[7 + 2 * p];
      const rx_u32 kind = i0 & 0xffu;
      if (kind == 0u) {         // Match: the threads behind it are less preferred
        m0 = cl->s0[k];
        m1 = cl->s1[k];
        matched = true;
        break;
      }
      if (at_end) continue;
      const bool takes = kind == 1u ? cp0 == arg : (cp0 < 0x110000u && rx_class_has(w, arg, cp0));
      if (takes) rx_add(w, *nl, st, i0 >> 8, q, at, cl->s0[k], cl->s1[k]);
    }
    if (at_end) break;
    pos = q;
    cp0 = cp1;
    len0 = len1;
    RxList* t = cl;
    cl = nl;
    nl = t;
    if (!matched) rx_add(w, *cl, st, entry, pos, at, -1, -1);      // a search starting here: behind every thread that started earlier
    if (cl->n == 0 && matched) break;
  }
  if (matched && (m0 < 0 || m1 < m0)) m0 = m1 = -1;
  return matched;
}

template <class P>
RXVM_FN bool rx_extract(const rx_u32* w, P text, rx_i32 n, rx_i32& m0, rx_i32& m1) {
  return rx_search(w, text, n, 0, m0, m1);
}

// split(str, pattern, limit) (string_funcs/split.rs:434-472 over the crate's Regex::split / find_iter): the pieces between successive
// matches of a group-0 program.  find_iter's rule for empty matches: one that ends where the previous match ended is dropped and the
// search resumes one character on.  limit > 0: at most limit − 1 cuts; limit = 0: trailing empty pieces are dropped (nothing left = one
// empty piece); limit < 0: every piece.  `emit(k, start, len)` sees piece k; → the number of pieces the list holds.  With `max_emit` the
// caller bounds what is emitted (the writing pass passes the count the counting pass returned).
template <class P, class F>
RXVM_FN rx_i32 rx_split(const rx_u32* w, P text, rx_i32 n, rx_i32 limit, rx_i32 max_emit, F emit) {
  rx_i32 last = 0, at = 0, last_end = -1, k = 0, kept = 0;
  while (true) {
    if (limit > 0 && k >= limit - 1) break;
    rx_i32 m0 = -1, m1 = -1;
    if (!rx_search(w, text, n, at, m0, m1) || m0 < 0) break;
    if (m0 == m1 && m1 == last_end) {
      // an empty match right behind the previous match: the search resumes behind the next character
      at = at + 1;
      while (at < n && (text[at] & 0xC0u) == 0x80u) at++;
      if (!rx_search(w, text, n, at, m0, m1) || m0 < 0) break;
    }
    if (k < max_emit) emit(k, last, m0 - last);
    if (m0 > last) kept = k + 1;
    k++;
    last = m1;
    at = m1;
    last_end = m1;
  }
  if (k < max_emit) emit(k, last, n - last);
  if (n > last) kept = k + 1;
  k++;
  if (limit == 0) return kept == 0 ? 1 : kept;
  return k;
}

// regexp_extract_all(str, pattern, idx) (string_funcs/regexp_extract_all.rs:79-108 over the crate's captures_iter = find_iter's matches): group idx of
// EVERY match.  `w0` is the pattern's group-0 program (it drives the iteration), `wg` the wanted group's (the same program when idx = 0): searched
// from the same position both find the same match.  `emit(k, start, len)` sees match k's group (empty when it took no part); → the number of matches.
template <class P, class F>
RXVM_FN rx_i32 rx_find_all(const rx_u32* w0, const rx_u32* wg, P text, rx_i32 n, rx_i32 max_emit, F emit) {
  rx_i32 at = 0, last_end = -1, k = 0;
  while (true) {
    rx_i32 m0 = -1, m1 = -1;
    if (!rx_search(w0, text, n, at, m0, m1) || m0 < 0) break;
    if (m0 == m1 && m1 == last_end) {
      at = at + 1;
      while (at < n && (text[at] & 0xC0u) == 0x80u) at++;
      if (!rx_search(w0, text, n, at, m0, m1) || m0 < 0) break;
    }
    if (k < max_emit) {
      rx_i32 g0 = m0, g1 = m1;
      if (wg != w0 && !rx_search(wg, text, n, at, g0, g1)) g0 = g1 = -1;
      emit(k, g0 < 0 ? 0 : g0, g0 < 0 ? 0 : g1 - g0);
    }
    k++;
    at = m1;
    last_end = m1;
  }
  return k;
}
