#include "CircuitVectorizer.hpp"

#include <cctype>
#include <map>
#include <stdexcept>

namespace {

struct Tok {
  enum Kind { Ident, Int, Punct } kind;
  std::string text;
};

std::vector<Tok> lex(const std::string &s) {
  std::vector<Tok> out;
  size_t i = 0;
  static const char *three[] = {"+++", "---", "***"};
  static const char *two[] = {"<=", ">=", "==", "!="};
  while (i < s.size()) {
    const char ch = s[i];
    if (std::isspace((unsigned char)ch)) { ++i; continue; }
    if (ch == '/' && i + 1 < s.size() && s[i + 1] == '/') {
      while (i < s.size() && s[i] != '\n') ++i;
      continue;
    }
    if (std::isalpha((unsigned char)ch) || ch == '_') {
      size_t j = i;
      while (j < s.size() && (std::isalnum((unsigned char)s[j]) || s[j] == '_')) ++j;
      out.push_back({Tok::Ident, s.substr(i, j - i)});
      i = j;
    } else if (std::isdigit((unsigned char)ch)) {
      size_t j = i;
      while (j < s.size() && std::isdigit((unsigned char)s[j])) ++j;
      out.push_back({Tok::Int, s.substr(i, j - i)});
      i = j;
    } else {
      std::string p(1, ch);
      for (const char *t : three)
        if (s.compare(i, 3, t) == 0) p = t;
      if (p.size() == 1)
        for (const char *t : two)
          if (s.compare(i, 2, t) == 0) p = t;
      out.push_back({Tok::Punct, p});
      i += p.size();
    }
  }
  return out;
}

bool isP(const Tok &t, const char *p) { return t.kind == Tok::Punct && t.text == p; }

std::string join(const std::vector<Tok> &toks) {
  std::string s;
  for (size_t i = 0; i < toks.size(); ++i) {
    const bool tight = i == 0 || isP(toks[i], ",") || isP(toks[i], ")") || isP(toks[i], "]") || isP(toks[i], "[") ||
                       isP(toks[i - 1], "(") || isP(toks[i - 1], "[") || isP(toks[i], ";") ||
                       (isP(toks[i], "(") && toks[i - 1].kind == Tok::Ident);
    if (!tight) s += " ";
    s += toks[i].text;
  }
  return s;
}

// one top-level statement (without its ';' unless it is a block statement such as a for loop)
struct Stmt {
  std::vector<Tok> toks;
  bool block = false;  // contains braces: passed through verbatim
  enum Kind { Other, Elem, Reduce } kind = Other;
  std::string target;       // Elem: vector written, Reduce: accumulator
  int slot = -1;            // Elem: slot written; Reduce: slot read
  bool constant = false;    // Elem: right-hand side is an integer literal
  long value = 0;           //       ... this one
  std::vector<Tok> shape;   // right-hand side (Reduce: the added term) with every `v[slot]` replaced by `v`
  std::set<std::string> idents;
};

// replaces `ident [ INT ]` by `ident`; all indices must be equal; returns false if an index is not a literal or they differ
bool abstractIndices(const std::vector<Tok> &rhs, std::vector<Tok> &shape, int &index, bool &any) {
  any = false;
  index = -1;
  for (size_t i = 0; i < rhs.size(); ++i) {
    if (rhs[i].kind == Tok::Ident && i + 1 < rhs.size() && isP(rhs[i + 1], "[")) {
      if (i + 3 >= rhs.size() || rhs[i + 2].kind != Tok::Int || !isP(rhs[i + 3], "]")) return false;
      const int idx = std::stoi(rhs[i + 2].text);
      if (any && idx != index) return false;
      any = true;
      index = idx;
      shape.push_back(rhs[i]);
      i += 3;
    } else {
      shape.push_back(rhs[i]);
    }
  }
  return true;
}

void classify(Stmt &s, const std::set<std::string> &scalars) {
  const auto &t = s.toks;
  for (const auto &k : t)
    if (k.kind == Tok::Ident) s.idents.insert(k.text);
  if (s.block || t.size() < 3 || t[0].kind != Tok::Ident) return;
  // x [ i ] = rhs
  if (t.size() >= 6 && isP(t[1], "[") && t[2].kind == Tok::Int && isP(t[3], "]") && isP(t[4], "=")) {
    const std::vector<Tok> rhs(t.begin() + 5, t.end());
    const int slot = std::stoi(t[2].text);
    const bool negLit = rhs.size() == 2 && isP(rhs[0], "-") && rhs[1].kind == Tok::Int;
    if ((rhs.size() == 1 && rhs[0].kind == Tok::Int) || negLit) {
      s.kind = Stmt::Elem; s.target = t[0].text; s.slot = slot; s.constant = true;
      s.value = negLit ? -std::stol(rhs[1].text) : std::stol(rhs[0].text);
      return;
    }
    std::vector<Tok> shape;
    int idx;
    bool any;
    if (abstractIndices(rhs, shape, idx, any) && any && idx == slot) {
      s.kind = Stmt::Elem; s.target = t[0].text; s.slot = slot; s.shape = shape;
    }
    return;
  }
  // s = s + term   (s a declared scalar accumulator)
  if (t.size() >= 5 && isP(t[1], "=") && t[2].kind == Tok::Ident && t[2].text == t[0].text && scalars.count(t[0].text) &&
      (isP(t[3], "+") || isP(t[3], "+++"))) {
    const std::vector<Tok> term(t.begin() + 4, t.end());
    for (const auto &k : term)  // the term must be one summand: no further top-level + or - (products and parentheses are fine)
      if (k.kind == Tok::Ident && k.text == t[0].text) return;
    int depth = 0;
    for (const auto &k : term) {
      if (isP(k, "(")) ++depth;
      if (isP(k, ")")) --depth;
      if (depth == 0 && (isP(k, "+") || isP(k, "-") || isP(k, "+++") || isP(k, "---"))) return;
    }
    std::vector<Tok> shape;
    int idx;
    bool any;
    if (abstractIndices(term, shape, idx, any) && any) {
      s.kind = Stmt::Reduce; s.target = t[0].text; s.slot = idx; s.shape = shape;
    }
  }
}

bool sameShape(const std::vector<Tok> &a, const std::vector<Tok> &b) {
  if (a.size() != b.size()) return false;
  for (size_t i = 0; i < a.size(); ++i)
    if (a[i].kind != b[i].kind || a[i].text != b[i].text) return false;
  return true;
}

std::string listOf(const std::vector<long> &v) {
  std::string s = "{";
  for (size_t i = 0; i < v.size(); ++i) s += (i ? ", " : "") + std::to_string(v[i]);
  return s + "}";
}

}  // namespace

std::string CircuitVectorizer::vectorize(const std::string &program) {
  elementwiseRuns = reductionRuns = 0;
  // ---- split into top-level statements ----
  const std::vector<Tok> toks = lex(program);
  std::vector<Stmt> stmts;
  {
    Stmt cur;
    int brace = 0, paren = 0;
    for (const Tok &t : toks) {
      if (isP(t, "{")) {
        // an expression list `{1, 2}` inside an expression is not a block: blocks follow `)` or start a statement
        const bool blockOpen = cur.toks.empty() || isP(cur.toks.back(), ")") || brace > 0;
        if (blockOpen || cur.block) { cur.block = true; ++brace; }
        else ++paren;  // treat the list's braces like parentheses
        cur.toks.push_back(t);
        continue;
      }
      if (isP(t, "}")) {
        cur.toks.push_back(t);
        if (cur.block && brace > 0) {
          if (--brace == 0) { stmts.push_back(cur); cur = Stmt(); }
        } else {
          --paren;
        }
        continue;
      }
      if (isP(t, "(")) ++paren;
      if (isP(t, ")")) --paren;
      if (isP(t, ";") && brace == 0 && paren == 0) {
        if (!cur.toks.empty()) stmts.push_back(cur);
        cur = Stmt();
        continue;
      }
      cur.toks.push_back(t);
    }
    if (!cur.toks.empty()) stmts.push_back(cur);
  }
  for (auto &s : stmts) classify(s, scalars);

  // ---- group runs ----
  std::vector<std::string> out;
  struct Run {
    Stmt::Kind kind;
    std::vector<const Stmt *> members;
  };
  std::map<std::string, Run> open;          // by target
  std::vector<std::string> openOrder;       // emission order = order in which the runs were opened

  auto passThrough = [&](const Stmt &s) { out.push_back(join(s.toks) + (s.block ? "" : ";")); };

  auto flush = [&](const std::string targetName) {  // by value: the caller's string may live in openOrder, which shrinks below
    const std::string &target = targetName;
    auto it = open.find(target);
    if (it == open.end()) return;
    const Run run = it->second;
    open.erase(it);
    for (size_t i = 0; i < openOrder.size(); ++i)
      if (openOrder[i] == target) { openOrder.erase(openOrder.begin() + i); break; }
    auto giveUp = [&] { for (const Stmt *m : run.members) passThrough(*m); };
    if (run.kind == Stmt::Elem) {
      // later writes to a slot win; shapes must agree; slots 0..n-1 must all be written
      std::map<int, const Stmt *> bySlot;
      for (const Stmt *m : run.members) bySlot[m->slot] = m;
      const int n = bySlot.rbegin()->first + 1;
      const Stmt *proto = nullptr;
      bool ok = (int)bySlot.size() == n && n >= 2;
      for (auto &kv : bySlot)
        if (!kv.second->constant) {
          if (!proto) proto = kv.second;
          else ok = ok && sameShape(proto->shape, kv.second->shape);
        }
      if (!ok || !proto) { giveUp(); return; }
      const auto len = lengths.find(target);
      if (len != lengths.end() && len->second < n) { giveUp(); return; }  // writes behind the declared end: not ours to fix
      const bool partial = len != lengths.end() && len->second > n;
      std::vector<long> mask(n, 1), add(n, 0);
      bool outliers = false;
      for (auto &kv : bySlot)
        if (kv.second->constant) { mask[kv.first] = 0; add[kv.first] = kv.second->value; outliers = true; }
      if (partial) {
        // the run stops short of the vector's end: merge under a mask (one more entry than the run, so that the padding of the
        // mask lists with their last element clears / keeps everything behind slot n-1)
        const std::string tmp = "__vt" + std::to_string(tempCounter++) + "__";
        std::vector<long> take(mask), keep(n + 1, 0), addp(add);
        take.push_back(0);
        keep[n] = 1;
        for (int i = 0; i < n; ++i) keep[i] = 0;
        addp.push_back(0);
        out.push_back("secret int " + tmp + " = " + join(proto->shape) + ";");
        out.push_back(tmp + " = " + tmp + " *** " + listOf(take) + ";");
        if (outliers) out.push_back(tmp + " = " + tmp + " +++ " + listOf(addp) + ";");
        out.push_back(target + " = " + target + " *** " + listOf(keep) + ";");
        out.push_back(target + " = " + target + " +++ " + tmp + ";");
        ++elementwiseRuns;
        return;
      }
      out.push_back(target + " = " + join(proto->shape) + ";");
      if (outliers) {
        out.push_back(target + " = " + target + " *** " + listOf(mask) + ";");
        out.push_back(target + " = " + target + " +++ " + listOf(add) + ";");
      }
      ++elementwiseRuns;
    } else {
      // indices must be exactly 0..k-1, one shape
      std::set<int> idx;
      bool ok = true;
      for (const Stmt *m : run.members) {
        ok = ok && idx.insert(m->slot).second && sameShape(run.members[0]->shape, m->shape);
      }
      const int k = (int)run.members.size();
      ok = ok && k >= 2 && *idx.begin() == 0 && *idx.rbegin() == k - 1;
      if (!ok) { giveUp(); return; }
      int pow2 = 1;
      while (pow2 < k) pow2 <<= 1;
      if (rowSlots && pow2 > rowSlots) { giveUp(); return; }  // the rotate-and-add tree must stay inside one row
      const std::string tmp = "__vt" + std::to_string(tempCounter++) + "__";
      out.push_back("secret int " + tmp + " = " + join(run.members[0]->shape) + ";");
      if (pow2 != k) {  // keep slots 0..k-1, clear what the padding replicated behind them
        std::vector<long> mask(k + 1, 1);
        mask[k] = 0;
        out.push_back(tmp + " = " + tmp + " *** " + listOf(mask) + ";");
      }
      for (int s = pow2 / 2; s >= 1; s /= 2) out.push_back(tmp + " = " + tmp + " +++ rotate(" + tmp + ", " + std::to_string(s) + ");");
      out.push_back(target + " = " + target + " +++ " + tmp + ";");
      ++reductionRuns;
    }
  };
  auto flushAll = [&] {
    while (!openOrder.empty()) flush(openOrder.front());
  };

  for (const Stmt &s : stmts) {
    if (s.kind == Stmt::Other) {
      flushAll();
      passThrough(s);
      continue;
    }
    // a statement that mentions the target of another open run ends that run first (its value is about to be used / changed)
    std::vector<std::string> toFlush;
    for (const auto &name : openOrder)
      if (name != s.target && s.idents.count(name)) toFlush.push_back(name);
    // ... and so does one that writes something an open run reads
    for (const auto &name : openOrder) {
      if (name == s.target) continue;
      for (const Stmt *m : open[name].members)
        if (m->idents.count(s.target)) { toFlush.push_back(name); break; }
    }
    for (const auto &name : toFlush) flush(name);
    auto it = open.find(s.target);
    if (it != open.end() && it->second.kind != s.kind) { flush(s.target); it = open.end(); }
    if (it == open.end()) {
      open[s.target] = Run{s.kind, {}};
      openOrder.push_back(s.target);
    }
    open[s.target].members.push_back(&s);
  }
  flushAll();

  std::string text;
  for (const auto &line : out) text += line + "\n";
  return text;
}
