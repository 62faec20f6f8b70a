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


// =====================================================================================================================
// ExpressionBatcher
// =====================================================================================================================
namespace {

// expression tree over leaves `name`, `name[int]`, integer literals; operators + - * (also the runtime's +++ --- ***)
struct Expr {
  enum Kind { Leaf, Lit, Bin } kind = Leaf;
  std::string name;   // Leaf: identifier; Bin: operator as written
  int index = -1;     // Leaf: element index, -1 for a scalar
  long value = 0;     // Lit
  std::vector<Expr> kids;
};

struct ExprParser {
  const std::vector<Tok> &t;
  size_t i = 0;
  bool ok = true;
  explicit ExprParser(const std::vector<Tok> &toks) : t(toks) {}
  bool at(const char *p) const { return i < t.size() && isP(t[i], p); }
  Expr primary() {
    Expr e;
    if (i >= t.size()) { ok = false; return e; }
    if (at("(")) {
      ++i;
      e = sum();
      if (!at(")")) ok = false; else ++i;
      return e;
    }
    if (t[i].kind == Tok::Int) { e.kind = Expr::Lit; e.value = std::stol(t[i].text); ++i; return e; }
    if (t[i].kind == Tok::Ident) {
      e.kind = Expr::Leaf; e.name = t[i].text; ++i;
      if (at("[")) {
        if (i + 2 >= t.size() || t[i + 1].kind != Tok::Int || !isP(t[i + 2], "]")) { ok = false; return e; }
        e.index = std::stoi(t[i + 1].text);
        i += 3;
      }
      if (at("(")) ok = false;  // calls are not ours
      return e;
    }
    ok = false;
    return e;
  }
  Expr product() {
    Expr l = primary();
    while (ok && (at("*") || at("***"))) {
      Expr b; b.kind = Expr::Bin; b.name = "*"; ++i;
      b.kids.push_back(l); b.kids.push_back(primary());
      l = b;
    }
    return l;
  }
  Expr sum() {
    Expr l = product();
    while (ok && (at("+") || at("+++") || at("-") || at("---"))) {
      Expr b; b.kind = Expr::Bin; b.name = (at("+") || at("+++")) ? "+" : "-"; ++i;
      b.kids.push_back(l); b.kids.push_back(product());
      l = b;
    }
    return l;
  }
};

bool sameTreeShape(const Expr &a, const Expr &b) {  // operators and literals equal, leaves free
  if (a.kind != b.kind) return false;
  if (a.kind == Expr::Lit) return a.value == b.value;
  if (a.kind == Expr::Leaf) return true;
  if (a.name != b.name || a.kids.size() != b.kids.size()) return false;
  for (size_t k = 0; k < a.kids.size(); ++k)
    if (!sameTreeShape(a.kids[k], b.kids[k])) return false;
  return true;
}
void leavesOf(const Expr &e, std::vector<const Expr *> &out) {
  if (e.kind == Expr::Leaf) out.push_back(&e);
  for (const auto &k : e.kids) leavesOf(k, out);
}
void flattenSum(const Expr &e, std::vector<const Expr *> &terms, bool &plusOnly) {
  if (e.kind == Expr::Bin && (e.name == "+" || e.name == "-")) {
    if (e.name == "-") plusOnly = false;
    flattenSum(e.kids[0], terms, plusOnly);
    flattenSum(e.kids[1], terms, plusOnly);
  } else {
    terms.push_back(&e);
  }
}
// the tree with its leaves replaced, in order, by the given names
std::string render(const Expr &e, const std::vector<std::string> &names, size_t &next, bool top = true) {
  if (e.kind == Expr::Lit) return std::to_string(e.value);
  if (e.kind == Expr::Leaf) return names[next++];
  const std::string l = render(e.kids[0], names, next, false), r = render(e.kids[1], names, next, false);
  const std::string body = l + " " + e.name + " " + r;
  return top ? body : "(" + body + ")";
}
std::string leafText(const Expr &l) { return l.index < 0 ? l.name : l.name + "[" + std::to_string(l.index) + "]"; }

struct Assign {
  std::string target;
  int slot = -1;  // x[slot] = ..., -1: x = ...
  Expr rhs;
};
bool parseAssign(const std::vector<Tok> &t, Assign &a) {
  if (t.size() < 3 || t[0].kind != Tok::Ident) return false;
  size_t eq = 1;
  a.target = t[0].text;
  if (isP(t[1], "[")) {
    if (t.size() < 6 || t[2].kind != Tok::Int || !isP(t[3], "]")) return false;
    a.slot = std::stoi(t[2].text);
    eq = 4;
  }
  if (!isP(t[eq], "=")) return false;
  const std::vector<Tok> rhs(t.begin() + eq + 1, t.end());
  ExprParser p(rhs);
  a.rhs = p.sum();
  return p.ok && p.i == rhs.size();
}

}  // namespace

ExpressionBatcher::Result ExpressionBatcher::batch(const std::string &program) {
  Result res;
  // top-level statements (no blocks: anything with braces is not ours)
  std::vector<std::vector<Tok>> stmts(1);
  for (const Tok &t : lex(program)) {
    if (isP(t, "{") || isP(t, "}")) return res;
    if (isP(t, ";")) { if (!stmts.back().empty()) stmts.emplace_back(); continue; }
    stmts.back().push_back(t);
  }
  if (stmts.back().empty()) stmts.pop_back();
  std::vector<Assign> as(stmts.size());
  for (size_t k = 0; k < stmts.size(); ++k)
    if (!parseAssign(stmts[k], as[k])) return res;
  if (as.empty()) return res;
  auto tmp = [&] { return "__eb" + std::to_string(tempCounter++) + "__"; };
  auto listOfL = [](const std::vector<long> &v) { return listOf(v); };

  // ---- 1. matrix-vector product ----
  {
    const int R = (int)as.size();
    bool ok = R >= 2;
    int C = 0;
    std::string A, B;
    for (int k = 0; ok && k < R; ++k) {
      ok = as[k].target == as[0].target && as[k].slot == k;
      std::vector<const Expr *> terms;
      bool plusOnly = true;
      if (ok) flattenSum(as[k].rhs, terms, plusOnly);
      ok = ok && plusOnly && (k == 0 ? terms.size() >= 2 : (int)terms.size() == C);
      if (k == 0) C = (int)terms.size();
      for (int j = 0; ok && j < C; ++j) {
        const Expr &t = *terms[j];
        ok = t.kind == Expr::Bin && t.name == "*" && t.kids[0].kind == Expr::Leaf && t.kids[1].kind == Expr::Leaf &&
             t.kids[0].index >= 0 && t.kids[1].index >= 0;
        if (!ok) break;
        const Expr *a = &t.kids[0], *b = &t.kids[1];
        if (!(a->index == k * C + j && b->index == j)) std::swap(a, b);  // either operand order
        ok = a->index == k * C + j && b->index == j;
        if (k == 0 && j == 0) { A = a->name; B = b->name; }
        ok = ok && a->name == A && b->name == B && A != B && A != as[0].target && B != as[0].target;
      }
    }
    if (ok) {
      const std::string c = as[0].target, bm = tmp(), bb = tmp(), t = tmp(), s = tmp();
      std::vector<long> maskC(C + 1, 1);
      maskC[C] = 0;
      std::string p;
      p += "secret int " + bm + " = " + B + " *** " + listOfL(maskC) + ";\n";  // the runtime pads with the last value: clear it
      p += "secret int " + bb + " = " + bm + ";\n";
      for (int r = 1; r < R; ++r) p += bb + " = " + bb + " +++ rotate(" + bm + ", " + std::to_string(-r * C) + ");\n";
      p += "secret int " + t + " = " + A + " *** " + bb + ";\n";  // slot kC + j = a[kC + j] b[j]
      p += "secret int " + s + " = " + t + ";\n";
      for (int d = 1; d < C; ++d) p += s + " = " + s + " +++ rotate(" + t + ", " + std::to_string(d) + ");\n";  // slot kC = c_k
      for (int k = 0; k < R; ++k) {  // compaction: slot kC -> slot k
        std::vector<long> unit(k * C + 2, 0);
        unit[k * C] = 1;
        const std::string u = tmp();
        p += "secret int " + u + " = " + s + " *** " + listOfL(unit) + ";\n";
        if (k == 0) p += c + " = " + u + ";\n";
        else p += c + " = " + c + " +++ rotate(" + u + ", " + std::to_string(k * (C - 1)) + ");\n";
      }
      res.program = p;
      res.aux = c + " = {" + c + "[0], .., " + c + "[" + std::to_string(R - 1) + "]};\n";
      res.batched = true;
      res.rule = "matrix-vector";
      return res;
    }
  }

  // ---- 2. same-shaped statements over scalar leaves: one input vector per leaf position ----
  if (as.size() >= 2) {
    bool ok = true;
    std::vector<std::vector<const Expr *>> leaves(as.size());
    for (size_t k = 0; ok && k < as.size(); ++k) {
      ok = as[k].target == as[0].target && as[k].slot == (int)k && sameTreeShape(as[0].rhs, as[k].rhs);
      if (ok) leavesOf(as[k].rhs, leaves[k]);
      ok = ok && !leaves[k].empty() && leaves[k].size() == leaves[0].size();
      for (const Expr *l : leaves[k]) ok = ok && l->index < 0;  // scalars only: indexed leaves are the CircuitVectorizer's patterns
    }
    if (ok) {
      std::vector<std::string> names;
      for (size_t pos = 0; pos < leaves[0].size(); ++pos) {
        const std::string in = "__input" + std::to_string(pos) + "__";
        names.push_back(in);
        res.aux += in + " = {";
        for (size_t k = 0; k < as.size(); ++k) res.aux += (k ? ", " : "") + leafText(*leaves[k][pos]);
        res.aux += "};\n";
      }
      size_t next = 0;
      res.program = as[0].target + " = " + render(as[0].rhs, names, next) + ";\n";
      res.batched = true;
      res.rule = "statements";
      return res;
    }
  }

  // ---- 3. one expression = a sum of same-shaped terms ----
  if (as.size() == 1 && as[0].slot < 0) {
    std::vector<const Expr *> terms;
    bool plusOnly = true;
    flattenSum(as[0].rhs, terms, plusOnly);
    bool ok = plusOnly && terms.size() >= 2;
    std::vector<std::vector<const Expr *>> leaves(terms.size());
    for (size_t j = 0; ok && j < terms.size(); ++j) {
      ok = sameTreeShape(*terms[0], *terms[j]);
      if (ok) leavesOf(*terms[j], leaves[j]);
      ok = ok && !leaves[j].empty() && leaves[j].size() == leaves[0].size();
      for (const Expr *l : leaves[j]) ok = ok && l->index < 0;
    }
    if (ok) {
      const int k = (int)terms.size();
      std::vector<std::string> names;
      for (size_t pos = 0; pos < leaves[0].size(); ++pos) {
        const std::string in = "__input" + std::to_string(pos) + "__";
        names.push_back(in);
        res.aux += in + " = {";
        for (int j = 0; j < k; ++j) res.aux += (j ? ", " : "") + leafText(*leaves[j][pos]);
        res.aux += "};\n";
      }
      res.aux += as[0].target + " = __input0__[0];\n";
      size_t next = 0;
      std::string p = "__input0__ = " + render(*terms[0], names, next) + ";\n";
      int pow2 = 1;
      while (pow2 < k) pow2 <<= 1;
      if (pow2 != k) {  // clear what the padding replicated behind the k terms
        std::vector<long> mask(k + 1, 1);
        mask[k] = 0;
        p += "__input0__ = __input0__ *** " + listOfL(mask) + ";\n";
      }
      for (int sft = pow2 / 2; sft >= 1; sft /= 2) p += "__input0__ = __input0__ + rotate(__input0__, " + std::to_string(sft) + ");\n";
      res.program = p;
      res.batched = true;
      res.rule = "sum-of-terms";
      return res;
    }
  }
  return res;
}
