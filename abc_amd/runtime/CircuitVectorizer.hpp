// CircuitVectorizer -- the batching front end the reference only sketches: a source-to-source pass over the runtime's input
// language that turns slot-wise scalar code into the batched circuit the ciphertext plugin surface can execute.
//
// Upstream: SpecialVectorizer (src/visitor/Vectorizer.cpp:17-84) records assignments as "ComplexValue"s and deletes them --
// the emit step is a TODO (:28-35) -- and ExpressionBatcher (src/visitor/ExpressionBatcher.cpp:141-221,272-347) stops at
// computing batchability; every test of both is DISABLED (test/visitor/VectorizerTest.cpp:7-225).  Those tests state what
// the pass is meant to produce, and this class produces it:
//   * runs of  x[i] = E(i);  with one expression shape over the same slot index  ->  x = E;      (trivialVectors, :7-38)
//     interleaved runs on different targets are separated                                       (trivialInterleaved, :64-94)
//     slots whose value is a literal instead become a mask multiply + constant add              (singleOutlierVector, :96-124)
//   * runs of  s = s + E(i);  i = 0..k-1  ->  a rotate-and-add tree over the batched operand    (sumStatements*, :140-214)
//     (k not a power of two: the operand is first masked to k slots, the tree runs over the next power of two)
// Two deliberate differences from the (unfinished, mutually inconsistent) expected texts upstream:
//   the old value of the accumulator is kept (`sum = sum + t` at the end; upstream's text drops it), and masks have one
//   entry per slot (upstream's sample has one too few).  The reduced value sits in slot 0 of the accumulator.
// A secret vector cannot be indexed at run time (RuntimeVisitor.cpp:268-298 throws, pointing at the Vectorizer), so the
// output of this pass is what makes such programs executable at all: `CircuitRuntime::executeAst(vectorize(program))`.
// An element-wise run is taken to cover its target: x[0..n-1] = E(i) becomes x = E, which also rewrites whatever x held behind
// slot n-1.  Where the declared length of x is known (vectorLengths) and LONGER than the run, the rewrite is a masked merge
// instead -- x = x *** {0,..,0,1} +++ E *** {1,..,1,0} -- so a partial update leaves the other slots alone.  A reduction over k
// terms rotates within one row of the ciphertext: with slotsPerRow set (BFV: N/2), runs whose tree would not fit are passed
// through unchanged.
#pragma once

#include <map>
#include <set>
#include <string>
#include <vector>

class CircuitVectorizer {
 public:
  // identifiers that name scalars (accumulators): `s = s + x[i]` is a reduction only if s is one of them
  explicit CircuitVectorizer(std::set<std::string> scalarAccumulators = {}, std::map<std::string, int> vectorLengths = {},
                             int slotsPerRow = 0)
      : scalars(std::move(scalarAccumulators)), lengths(std::move(vectorLengths)), rowSlots(slotsPerRow) {}
  // returns the transformed program; statements that match no pattern are passed through unchanged, in order
  std::string vectorize(const std::string &program);
  // number of runs rewritten by the last call (tests)
  int elementwiseRuns = 0, reductionRuns = 0;

 private:
  std::set<std::string> scalars;
  std::map<std::string, int> lengths;  // declared length of a vector (optional)
  int rowSlots = 0;                    // 0: unknown
  int tempCounter = 0;
};

// ExpressionBatcher -- batching of expression TREES, the part of the front end that ref:src/visitor/ExpressionBatcher.cpp:141-221,
// 272-347 stops short of (it computes batchability and never emits) and whose expectations are the DISABLED cases
// ref:test/visitor/ExpressionBatcherTest.cpp:8-41 and ref:test/visitor/VectorizerTest.cpp:370-526.  Three rewrites, tried in order:
//   1. matrix-vector product (VectorizerTest.cpp:370-433):  c[k] = a[kC] b[0] + ... + a[kC + C-1] b[C-1],  k = 0..R-1  (a row
//      major, "merged" indices as upstream's comment has them)  ->  b masked to C slots and replicated R times, ONE slot-wise
//      product, an in-group rotate-and-add, and a masked compaction that leaves c = {c_0, .., c_{R-1}}.  Upstream's expected
//      text is the first of these steps only, written as R products (c = a*b; c = c + a*rotate(b,-3); ...): it yields the R C
//      products, never sums them, and assumes b is zero behind its data whereas the runtime pads with the LAST value.
//   2. same-shaped statements (batchableExpressionVectorizable, :484-526):  x[k] = E_k  with one tree shape over scalar leaves
//      ->  one input vector per leaf position ({a,e,i,m}, {b,f,j,n}, ...) and ONE slot-wise evaluation x = E(inputs).  (Upstream's
//      "ideal" text packs all sixteen leaves into one ciphertext and multiplies it with rotations of itself by 4 and 2 -- which
//      does not pair a with b; its own comment says the pass would not produce it.)
//   3. one expression that is a sum of same-shaped terms (batchableExpression, ExpressionBatcherTest.cpp:8-41 = VectorizerTest.cpp:
//      434-480):  x = (a*b) + (c*d)  ->  inputs {a,c}, {b,d}; __input0__ = __input0__ * __input1__; a rotate-and-add tree; the
//      result in slot 0 -- upstream's expected text, token for token.
// `aux` is what upstream calls the auxiliary information: how the CALLER must pack the scalar inputs (`__inputN__ = {a, c};`)
// and where results come out (`x = __input0__[0];`); `program` is the batched circuit in the runtime's input language
// (rotate() takes a variable, so sub-expressions that are rotated get named temporaries).
class ExpressionBatcher {
 public:
  struct Result {
    std::string aux, program;
    bool batched = false;
    std::string rule;  // "matrix-vector", "statements", "sum-of-terms"
  };
  Result batch(const std::string &program);

 private:
  int tempCounter = 0;
};
