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
