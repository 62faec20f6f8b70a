// GraphCapable -- optional extension of a ciphertext factory: record the device work of an operation sequence once and
// replay it (a recorded circuit in place of the eager per-call dispatch of SpecialRuntimeVisitor,
// src/runtime/RuntimeVisitor.cpp:40-159).  HipCiphertextFactory implements it over abc_hip_graph_*; CircuitRuntime::compile
// uses it when the factory offers it.  Not part of the reference's plugin surface: a factory without it is simply
// interpreted eagerly, as upstream does.
#pragma once

#include <cstdint>
#include <vector>

class AbstractCiphertext;

class GraphCapable {
 public:
  virtual ~GraphCapable() = default;
  virtual void graphBegin() const = 0;                 // everything issued from here on is recorded, not executed
  virtual void *graphEnd() const = 0;                  // returns the executable recording
  virtual void graphAbort() const = 0;                 // leave recording mode after an error, discarding what was recorded
  virtual void graphLaunch(void *graph) const = 0;
  virtual void graphDestroy(void *graph) const = 0;
  virtual void synchronize() const = 0;
  // new contents for an EXISTING ciphertext without changing its device address (the recording has the address baked in)
  virtual void rewriteCiphertext(AbstractCiphertext &target, const std::vector<int64_t> &values) const = 0;
  virtual void rewriteCiphertextBatch(AbstractCiphertext &target, const std::vector<std::vector<int64_t>> &perInstance) const = 0;
};
