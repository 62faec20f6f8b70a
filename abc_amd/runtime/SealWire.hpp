// SealWire -- reader / writer for the byte format of SEAL 3.6 `save` / `load` (ciphertexts, secret / public / relinearisation /
// Galois keys), so that a client can keep key generation and encryption in SEAL and hand the evaluation to this backend, and so
// that ciphertexts of the two backends can be compared bit for bit wherever a SEAL build exists.
//
// The reference never serialises anything (it calls none of SEAL's save / load; its Dockerfile pins SEAL 3.6.5:
// Docker/Dockerfile:9, CMakeLists.txt:57), and SEAL itself is not in this image: PARITY UNPINNED.  The layout below restates
// SEAL 3.6's published serialisation (native/src/seal/serialization.h: SEALHeader; ciphertext.cpp / plaintext.cpp /
// kswitchkeys.cpp: save_members; dynarray.h; encryptionparams.cpp: compute_parms_id; util/hash.h: BLAKE2b-256):
//   SEALHeader (16 bytes): u16 magic 0xA15E | u8 header_size 0x10 | u8 version_major | u8 version_minor | u8 compr_mode |
//                          u16 reserved 0 | u64 size (whole object, header included).  compr_mode 0 none, 1 zlib, 2 zstd; the
//                          body behind the header is compressed as one stream.
//   DynArray<u64>        : SEALHeader(none) | u64 count | count words
//   Ciphertext           : parms_id (4 x u64) | u8 is_ntt_form | u64 size | u64 poly_modulus_degree | u64 coeff_modulus_size |
//                          f64 scale | DynArray data [size][coeff_modulus_size][N]
//   Plaintext (SecretKey): parms_id | u64 coeff_count | f64 scale | DynArray data
//   PublicKey            : a Ciphertext (size 2, key level, NTT form)
//   KSwitchKeys          : parms_id | u64 dim1 | per entry: u64 dim2 | dim2 x PublicKey, each with its own SEALHeader(none)
//                          RelinKeys: dim1 = 1; GaloisKeys: dim1 = N, entry (galois_elt - 1) / 2
//   parms_id             : BLAKE2b-256 over the u64 words { scheme, N, q_0 .. q_{k-1}, t (only if non-zero) }
// What is tested here: BLAKE2b against RFC 7693 / Python's hashlib, the header bytes, round trips through every object type
// and compression mode, and key transport between two contexts on the device (tests/cpp/test_seal_wire.cpp,
// tests/test_seal_wire.py).  Seed-compressed objects (SEAL's Serializable<> results, which carry one polynomial and a PRNG
// seed) are refused: expanding them needs SEAL's generator.
#pragma once

#include <array>
#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>

namespace sealwire {

typedef std::array<uint64_t, 4> ParmsId;

struct Parms {
  uint8_t scheme = 1;            // SEAL scheme_type: 1 BFV, 2 CKKS
  uint64_t ringDegree = 0;       // N
  std::vector<uint64_t> primes;  // key level: data primes, then the special prime
  uint64_t plainModulus = 0;     // BFV t; 0 for CKKS
};
// identifier of the parameter set that keeps the first `primesUsed` primes (key level: all of them; top data level: all but
// the special prime, unless there is only one prime)
ParmsId parmsId(const Parms &parms, size_t primesUsed);

enum Compression : uint8_t { None = 0, Zlib = 1, Zstd = 2 };

struct CiphertextImage {
  ParmsId id{};
  bool nttForm = false;
  uint64_t size = 0, ringDegree = 0, limbs = 0;
  double scale = 1.0;
  std::vector<uint64_t> data;  // [size][limbs][N]
};
struct PlaintextImage {
  ParmsId id{};
  uint64_t coeffCount = 0;
  double scale = 1.0;
  std::vector<uint64_t> data;
};
struct KSwitchImage {
  ParmsId id{};
  std::vector<std::vector<CiphertextImage>> keys;  // [dim1][dim2]
};

// all of these throw std::runtime_error (bad magic, unsupported version or compression, truncated stream, inconsistent sizes,
// seed-compressed object); `limitWords` bounds what a load may allocate (default 2^27 words = 1 GiB: a Galois key set of
// N = 2^15 with 15 limbs is 0.5 GiB; pass more for larger objects)
void save(std::ostream &out, const CiphertextImage &ct, Compression mode = None);
void save(std::ostream &out, const PlaintextImage &pt, Compression mode = None);
void save(std::ostream &out, const KSwitchImage &keys, Compression mode = None);
void load(std::istream &in, CiphertextImage &ct, uint64_t limitWords = (uint64_t)1 << 27);
void load(std::istream &in, PlaintextImage &pt, uint64_t limitWords = (uint64_t)1 << 27);
void load(std::istream &in, KSwitchImage &keys, uint64_t limitWords = (uint64_t)1 << 27);

// BLAKE2b (RFC 7693), unkeyed, outLen <= 64
void blake2b(void *out, size_t outLen, const void *in, size_t inLen);

constexpr uint8_t kVersionMajor = 3, kVersionMinor = 6;

}  // namespace sealwire
