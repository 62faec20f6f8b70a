// SealWire.cpp -- see SealWire.hpp (layout, provenance, what is and is not pinned)
#include "SealWire.hpp"

#include <dlfcn.h>
#include <zlib.h>

#include <cstring>
#include <istream>
#include <ostream>
#include <sstream>
#include <stdexcept>

namespace sealwire {

// ---- BLAKE2b (RFC 7693) ----
namespace {
const uint64_t kIv[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                         0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
const uint8_t kSigma[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
inline uint64_t rotr(uint64_t x, int r) { return (x >> r) | (x << (64 - r)); }
void compress(uint64_t h[8], const uint8_t block[128], uint64_t t, bool last) {
  uint64_t m[16], v[16];
  std::memcpy(m, block, 128);  // little-endian host (x86-64)
  for (int i = 0; i < 8; i++) {
    v[i] = h[i];
    v[8 + i] = kIv[i];
  }
  v[12] ^= t;  // the high counter word stays zero: inputs here are far below 2^64 bytes
  if (last) v[14] = ~v[14];
  auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
    v[a] = v[a] + v[b] + x;
    v[d] = rotr(v[d] ^ v[a], 32);
    v[c] = v[c] + v[d];
    v[b] = rotr(v[b] ^ v[c], 24);
    v[a] = v[a] + v[b] + y;
    v[d] = rotr(v[d] ^ v[a], 16);
    v[c] = v[c] + v[d];
    v[b] = rotr(v[b] ^ v[c], 63);
  };
  for (int r = 0; r < 12; r++) {
    const uint8_t *s = kSigma[r];
    G(0, 4, 8, 12, m[s[0]], m[s[1]]);
    G(1, 5, 9, 13, m[s[2]], m[s[3]]);
    G(2, 6, 10, 14, m[s[4]], m[s[5]]);
    G(3, 7, 11, 15, m[s[6]], m[s[7]]);
    G(0, 5, 10, 15, m[s[8]], m[s[9]]);
    G(1, 6, 11, 12, m[s[10]], m[s[11]]);
    G(2, 7, 8, 13, m[s[12]], m[s[13]]);
    G(3, 4, 9, 14, m[s[14]], m[s[15]]);
  }
  for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
}
}  // namespace

void blake2b(void *out, size_t outLen, const void *in, size_t inLen) {
  if (outLen == 0 || outLen > 64) throw std::runtime_error("blake2b: digest length must be 1..64");
  uint64_t h[8];
  for (int i = 0; i < 8; i++) h[i] = kIv[i];
  h[0] ^= 0x01010000ull ^ (uint64_t)outLen;
  const uint8_t *p = static_cast<const uint8_t *>(in);
  uint64_t t = 0;
  while (inLen > 128) {
    t += 128;
    compress(h, p, t, false);
    p += 128;
    inLen -= 128;
  }
  uint8_t block[128] = {0};
  if (inLen) std::memcpy(block, p, inLen);
  t += inLen;
  compress(h, block, t, true);
  std::memcpy(out, h, outLen);
}

ParmsId parmsId(const Parms &parms, size_t primesUsed) {
  if (primesUsed < 1 || primesUsed > parms.primes.size()) throw std::runtime_error("parmsId: prime count out of range");
  std::vector<uint64_t> words;
  words.push_back(parms.scheme);
  words.push_back(parms.ringDegree);
  for (size_t i = 0; i < primesUsed; i++) words.push_back(parms.primes[i]);
  if (parms.plainModulus) words.push_back(parms.plainModulus);  // a zero modulus occupies no words
  ParmsId id{};
  blake2b(id.data(), 32, words.data(), words.size() * 8);
  return id;
}

// ---- byte-level helpers ----
namespace {
constexpr uint16_t kMagic = 0xA15E;
constexpr size_t kHeaderBytes = 16;

template <class T>
void put(std::string &s, T v) {
  s.append(reinterpret_cast<const char *>(&v), sizeof(T));
}
void putHeader(std::string &s, uint8_t mode, uint64_t totalSize) {
  put<uint16_t>(s, kMagic);
  put<uint8_t>(s, (uint8_t)kHeaderBytes);
  put<uint8_t>(s, kVersionMajor);
  put<uint8_t>(s, kVersionMinor);
  put<uint8_t>(s, mode);
  put<uint16_t>(s, 0);
  put<uint64_t>(s, totalSize);
}
void putArray(std::string &s, const std::vector<uint64_t> &data) {
  putHeader(s, None, kHeaderBytes + 8 + 8 * (uint64_t)data.size());
  put<uint64_t>(s, data.size());
  s.append(reinterpret_cast<const char *>(data.data()), data.size() * 8);
}

struct Reader {
  const char *p;
  size_t left;
  template <class T>
  T get() {
    if (left < sizeof(T)) throw std::runtime_error("SEAL object: truncated");
    T v;
    std::memcpy(&v, p, sizeof(T));
    p += sizeof(T);
    left -= sizeof(T);
    return v;
  }
  void bytes(void *dst, size_t n) {
    if (left < n) throw std::runtime_error("SEAL object: truncated");
    std::memcpy(dst, p, n);
    p += n;
    left -= n;
  }
};
struct Header {
  uint8_t mode;
  uint64_t size;
};
Header getHeader(Reader &r) {
  if (r.get<uint16_t>() != kMagic) throw std::runtime_error("SEAL object: bad magic (not a SEAL 3.6 stream)");
  if (r.get<uint8_t>() != kHeaderBytes) throw std::runtime_error("SEAL object: unexpected header size");
  const uint8_t major = r.get<uint8_t>();
  const uint8_t minor = r.get<uint8_t>();
  if (major != kVersionMajor) throw std::runtime_error("SEAL object: written by SEAL " + std::to_string(major) + "." + std::to_string(minor) + ", this reader follows 3.x");
  Header h;
  h.mode = r.get<uint8_t>();
  (void)r.get<uint16_t>();
  h.size = r.get<uint64_t>();
  if (h.size < kHeaderBytes) throw std::runtime_error("SEAL object: size field smaller than the header");
  return h;
}
void getArray(Reader &r, std::vector<uint64_t> &data, uint64_t limitWords) {
  const Header h = getHeader(r);
  if (h.mode != None) throw std::runtime_error("SEAL object: compressed inner array");
  const uint64_t count = r.get<uint64_t>();
  if (count > limitWords) throw std::runtime_error("SEAL object: array larger than the caller's limit");
  if (h.size != kHeaderBytes + 8 + 8 * count) throw std::runtime_error("SEAL object: array size field inconsistent");
  if (r.left / 8 < count) throw std::runtime_error("SEAL object: array longer than the bytes that follow it");  // before any allocation
  data.resize(count);
  r.bytes(data.data(), count * 8);
}

// ---- compression of a whole body ----
std::string zlibDeflate(const std::string &raw) {
  uLongf cap = compressBound((uLong)raw.size());
  std::string out(cap, '\0');
  if (compress2(reinterpret_cast<Bytef *>(&out[0]), &cap, reinterpret_cast<const Bytef *>(raw.data()), (uLong)raw.size(),
                Z_DEFAULT_COMPRESSION) != Z_OK)
    throw std::runtime_error("zlib: deflate failed");
  out.resize(cap);
  return out;
}
std::string zlibInflate(const char *src, size_t n, uint64_t limitBytes) {
  z_stream zs;
  std::memset(&zs, 0, sizeof(zs));
  if (inflateInit(&zs) != Z_OK) throw std::runtime_error("zlib: inflateInit failed");
  zs.next_in = reinterpret_cast<Bytef *>(const_cast<char *>(src));
  zs.avail_in = 0;
  size_t fed = 0;  // input goes in bounded pieces: avail_in is 32 bits wide, a body may be longer
  std::string out;
  char buf[1 << 16];
  int rc = Z_OK;
  while (rc != Z_STREAM_END) {
    if (zs.avail_in == 0 && fed < n) {
      const size_t piece = (n - fed < ((size_t)1 << 30)) ? n - fed : ((size_t)1 << 30);
      zs.next_in = reinterpret_cast<Bytef *>(const_cast<char *>(src + fed));
      zs.avail_in = (uInt)piece;
      fed += piece;
    }
    zs.next_out = reinterpret_cast<Bytef *>(buf);
    zs.avail_out = sizeof(buf);
    rc = inflate(&zs, Z_NO_FLUSH);
    if (rc != Z_OK && rc != Z_STREAM_END) {
      inflateEnd(&zs);
      throw std::runtime_error("zlib: corrupt stream");
    }
    out.append(buf, sizeof(buf) - zs.avail_out);
    if (out.size() > limitBytes) {
      inflateEnd(&zs);
      throw std::runtime_error("SEAL object: inflated body larger than the caller's limit");
    }
    if (rc == Z_OK && zs.avail_in == 0 && fed == n && zs.avail_out != 0) {
      inflateEnd(&zs);
      throw std::runtime_error("zlib: truncated stream");
    }
  }
  inflateEnd(&zs);
  return out;
}
// Zstandard: the image carries the runtime library only (no header), so the four stable entry points are bound at run time
struct ZstdApi {
  size_t (*compressBound)(size_t) = nullptr;
  size_t (*compress)(void *, size_t, const void *, size_t, int) = nullptr;
  unsigned (*isError)(size_t) = nullptr;
  void *(*createDStream)() = nullptr;
  size_t (*freeDStream)(void *) = nullptr;
  size_t (*initDStream)(void *) = nullptr;
  struct Buf {
    void *ptr;
    size_t size, pos;
  };
  size_t (*decompressStream)(void *, Buf *out, Buf *in) = nullptr;
  bool ok = false;
  ZstdApi() {
    void *lib = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return;
    compressBound = reinterpret_cast<decltype(compressBound)>(dlsym(lib, "ZSTD_compressBound"));
    compress = reinterpret_cast<decltype(compress)>(dlsym(lib, "ZSTD_compress"));
    isError = reinterpret_cast<decltype(isError)>(dlsym(lib, "ZSTD_isError"));
    createDStream = reinterpret_cast<decltype(createDStream)>(dlsym(lib, "ZSTD_createDStream"));
    freeDStream = reinterpret_cast<decltype(freeDStream)>(dlsym(lib, "ZSTD_freeDStream"));
    initDStream = reinterpret_cast<decltype(initDStream)>(dlsym(lib, "ZSTD_initDStream"));
    decompressStream = reinterpret_cast<decltype(decompressStream)>(dlsym(lib, "ZSTD_decompressStream"));
    ok = compressBound && compress && isError && createDStream && freeDStream && initDStream && decompressStream;
  }
};
const ZstdApi &zstd() {
  static const ZstdApi api;
  if (!api.ok) throw std::runtime_error("SEAL object: Zstandard-compressed, and libzstd.so.1 is not available (save with compr_mode_type::none or zlib)");
  return api;
}
std::string zstdDeflate(const std::string &raw) {
  const ZstdApi &z = zstd();
  std::string out(z.compressBound(raw.size()), '\0');
  const size_t n = z.compress(&out[0], out.size(), raw.data(), raw.size(), 3);
  if (z.isError(n)) throw std::runtime_error("zstd: compression failed");
  out.resize(n);
  return out;
}
std::string zstdInflate(const char *src, size_t n, uint64_t limitBytes) {
  const ZstdApi &z = zstd();
  void *ds = z.createDStream();
  if (!ds) throw std::runtime_error("zstd: no decompression context");
  z.initDStream(ds);
  std::string out;
  char buf[1 << 16];
  ZstdApi::Buf in{const_cast<char *>(src), n, 0};
  size_t rc = 1;
  while (rc != 0) {
    ZstdApi::Buf ob{buf, sizeof(buf), 0};
    const size_t before = in.pos;
    rc = z.decompressStream(ds, &ob, &in);
    if (z.isError(rc)) {
      z.freeDStream(ds);
      throw std::runtime_error("zstd: corrupt stream");
    }
    out.append(buf, ob.pos);
    if (out.size() > limitBytes) {
      z.freeDStream(ds);
      throw std::runtime_error("SEAL object: inflated body larger than the caller's limit");
    }
    if (rc != 0 && in.pos == in.size && ob.pos == 0 && before == in.pos) {
      z.freeDStream(ds);
      throw std::runtime_error("zstd: truncated stream");
    }
  }
  z.freeDStream(ds);
  return out;
}

void emit(std::ostream &out, const std::string &body, Compression mode) {
  std::string packed;
  const std::string *payload = &body;
  if (mode == Zlib) {
    packed = zlibDeflate(body);
    payload = &packed;
  } else if (mode == Zstd) {
    packed = zstdDeflate(body);
    payload = &packed;
  } else if (mode != None) {
    throw std::runtime_error("SEAL object: unknown compression mode");
  }
  std::string head;
  putHeader(head, (uint8_t)mode, kHeaderBytes + payload->size());
  out.write(head.data(), (std::streamsize)head.size());
  out.write(payload->data(), (std::streamsize)payload->size());
  if (!out) throw std::runtime_error("SEAL object: write failed");
}
// reads one whole object from the stream and returns its (inflated) body
std::string absorb(std::istream &in, uint64_t limitWords) {
  char head[kHeaderBytes];
  in.read(head, kHeaderBytes);
  if ((size_t)in.gcount() != kHeaderBytes) throw std::runtime_error("SEAL object: truncated header");
  Reader hr{head, kHeaderBytes};
  const Header h = getHeader(hr);
  const uint64_t limitBytes = limitWords * 8 + (1u << 20);
  if (h.size - kHeaderBytes > limitBytes) throw std::runtime_error("SEAL object: larger than the caller's limit");
  std::string raw(h.size - kHeaderBytes, '\0');
  in.read(&raw[0], (std::streamsize)raw.size());
  if ((uint64_t)in.gcount() != raw.size()) throw std::runtime_error("SEAL object: truncated body");
  if (h.mode == None) return raw;
  if (h.mode == Zlib) return zlibInflate(raw.data(), raw.size(), limitBytes);
  if (h.mode == Zstd) return zstdInflate(raw.data(), raw.size(), limitBytes);
  throw std::runtime_error("SEAL object: unknown compression mode " + std::to_string(h.mode));
}

void putCiphertextMembers(std::string &s, const CiphertextImage &ct) {
  if (ct.data.size() != ct.size * ct.limbs * ct.ringDegree) throw std::runtime_error("SEAL ciphertext: data does not match size x limbs x N");
  for (uint64_t w : ct.id) put<uint64_t>(s, w);
  put<uint8_t>(s, ct.nttForm ? 1 : 0);
  put<uint64_t>(s, ct.size);
  put<uint64_t>(s, ct.ringDegree);
  put<uint64_t>(s, ct.limbs);
  put<double>(s, ct.scale);
  putArray(s, ct.data);
}
void getCiphertextMembers(Reader &r, CiphertextImage &ct, uint64_t limitWords) {
  for (auto &w : ct.id) w = r.get<uint64_t>();
  const uint8_t ntt = r.get<uint8_t>();
  if (ntt > 1) throw std::runtime_error("SEAL ciphertext: is_ntt_form is neither 0 nor 1");
  ct.nttForm = ntt != 0;
  ct.size = r.get<uint64_t>();
  ct.ringDegree = r.get<uint64_t>();
  ct.limbs = r.get<uint64_t>();
  ct.scale = r.get<double>();
  if (ct.size > 16 || ct.limbs > 64 || ct.ringDegree > (1u << 20) || (ct.ringDegree & (ct.ringDegree - 1)))
    throw std::runtime_error("SEAL ciphertext: implausible dimensions");
  getArray(r, ct.data, limitWords);
  const uint64_t want = ct.size * ct.limbs * ct.ringDegree;
  if (ct.data.size() == ct.limbs * ct.ringDegree && ct.size >= 2)
    throw std::runtime_error("SEAL ciphertext: seed-compressed object (one polynomial + PRNG seed); save the expanded object instead");
  if (ct.data.size() != want) throw std::runtime_error("SEAL ciphertext: data does not match size x limbs x N");
}
}  // namespace

void save(std::ostream &out, const CiphertextImage &ct, Compression mode) {
  std::string body;
  putCiphertextMembers(body, ct);
  emit(out, body, mode);
}
void load(std::istream &in, CiphertextImage &ct, uint64_t limitWords) {
  const std::string body = absorb(in, limitWords);
  Reader r{body.data(), body.size()};
  getCiphertextMembers(r, ct, limitWords);
  if (r.left) throw std::runtime_error("SEAL ciphertext: trailing bytes inside the object");
}

void save(std::ostream &out, const PlaintextImage &pt, Compression mode) {
  if (pt.data.size() != pt.coeffCount) throw std::runtime_error("SEAL plaintext: data does not match coeff_count");
  std::string body;
  for (uint64_t w : pt.id) put<uint64_t>(body, w);
  put<uint64_t>(body, pt.coeffCount);
  put<double>(body, pt.scale);
  putArray(body, pt.data);
  emit(out, body, mode);
}
void load(std::istream &in, PlaintextImage &pt, uint64_t limitWords) {
  const std::string body = absorb(in, limitWords);
  Reader r{body.data(), body.size()};
  for (auto &w : pt.id) w = r.get<uint64_t>();
  pt.coeffCount = r.get<uint64_t>();
  pt.scale = r.get<double>();
  getArray(r, pt.data, limitWords);
  if (pt.data.size() != pt.coeffCount) throw std::runtime_error("SEAL plaintext: data does not match coeff_count");
  if (r.left) throw std::runtime_error("SEAL plaintext: trailing bytes inside the object");
}

void save(std::ostream &out, const KSwitchImage &keys, Compression mode) {
  std::string body;
  for (uint64_t w : keys.id) put<uint64_t>(body, w);
  put<uint64_t>(body, keys.keys.size());
  for (const auto &entry : keys.keys) {
    put<uint64_t>(body, entry.size());
    for (const auto &pk : entry) {
      std::string inner;
      putCiphertextMembers(inner, pk);
      putHeader(body, None, kHeaderBytes + inner.size());
      body += inner;
    }
  }
  emit(out, body, mode);
}
void load(std::istream &in, KSwitchImage &keys, uint64_t limitWords) {
  const std::string body = absorb(in, limitWords);
  Reader r{body.data(), body.size()};
  for (auto &w : keys.id) w = r.get<uint64_t>();
  const uint64_t dim1 = r.get<uint64_t>();
  if (dim1 > (1u << 20)) throw std::runtime_error("SEAL key-switching keys: implausible entry count");
  keys.keys.assign(dim1, {});
  for (uint64_t i = 0; i < dim1; i++) {
    const uint64_t dim2 = r.get<uint64_t>();
    if (dim2 > 64) throw std::runtime_error("SEAL key-switching keys: implausible decomposition count");
    keys.keys[i].resize(dim2);
    for (uint64_t j = 0; j < dim2; j++) {
      const Header h = getHeader(r);
      if (h.mode != None) throw std::runtime_error("SEAL key-switching keys: compressed inner key");
      const size_t before = r.left;
      getCiphertextMembers(r, keys.keys[i][j], limitWords);
      if (before - r.left != h.size - kHeaderBytes) throw std::runtime_error("SEAL key-switching keys: inner size field inconsistent");
    }
  }
  if (r.left) throw std::runtime_error("SEAL key-switching keys: trailing bytes inside the object");
}

}  // namespace sealwire
