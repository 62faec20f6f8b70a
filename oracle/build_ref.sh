#!/bin/sh
# oracle/build_ref.sh -- compiles the reference's OWN runtime (RuntimeVisitor + DummyCiphertextFactory +
# parser + AST, no SEAL: those two translation units are empty without HAVE_SEAL_BFV) from the sources
# where they lie under /root/reference, with plain g++ (the reference's CMake build is not run).
# Outputs ONLY under oracle/_ref/ (git-ignored, travels to the GPU box like the other built binaries).
# nlohmann/json: the image ships the single header at /opt/conda/include/json.hpp; an include directory
# with a symlink named nlohmann/json.hpp points at it (no stand-in is written).
# Test infrastructure only: used to produce tests/golden/ref_dummy_runtime.txt (config 1).
set -e
REF=/root/reference
HERE=$(cd "$(dirname "$0")" && pwd)
OUT="$HERE/_ref"
[ -d "$REF/src" ] || { echo "reference tree absent"; exit 0; }
[ -f /opt/conda/include/json.hpp ] || { echo "nlohmann json header absent: reference runtime unbuildable here"; exit 0; }
mkdir -p "$OUT/obj" "$OUT/inc/nlohmann"
ln -sf /opt/conda/include/json.hpp "$OUT/inc/nlohmann/json.hpp"
CXXFLAGS="-std=c++17 -O1 -w -I$REF/include -I$REF -I$OUT/inc"
find "$REF/src" -name '*.cpp' | sort > "$OUT/sources.txt"
# compile in parallel, one object per source (names flattened)
cat "$OUT/sources.txt" | xargs -P 8 -I{} sh -c 'o="'"$OUT"'/obj/$(echo {} | sed "s#/#_#g").o"; [ -f "$o" ] || g++ '"$CXXFLAGS"' -c {} -o "$o"'
ar rcs "$OUT/libabc_ref.a" "$OUT"/obj/*.o
g++ $CXXFLAGS "$HERE/ref_dummy_driver.cpp" "$OUT/libabc_ref.a" -o "$OUT/ref_dummy_driver"
echo "built $OUT/ref_dummy_driver"
# drop-in proof: the reference's RuntimeVisitor over this repo's HipCiphertextFactory, through ABC's real headers
RT="$HERE/../abc_amd/runtime"
if [ -f "$HERE/../abc_amd/libabc_hip.so" ]; then
  g++ $CXXFLAGS -DABC_HIP_USE_REFERENCE_HEADERS -I"$RT" "$HERE/ref_hip_dropin.cpp" "$RT/HipCiphertext.cpp" \
      "$RT/HipCiphertextFactory.cpp" "$RT/SealWire.cpp" "$OUT/libabc_ref.a" -L"$HERE/../abc_amd" -labc_hip -lz -ldl \
      -Wl,-rpath,'$ORIGIN/../../abc_amd' -o "$OUT/ref_hip_dropin"
  echo "built $OUT/ref_hip_dropin"
fi
