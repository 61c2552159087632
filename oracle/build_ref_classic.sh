#!/bin/bash
# TEST INFRASTRUCTURE ONLY.  The reference's classic image functions (ccv_resample, ccv_filter and what they need), compiled
# from where they lie WITHOUT OpenMP into oracle/_ref/libccv_classic.so: the OpenMP build of the classic half crashes in
# ccv_filter in this toolchain, and the pre-process oracle needs nothing from nnc.  -Bsymbolic: its internal calls must not bind
# to the same-named functions of libccv_ref.so when both live in one test process.  Same rules as build_ref.sh: nothing copied.
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
[ -d "$REF/lib" ] || { echo "reference not present; keeping prebuilt"; exit 0; }
L=$REF/lib
CC=${CC:-/opt/rocm/lib/llvm/bin/clang}
mkdir -p $HERE/_ref
$CC -w -O2 -fPIC -shared -I$L -DHAVE_SSE2 -DHAVE_PTHREAD \
  $L/ccv_numeric.c $L/3rdparty/kissfft/*.c $L/ccv_memory.c $L/ccv_cache.c $L/ccv_util.c $L/ccv_algebra.c $L/3rdparty/siphash/siphash24.c \
  $L/ccv_basic.c $L/ccv_image_processing.c $L/ccv_resample.c $L/ccv_io.c $L/3rdparty/dsfmt/dSFMT.c $L/3rdparty/sfmt/SFMT.c $L/ccv_transform.c \
  -Wl,-Bsymbolic -o $HERE/_ref/libccv_classic.so -lm -lpthread
echo "built $HERE/_ref/libccv_classic.so"
