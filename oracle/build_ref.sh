#!/bin/bash
# TEST INFRASTRUCTURE ONLY (see oracle/README.md).
# Compiles the reference's own CPU implementation (lib/ + lib/nnc CPU_REF / CPU_OPT
# backends) from the sources WHERE THEY LIE under $REF (default /root/reference) into
# oracle/_ref/libccv_ref.so.  No reference source is copied into this repo; only the
# built objects / .so land in oracle/_ref/ (git-ignored, but shipped to the GPU box).
# This replaces the reference's configure+make (needs ruby/autoconf state, a vendored
# sqlite3.c that is absent from the mount, and gcc-incompatible OpenMP loops):
#   * clang (ROCm's LLVM) because lib/ccv_internal.h:30 parallel_for needs clang's OpenMP
#   * system libsqlite3.so.0 instead of lib/3rdparty/sqlite3/sqlite3.c
#   * -DCblas* constants: cmd/blas/cpu_sys/_ccv_nnc_gemm_cpu_sys.c references them in
#     unused static-inline helpers even without HAVE_CBLAS.
# Usage: oracle/build_ref.sh [extra -D flags, e.g. -DHAVE_CUDA -DHAVE_CUDNN -DHAVE_NCCL] ; OUT=<dir> overrides output dir
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=${OUT:-$HERE/_ref}
NAME=${NAME:-libccv_ref.so}
CC=${CC:-/opt/rocm/lib/llvm/bin/clang}
[ -d "$REF/lib/nnc" ] || { echo "reference not present at $REF; keeping prebuilt $OUT"; exit 0; }
mkdir -p "$OUT/obj"
L=$REF/lib
COMMON="-O3 -fPIC -fopenmp -I$L -DHAVE_SSE2 -DHAVE_PTHREAD -DUSE_OPENMP -DCblasColMajor=102 -DCblasNoTrans=111 -DCblasTrans=112 -Wno-implicit-function-declaration -Wno-everything $*"
CORE="ccv_cache.c ccv_memory.c 3rdparty/siphash/siphash24.c 3rdparty/kissfft/kiss_fft.c 3rdparty/kissfft/kiss_fftnd.c 3rdparty/kissfft/kiss_fftr.c 3rdparty/kissfft/kiss_fftndr.c 3rdparty/kissfft/kissf_fft.c 3rdparty/kissfft/kissf_fftnd.c 3rdparty/kissfft/kissf_fftr.c 3rdparty/kissfft/kissf_fftndr.c 3rdparty/dsfmt/dSFMT.c 3rdparty/sfmt/SFMT.c ccv_io.c ccv_numeric.c ccv_algebra.c ccv_util.c ccv_basic.c ccv_image_processing.c ccv_resample.c ccv_transform.c ccv_classic.c ccv_daisy.c ccv_sift.c ccv_bbf.c ccv_mser.c ccv_swt.c ccv_dpm.c ccv_tld.c ccv_ferns.c ccv_icf.c ccv_scd.c ccv_convnet.c ccv_output.c"
NNC=$(cd $L/nnc && ls *.c)
CMD=$(cd $L/nnc/cmd && find . -name '*.c' -not -path '*/gpu/*' -not -path '*/mps/*' | sed 's|^\./||')
jobs=()
for f in $CORE; do echo "$L/$f|-ffast-math"; done > "$OUT/obj/list.txt"
for f in $NNC; do echo "$L/nnc/$f|"; done >> "$OUT/obj/list.txt"
for f in $CMD; do echo "$L/nnc/cmd/$f|-I$L/nnc/cmd"; done >> "$OUT/obj/list.txt"
compile_one() {
  src=${1%%|*}; extra=${1##*|}
  o="$OUT/obj/$(echo "${src#$L/}" | tr '/' '_' | sed 's/\.c$/.o/')"
  if [ ! -f "$o" ] || [ "$src" -nt "$o" ]; then $CC $COMMON $extra -c "$src" -o "$o" || exit 255; fi
}
export -f compile_one; export CC COMMON OUT L
cat "$OUT/obj/list.txt" | xargs -P ${JOBS:-8} -I{} bash -c 'compile_one "{}"'
$CC -shared -fopenmp -o "$OUT/$NAME" "$OUT"/obj/*.o ${EXTRA_LINK} /usr/lib/x86_64-linux-gnu/libsqlite3.so.0 -lm -lrt -lpthread -Wl,-rpath,/opt/rocm/lib/llvm/lib
echo "built $OUT/$NAME"
