#!/bin/bash
# TEST INFRASTRUCTURE ONLY.  The drop-in proof: the reference's OWN host (lib/ + lib/nnc/*.c, unmodified, compiled from
# where it lies under $REF with its GPU configuration macros) linked against libnnc_mi355x.so, plus the reference's OWN
# GPU integration tests (test/int/nnc/*.tests.c, compiled from where they lie) linked against that host.
#   oracle/_ref/libccv_host_gpu.so   host + product library          (runs on the MI355X box)
#   oracle/_ref/libccv_host_emu.so   host + CPU-emulator build       (runs in this container: CPU test tier)
#   oracle/_ref/int/<name>.gpu|.emu  the reference's int tests; argv[1] = filter on the test-case name (a substring; the exact name under NNC_CASE_EXACT, oracle/case_exact.c)
# The macro names (-DHAVE_CUDA ...) are the reference host's spelling of "a GPU backend is linked in"; no vendor library
# or header is involved: the host's GPU half is pure C (lib/nnc/gpu/ccv_nnc_compat.h:17-62) and every symbol it needs
# comes from ccv_amd/csrc (INTEGRATION.md lists them).  Nothing from $REF is copied into the repo.
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
[ -d "$REF/lib/nnc" ] || { echo "reference not present at $REF; keeping prebuilt files"; exit 0; }
CC=${CC:-/opt/rocm/lib/llvm/bin/clang}
OUT=$HERE/_ref
# 1. host objects (shared by both links)
OUT=$OUT/hostgpu NAME=host_objs_only.so EXTRA_LINK="-Wl,--unresolved-symbols=ignore-all" "$HERE/build_ref.sh" -DHAVE_CUDA -DHAVE_CUDNN -DHAVE_NCCL > /dev/null
rm -f $OUT/hostgpu/host_objs_only.so
OBJS="$OUT/hostgpu/obj/*.o"
LIBS="/usr/lib/x86_64-linux-gnu/libsqlite3.so.0 -lm -lrt -lpthread"
# 2. the two host libraries
$CC -shared -fopenmp -o $OUT/libccv_host_gpu.so $OBJS -L$ROOT/ccv_amd/lib -lnnc_mi355x $LIBS -Wl,-rpath,'$ORIGIN/../../ccv_amd/lib' -Wl,-rpath,/opt/rocm/lib/llvm/lib -Wl,-rpath,/opt/rocm/lib
if [ -f $ROOT/tests/emu/_build/libnnc_mi355x_emu.so ]; then
  # (a link, not a copy: the host binaries of the CPU tier must run the emulator library as it is NOW -- a copy went stale whenever only `make emu` was rerun)
  ln -sf ../../tests/emu/_build/libnnc_mi355x_emu.so $OUT/libnnc_mi355x_emu.so
  $CC -shared -fopenmp -o $OUT/libccv_host_emu.so $OBJS -L$OUT -lnnc_mi355x_emu $LIBS -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib/llvm/lib
fi
# 3. the reference's int tests
mkdir -p $OUT/int
TESTS=${TESTS:-"cudnn cublas sgd tensor schedule datatype transform loss reduce adam rmsprop gelu leaky_relu swish smooth_l1 compare index pad upsample lamb random concat cnnp.core dynamic.graph parallel nccl nms roi_align compression lstm palettize"}  # graph.vgg.d / symbolic.graph.vgg.d have no test case without libpng (their bodies are #ifdef HAVE_LIBPNG): nothing to link
TFLAGS="-O2 -fopenmp -I$REF/lib -I$REF/test -DHAVE_SSE2 -DHAVE_PTHREAD -DUSE_OPENMP -DHAVE_CUDA -DHAVE_CUDNN -DHAVE_NCCL -Wno-everything"
# exact-name case selection for the reference's runner (oracle/case_exact.c)
$CC -O2 -fPIC -c $HERE/case_exact.c -o $OUT/int/case_exact.o
EXACT="-Dstrstr=nnc_case_strstr $OUT/int/case_exact.o"
for t in $TESTS; do
  src=$REF/test/int/nnc/$t.tests.c
  [ -f $src ] || continue
  $CC $TFLAGS $EXACT $src -o $OUT/int/$t.gpu -L$OUT -lccv_host_gpu -L$ROOT/ccv_amd/lib -lnnc_mi355x $LIBS -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,'$ORIGIN/../../../ccv_amd/lib' -Wl,-rpath,/opt/rocm/lib/llvm/lib -Wl,-rpath,/opt/rocm/lib &
  if [ -f $OUT/libccv_host_emu.so ]; then
    $CC $TFLAGS $EXACT $src -o $OUT/int/$t.emu -L$OUT -lccv_host_emu -lnnc_mi355x_emu $LIBS -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,/opt/rocm/lib/llvm/lib &
  fi
done
# 3b. the VGG-D training step through the reference host (tools/host_vgg_bench.c: our client of the reference's public API;
#     bench.py --via-host runs it on the MI355X, tests/test_via_host.py the emulator build at a small size)
$CC $TFLAGS $ROOT/tools/host_vgg_bench.c -o $OUT/host_vgg_bench.gpu -L$OUT -lccv_host_gpu -L$ROOT/ccv_amd/lib -lnnc_mi355x $LIBS -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../ccv_amd/lib' -Wl,-rpath,/opt/rocm/lib/llvm/lib -Wl,-rpath,/opt/rocm/lib &
$CC $TFLAGS $ROOT/tools/host_resnet_bench.c -o $OUT/host_resnet_bench.gpu -L$OUT -lccv_host_gpu -L$ROOT/ccv_amd/lib -lnnc_mi355x $LIBS -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../ccv_amd/lib' -Wl,-rpath,/opt/rocm/lib/llvm/lib -Wl,-rpath,/opt/rocm/lib &
$CC $TFLAGS $ROOT/tools/host_lstm_check.c -o $OUT/host_lstm_check.gpu -L$OUT -lccv_host_gpu -L$ROOT/ccv_amd/lib -lnnc_mi355x $LIBS -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../ccv_amd/lib' -Wl,-rpath,/opt/rocm/lib/llvm/lib -Wl,-rpath,/opt/rocm/lib &
# the same harness on the reference's OWN CPU backends (libccv_ref.so, no GPU backend linked): the other side of the whole-network parity tests
[ -f $OUT/libccv_ref.so ] && $CC -O2 -fopenmp -I$REF/lib -DHAVE_SSE2 -DHAVE_PTHREAD -DUSE_OPENMP -Wno-everything -DHOST_BENCH_CPU $ROOT/tools/host_resnet_bench.c -o $OUT/host_resnet_bench.cpu -L$OUT -lccv_ref $LIBS -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib/llvm/lib &
# the GPU data pipeline bound into the reference's dataframe (integration/nnc_mi355x_dataframe.c: host-side glue a maintainer compiles into the host) + its test client
DFLAGS="$TFLAGS -I$ROOT/include"
$CC $DFLAGS $ROOT/tools/host_dataframe_test.c $ROOT/integration/nnc_mi355x_dataframe.c -o $OUT/host_dataframe_test.gpu -L$OUT -lccv_host_gpu -L$ROOT/ccv_amd/lib -lnnc_mi355x $LIBS -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../ccv_amd/lib' -Wl,-rpath,/opt/rocm/lib/llvm/lib -Wl,-rpath,/opt/rocm/lib &
if [ -f $OUT/libccv_host_emu.so ]; then
  $CC $DFLAGS $ROOT/tools/host_dataframe_test.c $ROOT/integration/nnc_mi355x_dataframe.c -o $OUT/host_dataframe_test.emu -L$OUT -lccv_host_emu -lnnc_mi355x_emu $LIBS -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib/llvm/lib &
  $CC $TFLAGS $ROOT/tools/host_resnet_bench.c -o $OUT/host_resnet_bench.emu -L$OUT -lccv_host_emu -lnnc_mi355x_emu $LIBS -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib/llvm/lib &
  $CC $TFLAGS $ROOT/tools/host_lstm_check.c -o $OUT/host_lstm_check.emu -L$OUT -lccv_host_emu -lnnc_mi355x_emu $LIBS -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib/llvm/lib &
  $CC $TFLAGS $ROOT/tools/host_vgg_bench.c -o $OUT/host_vgg_bench.emu -L$OUT -lccv_host_emu -lnnc_mi355x_emu $LIBS -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib/llvm/lib &
fi
wait
ls $OUT/int | tr '\n' ' '; echo
# 4. the reference's CPU unit tests (test/unit/nnc/*.tests.c) against the SAME host + backend libraries: linking the GPU
#    backend in must leave every CPU-tensor code path of the host (graphs, autograd, cnnp, CPU backends) as it was.
#    They run from $OUT/unit/run/test/unit/nnc, a scratch mirror whose data/ and samples/ point back into $REF.
mkdir -p $OUT/unit/run/test/unit/nnc/gen
ln -sfn $REF/test/unit/nnc/data $OUT/unit/run/test/unit/nnc/data
ln -sfn $REF/samples $OUT/unit/run/samples
UNIT_SKIP=${UNIT_SKIP:-"cblas"}
for src in $REF/test/unit/nnc/*.tests.c; do
  t=$(basename $src .tests.c)
  case " $UNIT_SKIP " in *" $t "*) continue;; esac
  $CC $TFLAGS $src -o $OUT/unit/$t.gpu -L$OUT -lccv_host_gpu -L$ROOT/ccv_amd/lib -lnnc_mi355x $LIBS -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,'$ORIGIN/../../../ccv_amd/lib' -Wl,-rpath,/opt/rocm/lib/llvm/lib -Wl,-rpath,/opt/rocm/lib 2> $OUT/unit/$t.gpu.log &
  if [ -f $OUT/libccv_host_emu.so ]; then
    $CC $TFLAGS $src -o $OUT/unit/$t.emu -L$OUT -lccv_host_emu -lnnc_mi355x_emu $LIBS -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,/opt/rocm/lib/llvm/lib 2> $OUT/unit/$t.emu.log &
  fi
done
wait
find $OUT/unit -name "*.log" -size 0 -delete
echo "unit tests built: $(ls $OUT/unit/*.emu 2>/dev/null | wc -l) emu, $(ls $OUT/unit/*.gpu 2>/dev/null | wc -l) gpu"
echo "built $OUT/libccv_host_gpu.so"
