#!/bin/bash
# rebuild only tools/host_resnet_bench.c's three binaries (oracle/build_ref_host.sh builds everything; this is the quick path while iterating on the harness)
set -e
REF=${REF:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/oracle/_ref
CC=/opt/rocm/lib/llvm/bin/clang
LIBS="/usr/lib/x86_64-linux-gnu/libsqlite3.so.0 -lm -lrt -lpthread"
TFLAGS="-O2 -fopenmp -I$REF/lib -I$REF/test -DHAVE_SSE2 -DHAVE_PTHREAD -DUSE_OPENMP -DHAVE_CUDA -DHAVE_CUDNN -DHAVE_NCCL -Wno-everything"
$CC $TFLAGS $ROOT/tools/host_resnet_bench.c -o $OUT/host_resnet_bench.gpu -L$OUT -lccv_host_gpu -L$ROOT/ccv_amd/lib -lnnc_mi355x $LIBS -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../ccv_amd/lib' -Wl,-rpath,/opt/rocm/lib/llvm/lib -Wl,-rpath,/opt/rocm/lib &
$CC -O2 -fopenmp -I$REF/lib -DHAVE_SSE2 -DHAVE_PTHREAD -DUSE_OPENMP -Wno-everything -DHOST_BENCH_CPU $ROOT/tools/host_resnet_bench.c -o $OUT/host_resnet_bench.cpu -L$OUT -lccv_ref $LIBS -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib/llvm/lib &
$CC $TFLAGS $ROOT/tools/host_resnet_bench.c -o $OUT/host_resnet_bench.emu -L$OUT -lccv_host_emu -lnnc_mi355x_emu $LIBS -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib/llvm/lib &
wait
