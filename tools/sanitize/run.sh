#!/bin/bash
# Host sanitizers over the gate codec (see gatestream_harness.cpp).  usage: tools/sanitize/run.sh [thread|address]   (default: both)
set -e
cd "$(dirname "$0")/../.."
out=/tmp/fk_sanitize; mkdir -p $out
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
for san in ${1:-thread address}; do
  flags="-fsanitize=$san"; [ $san = address ] && flags="-fsanitize=address,undefined -fno-sanitize-recover=undefined"
  $HIPCC -O1 -g -std=c++17 --offload-arch=gfx950 --cuda-host-only $flags -Wno-option-ignored -c fawkes-crypto_amd/csrc/gatestream.hip -o $out/gatestream_$san.o
  $CXX -O1 -g -std=c++17 $flags -c tools/sanitize/gatestream_harness.cpp -o $out/harness_$san.o
  $CXX $flags $out/harness_$san.o $out/gatestream_$san.o -o $out/harness_$san -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 -ldl -lpthread
  for t in 1 6; do
    echo "== $san, FK_HOST_THREADS=$t"
    TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1" ASAN_OPTIONS="detect_leaks=1" $out/harness_$san $t
  done
done
echo "sanitizers: clean"
