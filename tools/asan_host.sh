#!/bin/bash
# Host side of libadvx_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: the GPU pool offers no
# device ASan).  Builds the library with the HOST code instrumented (-fno-gpu-sanitize leaves the gfx950 code alone), puts
# it where the package loads it from, runs the host-logic tests of the CPU tier and a random-geometry fuzz of the plan
# builder (tools/fuzz_host_geometry.py), and restores the product build.
#   tools/asan_host.sh [fuzz cases]      -> /tmp/asan_host.log (copy the tail to profiles/<round>/)
set -e -o pipefail
cd "$(dirname "$0")/.."
N=${1:-3000}
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
LIB=adversarialvlm_amd/libadvx_hip.so
/opt/rocm/bin/hipcc -O1 -g --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 \
    -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -o /tmp/libadvx_asan.so adversarialvlm_amd/csrc/advx.hip
cp $LIB /tmp/libadvx_hip.real.so
trap 'cp /tmp/libadvx_hip.real.so '$LIB EXIT
cp /tmp/libadvx_asan.so $LIB
export LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
{
  python -m pytest tests/test_cabi_host.py tests/test_index_tensors.py tests/test_host_logic.py tests/test_probe_encode.py -q \
      -m "not gpu" --deselect tests/test_cabi_host.py::test_plain_c_host_compiles_and_links_against_the_header
  python tools/fuzz_host_geometry.py $N
} 2>&1 | tee /tmp/asan_host.log
if grep -q "runtime error\|AddressSanitizer" /tmp/asan_host.log; then echo "SANITIZER FINDINGS"; exit 1; fi
echo "asan/ubsan: clean"
