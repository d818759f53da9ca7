#!/bin/bash
# Host-only AddressSanitizer + UndefinedBehaviorSanitizer run of the library's host code on an MI355X box
# (GPU sanitizers are not available on this pool): builds tools/libaesw_asan.so and tools/asan_driver.c against it,
# then runs the driver -- every host-pointer entry point in the three layouts, geometry / selector helpers and the C++
# mirror in all assign modes, the device-pointer streaming entry points and the one-rank gather.  Usage (from the repo root): gpurun -- 'bash tools/sanitize.sh'
set -eu
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
SRC="halo2-aes_amd/csrc/aesw_kernels.hip halo2-aes_amd/csrc/aesw_api.cpp halo2-aes_amd/csrc/aesw_arena.cpp halo2-aes_amd/csrc/aesw_comm.cpp halo2-aes_amd/host/host_capi.cpp"
if [ ! -f tools/libaesw_asan.so ] || [ -n "$(find $SRC halo2-aes_amd/csrc/*.h halo2-aes_amd/host/*.hpp -newer tools/libaesw_asan.so)" ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined \
        -Xarch_host -fno-omit-frame-pointer -Xarch_host -g -o tools/libaesw_asan.so $SRC
fi
/opt/rocm/lib/llvm/bin/clang -O1 -g -fsanitize=address,undefined -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tools/asan_driver.c \
    -o tools/asan_driver tools/libaesw_asan.so -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
# detect_leaks=0: the HIP runtime keeps its allocations; protect_shadow_gap=0: the GPU driver maps into the gap
ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 timeout -k 10 300 ./tools/asan_driver

# ThreadSanitizer (host only): three threads, one context each, the host-pointer path in three layouts at once.
# The HIP / HSA runtimes are not instrumented, so their frames are suppressed (tools/tsan.supp).
if [ ! -f tools/libaesw_tsan.so ] || [ -n "$(find $SRC halo2-aes_amd/csrc/*.h halo2-aes_amd/host/*.hpp -newer tools/libaesw_tsan.so)" ]; then
    hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -shared -Xarch_host -fsanitize=thread -Xarch_host -g -o tools/libaesw_tsan.so $SRC
fi
/opt/rocm/lib/llvm/bin/clang -O1 -g -fsanitize=thread -Iinclude tools/tsan_driver.c -o tools/tsan_driver tools/libaesw_tsan.so -lpthread \
    -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
TSAN_OPTIONS="report_signal_unsafe=0 suppressions=tools/tsan.supp exitcode=66" timeout -k 10 300 ./tools/tsan_driver
