#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the HOST code: the CPU oracle (gcc) and the host side of
# libhg_sim.so (hipcc, -Xarch_host: GPU sanitizers are not available on this pool and are not requested).
# Runs here on the CPU box; no GPU needed (hg_create must stop at HG_ENODEVICE after its host-side model checks).
# usage: bash tools/host_sanitize.sh          -> exit code 0 and no sanitizer report = clean
set -eu
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
python3 "$root/tools/sanitize/dump_models.py" "$tmp/models"
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g"
gcc -O1 -std=c11 -fopenmp $SAN -I"$root/include" -c "$root/oracle/hg_oracle.c" -o "$tmp/hg_oracle.o"
gcc -O1 -std=c11 $SAN -I"$root/include" -c "$root/tools/sanitize/san_main.c" -o "$tmp/san_main.o"
(cd "$root/humanoid_gym_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -std=c++17 -fno-slp-vectorize -fPIC \
    -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer \
    -c hg_sim.hip -o "$tmp/hg_sim.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address -fsanitize=undefined -fno-gpu-sanitize -fopenmp \
    "$tmp/san_main.o" "$tmp/hg_oracle.o" "$tmp/hg_sim.o" -lm -o "$tmp/san_main"
printf "leak:___kmp_allocate\nleak:libomp.so\nleak:libamdhip64\nleak:libhsa-runtime64\n" > "$tmp/lsan.supp"      # the runtimes' own one-time allocations
LSAN_OPTIONS=suppressions="$tmp/lsan.supp":print_suppressions=0 ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=4 \
    "$tmp/san_main" "$tmp"/models/*.bin
echo "host sanitizers: clean"
