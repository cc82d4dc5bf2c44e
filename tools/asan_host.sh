#!/bin/bash
# AddressSanitizer + UBSan run of the HOST side of the library (plan construction, table algebra, the C ABI) on a box
# without a GPU: the five host translation units are rebuilt with -fsanitize=address,undefined (-fno-gpu-sanitize: the
# device code objects stay as they are; GPU ASan is not available on this pool), linked with the regular kernel
# objects, and the host-plan tests (RF_DEVICE_HOST_ONLY plans: every table of every path) run against that build.
#   bash tools/asan_host.sh            -> prints the pytest summary; any sanitizer report fails the run
set -eu
root=$(cd "$(dirname "$0")/.." && pwd)
out=${TMPDIR:-/tmp}/recfilter_asan
mkdir -p "$out"
make -C "$root/recfilter_amd/csrc" -j8 > /dev/null
cd "$root/recfilter_amd/csrc"
for f in plan.cpp plan_fused.cpp plan_overlap.cpp plan_matrix.cpp capi.cpp; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fno-slp-vectorize -fsanitize=address,undefined \
        -fno-gpu-sanitize -fno-omit-frame-pointer -x hip -c $f -o "$out/${f%.cpp}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize \
    -o "$out/librecfilter_amd_asan.so" kernels_*.o "$out"/plan.o "$out"/plan_fused.o "$out"/plan_overlap.o "$out"/plan_matrix.o "$out"/capi.o
rt=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd "$root"
log="$out/asan_host.log"
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1 \
    LD_PRELOAD=$rt RECFILTER_AMD_LIB="$out/librecfilter_amd_asan.so" \
    python3 -c "
import sys, pytest
sys.path.insert(0, '.')
import recfilter_amd.capi as c
assert 'asan' in c.LIB_PATH, c.LIB_PATH
c.lib()
assert any('librecfilter_amd_asan.so' in l for l in open('/proc/self/maps')), 'sanitized build not mapped'
print('sanitized library mapped:', c.LIB_PATH)
sys.exit(pytest.main(['tests/test_host_plan.py', 'tests/test_dist_cpu.py', '-q']))
" > "$log" 2>&1 || { tail -30 "$log"; exit 1; }
reports=$(grep -c "runtime error\|AddressSanitizer" "$log" || true)
tail -4 "$log"
echo "sanitizer reports: $reports"
[ "$reports" = "0" ]
