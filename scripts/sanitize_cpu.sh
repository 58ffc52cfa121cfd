#!/bin/bash
# The host side of libgscan.so / libgrabhost.so (pattern compiler, matcher, the VM and its compiler, the report walk, the C facade)
# under AddressSanitizer + UndefinedBehaviorSanitizer, driven by the CPU test suite and a differential fuzz campaign.  The device code
# is NOT instrumented (-fno-gpu-sanitize: GPU ASan needs xnack+, which the pool refuses); a copy of the tree is built in /tmp so the
# product's own .so files stay as they are.
#     scripts/sanitize_cpu.sh [draws per grammar, default 3000]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${SANITIZE_DIR:-/tmp/grab_sanitize}
DRAWS=${1:-3000}
CLANG=/opt/rocm/lib/llvm/bin/clang++
SAN="-fsanitize=address,undefined -fno-sanitize=vptr,function -fno-gpu-sanitize -g -fno-omit-frame-pointer"
rm -rf "$W" && mkdir -p "$W/repo" "$W/logs"
(cd "$ROOT" && tar --exclude=.git --exclude=gpurun_out --exclude=profiles --exclude=grab_amd/csrc/obj --exclude=grab_amd/lib --exclude=grab_amd/bin --exclude=__pycache__ --exclude=oracle/_ref -cf - .) | tar -xf - -C "$W/repo"
cd "$W/repo/grab_amd/csrc" || exit 1
make -j6 ../lib/libgscan.so ../lib/libgrabhost.so \
    HIPFLAGS="--offload-arch=gfx950 -O1 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -Wno-unused-function -Wno-unused-value -Wno-unused-result $SAN" \
    CXX=$CLANG CXXFLAGS="-O1 -std=c++17 -fPIC -pthread $SAN" > "$W/logs/build.txt" 2>&1 || { tail -20 "$W/logs/build.txt"; exit 1; }
mkdir -p ../bin
$CLANG -O1 -std=c++17 -fPIC -pthread ${SAN/-fno-gpu-sanitize/} -shared-libasan grab_cli.cc -o ../bin/grab -L../lib -lgrabhost -lgscan -Wl,-rpath,'$ORIGIN/../lib' >> "$W/logs/build.txt" 2>&1
printf '#!/bin/sh\n' > ../bin/gscan_sweep && chmod +x ../bin/gscan_sweep   # (a HIP tool: not part of this)
cd "$W/repo" || exit 1
export LD_PRELOAD=$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0:log_path=$W/logs/asan UBSAN_OPTIONS=print_stacktrace=1:log_path=$W/logs/ubsan
# (deselected: a test of the matcher's give-up depth on a 1 MiB thread stack -- ASan's frames are several times the size --, and
# the recipe that links the reference's main.cc against the product with gcc -- there is no gcc ASan runtime for clang's objects)
python -m pytest tests -q -m "not gpu" -p no:cacheprovider -n 5 \
    --deselect tests/test_pattern.py::test_deep_group_repeats_and_small_stacks --deselect tests/test_integration.py::test_recipe_a_compiles_and_links 2>&1 | tail -3
for g in plain calls binary; do python scripts/fuzz_campaign.py --seed0 62000000 --procs 6 --draws "$DRAWS" --grammar $g 2>&1 | grep -E '"(grammar|draws|compared|differences)"' | tr -d '\n'; echo; done
echo "sanitizer reports (files under $W/logs other than the makecontext notice):"
grep -L "makecontext" "$W"/logs/asan.* "$W"/logs/ubsan.* 2>/dev/null | head
grep -h -A12 "ERROR: AddressSanitizer\|runtime error" "$W"/logs/asan.* "$W"/logs/ubsan.* 2>/dev/null | head -60
echo "end of sanitizer reports"

# ThreadSanitizer over the part of the host side that is multi-threaded WITHOUT a device: the parallel tree walk of `grab -n`
# (walk.cc: walker threads sharing the tree directory by directory) and the worker placement, through tests/test_host_cpu.py.
# (The engine's reader pools and the -n workers need a HIP device: their races are the GPU suite's to find -- the reader pools'
# lost wake-up of round 6 was found there.)
unset LD_PRELOAD
cd "$W/repo/grab_amd/csrc" || exit 1
cp "$ROOT/grab_amd/lib/libgscan.so" ../lib/libgscan.so   # (the product's own build: the ASan one cannot be loaded beside TSan's runtime; the walk never enters it)
$CLANG -O1 -g -std=c++17 -fPIC -pthread -fsanitize=thread -DGRAB_PCRE_VALIDATE -idirafter /opt/conda/include -shared filegrep.cc capi_host.cc walk.cc placement.cc \
    -o ../lib/libgrabhost.so -L../lib -lgscan /usr/lib/x86_64-linux-gnu/libpcre.so.3 -Wl,-rpath,'$ORIGIN' >> "$W/logs/build.txt" 2>&1 || { tail "$W/logs/build.txt"; exit 1; }
sleep 1; touch ../bin/grab ../bin/gscan_sweep
cd "$W/repo" || exit 1
LD_PRELOAD=$($CLANG -print-file-name=libclang_rt.tsan-x86_64.so) TSAN_OPTIONS="log_path=$W/logs/tsan:report_signal_unsafe=0" \
    python -m pytest tests/test_host_cpu.py -q -p no:cacheprovider -k "walk or place" 2>&1 | tail -2
echo "thread sanitizer reports:"; cat "$W"/logs/tsan.* 2>/dev/null | head -60; echo "end of thread sanitizer reports"
