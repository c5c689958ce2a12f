#!/usr/bin/env bash
# Host code under sanitizers (CPU build, no GPU needed).   tools/sanitize_host.sh [asan|tsan|all]
#
# Builds every HOST translation unit of the library -- lc_ctx.cpp, lc_comm.cpp, lc_engine.cpp, lc_topic.cpp,
# lc_capi.cpp and the host halves of the five .hip files (--cuda-host-only: launch planners, grids, LDS grants) --
# twice, with -fsanitize=address,undefined and with -fsanitize=thread, against tools/sanitize/hip_host_stub.cpp (a
# host-memory stand-in for the HIP runtime: kernels do not run, launches fail), and runs
#   * tools/sanitize/host_hammer.cpp: M-step pool, block cache / cache_release_thread, heap and shared-memory all-reduce
#     with 1-8 ranks incl. one rank aborting and a left-over rendezvous object, the failing-shard path of learn_sharded;
#   * (asan) pytest -m "not gpu" with the sanitized library loaded through LC_LIB_PATH.
# Logs: profiles/r06_sanitize_{asan_ubsan,tsan}.log.  Exit status 0 = both clean.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/tools/sanitize/_build"
CSRC="$ROOT/libcluster_amd/csrc"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
CLANG_RT="$(dirname "$($HIPCC --print-file-name=libclang_rt.asan-x86_64.so 2>/dev/null || true)")"
[ -f "$CLANG_RT/libclang_rt.asan-x86_64.so" ] || CLANG_RT="$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)"
WHAT="${1:-all}"
mkdir -p "$OUT" "$ROOT/profiles"
HASH="$(cd "$ROOT" && python3 -c 'from libcluster_amd import build; print(build.source_hash())')"
status=0

build() {  # $1 = tag, $2 = sanitizer flags
  local tag="$1" san="$2" d="$OUT/$1"
  mkdir -p "$d"
  local common="-O1 -g -fno-omit-frame-pointer -std=c++17 -fPIC -I$ROOT/include -I$CSRC $san"
  local pids=()
  for f in lc_kernels_estep lc_kernels_suffstat lc_kernels_diag lc_kernels_aux lc_kernels_fused; do
    $HIPCC --offload-arch=gfx950 --cuda-host-only $common -c "$CSRC/$f.hip" -o "$d/$f.o" 2>>"$d/build.log" & pids+=($!)
  done
  local cxx="/opt/rocm/lib/llvm/bin/clang++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"  # (plain C++: these are host sources)
  for f in lc_ctx lc_comm lc_engine lc_topic lc_capi; do
    $cxx $common -DLC_SOURCE_HASH="\"$HASH\"" -c "$CSRC/$f.cpp" -o "$d/$f.o" 2>>"$d/build.log" & pids+=($!)
  done
  $cxx $common -c "$ROOT/tools/sanitize/hip_host_stub.cpp" -o "$d/hip_host_stub.o" 2>>"$d/build.log" & pids+=($!)
  $cxx $common -c "$ROOT/tools/sanitize/host_hammer.cpp" -o "$d/host_hammer.o" 2>>"$d/build.log" & pids+=($!)
  for p in "${pids[@]}"; do wait "$p" || { echo "compile failed ($tag), see $d/build.log"; tail -20 "$d/build.log"; return 1; }; done
  # the host halves refer to their (absent) device images by name
  nm -u "$d"/lc_kernels_*.o | grep -o '__hip_fatbin_[0-9a-f]*' | sort -u |
    awk '{print "const char " $1 "[8] __attribute__((section(\".hip_fatbin\"))) = {0};"}' > "$d/fatbin_syms.c"
  gcc -c "$d/fatbin_syms.c" -o "$d/fatbin_syms.o" || return 1
  local objs="$d/lc_kernels_estep.o $d/lc_kernels_suffstat.o $d/lc_kernels_diag.o $d/lc_kernels_aux.o $d/lc_kernels_fused.o \
    $d/lc_ctx.o $d/lc_comm.o $d/lc_engine.o $d/lc_topic.o $d/lc_capi.o $d/hip_host_stub.o $d/fatbin_syms.o"
  # (clang++ directly: hipcc would add the real libamdhip64 to the link)
  /opt/rocm/lib/llvm/bin/clang++ $san -shared-libsan -shared -o "$d/libcluster_hip.so" $objs -lpthread -ldl -lrt -Wl,-rpath,"$CLANG_RT" 2>>"$d/build.log" || { tail -20 "$d/build.log"; return 1; }
  /opt/rocm/lib/llvm/bin/clang++ $san -shared-libsan -o "$d/host_hammer" "$d/host_hammer.o" $objs -lpthread -ldl -lrt \
    -Wl,-rpath,"$CLANG_RT" 2>>"$d/build.log" || { tail -20 "$d/build.log"; return 1; }
}

if [ "$WHAT" = asan ] || [ "$WHAT" = all ]; then
  LOG="$ROOT/profiles/r06_sanitize_asan_ubsan.log"
  {
    echo "# tools/sanitize_host.sh asan -- $(date -u +%FT%TZ) -- source hash $HASH"
    echo "# -fsanitize=address,undefined over the host translation units + host halves of the .hip files, HIP runtime = tools/sanitize/hip_host_stub.cpp"
  } > "$LOG"
  if build asan "-fsanitize=address,undefined -fno-sanitize-recover=undefined"; then
    export ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=23" UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
    echo "## host_hammer all" >> "$LOG"
    ( cd "$ROOT" && timeout 900 "$OUT/asan/host_hammer" all ) >> "$LOG" 2>&1 || { echo "host_hammer (asan) FAILED"; status=1; }
    echo "## pytest -m 'not gpu' with LC_LIB_PATH=libcluster_hip.so (LD_PRELOAD of the ASan runtime; leak check off: CPython)" >> "$LOG"
    ( cd "$ROOT" && ASAN_OPTIONS="detect_leaks=0:exitcode=23" LD_PRELOAD="$CLANG_RT/libclang_rt.asan-x86_64.so" \
        LC_LIB_PATH="$OUT/asan/libcluster_hip.so" timeout 1500 python3 -m pytest tests -x -q -m "not gpu" -p no:cacheprovider ) >> "$LOG" 2>&1 \
      || { echo "pytest under ASan FAILED"; status=1; }
  else
    status=1
  fi
  grep -c "ERROR: AddressSanitizer\|runtime error:" "$LOG" | sed 's/^/# sanitizer reports in the log: /' >> "$LOG"
  tail -12 "$LOG"
fi

if [ "$WHAT" = tsan ] || [ "$WHAT" = all ]; then
  LOG="$ROOT/profiles/r06_sanitize_tsan.log"
  {
    echo "# tools/sanitize_host.sh tsan -- $(date -u +%FT%TZ) -- source hash $HASH"
    echo "# -fsanitize=thread over the same objects"
  } > "$LOG"
  if build tsan "-fsanitize=thread"; then
    export TSAN_OPTIONS="halt_on_error=0:exitcode=24:second_deadlock_stack=1"
    for part in shm pool cache local capi; do
      echo "## host_hammer $part" >> "$LOG"
      ( cd "$ROOT" && timeout 1500 "$OUT/tsan/host_hammer" $part ) >> "$LOG" 2>&1 || { echo "host_hammer $part (tsan) FAILED"; status=1; }
    done
  else
    status=1
  fi
  grep -c "WARNING: ThreadSanitizer" "$LOG" | sed 's/^/# sanitizer reports in the log: /' >> "$LOG"
  tail -12 "$LOG"
fi
exit $status
