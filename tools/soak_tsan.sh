#!/bin/bash
# The manager's soak (tools/soak_manager.py) over ThreadSanitizer builds of the two libraries' host sources: libgarage_block and
# libgarage_ec's C ABI + CPU backend (no HIP backend: tests/c/ec_nodevice.cpp stands in, as in the other sanitizer builds), loaded
# into an ordinary python with libtsan preloaded.  Everything happens in a scratch copy of the tree; the tree itself is not touched.
# SAN=address: the same with AddressSanitizer + UBSan builds (use-after-free of a recycled pinned buffer, overruns; python's own
# leaks are not looked at).
# usage: [SAN=thread|address] tools/soak_tsan.sh [seconds] [seed] [devices] [directory-nodes root or ""]
# exit 0 = the soak passed AND the sanitizer reported nothing
set -eu
R="$(cd "$(dirname "$0")/.." && pwd)"
SECS="${1:-10}"; SEED="${2:-3}"; NDEV="${3:-1}"; ROOT="${4:-}"
W="$(mktemp -d /tmp/soak_tsan.XXXXXX)"
trap 'rm -rf "$W"' EXIT
mkdir -p "$W/garage_amd/csrc" "$W/tests/c" "$W/tools" "$W/include" "$W/oracle" "$W/tests"
cp -r "$R/garage_amd/"*.py "$W/garage_amd/"
cp "$R/garage_amd/csrc/"*.cpp "$R/garage_amd/csrc/"*.hpp "$W/garage_amd/csrc/"
cp "$R/tests/c/ec_nodevice.cpp" "$W/tests/c/"
cp "$R/tests/__init__.py" "$W/tests/" 2>/dev/null || true
cp "$R/tests/patterns.py" "$W/tests/"
cp "$R/tools/soak_manager.py" "$W/tools/"
cp "$R/include/"*.h "$W/include/"
cp -r "$R/oracle/"*.py "$R/oracle/"*.so "$W/oracle/" 2>/dev/null || true
cat > "$W/stubs.cpp" <<'STUBS'
// the HIP-only exports of libgarage_ec, absent from a host-only link (the python binding resolves every symbol when it loads)
extern "C" {
#define STUB(name) int name() { return -100; }
STUB(gec_get_kernel_variant) STUB(gec_group_allgather_decode) STUB(gec_group_alltoall_decode) STUB(gec_group_bytes_exchanged)
STUB(gec_group_create) STUB(gec_group_create_with_transport) STUB(gec_group_create_with_transport2) STUB(gec_group_destroy)
STUB(gec_group_peer_decode) STUB(gec_ipc_export) STUB(gec_ipc_open) STUB(gec_ipc_close) STUB(gec_group_rank) STUB(gec_group_size) STUB(gec_group_slots) STUB(gec_group_unique_id) STUB(gec_launch_geometry) STUB(gec_set_kernel_variant)
}
STUBS
SAN="${SAN:-thread}"
if [ "$SAN" = "address" ]; then
  F="-O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -std=c++17 -fPIC -shared"
  PRE="$(gcc -print-file-name=libasan.so)"
  export ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:abort_on_error=0:exitcode=99" UBSAN_OPTIONS="print_stacktrace=1"
  PAT="ERROR: AddressSanitizer\|runtime error:"
else
  F="-O1 -g -fsanitize=thread -fno-omit-frame-pointer -std=c++17 -fPIC -shared"
  PRE="$(gcc -print-file-name=libtsan.so)"
  # One report is the runtime's own and not looked at: when a thread that used a thread_local vector of the libraries ends,
  # its destructor reads the vector (__call_tls_dtors); whoever joins that thread later frees its TLS block inside ld.so
  # (_dl_deallocate_tls), and gcc 11's libtsan does not see the join between the two -- "data race ... in _dl_deallocate_tls",
  # every frame in libc / ld.so / ~vector, once in ~20 runs of 120 s (the soak's reader and writer threads end at different times).
  printf 'race:_dl_deallocate_tls\n' > "$W/tsan.supp"
  export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=0 suppressions=$W/tsan.supp"
  PAT="WARNING: ThreadSanitizer"
fi
( cd "$W/garage_amd/csrc" && cp ../../tests/c/ec_nodevice.cpp "$W/stubs.cpp" . &&
  CF="${F% -shared}" && pids="" &&
  for src in ec_api ec_env ec_cpu ec_nodevice stubs bm_core bm_node bm_gather bm_rw bm_stream bm_resync bm_scrub bm_scrub_worker bm_batcher; do   # side by side: fourteen units, ~12 s instead of ~40
    g++ $CF -c -o $src.o $src.cpp &
    pids="$pids $!"
  done &&
  for p in $pids; do wait $p; done &&
  g++ $F -o ../libgarage_ec.so ec_api.o ec_env.o ec_cpu.o ec_nodevice.o stubs.o -lpthread -ldl &&
  g++ $F -o ../libgarage_block.so bm_core.o bm_node.o bm_gather.o bm_rw.o bm_stream.o bm_resync.o bm_scrub.o bm_scrub_worker.o bm_batcher.o -L.. -lgarage_ec -lpthread -ldl -Wl,-rpath,'$ORIGIN' )
cd "$W"
set +e
SOAK_TIME_SCALE="${SOAK_TIME_SCALE:-5}" LD_PRELOAD="$PRE" python tools/soak_manager.py "$SECS" cpu 60000 "$SEED" "$NDEV" "$ROOT" > "$W/out.log" 2>&1
RC=$?
set -e
N=$(grep -c "$PAT" "$W/out.log" || true)
if grep -q "FATAL: ThreadSanitizer\|Shadow memory range interleaves\|ASan runtime does not come first" "$W/out.log"; then
  echo "soak_tsan: the sanitizer cannot run here"; grep "FATAL\|Shadow memory\|does not come first" "$W/out.log" | head -2; exit 77; fi
tail -1 "$W/out.log" | cut -c1-400
if [ "$RC" != "0" ]; then echo "--- the soak's last lines:"; tail -40 "$W/out.log" | cut -c1-600; fi
if [ "$N" != "0" ]; then grep -A18 "$PAT" "$W/out.log" | head -90; fi
echo "soak_tsan: soak exit $RC, $SAN sanitizer reports: $N"
[ "$RC" = "0" ] && [ "$N" = "0" ]
