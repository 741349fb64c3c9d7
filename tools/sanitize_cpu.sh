#!/bin/bash
# CPU-side sanitizer pass (AddressSanitizer + UBSan + LeakSanitizer; GPU sanitizers are not available on this pool): the host C code
# (h264e_host.c, encode_app.c) and the kernel sources in their test-only lane-loop emulation build, run through the emulation parity
# tests, the per-stage fixtures, the shard tests and the CLI (reader thread, staging buffers, input ring, shards).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/tests/emu/build_asan
S=$R/h264-lab_amd/csrc
mkdir -p $B
F="-O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer"
gcc $F -Wno-format-truncation -c $S/h264e_host.c -o $B/host.o
(cd $R/tests/emu && g++ -std=c++17 $F -DH264E_EMU -c emu_backend.cpp -o $B/emu.o)
g++ -shared -fsanitize=address,undefined -o $B/libh264e_emu_asan.so $B/emu.o $B/host.o
gcc -O1 -g -fsanitize=address,undefined -o $B/encode_app_asan $S/encode_app.c -L$B -lh264e_emu_asan -Wl,-rpath,'$ORIGIN' -lm -lpthread
cd $R
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=0 \
  H264E_EMU_LIB_OVERRIDE=$B/libh264e_emu_asan.so python -m pytest tests/test_emu_parity.py tests/test_stages.py tests/test_multirank.py -x -q -s -m "not gpu" -k "not two_ranks" 2>&1 \
  | grep -E "runtime error|AddressSanitizer|passed|failed" | sort | uniq -c
python3 - <<PY
import sys; sys.path.insert(0, "$R/tests")
import clips
clips.make("scene", 176, 144, 10).tofile("/tmp/asan_176x144.yuv")
PY
for mode in "" "--clip 0" "--threads 3" "--kbps 200" "--gpus 2 --gop 3" "--psnr x"; do
  ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=halt_on_error=0 H264E_APP_STAGE_KB=100 H264E_APP_RING_KB=300 H264E_APP_OUT_KB=16 \
    $B/encode_app_asan --input /tmp/asan_176x144.yuv --output /tmp/asan_o.264 --qp 28 --stats x $mode 2>&1 | grep -E "runtime error|Sanitizer|leak" || echo "encode_app $mode: clean"
done
