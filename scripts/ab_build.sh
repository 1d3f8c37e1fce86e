#!/bin/sh
# Build librtv_hip.so of another git revision into scripts/micro/librtv_<rev>.so for same-box A/B runs:
#   scripts/ab_build.sh HEAD~1 ;  RTV_LIB_PATH=scripts/micro/librtv_HEAD~1.so python scripts/attn_bench.py
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" realtime_video_amd/csrc include | tar -x -C "$TMP"
make -C "$TMP/realtime_video_amd/csrc" -j8 > /dev/null
cp "$TMP/realtime_video_amd/librtv_hip.so" "$ROOT/scripts/micro/librtv_$(echo $REV | tr '/~^' '___').so"
rm -rf "$TMP"
