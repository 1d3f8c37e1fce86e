#!/bin/sh
# Lab build of the product attention kernel with s_memtime phase stamps: scripts/micro/libattn_tr.so (scripts/attn_trace.py).
set -e
cd "$(dirname "$0")"
C=../../realtime_video_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRTV_ATTN_TRACE $ATT_LAB_FLAGS -Wno-unused-value \
  $C/attn_fwd.hip $C/runtime.hip $C/gemm.hip $C/gemm8.hip -o ${ATT_LAB_OUT:-libattn_tr.so}
