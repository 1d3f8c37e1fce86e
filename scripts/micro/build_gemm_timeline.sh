#!/bin/sh
# Lab build of the product GEMM (csrc/gemm8.hip + gemm.hip + runtime.hip) with per-workgroup timeline stamps:
# scripts/micro/libgemm_tl.so, driven by scripts/gemm_timeline.py.  Not part of librtv_hip.so.
set -e
cd "$(dirname "$0")"
C=../../realtime_video_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRTV_GEMM_TIMELINE -Wno-unused-value \
  $C/gemm8.hip $C/gemm.hip $C/runtime.hip -o libgemm_tl.so
