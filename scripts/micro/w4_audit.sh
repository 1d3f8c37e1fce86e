#!/bin/bash
# Static audit of the one-wave-per-SIMD attention kernel (csrc/attn_w4.hip): compiles it to ISA (no GPU needed) and checks what
# the asm-owned accumulation registers require of the compiler: no v_accvgpr_* outside ASMSTART/ASMEND, no scratch, no VGPR spills.
#   scripts/micro/w4_audit.sh [variant=0] [extra hipcc flags]
# HIPCC / ARCH: the compiler and target the library itself was built with (csrc/Makefile honours the same variables).
set -e
cd "$(dirname "$0")/../../realtime_video_amd/csrc"
VAR=${1:-0}; shift || true
OUT=${W4_AUDIT_DIR:-/tmp/w4_audit}; mkdir -p $OUT
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=${ARCH:-gfx950} -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -S --cuda-device-only "$@" attn_w4.hip -o $OUT/w4.s 2>&1 | grep -E "error|warning:" || true
K="_ZN3rtv18attn_fwd_w4_kernelILb0ELi${VAR}EEEvNS_10AttnParamsE"
awk "/^$K:/,/^.Lfunc_end/" $OUT/w4.s > $OUT/k.s
grep -A40 "\.name: *$K" $OUT/w4.s | grep -E "vgpr_count|agpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|sgpr_count" | tr -s ' ' | tr '\n' ' '; echo
echo "lines $(wc -l < $OUT/k.s)  mfma $(grep -c v_mfma $OUT/k.s)  scratch $(grep -c scratch_ $OUT/k.s)  s_load $(grep -c 's_load' $OUT/k.s)"
awk '/ASMSTART/{a=1} /ASMEND/{a=0} /v_accvgpr/{ if(!a) n++ } END{print "v_accvgpr outside asm:", n+0}' $OUT/k.s
awk '/s_barrier/{printf "%6d: barrier  mfma=%d scratch=%d v_mov=%d lane=%d s_nop=%d lines=%d\n", NR, m, sc, vm, ln, sn, NR-last; m=0; sc=0; vm=0; ln=0; sn=0; last=NR} /v_mfma/{m++} /scratch_/{sc++} /v_mov_b32/{vm++} /v_readlane|v_writelane/{ln++} /s_nop/{sn++}' $OUT/k.s
awk '/ASMSTART/{a=1} /ASMEND/{a=0} { if(!a && ($0 ~ /[ ,\[]a[0-9]+[ ,\]:]|[ ,]a\[[0-9]/) && $0 !~ /^[ \t]*;/ && $0 !~ /\.amdhsa|\.sgpr|\.vgpr|\.agpr/) { n++; if (n<=5) print "  AGPR outside asm: " $0 } } END{print "instructions naming an accumulation register outside asm:", n+0}' $OUT/k.s
