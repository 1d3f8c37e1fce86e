#!/bin/bash
# Static audit of the 160-row one-wave-per-SIMD GEMM (csrc/gemm5.hip), no GPU needed: its accumulators and fragments live in
# accumulation registers named literally in inline asm, so the compiler must not place anything of its own there while they are
# live - i.e. in front of the LAST accumulator read-out of the kernel (behind it the registers are dead and hipcc may, and does,
# use them as spill space for the epilogue).  Also: no scratch, no spills, 3 x 40 matrix instructions in the K loop.
# HIPCC / ARCH: the compiler and target the library itself was built with (csrc/Makefile honours the same variables).
set -e
cd "$(dirname "$0")/../../realtime_video_amd/csrc"
OUT=${G5_AUDIT_DIR:-/tmp/g5_audit}; mkdir -p $OUT
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=${ARCH:-gfx950} -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -S --cuda-device-only "$@" gemm5.hip -o $OUT/g5.s 2>&1 | grep -E "error|warning:" || true
for F in 0 1; do
  K="_ZN3rtv12gemm5_kernelILb${F}EEEvNS_10GemmParamsENS_9SplitArgsE"
  awk "/^$K:/,/^.Lfunc_end/" $OUT/g5.s > $OUT/k$F.s
  SP=$(awk "/\\.name: *$K/{f=1} f && /vgpr_spill_count:/{print \$2; exit}" $OUT/g5.s)
  PS=$(awk "/\\.name: *$K/{f=1} f && /private_segment_fixed_size:/{print \$2; exit}" $OUT/g5.s)
  L=$(grep -n "v_accvgpr_read_b32 v[0-9]*, a\[159\]" $OUT/k$F.s | tail -1 | cut -d: -f1)
  N=$(awk -v L=$L '/ASMSTART/{a=1} /ASMEND/{a=0} { if(!a && NR<L && ($0 ~ /[ ,\[]a[0-9]+[ ,\]:]|[ ,]a\[[0-9]/) && $0 !~ /^[ \t]*;/) n++ } END{print n+0}' $OUT/k$F.s)
  echo "AUDIT gemm5 F16=$F: vgpr_spills ${SP:-?} private_segment ${PS:-?} scratch_ops $(grep -c scratch_ $OUT/k$F.s) mfma $(grep -c v_mfma $OUT/k$F.s) compiler_acc_refs_before_last_accumulator_read $N (last read at line $L of $(wc -l < $OUT/k$F.s))"
done
