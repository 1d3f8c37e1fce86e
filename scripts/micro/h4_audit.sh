#!/bin/bash
# Static audit of the one-wave-per-SIMD halo-tile convolution kernels (csrc/vae_conv.hip: conv_halo4_kernel, conv_halo4p_kernel), no GPU
# needed: their accumulators and fragments live in accumulation registers named literally in inline asm, so the compiler must not place
# anything of its own there (no accumulation-register reference outside ASMSTART / ASMEND), and there must be no scratch and no spills.
# HIPCC / ARCH: the compiler and target the library itself was built with (csrc/Makefile honours the same variables).
set -e
cd "$(dirname "$0")/../../realtime_video_amd/csrc"
OUT=${H4_AUDIT_DIR:-/tmp/h4_audit}; mkdir -p $OUT
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=${ARCH:-gfx950} -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -S --cuda-device-only "$@" vae_conv.hip -o $OUT/vc.s 2>&1 | grep -E "error|warning: [a-z]" || true
for K in $(grep -o "^_ZN3rtv1[78]conv_halo4p\?_kernel[A-Za-z0-9_]*:" $OUT/vc.s | tr -d ':' | sort -u); do
  awk "/^$K:/,/^.Lfunc_end/" $OUT/vc.s > $OUT/k.s
  SP=$(awk "/\\.name: *$K\$/{f=1} f && /vgpr_spill_count:/{print \$2; exit}" $OUT/vc.s)
  SS=$(awk "/\\.name: *$K\$/{f=1} f && /sgpr_spill_count:/{print \$2; exit}" $OUT/vc.s)
  PS=$(awk "/\\.name: *$K\$/{f=1} f && /private_segment_fixed_size:/{print \$2; exit}" $OUT/vc.s)
  AO=$(grep -A40 "^\s*\.amdhsa_kernel $K\$" $OUT/vc.s | awk '/amdhsa_accum_offset/{print $2; exit}')
  # compiler references to accumulation registers (outside ASMSTART / ASMEND).  The persistent kernel's own map starts at a8 (a[0:7] were
  # left to the compiler while its register pressure was being brought down): references to a0..a7 are counted separately - both must be 0
  N=$(awk '/ASMSTART/{a=1} /ASMEND/{a=0} { if(!a && ($0 ~ /[ ,\[]a[0-9]+[ ,\]:]|[ ,]a\[[0-9]/) && $0 !~ /^[ \t]*;/) { if ($0 ~ /[ ,\[]a([89]|[1-9][0-9]+)[ ,\]:]|a\[[0-9]+:([89]|[1-9][0-9]+)\]/) hi++; else lo++ } } END{print hi+0 " (a0-a7: " lo+0 ")"}' $OUT/k.s)
  echo "AUDIT $K: arch_vgprs ${AO:-?} vgpr_spills ${SP:-?} sgpr_spills ${SS:-?} private_segment ${PS:-?} scratch_ops $(grep -c scratch_ $OUT/k.s) mfma $(grep -c v_mfma $OUT/k.s) compiler_acc_refs $N"
done
