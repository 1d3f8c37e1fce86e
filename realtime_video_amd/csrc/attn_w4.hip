// One-wave-per-SIMD flash attention forward for gfx950 (head_dim 128, bf16 / f16, BLHD, in-place strided K / V): the kernel
// behind `attention()` (wan/modules/attention.py:150-212) and the cached self-attention call (causal_model.py:386-392) for the
// launches that fill the chip.  Same math, same summation order and the same bits as attn_fwd_pp_kernel (attn_fwd.hip).
//
// Why another kernel: the four-phase kernel runs two waves of 32 query rows per SIMD; per 64-key tile each wave reads the whole
// K and V tile out of LDS (2 x 16 KiB), issues its share of the staging and does a full softmax - ~1750 cycles of non-matrix
// work against 2 x 620 cycles of MFMA, and the partner wave can hide only one under the other (profiles/r02_attn_four_phase_
// trace_and_pmc.log: tile period 4400 cycles for 2048 of MFMA).  Here a workgroup is FOUR waves, one per SIMD, each owning 64
// query rows (two 32-row blocks A / B) and the whole 512-register file:
//   * every K / V^T fragment read from LDS feeds TWO MFMAs (one per query block): half the LDS reads per flop;
//   * one instruction stream per SIMD: every non-matrix instruction is placed BEHIND a fixed MFMA (at most ~5 per MFMA gap:
//     one LDS read or one DMA piece plus ~4 VALU), pinned with sched_barrier - nothing waits in front of the matrix pipe;
//   * the softmax of tile j+1 is software-pipelined across the matrix phases of tiles j and j+1 (two S^T buffers):
//         phase A(j):  32 MFMAs  S^T(j+1) = K(j+1) . Q^T      | V^T(j) fragment reads, second half of softmax(j)
//         phase B(j):  32 MFMAs  O^T    += V^T(j) . P^T(j)    | K(j+2) fragment reads, the tile's 8 DMA pieces,
//                                                                row maxima / rescale decision / first half of softmax(j+1)
//     one s_barrier per tile (end of B);
//   * K and V^T fragments share ONE 64-register block: the V^T fragment f is read into the registers of K fragment f the moment
//     both of its MFMAs have issued, and vice versa;
//   * register file: O^T a[0:127], Q^T a[128:191], fragments a[192:255] - accumulation registers named LITERALLY in inline asm
//     (every MFMA, LDS fragment read and accumulator move of this kernel is an asm statement; the compiler never sees these
//     values, so it cannot copy, spill or re-home them) - and S^T 2 x 64 + P^T 32 + state in the 256 architectural registers,
//     where the compiler schedules the softmax.  (Left to hipcc, one wave per SIMD puts EVERY MFMA result into accumulation
//     registers - 64 v_accvgpr_read per tile for the softmax -, -amdgpu-mfma-vgpr-form every one into architectural registers
//     with O^T spilling, and fragments built from two transpose reads are copied through architectural registers: measured in
//     the ISA, round 5.)  Audit after every edit (scripts/micro/w4_audit.sh): no v_accvgpr_* outside ASMSTART / ASMEND, no
//     scratch, vgpr_spill_count 0;
//   * K / V tiles by LDS DMA into 4-slot rings (K three tiles ahead, V two), the tile loop unrolled by four so that every LDS
//     offset is an immediate; retired by ONE counted vmcnt per tile.
// The lazy rescale of O^T (rows whose maximum grew by more than 2^8) is decided beside PV(j) but applied after it - a rescale in
// the middle of a tile's P.V would mix two reference points.
#include "attn_common.h"

namespace rtv {
namespace w4 {

constexpr int NSLOT = 4;                                   // ring slots per operand
constexpr int LDS_BYTES = 2 * NSLOT * ATT_TILE_BYTES;      // 128 KiB
constexpr int V_BASE = NSLOT * ATT_TILE_BYTES;
constexpr int QT = 256;                                    // query rows per workgroup
constexpr int QW = 64;                                     // query rows per wave

// ------------------------------------------------------------------------------------------------------------------------
// The softmax of ONE tile as a stream of single VALU instructions, in issue order.  Per wave two query blocks qb; a block's 32
// scores per lane are S[qb][h][0..15], h = 32-key block; pair u = 2 w + qb, w = 0..15 = elements 2t, 2t+1 of block kbk
// (w = 8 kbk + t) -> half a P^T register.
//   MAX (qb, h, k) k = 0..7  chain (qb, h): acc = max3(acc, v, v')      four chains alternate: no instruction depends on one
//   CMB (qb)                 mx = max of the block's two chains          of the three in front of it
//   XCH0 / XCH1 (qb)         the other half of the row lives in lane ^ 32
//   DEC (qb, k) k = 0..5(6)  reference point, lazy-rescale decision, alpha (, l *= alpha)
//   F0 / F1 (u)              x = s * c - m    for the two elements of the pair
//   E0 / E1 (u)              e = exp2(x)
//   A0 / A1 (u)              row-sum partials += e                       (not with RS)
//   CV (u)                   pack the pair
//   SUM (qb, k) k = 0..1     l += partial sums                           (not with RS)
// F / E / {A, CV} are skewed by one pair each - F(k), E(k-1), A/CV(k-2) form a time step.
// RS (row sums by the matrix pipe): l^T += 1 . P^T - one more MFMA per query block and 16-key step, all-ones A operand - instead
// of the 64 additions + 4 of a tile: 8 MFMAs (256 matrix cycles) for 68 issue slots of a stream that is bound by its issue slots,
// not by the matrix pipe (profiles/r05_attn_w4_ablation.log).  The sums are then taken over the bf16-ROUNDED probabilities, the
// numbers O^T is accumulated from (the four-phase kernel sums the unrounded ones: the two kernels differ by rounding noise of
// the normaliser, ~2^-9 / sqrt(keys) relative, instead of being bit-identical).
enum OpKind { OP_MAX = 0, OP_CMB, OP_XCH0, OP_XCH1, OP_DEC, OP_F, OP_E, OP_A, OP_CV, OP_SUM };
struct Op {
  int kind, qb, idx, sub, cost;
};
constexpr int N_PAIRS = 32;
constexpr int MAX_OPS = 300;
struct OpTable {
  Op op[MAX_OPS];
  int n;
};
template <bool RS>
constexpr OpTable make_ops() {
  OpTable t{};
  int n = 0;
  for (int k = 0; k < 8; ++k)
    for (int h = 0; h < 2; ++h)
      for (int qb = 0; qb < 2; ++qb) t.op[n++] = Op{OP_MAX, qb, k, h, 1};
  for (int qb = 0; qb < 2; ++qb) t.op[n++] = Op{OP_CMB, qb, 0, 0, 1};
  for (int qb = 0; qb < 2; ++qb) t.op[n++] = Op{OP_XCH0, qb, 0, 0, 2};
  for (int qb = 0; qb < 2; ++qb) t.op[n++] = Op{OP_XCH1, qb, 0, 0, 1};
  for (int k = 0; k < (RS ? 6 : 7); ++k)
    for (int qb = 0; qb < 2; ++qb) t.op[n++] = Op{OP_DEC, qb, k, 0, k == 3 ? 2 : 1};
  for (int k = 0; k < N_PAIRS + 2; ++k) {
    if (k < N_PAIRS) {
      t.op[n++] = Op{OP_F, k & 1, k, 0, 1};
      t.op[n++] = Op{OP_F, k & 1, k, 1, 1};
    }
    if (k >= 1 && k - 1 < N_PAIRS) {
      t.op[n++] = Op{OP_E, (k - 1) & 1, k - 1, 0, 1};
      t.op[n++] = Op{OP_E, (k - 1) & 1, k - 1, 1, 1};
    }
    if (k >= 2) {
      if (!RS) {
        t.op[n++] = Op{OP_A, (k - 2) & 1, k - 2, 0, 1};
        t.op[n++] = Op{OP_A, (k - 2) & 1, k - 2, 1, 1};
      }
      t.op[n++] = Op{OP_CV, (k - 2) & 1, k - 2, 0, 1};
    }
  }
  if (!RS)
    for (int k = 0; k < 2; ++k)
      for (int qb = 0; qb < 2; ++qb) t.op[n++] = Op{OP_SUM, qb, k, 0, 1};
  t.n = n;
  return t;
}

// ---- the matrix instructions of phase B, in issue order.  Without RS: slot m = (step, db, qb) = (m >> 3, (m >> 1) & 3, m & 1);
// with RS every 16-key step is followed by its two row-sum MFMAs: slot m = 10 step + j, j < 8 as before, j = 8, 9: l^T of qb j - 8.
template <bool RS>
struct PhaseB {
  static constexpr int N = RS ? 40 : 32;
  static constexpr int PER = RS ? 10 : 8;
  static constexpr int step(int m) { return m / PER; }
  static constexpr bool is_pv(int m) { return m % PER < 8; }
  static constexpr int db(int m) { return (m % PER) >> 1; }
  static constexpr int qb(int m) { return is_pv(m) ? (m % PER) & 1 : m % PER - 8; }
  static constexpr int frag(int m) { return step(m) * 4 + db(m); }                              // V^T fragment of a PV slot
  static constexpr int kread(int m) { return is_pv(m) && ((m % PER) & 1) ? frag(m) : -1; }    // K fragment read behind slot m
  static constexpr int last_reader(int st) { return st * PER + PER - 1; }                      // last slot that reads P^T step st
  static constexpr int first_slot_of_frag(int f) { return (f >> 2) * PER + 2 * (f & 3); }
};

// Issue slots of one tile's softmax: phase B of the previous tile's iteration (slots 0 .. NB-1) then phase A (NB .. NB+31).
// What else a slot carries: the K fragment read behind the second PV MFMA of a V^T fragment, the DMA pieces (dma_at), in A the
// two transpose reads of a V^T fragment behind odd slots.  The stream is cut so that every slot carries the same number of
// instructions (weights in quarter instructions; a variant may price a DMA piece higher than a VALU instruction).
struct SchedTable {
  int begin[80];     // ops of slot s: [begin[s], begin[s + 1])
  bool ok;
};
template <int VAR, bool RS>
struct Sched {
  using PB = PhaseB<RS>;
  static constexpr int NB = PB::N, NS = NB + 32;
  static constexpr OpTable OPS = make_ops<RS>();
  // slot of B behind which DMA piece i (0-3 K, 4-7 V) is issued: behind the FIRST MFMA of a fragment pair (the second carries
  // the K fragment read); variant 1: late in the phase; variant 2: together with the K reads
  static constexpr int dma_at(int i) {
    const int f = VAR == 1 ? 8 + i : i + 1;                 // fragment pair whose first (2) / second slot carries the piece
    return PB::first_slot_of_frag(f) + (VAR == 2 ? 1 : 0);
  }
  static constexpr int dma_piece(int slot) {
    for (int i = 0; i < 8; ++i)
      if (dma_at(i) == slot) return i;
    return -1;
  }
  static constexpr int W_DMA = VAR == 3 ? 12 : 6, W_K = 4, W_TR = 4;   // quarter instructions (a piece = s_add m0 + the load)
  static constexpr int others4(int s) {
    if (s < NB) return (PB::kread(s) >= 0 ? W_K : 0) + (dma_piece(s) >= 0 ? W_DMA : 0);
    return ((s - NB) & 1) ? 2 * W_TR : 0;
  }
  static constexpr SchedTable make() {
    SchedTable t{};
    int tot = 0;
    for (int i = 0; i < OPS.n; ++i) tot += 4 * OPS.op[i].cost;
    for (int s = 0; s < NS; ++s) tot += others4(s);
    // slot s ends where the cumulative cost reaches (s + 1) / NS of the total
    int g = 0, cum = 0, cum_other = 0;
    for (int s = 0; s < NS; ++s) {
      t.begin[s] = g;
      cum_other += others4(s);
      const int target = (tot * (s + 1)) / NS - cum_other;       // VALU quarter-instructions behind slots 0..s
      while (g < OPS.n && cum + 2 * OPS.op[g].cost <= target) {  // an instruction belongs to the slot its midpoint falls in
        cum += 4 * OPS.op[g].cost;
        ++g;
      }
    }
    t.begin[NS] = OPS.n;   // (rounding) the last slot takes what is left
    t.ok = true;
    // P^T hazard: the pack CV(u) writes P^T[qb][kbk][t >> 2] of the NEXT tile while phase B of the current one still reads
    // that key step: the pack must sit behind the step's last reader
    for (int s = 0; s < NS; ++s)
      for (int gg = t.begin[s]; gg < t.begin[s + 1]; ++gg) {
        const Op G = OPS.op[gg];
        if (G.kind != OP_CV) continue;
        const int w = G.idx >> 1, st = (w >> 3) * 2 + ((w & 7) >> 2);
        if (s < PB::last_reader(st)) t.ok = false;
      }
    return t;
  }
  static constexpr SchedTable tab = make();
  static constexpr int begin(int s) { return tab.begin[s]; }
};

// ---- fragment waits.  LDS operations of a wave retire in order.  Phase B issues the 16 K fragment reads of the next tile (one
// behind the second PV MFMA of each V^T fragment), phase A the 32 transpose reads of the tile's V^T fragments (two behind every
// odd slot).  Phase A consumes K fragment f in slots 2f, 2f+1:
//   slot 0: lgkmcnt(14) - fragments 0, 1 have landed (14 younger K reads may be in flight);
//   slot 4: lgkmcnt(15) - fragments 2, 3 (12 younger K reads + the 4 transpose reads of slots 1, 3 = 16 > 15: conservative);
//   slot 8: lgkmcnt(8)  - every K read (only the 8 transpose reads of slots 1..7 are younger; the last K read is 9 slots old).
// Phase B consumes V^T fragment f in its slots first_slot_of_frag(f), +1:
//   slot 0: lgkmcnt(15) - of the 32 transpose reads the oldest 17 have landed: fragments 0..7 (key steps 0, 1);
//   first slot of key step 2: lgkmcnt(8) - every transpose read (younger: the 8 K reads of steps 0, 1).
constexpr int wait_a(int n) { return n == 0 ? 14 : (n == 4 ? 15 : (n == 8 ? 8 : -1)); }
template <bool RS>
constexpr int wait_b(int m) { return m == 0 ? 15 : (m == PhaseB<RS>::first_slot_of_frag(8) ? 8 : -1); }

// ---- accumulation-register map (asm-owned)
constexpr int A_O = 0;       // O^T [qb][db]: a[(qb*4 + db)*16 .. +15]
constexpr int A_Q = 128;     // Q^T [qb][dc]: a[128 + (qb*8 + dc)*4 .. +3]
constexpr int A_KV = 192;    // fragment f:   a[192 + 4 f .. +3]
// The compiler must never place a value of its own in an accumulation register: nothing in the asm statements tells it that
// they are occupied (naming them as clobbers on every statement works - and costs an s_nop between any two statements, 3 per
// MFMA).  What keeps it out is that it has no reason to enter: the architectural file holds everything it allocates with ~50
// registers to spare (left to itself under pressure, its allocator parks long-lived values - LDS addresses, loop invariants - in
// "free" accumulation registers and reloads them with v_accvgpr_read, on top of O^T: seen in the ISA of an earlier version).
// scripts/micro/w4_audit.sh is the check, run by __graft_entry__.build() and by the CPU test suite: no accumulation register
// in any instruction outside ASMSTART / ASMEND, no scratch, no spills.  The one clobber list below (kernel entry) makes the
// kernel descriptor allocate all 256.
#define RTV_W4_ACC \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
  "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
  "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
  "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
  "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
  "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
  "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
  "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", \
  "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", \
  "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", \
  "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", \
  "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", \
  "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", \
  "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", \
  "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", \
  "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"

// S^T(VGPR) = / += K fragment (AGPR) . Q^T fragment (AGPR)
template <bool F16, int KF, int QF, bool FIRST>
__device__ __forceinline__ void mfma_qk(f32x16& s) {
  if constexpr (FIRST) {
    if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%c1:%c2], a[%c3:%c4], 0" : "=v"(s) : "n"(KF), "n"(KF + 3), "n"(QF), "n"(QF + 3));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], 0" : "=v"(s) : "n"(KF), "n"(KF + 3), "n"(QF), "n"(QF + 3));
  } else {
    if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(s) : "n"(KF), "n"(KF + 3), "n"(QF), "n"(QF + 3));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(s) : "n"(KF), "n"(KF + 3), "n"(QF), "n"(QF + 3));
  }
}
// O^T(AGPR) += V^T fragment (AGPR) . P^T (VGPR)
template <bool F16, int OF, int VF>
__device__ __forceinline__ void mfma_pv(const u32x4& p_vgpr) {
  if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 a[%c1:%c2], a[%c3:%c4], %0, a[%c1:%c2]" ::"v"(p_vgpr), "n"(OF), "n"(OF + 15), "n"(VF), "n"(VF + 3));
  else asm volatile("v_mfma_f32_32x32x16_bf16 a[%c1:%c2], a[%c3:%c4], %0, a[%c1:%c2]" ::"v"(p_vgpr), "n"(OF), "n"(OF + 15), "n"(VF), "n"(VF + 3));
}
// l^T(VGPR) += ones (VGPR) . P^T (VGPR): every row of the 32 x 32 result is the row-sum vector of the 32 queries
template <bool F16>
__device__ __forceinline__ void mfma_rowsum(f32x16& l, const u32x4& ones, const u32x4& p_vgpr) {
  if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(l) : "v"(ones), "v"(p_vgpr));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(l) : "v"(ones), "v"(p_vgpr));
}
// max without the canonicalising v_max hipcc puts in front of fmaxf on values it cannot prove quiet; volatile: ordered like a pin
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm volatile("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int DST, int OFF>
__device__ __forceinline__ void lds_read128_a(uint32_t lds_addr) {
  asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%3" ::"v"(lds_addr), "n"(DST), "n"(DST + 3), "n"(OFF));
}
template <int DST, int OFF>
__device__ __forceinline__ void lds_tr_read_a(uint32_t lds_addr) {   // two halves of a V^T fragment: keys +0..3 and +8..11
  asm volatile("ds_read_b64_tr_b16 a[%c1:%c2], %0 offset:%5\n\tds_read_b64_tr_b16 a[%c3:%c4], %0 offset:%6" ::"v"(lds_addr), "n"(DST),
               "n"(DST + 1), "n"(DST + 2), "n"(DST + 3), "n"(OFF), "n"(OFF + 8 * 256));
}
template <int CNT>
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%c0)" ::"n"(CNT));
}
template <int DST>
__device__ __forceinline__ void acc_write(uint32_t v) {
  asm volatile("v_accvgpr_write_b32 a[%c1], %0" ::"v"(v), "n"(DST));
}
template <int SRC>
__device__ __forceinline__ float acc_read() {
  float r;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(r) : "n"(SRC));
  return r;
}
template <int REG>
__device__ __forceinline__ void acc_scale(float f) {   // a[REG] *= f
  float t;
  asm volatile("v_accvgpr_read_b32 %0, a[%c2]\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a[%c2], %0" : "=&v"(t) : "v"(f), "n"(REG));
}
// MFMA result -> accumulator move / accumulator write -> MFMA operand: the hazards the compiler cannot see through the asm
__device__ __forceinline__ void acc_settle() { asm volatile("s_nop 15\n\ts_nop 7" ); }

}  // namespace w4

#ifdef RTV_LAB   // cycle probe of the tile loop (lab build): per workgroup [shader cycles, 100 MHz ticks, tiles] of wave 0
__device__ unsigned long long* g_w4_probe = nullptr;
#endif

// VARL = schedule variant (0..3) + 10 x LAB + 100 x OPT.
// OPT bit 0: RS, row sums by the matrix pipe (see the op table); bit 1: DMA_IMM, one M0 write per operand and tile - a piece's
// 1-KiB step is the instruction's immediate offset, which the hardware adds to the LDS AND the memory address (the per-lane
// memory offsets carry the compensation).
// LAB > 0 (lab build only, timing experiments, garbage results): 1 = no softmax instructions in the tile loop, 2 = exponentials
// replaced by adds, 3 = no fragment reads, 4 = no DMA, 5 = no fragment waits, 6 = matrix instructions only (1 + 3 + 4 + 5).
template <bool F16, int VARL>
__global__ __launch_bounds__(256, 1) void attn_fwd_w4_kernel(AttnParams p) {
  using namespace w4;
  constexpr int VAR = VARL % 10, LAB = (VARL / 10) % 10, OPT = VARL / 100;
  constexpr bool RS = OPT & 1, DMA_IMM = OPT & 2, EPI_LDS = OPT & 4;   // bit 2: output rows through LDS (whole 256-byte rows per store)
  constexpr bool NO_VALU = LAB == 1 || LAB == 6, NO_EXP = LAB == 2, NO_FRAG = LAB == 3 || LAB == 6, NO_DMA = LAB == 4 || LAB == 6,
                 NO_WAIT = LAB == 5 || LAB == 6;
  using SC = Sched<VAR, RS>;
  using PB = PhaseB<RS>;
  constexpr int NB = PB::N;
  static_assert(SC::tab.ok, "softmax schedule violates the P^T hazard");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sK = smem;
  char* const sV = smem + V_BASE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;

  int bh, qt;
  {
    const int nbh = p.B * p.H;
    const int bid = blockIdx.x;
    if (nbh % 8 == 0) {   // all q tiles of a head on ONE XCD (block b runs on XCD b % 8): its K/V stream stays in that L2
      int xcd = bid & 7, slot = bid >> 3;
      bh = xcd + 8 * (slot / p.n_qtiles);
      qt = slot % p.n_qtiles;
    } else {
      bh = bid / p.n_qtiles;
      qt = bid % p.n_qtiles;
    }
  }
  const int b = bh / p.H, h = bh % p.H;
  const int q0 = qt * QT;
  const uint16_t* qb_ = p.q + (size_t)b * p.q_bs + (size_t)h * ATT_D;
  const uint16_t* kb = p.k + (size_t)b * p.k_bs + (size_t)h * ATT_D;
  const uint16_t* vb = p.v + (size_t)b * p.v_bs + (size_t)h * ATT_D;
  uint16_t* ob = p.o + (size_t)b * p.o_bs + (size_t)h * ATT_D;

  // ---- key-prefix limits (block-causal rule kv < ends[q], causal_model.py:134-136), per query block of the lane
  int q_row[2], kv_lim[2];
  int wave_min_lim = p.Lkv, wg_max_lim = p.Lkv;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    q_row[qb] = q0 + wave * QW + qb * 32 + l31;
    kv_lim[qb] = p.Lkv;
  }
  if (p.causal_block > 0) {
    const int cb = p.causal_block;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) kv_lim[qb] = min(p.Lkv, ((p.q_offset + min(q_row[qb], p.Lq - 1)) / cb + 1) * cb);
    const int first = min(q0 + wave * QW, p.Lq - 1);
    wave_min_lim = min(p.Lkv, ((p.q_offset + first) / cb + 1) * cb);
    const int last = min(q0 + QT - 1, p.Lq - 1);
    wg_max_lim = min(p.Lkv, ((p.q_offset + last) / cb + 1) * cb);
  }
  wave_min_lim = __builtin_amdgcn_readfirstlane(wave_min_lim);
  const int ntiles = (wg_max_lim + ATT_KT - 1) / ATT_KT;
  int t_lo, t_hi;
  split_tile_range(p, ntiles, &t_lo, &t_hi);
  t_lo = __builtin_amdgcn_readfirstlane(t_lo);
  t_hi = __builtin_amdgcn_readfirstlane(t_hi);
  const int n_it = t_hi - t_lo;

  // ---- DMA staging: wave w moves tile rows 16w .. 16w+15 of K and of V, four 1-KiB pieces (4 rows) each.  Lane l lands at
  //      byte 16 l of the piece = row (l >> 4), chunk position (l & 15) and fetches the chunk that belongs there:
  //      K: position = chunk ^ (row & 15);  V: position = chunk ^ ((row & 3) << 2)   (the layouts of the fragment reads).
  //      One row range only (the launcher keeps two-range ring windows on the four-phase kernel), so EVERY tile is a per-lane
  //      byte offset fixed for the whole kernel + a uniform tile offset in an SGPR: no per-tile VALU, no branch.  The buffer
  //      descriptors end behind the last row of the window: the rows of the ragged last tile that lie beyond it read as zeros
  //      (finite - their scores are masked, their P is 0); tiles past the end re-read the last tile into a slot nobody reads.
  const int k_rs = __builtin_amdgcn_readfirstlane((int)p.k_rs), v_rs = __builtin_amdgcn_readfirstlane((int)p.v_rs);
  const int w_lkv = __builtin_amdgcn_readfirstlane(p.Lkv);
  const __amdgpu_buffer_rsrc_t rsrcK =
      __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, (int)((uint32_t)(w_lkv - 1) * (uint32_t)k_rs * 2u + 256u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcV =
      __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, (int)((uint32_t)(w_lkv - 1) * (uint32_t)v_rs * 2u + 256u), 0x00020000);
  const int t_last = (w_lkv - 1) / ATT_KT;      // last tile that holds a row of the window
  const int st_row = wave * 16 + (lane >> 4);   // tile row of piece 0 (piece i: + 4 i)
  const int st_cp = lane & 15;
  uint32_t k_fast[4], v_fast[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = st_row + 4 * i;
    k_fast[i] = (uint32_t)(r * k_rs + (st_cp ^ (r & 15)) * 8) * 2u - (DMA_IMM ? 1024u * i : 0u);   // r >= 4 i, row >= 256 bytes
    v_fast[i] = (uint32_t)(r * v_rs + (st_cp ^ ((r & 3) << 2)) * 8) * 2u - (DMA_IMM ? 1024u * i : 0u);
  }
  auto tile_soff = [&](int t, int rs) __attribute__((always_inline)) { return (uint32_t)(min(t, t_last) * ATT_KT * rs) * 2u; };
  const uint32_t lds_wave = (uint32_t)(uintptr_t)(RTV_LDS char*)smem + (uint32_t)wave * 4096u;   // this wave's share of ring slot 0 of K
  // piece i (0-3) of a tile into ring slot `slot` of K (VOP = 0) or V (VOP = 1); `lds_w` = lds_wave, made opaque once per
  // iteration so that the 32 destination addresses of the unrolled loop are an s_add each, not 32 hoisted SGPRs
  auto piece = [&](auto vopc, auto slotc, auto ic, uint32_t lds_w, uint32_t soff) __attribute__((always_inline)) {
    constexpr int VOP = decltype(vopc)::value, SLOT = decltype(slotc)::value, I = decltype(ic)::value;
    constexpr int IMM = DMA_IMM ? I * 1024 : 0;
    RTV_LDS void* dst = (RTV_LDS void*)(uintptr_t)(lds_w + (uint32_t)(VOP * V_BASE + SLOT * ATT_TILE_BYTES + I * 1024 - IMM));
    if constexpr (VOP) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcV, dst, 16, v_fast[I], soff, IMM, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcK, dst, 16, k_fast[I], soff, IMM, 0);
  };
  auto stage_tile = [&](auto vopc, auto slotc, int t) __attribute__((always_inline)) {   // all four pieces (prologue, idle waves)
    const uint32_t soff = tile_soff(t, decltype(vopc)::value ? v_rs : k_rs);
    static_for<0, 4>([&](auto ic) __attribute__((always_inline)) { piece(vopc, slotc, ic, lds_wave, soff); });
  };

  // ---- prologue DMA: K(0), K(1), K(2), V(0) (tile i of this workgroup = global tile t_lo + i lives in ring slot i & 3)
  stage_tile(IntC<0>{}, IntC<0>{}, t_lo);
  stage_tile(IntC<0>{}, IntC<1>{}, t_lo + 1);
  stage_tile(IntC<0>{}, IntC<2>{}, t_lo + 2);
  stage_tile(IntC<1>{}, IntC<0>{}, t_lo);

  // A wave whose 64 query rows all lie beyond Lq keeps its DMA duty and its barriers, nothing else
  const bool idle_rows = p.skip_idle && q0 + wave * QW >= p.Lq;   // wave-uniform
  if (idle_rows) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    auto idle_it = [&](auto cc, int i) __attribute__((always_inline)) {   // iteration i of ring copy C = i & 3
      constexpr int C = decltype(cc)::value;
      stage_tile(IntC<0>{}, IntC<C>{}, t_lo + i + 4);
      stage_tile(IntC<1>{}, IntC<(C + 2) & 3>{}, t_lo + i + 2);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
    idle_it(IntC<3>{}, -1);
    for (int i = 0; i < n_it; i += 4) {
      idle_it(IntC<0>{}, i);
      if (i + 1 < n_it) idle_it(IntC<1>{}, i + 1);
      if (i + 2 < n_it) idle_it(IntC<2>{}, i + 2);
      if (i + 3 < n_it) idle_it(IntC<3>{}, i + 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (EPI_LDS && p.kv_splits <= 1) __syncthreads();   // the working waves' barrier in front of their epilogue image
    return;
  }

  // ---- all 256 accumulation registers belong to the asm statements of this kernel from here on (the clobber list also makes the
  //      kernel descriptor allocate them)
  asm volatile("" ::: RTV_W4_ACC);
  // ---- Q^T fragments (MFMA B operand) into a[128:191]: lane holds Q[q][dc*16 + g*8 .. +8] of both query blocks; O^T = 0
  {
    u32x4 qf[2][8];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const uint16_t* qp = qb_ + (size_t)min(q_row[qb], p.Lq - 1) * p.q_rs + g * 8;
#pragma unroll
      for (int dc = 0; dc < 8; ++dc) qf[qb][dc] = *(const u32x4*)(qp + dc * 16);
    }
    static_for<0, 64>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      acc_write<A_Q + i>(qf[i >> 5][(i >> 2) & 7][i & 3]);
    });
    static_for<0, 128>([&](auto ic) __attribute__((always_inline)) { acc_write<A_O + decltype(ic)::value>(0u); });
  }

  // ---- per-lane LDS read addresses (slot 0; slot / key block / k-step are immediates)
  // K operand (A): row = kbk*32 + l31, chunk = (dc*2 + g) ^ (row & 15)
  uint32_t k_rd32[8];
#pragma unroll
  for (int dc = 0; dc < 8; ++dc)
    k_rd32[dc] = (uint32_t)(uintptr_t)(RTV_LDS const char*)(sK + l31 * 256 + (((dc * 2 + g) ^ (l31 & 15)) << 4));
  // V^T operand (A) via transpose read: a 16-lane group gathers a [4 keys][16 dims] block
  const int i16 = lane & 15, h16 = (lane >> 4) & 1;
  uint32_t v_rd32[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
    v_rd32[db] = (uint32_t)(uintptr_t)(RTV_LDS const char*)(sV + (4 * g + (i16 >> 2)) * 256 + h16 * 32 + (i16 & 3) * 8 +
                                                            ((db ^ (i16 >> 2)) << 6));

  // ---- state
  f32x16 S[2][2][2];     // S^T [buffer][query block][32-key block]
  u32x4 P[2][2][2];      // P^T [query block][32-key block][16-key step]
#pragma unroll
  for (int bf = 0; bf < 2; ++bf)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int r = 0; r < 16; ++r) S[bf][qb][kb2][r] = 0.f;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int s = 0; s < 2; ++s) P[qb][kb2][s] = u32x4{0u, 0u, 0u, 0u};
  float m_run[2] = {-1e30f, -1e30f};   // reference point of the exponentials (>= running max - RESCALE_SLACK), log2 domain
  float l_run[2] = {0.f, 0.f};         // the lane's partial row sums                                   (not with RS)
  f32x16 L[2];                         // RS: l^T accumulators - every element of a lane's 16 = its query's row sum
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int r = 0; r < 16; ++r) L[qb][r] = 0.f;
  const uint32_t one2 = F16 ? 0x3c003c00u : 0x3f803f80u;
  u32x4 ones = {one2, one2, one2, one2};
  asm volatile("" : "+v"(ones));        // (a register quad, not four literals per MFMA)
  float alpha[2] = {1.f, 1.f};         // pending O^T rescale factor of the tile whose softmax has started
  float mx[2] = {0.f, 0.f}, mxc[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, dec_t[2] = {0.f, 0.f}, dec_n[2] = {0.f, 0.f};
  float ps[2][2] = {{0.f, 0.f}, {0.f, 0.f}};   // row-sum partials of the tile in flight (even / odd elements, as the pp kernel)
  float px[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, pe[2][2] = {{0.f, 0.f}, {0.f, 0.f}};   // pipeline registers of the F / E / C stages
  unsigned long long need_any = 0;     // lanes whose row needs the rescale (either block): wave-uniform
  const float c = p.scale_log2e;
  constexpr float RESCALE_SLACK = 8.f;   // lazy rescaling, see the lockstep kernel

  // one instruction of the softmax stream of the tile whose scores are in S[BUF].  Results are pinned behind their instruction
  // (an empty asm that "modifies" them, or the instruction itself as volatile asm): hipcc otherwise sinks the whole softmax out of
  // the matrix phases, sched_barrier notwithstanding.  A pinned value read by a VALU instruction less than two instructions
  // later costs an s_nop (hipcc's pad behind inline asm): the four-chain / two-block interleaving keeps readers three away.
#define RTV_PIN(x) asm volatile("" : "+v"(x))
  auto run_group = [&](auto bufc, auto gc) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
    constexpr Op G = SC::OPS.op[decltype(gc)::value];
    constexpr int qb = G.qb;
    if constexpr (G.kind == OP_MAX) {
      constexpr int k = G.idx, h = G.sub;
      if constexpr (k == 0) {
        mxc[qb][h] = fmaxf(fmaxf(S[BUF][qb][h][0], S[BUF][qb][h][1]), S[BUF][qb][h][2]);
        RTV_PIN(mxc[qb][h]);
      } else if constexpr (k < 7) {
        mxc[qb][h] = fmaxf(fmaxf(mxc[qb][h], S[BUF][qb][h][2 * k + 1]), S[BUF][qb][h][2 * k + 2]);
        RTV_PIN(mxc[qb][h]);
      } else {
        mxc[qb][h] = vmax(mxc[qb][h], S[BUF][qb][h][15]);
      }
    } else if constexpr (G.kind == OP_CMB) {
      mx[qb] = vmax(mxc[qb][0], mxc[qb][1]);
    } else if constexpr (G.kind == OP_XCH0) {
      const unsigned u = __float_as_uint(mx[qb]);
      const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // {lanes: [lo, lo], [hi, hi]}
      mx[qb] = __uint_as_float(sw[0]);
      dec_t[qb] = __uint_as_float(sw[1]);
      RTV_PIN(mx[qb]);
      RTV_PIN(dec_t[qb]);
    } else if constexpr (G.kind == OP_XCH1) {
      mx[qb] = vmax(mx[qb], dec_t[qb]);
    } else if constexpr (G.kind == OP_DEC) {
      // wave-uniform rare branch elsewhere, per-row decision here: rows that do not need it get alpha = exp2(0) = 1 exactly,
      // so a row's arithmetic never depends on which other rows share its wave (token-sharded == unsharded, bit for bit)
      constexpr int k = G.idx;
      if constexpr (k == 0) {
        dec_t[qb] = mx[qb] * c;
        RTV_PIN(dec_t[qb]);
      } else if constexpr (k == 1) {
        dec_t[qb] = vmax(m_run[qb], dec_t[qb]);                  // m_cand
      } else if constexpr (k == 2) {
        mx[qb] = dec_t[qb] - m_run[qb];
        RTV_PIN(mx[qb]);
      } else if constexpr (k == 3) {
        const bool need = mx[qb] > RESCALE_SLACK;
        need_any |= __builtin_amdgcn_ballot_w64(need);
        dec_n[qb] = need ? dec_t[qb] : m_run[qb];                 // the new reference point
        RTV_PIN(dec_n[qb]);
      } else if constexpr (k == 4) {
        mx[qb] = m_run[qb] - dec_n[qb];
        RTV_PIN(mx[qb]);
      } else if constexpr (k == 5) {
        alpha[qb] = __builtin_amdgcn_exp2f(mx[qb]);
        m_run[qb] = dec_n[qb];
        RTV_PIN(alpha[qb]);
      } else {
        l_run[qb] *= alpha[qb];
        RTV_PIN(l_run[qb]);
      }
    } else if constexpr (G.kind == OP_F) {
      constexpr int w = G.idx >> 1, kbk = w >> 3, t = w & 7, st = G.idx & 1;   // stage registers alternate per pair
      px[st][G.sub] = __builtin_fmaf(S[BUF][qb][kbk][2 * t + G.sub], c, -m_run[qb]);
      RTV_PIN(px[st][G.sub]);
    } else if constexpr (G.kind == OP_E) {
      constexpr int st = G.idx & 1;
      if constexpr (NO_EXP) pe[st][G.sub] = px[st][G.sub] + 1.0f;
      else pe[st][G.sub] = __builtin_amdgcn_exp2f(px[st][G.sub]);
      RTV_PIN(pe[st][G.sub]);
    } else if constexpr (G.kind == OP_A) {
      constexpr int st = G.idx & 1;
      ps[qb][G.sub] += pe[st][G.sub];
      RTV_PIN(ps[qb][G.sub]);
    } else if constexpr (G.kind == OP_CV) {
      constexpr int w = G.idx >> 1, kbk = w >> 3, t = w & 7, st = G.idx & 1;
      uint32_t pk = pack2<F16>(pe[st][0], pe[st][1]);
      RTV_PIN(pk);
      P[qb][kbk][t >> 2][t & 3] = pk;
    } else if constexpr (G.kind == OP_SUM) {
      if constexpr (G.idx == 0) {
        ps[qb][0] += ps[qb][1];
        RTV_PIN(ps[qb][0]);
      } else {
        l_run[qb] += ps[qb][0];
        RTV_PIN(l_run[qb]);
        ps[qb][0] = 0.f;
        ps[qb][1] = 0.f;
      }
    }
  };
#undef RTV_PIN
  // the instructions of slot SLOT (0 .. NB-1: phase B, NB .. NB+31: phase A) on the scores in S[BUF]
  auto run_slot = [&](auto bufc, auto slotc) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slotc)::value;
    if constexpr (!NO_VALU) static_for<SC::begin(SLOT), SC::begin(SLOT + 1)>([&](auto gc) __attribute__((always_inline)) { run_group(bufc, gc); });
  };
  // mask of the tile whose scores are in S[BUF] (global tile t): only on tiles that cross a limit of this wave.  Branch-free
  // integer form - s = min(s, kv < lim ? +inf : -inf) - so that the 64 decisions do not become 64 SGPR pairs
  auto mask_tile = [&](auto bufc, int t, bool all) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
    int g4 = 4 * g + t * ATT_KT;
    asm volatile("" : "+v"(g4));   // rare path: its per-lane key indices must not be hoisted out of the tile loop
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int lim1 = (all ? 0 : kv_lim[qb]) - 1 - g4;    // key kvi is masked iff lim - 1 - kvi < 0
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int d = lim1 - (kbk * 32 + (r & 3) + 8 * (r >> 2));
          const uint32_t w = ((uint32_t)(d >> 31) & 0x80000000u) | 0x7f800000u;   // -inf where masked, +inf elsewhere
          S[BUF][qb][kbk][r] = fminf(S[BUF][qb][kbk][r], __uint_as_float(w));
        }
    }
  };

  // ---- phase A of iteration i (ring copy C = i & 3): S[NX] = K(i+1) . Q^T, fillers: V^T(i) reads, second half of softmax(i)
  auto phase_a = [&](auto cc, auto fillc) __attribute__((always_inline)) {
    constexpr int C = decltype(cc)::value, CUR = C & 1, NX = CUR ^ 1;
    constexpr bool FILL = decltype(fillc)::value;
    uint32_t va[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) va[db] = v_rd32[db];
    static_for<0, 32>([&](auto nc) __attribute__((always_inline)) {
      constexpr int n = decltype(nc)::value, dc = n >> 2, kbk = (n >> 1) & 1, qb = n & 1, f = n >> 1;
      if constexpr (wait_a(n) >= 0 && !(NO_WAIT && FILL)) lds_wait<wait_a(n)>();
      mfma_qk<F16, A_KV + 4 * f, A_Q + (qb * 8 + dc) * 4, dc == 0>(S[NX][qb][kbk]);
      if constexpr (FILL) {
        __builtin_amdgcn_sched_barrier(0);
        // V^T fragment idx = (kbk*2 + s)*4 + db of tile i (ring slot C) into the registers K fragment idx has just left
        if constexpr ((n & 1) && !NO_FRAG) {
          constexpr int idx = n >> 1;
          constexpr int off = C * ATT_TILE_BYTES + ((idx >> 3) * 32 + ((idx >> 2) & 1) * 16) * 256;
          lds_tr_read_a<A_KV + 4 * idx, off>(va[idx & 3]);
        }
        run_slot(IntC<CUR>{}, IntC<NB + n>{});
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- phase B of iteration i: O^T += V^T(i) . P^T(i) (PV = false: the virtual iteration in front of tile 0), fillers:
  //      K(i+2) fragment reads, the DMA pieces of K(i+4) / V(i+2), first half of softmax(i+1) on S[NX]
  auto phase_b = [&](auto cc, auto pvc, uint32_t lds_w, uint32_t soff_k, uint32_t soff_v) __attribute__((always_inline)) {
    constexpr int C = decltype(cc)::value, CUR = C & 1, NX = CUR ^ 1;
    constexpr bool PV = decltype(pvc)::value;
    uint32_t ka[8];
#pragma unroll
    for (int dc = 0; dc < 8; ++dc) ka[dc] = k_rd32[dc];
    static_for<0, NB>([&](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value, step = PB::step(m), qb = PB::qb(m);
      if constexpr (PV) {
        if constexpr (wait_b<RS>(m) >= 0 && !NO_WAIT) lds_wait<wait_b<RS>(m)>();
        if constexpr (PB::is_pv(m)) mfma_pv<F16, A_O + (qb * 4 + PB::db(m)) * 16, A_KV + 4 * PB::frag(m)>(P[qb][step >> 1][step & 1]);
        else mfma_rowsum<F16>(L[qb], ones, P[qb][step >> 1][step & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (PB::kread(m) >= 0 && !(NO_FRAG && PV)) {   // K fragment [dc][kbk] = n of tile i+2 (ring slot (C + 2) & 3) into the
        constexpr int n = PB::kread(m);                         // registers V^T fragment n has just left
        lds_read128_a<A_KV + 4 * n, ((C + 2) & 3) * ATT_TILE_BYTES + (n & 1) * 32 * 256>(ka[n >> 1]);
      }
      if constexpr (SC::dma_piece(m) >= 0 && !(NO_DMA && PV)) {
        constexpr int i = SC::dma_piece(m);
        if constexpr (i < 4) piece(IntC<0>{}, IntC<C>{}, IntC<i>{}, lds_w, soff_k);                 // K(i+4) into the slot of K(i)
        else piece(IntC<1>{}, IntC<(C + 2) & 3>{}, IntC<i - 4>{}, lds_w, soff_v);                    // V(i+2)
      }
      run_slot(IntC<NX>{}, IntC<m>{});
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // the lazy rescale of O^T, decided beside PV(i), applied behind it (rare: the first tiles of a row)
  auto rescale_o = [&]() __attribute__((always_inline)) {
    if (need_any != 0) {
      acc_settle();
      static_for<0, 128>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        acc_scale<A_O + i>(alpha[i >> 6]);
      });
      if constexpr (RS) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int r = 0; r < 16; ++r) L[qb][r] *= alpha[qb];
      }
      acc_settle();
      need_any = 0;
    }
  };
  auto end_of_iteration = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NO_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // everything but this iteration's 8 pieces has landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // iteration i of ring copy C: tile t = t_lo + i
  auto iteration = [&](auto cc, int i) __attribute__((always_inline)) {
    constexpr int C = decltype(cc)::value, NX = (C & 1) ^ 1;
    const int t = t_lo + i;
    phase_a(cc, IntC<1>{});
    if ((t + 2) * ATT_KT > wave_min_lim || i + 1 >= n_it) mask_tile(IntC<NX>{}, t + 1, i + 1 >= n_it);
    uint32_t lds_w = lds_wave;
    asm volatile("" : "+s"(lds_w));
    phase_b(cc, IntC<1>{}, lds_w, tile_soff(t + 4, k_rs), tile_soff(t + 2, v_rs));
    rescale_o();
    end_of_iteration();
  };

  // ---- prologue: everything staged so far has landed; K(0) fragments; the virtual iteration -1 (ring copy 3): S[0] = K(0) . Q^T,
  //      then phase B without PV = K(1) fragment reads, DMA of K(3) / V(1), first half of softmax(0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  static_for<0, 16>([&](auto nc) __attribute__((always_inline)) {
    constexpr int n = decltype(nc)::value;
    lds_read128_a<A_KV + 4 * n, (n & 1) * 32 * 256>(k_rd32[n >> 1]);
  });
  lds_wait<0>();
  acc_settle();   // accumulator writes (Q^T, O^T = 0) -> first MFMA
  phase_a(IntC<3>{}, IntC<0>{});
  // An MFMA result may be read by a VALU instruction 11 wait states behind the MFMA at the earliest, and the compiler does not know
  // that the asm statements above are MFMAs.  In the tile loop the first reader of S^T(j+1) sits behind the fillers of the last
  // matrix slot, the scalar limit test, a wait and a PV MFMA that cannot issue before the last QK MFMA has left the pipe; here,
  // with no PV in between, the distance is made explicit.
  acc_settle();
  if ((t_lo + 1) * ATT_KT > wave_min_lim) mask_tile(IntC<0>{}, t_lo, false);
  phase_b(IntC<3>{}, IntC<0>{}, lds_wave, tile_soff(t_lo + 3, k_rs), tile_soff(t_lo + 1, v_rs));
  need_any = 0;   // O^T is still zero: nothing to rescale
  end_of_iteration();

#ifdef RTV_LAB
  const unsigned long long probe_c0 = __builtin_readcyclecounter(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- the tile loop, unrolled by the ring length
  for (int i = 0; i < n_it; i += 4) {
    iteration(IntC<0>{}, i);
    if (i + 1 >= n_it) break;
    iteration(IntC<1>{}, i + 1);
    if (i + 2 >= n_it) break;
    iteration(IntC<2>{}, i + 2);
    if (i + 3 >= n_it) break;
    iteration(IntC<3>{}, i + 3);
  }
#ifdef RTV_LAB
  if (g_w4_probe != nullptr && wave == 0 && lane == 0 && blockIdx.x < 1024) {
    g_w4_probe[blockIdx.x * 3 + 0] = __builtin_readcyclecounter() - probe_c0;
    g_w4_probe[blockIdx.x * 3 + 1] = __builtin_amdgcn_s_memrealtime() - probe_r0;
    g_w4_probe[blockIdx.x * 3 + 2] = (unsigned long long)n_it;
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pieces of the tiles past the end must not outlive the workgroup's LDS

  // ---------------- epilogue: O = O^T / l, lane owns row q and dims db*32 + 8*i + 4*g + {0..3}
  acc_settle();
  static_for<0, 2>([&](auto qc) __attribute__((always_inline)) {
    constexpr int qb = decltype(qc)::value;
    f32x16 oq[4];
    static_for<0, 64>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      oq[i >> 4][i & 15] = acc_read<A_O + qb * 64 + i>();
    });
    float l_tot;
    if constexpr (RS) l_tot = L[qb][0];   // the matrix pipe has summed all 64 keys of every tile, both lane halves
    else l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
    if (p.kv_splits > 1) {
      store_partial(p, bh, q_row[qb], g, oq, m_run[qb], l_tot);
    } else if constexpr (EPI_LDS) {
      // Through a wave-private LDS image of 64 rows x 256 bytes (the K / V rings are dead; this wave's last fragment reads and
      // DMA pieces have retired - vmcnt(0) above, its reads were consumed - and the image lies in the wave's OWN quarter of the K
      // ring only if no other wave still reads it: hence the workgroup barrier in front of the first block).  The register layout
      // gives a lane one row and 4 consecutive dims per quad: stored directly that is 8-byte accesses at a row stride, 32
      // different cache lines per instruction; from the image every lane stores 16 contiguous bytes and 16 lanes a whole row.
      // 16-byte chunk c of row r at chunk c ^ (r & 15): the 8-byte writes of 16 consecutive rows and the row-contiguous 16-byte
      // reads are both conflict-free.
      const float inv = 1.0f / l_tot;
      char* img = smem + wave * (64 * 256);
      if constexpr (qb == 0) __syncthreads();
      const int row = qb * 32 + l31;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u32x2 w;
          w[0] = pack2<F16>(oq[db][4 * i + 0] * inv, oq[db][4 * i + 1] * inv);
          w[1] = pack2<F16>(oq[db][4 * i + 2] * inv, oq[db][4 * i + 3] * inv);
          *(u32x2*)(img + row * 256 + (((db * 4 + i) ^ (row & 15)) << 4) + g * 8) = w;
        }
      if constexpr (qb == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const int rsub = lane >> 4, cpos = lane & 15;
#pragma unroll
        for (int ps = 0; ps < 16; ++ps) {
          const int r = ps * 4 + rsub;
          const u32x4 t = *(const u32x4*)(img + r * 256 + ((cpos ^ (r & 15)) << 4));
          const int qr = q0 + wave * QW + r;
          if (qr < p.Lq) *(u32x4*)(ob + (size_t)qr * p.o_rs + cpos * 8) = t;
        }
      }
    } else {
      const float inv = 1.0f / l_tot;
      if (q_row[qb] < p.Lq) {
        uint16_t* op = ob + (size_t)q_row[qb] * p.o_rs + 4 * g;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            u32x2 w;
            w[0] = pack2<F16>(oq[db][4 * i + 0] * inv, oq[db][4 * i + 1] * inv);
            w[1] = pack2<F16>(oq[db][4 * i + 2] * inv, oq[db][4 * i + 3] * inv);
            *(u32x2*)(op + db * 32 + i * 8) = w;
          }
      }
    }
  });
}

#ifdef RTV_LAB
}  // namespace rtv
extern "C" int rtv_attn_w4_probe(unsigned long long* buf) {   // lab: [1024][3] u64, or null to switch the probe off
  return hipMemcpyToSymbol(HIP_SYMBOL(rtv::g_w4_probe), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
namespace rtv {
#endif

// ---- launcher (called by attn_fwd_impl, attn_fwd.hip): p.n_qtiles = ceil(Lq / 256), grid and split set up by the caller
int launch_attn_w4(const AttnParams& p, bool f16, int variant, dim3 grid, hipStream_t stream) {
  static LdsAttr attr[800];
  if (variant < 0 || variant >= 800) return set_error(-1, "attn_w4: variant out of range");
  if (f16) return set_error(-1, "attn_w4: bf16 only (f16 launches stay on the four-phase kernel)");
#define RTV_W4_CASE(V)                                                                                        \
  case V: {                                                                                                   \
    const void* kp = (const void*)attn_fwd_w4_kernel<false, V>;                                               \
    if (int st = ensure_dynamic_lds(kp, w4::LDS_BYTES, &attr[V], "attn_w4")) return st;                       \
    hipLaunchKernelGGL((attn_fwd_w4_kernel<false, V>), grid, dim3(256), w4::LDS_BYTES, stream, p);            \
    break;                                                                                                    \
  }
  switch (variant) {
    RTV_W4_CASE(600)   // 200 + output rows through LDS
    RTV_W4_CASE(200)   // plain row sums (bit-identical with the four-phase kernel), one M0 write per operand
    RTV_W4_CASE(0)     // the same with one M0 write per DMA piece
#ifdef RTV_LAB
    RTV_W4_CASE(100)
    RTV_W4_CASE(300)
    RTV_W4_CASE(1)
    RTV_W4_CASE(2)
    RTV_W4_CASE(3)
    RTV_W4_CASE(101)
    RTV_W4_CASE(102)
    RTV_W4_CASE(103)
    RTV_W4_CASE(301)
    RTV_W4_CASE(302)
    RTV_W4_CASE(310)
    RTV_W4_CASE(320)
    RTV_W4_CASE(330)
    RTV_W4_CASE(340)
    RTV_W4_CASE(360)
#endif
    default:
      return set_error(-1, "attn_w4: variant = schedule (0..3) + 10 x lab experiment + 100 x options; this build has 200 and 0 "
                           "(lab build: more)");
  }
#undef RTV_W4_CASE
  return 0;
}

}  // namespace rtv
