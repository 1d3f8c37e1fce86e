// In-launch split-K of the last, partial round of 256x256 output tiles (gemm8.hip).
//
// T tiles on G CUs run as floor(T/G) full rounds plus R = T % G tiles; those R tiles would keep the chip at R/G
// occupancy for a whole tile time.  Instead each of them is cut along K into S segments ("units", S = min(8, G/R)),
// so the tail round runs R*S workgroups for 1/S of a tile time (with T < G every tile is a tail tile).  Partial accumulators go to an fp32 slab in a
// caller-provided workspace (rtv_gemm_set_workspace); the last arriver of a tile (agent-scope release / acquire around
// an arrival counter, no spinning) sums the slabs in a FIXED order (arrival order never shows in the result), resets the
// counter and runs the fused epilogue.
#pragma once
#include "rtv_common.h"

namespace rtv {

constexpr int SPLIT_SLAB_FLOATS = 256 * 256;  // fp32 partial tile (256 KiB)
constexpr int SPLIT_MAX_UNITS = 256;

struct SplitArgs {
  int first_unit;   // block ids >= first_unit are split units; full tiles before
  int S;            // K segments per split tile (1 = no splitting)
  float* slabs;     // [units][SPLIT_SLAB_FLOATS]
  int* counters;    // [split tiles]: zero between launches (zeroed when the workspace is attached; the reducer of a
                    // tile puts its counter back to zero)
  int tail_tiles;   // R: number of split tiles (block ids first_unit .. first_unit + R * S)
  int split_first;  // 1: the split units get the lowest block ids (dispatched first): chosen when there are few of them
  int half_tail;    // 1 (gemm8 only): the R tail tiles run as 2 R units of 128 x 256 (rows 0-127 / 128-255 of the tile, full K, the
                    // 128-row ping-pong body) instead of K segments: no slabs, no reduction; S is 1 then
  int pair_units;   // > 0 (gemm8 only, r04): the ragged LAST row of tiles (<= 128 real rows) is not part of the tile grid (tiles_m is
                    // one less); it runs as `pair_units` strips of 128 x 512 - two horizontally adjacent half tiles through the 128-row
                    // body, one after the other - in block ids [0, pair_units); ids up to pair_pad (a multiple of 8) are padding
                    // and every later id is shifted by pair_pad
  int pair_pad;
  int pair_m0;      // first row of the ragged row of tiles
};

// host: decide the split for T tiles of nk K-tiles each (defined in gemm8.hip, which owns the workspace pointers).
// Fills *sp / *grid.  Returns 0 or an error status.
// `extra_units`: workgroup slots of about one tile time that run beside the T tiles (gemm8's ragged-row strips incl. their padding
// ids): they count for the round arithmetic (R = (T + extra_units) % G) but are not tiles.
int plan_split_k(int T, int nk, bool allow_split, SplitArgs* sp, int* grid, hipStream_t stream, bool allow_half = false,
                 int extra_units = 0);

// device: block id -> (tile, K segment).  Returns true when this workgroup is a split unit.
// The units of the tail round are laid out for L2 locality like the full tiles are: every XCD gets a contiguous chunk of the
// (segment-major) unit list, i.e. ~R*S/8 consecutive tiles of ONE K segment - a compact block of the supertile order whose
// workgroups share A / W panels through that XCD's L2.  (Unit -> XCD round robin, the round-1 layout, put the two K halves of
// a tile and unrelated tiles on every XCD: each unit streamed its own panels, ~11 TB/s of L2 misses for the 232 units of the
// QKV projection, and the split tail was slower than an unsplit fifth round: profiles/r02_gemm_shapes_*.log.)
__device__ __forceinline__ bool split_unit_of_block(const SplitArgs& sp, int bid, int nk_total, int* tile_id, int* unit,
                                                    int* seg, int* kt_begin, int* kt_end) {
  *seg = 0;
  *unit = -1;
  if (sp.split_first) {
    const int nsu = sp.tail_tiles * sp.S;
    bid = bid < nsu ? sp.first_unit + bid : bid - nsu;
  }
  if (bid < sp.first_unit) {
    *tile_id = xcd_remap(bid, sp.first_unit);
  } else {
    const int v = xcd_remap(bid - sp.first_unit, sp.tail_tiles * sp.S);
    *seg = v / sp.tail_tiles;
    const int tl = v - *seg * sp.tail_tiles;
    *tile_id = sp.first_unit + tl;
    *unit = tl * sp.S + *seg;      // slab index: the S slabs of a tile are adjacent
  }
  const bool is_split = *unit >= 0 && sp.S > 1;
  *kt_begin = is_split ? (int)((long)nk_total * *seg / sp.S) : 0;
  *kt_end = is_split ? (int)((long)nk_total * (*seg + 1) / sp.S) : nk_total;
  return is_split;
}

// device: publish this unit's partial 128x64-per-wave accumulators; the last arriver of the tile returns true with
// the full sum in `acc` (placement-independent: the slab is a per-lane register image, the reduce is elementwise).
// `smem` needs 4 free bytes at offset 0 (the K loop is over).  8 waves, acc = [4][2] blocks of 32x32 per wave.
template <int TM>   // 32x32 blocks per wave: TM x 2 (the 256-row kernel: 4, the 128-row kernel: 2)
__device__ __forceinline__ bool split_k_reduce(f32x16 (&acc)[TM][2], const SplitArgs& sp, int unit, int seg, int tile_id,
                                               char* smem, int tid, int wave, int lane) {
  constexpr int NB = TM * 2;
  // Publish with WRITE-THROUGH (sc1) 16-byte stores + a per-wave drain, then ONE relaxed agent-scope ticket: no release
  // fence.  A release fence is `buffer_wbl2`, which writes back every dirty line of the XCD's L2 - at the end of a GEMM that
  // is megabytes of other workgroups' freshly written output tiles (measured: the 16 split units of the FFN-in GEMM cost
  // 64 us instead of ~30; profiles/r02_gemm_shapes.log).  The reducer takes one agent-scope acquire and reads plain.
  __amdgpu_buffer_rsrc_t slab =
      __builtin_amdgcn_make_buffer_rsrc((void*)(sp.slabs + (size_t)unit * SPLIT_SLAB_FLOATS), 0, SPLIT_SLAB_FLOATS * 4, 0x00020000);
#pragma unroll
  for (int blk = 0; blk < NB; ++blk)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x16& a = acc[blk >> 1][blk & 1];
      const f32x4 v = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), slab, (((wave * NB + blk) * 4 + q) * 64 + lane) * 16, 0,
                                             /*aux: sc1*/ 16);
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
  __syncthreads();
  int* flag = (int*)smem;
  if (tid == 0)
    *flag = __hip_atomic_fetch_add(sp.counters + (tile_id - sp.first_unit), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int ticket = *flag;
  if (ticket != sp.S - 1) return false;  // not the last arriver: done
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // all S units have arrived: nobody touches this counter again in this launch -> leave it zero for the next one
    // (saves the per-launch memset node: 3400 x 5 us per block of the 14B model)
    __hip_atomic_store(sp.counters + (tile_id - sp.first_unit), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int unit0 = unit - seg;
  // The order of the sum must not depend on WHICH unit arrived last (fp32 addition is not associative; a context-parallel rank on
  // another GPU, or the same launch tomorrow, sees another arrival order).  S == 2: own + other is the same number either way.
  // S > 2: this unit's own slab is in memory as well (it was published before the ticket was drawn), so the reducer starts from
  // zero and adds all S slabs in index order.
  if (sp.S > 2) {
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[blk >> 1][blk & 1][r] = 0.f;
  }
  for (int s = 0; s < sp.S; ++s) {
    if (s == seg && sp.S == 2) continue;
    const float4* other = (const float4*)(sp.slabs + (size_t)(unit0 + s) * SPLIT_SLAB_FLOATS);
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = other[((wave * NB + blk) * 4 + q) * 64 + lane];
        f32x16& a = acc[blk >> 1][blk & 1];
        a[4 * q] += v.x;
        a[4 * q + 1] += v.y;
        a[4 * q + 2] += v.z;
        a[4 * q + 3] += v.w;
      }
  }
  return true;
}

}  // namespace rtv
