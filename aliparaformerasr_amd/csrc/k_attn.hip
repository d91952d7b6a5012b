// k_attn.hip — softmax(q k^T) v for the SAN-M self-attention (T x T) and the decoder's
// cross-attention (L x T); head dim 128, f16 operands, fp32 softmax / accumulation.
//
// Replaces the MatMul-Softmax-MatMul subgraphs ONNX Runtime executes inside
// InferenceSession.Run (AliParaformerAsr/OfflineProjOfParaformer.cs:68).  The reference
// feeds speech_lengths = Tmax for every row (OfflineProjOfParaformer.cs:56-60, quirk Q2),
// so the additive (1-mask)*-10000 term of the graph is identically zero and is not
// materialised; only the tile tail beyond Lk is masked.
//
// Structure (flash attention, one pass over K/V, nothing T x T touches HBM):
//   grid  = (ceil(Lq/128), B*H); block = 4 wavefronts; each wavefront owns 32 query rows.
//   K/V tiles of 64 keys are staged HBM -> LDS with global_load_lds_dwordx4, double buffered.
//   S^T = K Q^T  (v_mfma_f32_32x32x16_f16, A = K rows from LDS, B = Q rows held in VGPRs):
//        the C layout then gives every lane ONE query (column) and 16 of the 32 keys, so the
//        row max / row sum are in-lane reductions plus one exchange with lane^32.
//   O^T = V^T P^T: the f16 P fragment is exactly the lane's own S registers (no cross-lane
//        movement); V^T fragments come from the row-major V tile via ds_read_b64_tr_b16.
//   The online-softmax rescale of O^T is a per-lane scalar multiply.
// LDS swizzles (applied on the global_load_lds SOURCE address and on the read address):
//   K: 16-byte chunk ^= (key & 15)          -> conflict-free ds_read_b128 of 32 keys
//   V: 16-byte chunk ^= ((key & 3) << 2)    -> the 4 key rows of a tr-read hit 4 bank quarters
#include "kernels.h"

#include <cstdlib>
#include <mutex>

namespace pf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f16x __attribute__((ext_vector_type(16)));

#define ATT_DK 128
// compile-time ablation bits for tools/attn_abl.sh (timing experiments only; 0 in every shipped build):
// 1 no exp (P = t), 2 no PV MFMAs/V reads, 4 no QK, 8 no steady-state DMA, 16 no O rescale / max tracking
#ifndef ATT_ABL
#define ATT_ABL 0
#endif
#ifndef ATT_DMA
#define ATT_DMA 0
#endif
#define ATT_BQ 128
#define ATT_BK 64
#define ATT_TILE_BYTES (ATT_BK * ATT_DK * 2)      // 16 KiB
#define ATT_STAGE_BYTES (2 * ATT_TILE_BYTES)      // one ring stage: K tile | V tile

// -DATT_TIMING (tools/attn_timing.sh; timing experiments only): every wave sums, per phase of the tile loop, the shader-clock time
// it spent there (s_memtime differences, scalar) and leaves the sums in att_tm[workgroup][wave][phase]:
//   0 counted DMA wait | 1 workgroup barrier | 2 issue of the next tile's DMA | 3 S^T = K Q^T issue (+ K fragment reads) |
//   4 tile-tail mask + online softmax (first use of S: includes the wait for the MFMAs) | 5 O^T += V^T P^T (V reads, issue) |
//   6 prologue (Q fragments, first stages) | 7 whole kernel
#ifdef ATT_TIMING
__device__ unsigned long long att_tm[1024 * 8 * 8];
#define ATT_TS(i) { const unsigned long long t_ = __builtin_readcyclecounter(); tm_acc[i] += (unsigned)(t_ - t_prev); t_prev = t_; }
#else
#define ATT_TS(i)
#endif

struct AttnDev {
  const half_t* q; const half_t* k; const half_t* v; half_t* o;
  int64_t q_bs, k_bs, v_bs, o_bs;
  int q_rs, k_rs, v_rs, o_rs;
  int H, Lq, Lk;
  float* range;      // null, or 256 {min, max} pairs (int8 path: the range of the context for its DynamicQuantizeLinear)
  // Q and K in ONE blocked [rows, 8 * blk_groups] matrix (kernels.h; written by gemm_qkvp_kernel): q = k = its base, batch b
  // starts at row b * blk_brows, head h's Q columns are groups h * 16 .., its K columns groups blk_kgrp + h * 16 ..
  int blk, blk_groups, blk_brows, blk_kgrp;
};

typedef float f2 __attribute__((ext_vector_type(2)));
template <int V>
struct att_ic { static constexpr int value = V; };

__device__ __forceinline__ void att_glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// Per wave and tile kt: S(kt) = K(kt) Q^T, softmax arithmetic, O^T += V(kt)^T P(kt)^T.  K and V tiles live in a ring of
// NS stages with NS - 1 tiles in flight (round 4: with two stages the only tile in flight had exactly one tile's compute
// time — about a microsecond — to arrive from HBM, where the QKV GEMM's write-through stores left it, and every tile
// began with an exposed wait), one workgroup barrier per tile.  (Issuing
// S(kt+1) ahead of the softmax of S(kt) was tried: +32 VGPRs, no gain — the two workgroups per CU already
// interleave their phases.)
// NW: wavefronts per workgroup (32 queries each).  4: two workgroups per CU; 8: one workgroup of 256 queries per CU — the
// K/V tiles of a (batch, head) are then staged by half as many workgroups (half the L2 -> LDS traffic).
template <int NW, int NS>
__global__ __launch_bounds__(64 * NW, (NS > 2 || NW == 8) ? NW / 4 : 2) void attn_kernel(AttnDev p) {
  constexpr int PPW = 16 / NW;                       // 1 KiB staging pieces (4 key rows) per wave, tile and operand
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5, lc = lane & 31;
#ifdef ATT_TIMING
  unsigned tm_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_begin = __builtin_readcyclecounter();
  unsigned long long t_prev = t_begin;
#endif
  // XCD-aware block order: hardware deals consecutive workgroup ids round-robin over the 8 XCDs; remap
  // so that the query tiles of one (batch, head) — which stream the same K/V — are consecutive on ONE
  // XCD and share its L2 (otherwise every K/V byte is fetched from HBM once per query tile)
  const int nqt = gridDim.x, total = gridDim.x * gridDim.y;
  int lid = blockIdx.x + nqt * blockIdx.y;
  {
    const int q8 = total >> 3, r8 = total & 7, xcd = lid & 7, idx = lid >> 3;
    lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  const int bh = lid / nqt, qt = lid - bh * nqt;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = qt * (32 * NW) + wave * 32;

  const half_t* qb = p.q + b * p.q_bs + h * ATT_DK;
  const half_t* kb_ = p.k + b * p.k_bs + h * ATT_DK;
  const half_t* vb = p.v + b * p.v_bs + h * ATT_DK;
  half_t* ob = p.o + b * p.o_bs + h * ATT_DK;

  // ---- Q fragments (B operand): lane supplies Q[q0 + lc][16*ds + 8*lh + 0..7]
  h8 qf[8];
  {
    int qr = q0 + lc;
    qr = qr < p.Lq ? qr : p.Lq - 1;
    if (p.blk) {
      // blocked: the 32 lanes of a half-wave read one 512-byte block per fragment
      const int64_t R = (int64_t)b * p.blk_brows + qr;
      const half_t* qp = p.q + (((R >> 5) * p.blk_groups + h * 16 + lh) * 32 + (R & 31)) * 8;
#pragma unroll
      for (int ds = 0; ds < 8; ++ds) qf[ds] = *reinterpret_cast<const h8*>(qp + ds * 512);
    } else {
      const half_t* qp = qb + (int64_t)qr * p.q_rs + 8 * lh;
#pragma unroll
      for (int ds = 0; ds < 8; ++ds) qf[ds] = *reinterpret_cast<const h8*>(qp + 16 * ds);
    }
  }

  // ---- staging: wave-instruction i of this wave covers tile rows (wave*4+i)*4 .. +3
  // (blocked K: piece pc = wave*PPW + i is key half pc >> 3, column groups 2 * (pc & 7) + (lane >> 5), key lane & 31 —
  // 1 KiB that is contiguous in memory when the half does not straddle a 32-row block, and whose LDS image
  // [half][group][key][16 B] is read conflict-free without a swizzle)
  const int srow = lane >> 4, schunk = lane & 15;
  unsigned k_src[PPW], v_src[PPW];                 // byte offsets inside a 64-key tile
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int row = (wave * PPW + i) * 4 + srow;
    k_src[i] = (unsigned)(row * p.k_rs + ((schunk ^ (row & 15)) << 3)) * 2u;
    v_src[i] = (unsigned)(row * p.v_rs + ((schunk ^ ((row & 3) << 2)) << 3)) * 2u;
  }
  auto blk_k_addr = [&](int i, int kt) __attribute__((always_inline)) -> const char* {
    const int pc = wave * PPW + i;
    int key = kt * ATT_BK + (pc >> 3) * 32 + (lane & 31);
    key = key < p.Lk ? key : p.Lk - 1;             // tail: re-read the last valid key (masked in the softmax)
    const int64_t R = (int64_t)b * p.blk_brows + key;
    const int grp = p.blk_kgrp + h * 16 + 2 * (pc & 7) + (lane >> 5);
    return reinterpret_cast<const char*>(p.k) + (((R >> 5) * p.blk_groups + grp) * 32 + (R & 31)) * 16;
  };
  auto stage_k = [&](int slot, int kt) __attribute__((always_inline)) {
    char* kl = smem + slot * ATT_STAGE_BYTES;
    if (p.blk) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) att_glds16(blk_k_addr(i, kt), kl + (wave * PPW + i) * 1024);
      return;
    }
    const char* kg = reinterpret_cast<const char*>(kb_ + (int64_t)kt * ATT_BK * p.k_rs);
    if ((kt + 1) * ATT_BK <= p.Lk) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) att_glds16(kg + k_src[i], kl + (wave * PPW + i) * 1024);
    } else {
      // tail tile: rows >= Lk are re-reads of row Lk-1.  They are masked out of the softmax anyway, but
      // what lies behind the last key in memory is not ours (another utterance, or stale workspace bytes
      // of another dtype, i.e. possibly NaN/Inf bit patterns) and 0 * NaN would poison the P V product.
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int over = kt * ATT_BK + (wave * PPW + i) * 4 + srow - (p.Lk - 1);     // rows to step back
        att_glds16(kg + (k_src[i] - (unsigned)((over > 0 ? over : 0) * p.k_rs * 2)), kl + (wave * PPW + i) * 1024);
      }
    }
  };
  auto stage_v = [&](int slot, int kt) __attribute__((always_inline)) {
    char* vl = smem + slot * ATT_STAGE_BYTES + ATT_TILE_BYTES;
    const char* vg = reinterpret_cast<const char*>(vb + (int64_t)kt * ATT_BK * p.v_rs);
    if ((kt + 1) * ATT_BK <= p.Lk) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) att_glds16(vg + v_src[i], vl + (wave * PPW + i) * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int over = kt * ATT_BK + (wave * PPW + i) * 4 + srow - (p.Lk - 1);
        att_glds16(vg + (v_src[i] - (unsigned)((over > 0 ? over : 0) * p.v_rs * 2)), vl + (wave * PPW + i) * 1024);
      }
    }
  };

  // ---- per-lane LDS read addresses, hoisted (buffer / key-block / row-group parts are immediates)
  // K (A operand of S^T): key row kb*32 + lc, 16-byte chunk (2*ds + lh) ^ (row & 15)
  typedef const __attribute__((address_space(3))) char* lds_cptr;
  const lds_cptr smem3 = (lds_cptr)smem;
  unsigned k_rd[8];
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) k_rd[ds] = p.blk ? ((2 * ds + lh) * 32 + lc) * 16 : lc * 256 + (((2 * ds + lh) ^ (lc & 15)) << 4);
  // V tr-read: key row = kb*32 + 16*s2 + 8*jj + 4*lh + (i>>2), element col = db*32 + 16*g + 4*(i&3)
  const int vi = lane & 15, vg_ = (lane >> 4) & 1;
  const int v_row_base = 4 * lh + (vi >> 2);
  const int v_col_base = 16 * vg_ + 4 * (vi & 3);
  const int v_rswz = (v_row_base & 3) << 2;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned vaddr[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
    vaddr[db] = lds0 + v_row_base * 256 + ((((v_col_base >> 3) + ((4 * db) ^ v_rswz))) << 4) + ((v_col_base & 7) << 1);

  f16x o_acc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o_acc[d][e] = 0.f;
  // running max (log2 domain) and sum; the max is only raised when some lane's tile max exceeds it by
  // more than 2^8 (P then stays <= 256, exact in f16/fp32), which skips most O^T rescales
  float m_run = -INFINITY, l_run = 0.f;
  const float LOG2E = 1.4426950408889634f;
  const f16x zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int nkt = (p.Lk + ATT_BK - 1) / ATT_BK;

  // S^T = K Q^T for the K tile in buffer KB (compile-time): 8 fragments in flight per 8-MFMA chain
  auto qk = [&](const unsigned kbase, f16x(&s)[2]) __attribute__((always_inline)) {
    typedef const __attribute__((address_space(3))) h8* lds_h8;
    h8 kf0[8], kf1[8];
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) kf0[ds] = *(lds_h8)(smem3 + k_rd[ds] + kbase);
    __builtin_amdgcn_sched_barrier(0);
    // chain 0; the fragments of chain 1 are requested one MFMA behind, into the registers chain 0 frees
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) {
      s[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf0[ds], qf[ds], ds == 0 ? zero16 : s[0], 0, 0, 0);
      kf1[ds] = *(lds_h8)(smem3 + k_rd[ds] + kbase + 32 * 256);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ds = 0; ds < 8; ++ds)
      s[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1[ds], qf[ds], ds == 0 ? zero16 : s[1], 0, 0, 0);
  };

#define ATT_TR(dst, db, off) \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(vcur[db]), "n"(off))
#define ATT_ISSUE(dst, db, VB)                                                                                   \
  ATT_TR(dst[0], db, VB + 0);    ATT_TR(dst[1], db, VB + 2048);  ATT_TR(dst[2], db, VB + 4096);  ATT_TR(dst[3], db, VB + 6144); \
  ATT_TR(dst[4], db, VB + 8192); ATT_TR(dst[5], db, VB + 10240); ATT_TR(dst[6], db, VB + 12288); ATT_TR(dst[7], db, VB + 14336);
#define ATT_WAIT(cur, n)                                                                                   \
  asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                 \
               : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]),        \
                 "+v"(cur[6]), "+v"(cur[7]));
#define ATT_PV(cur, db)                                                                                    \
  {                                                                                                        \
    _Pragma("unroll") for (int f = 0; f < 4; ++f) {                                                         \
      const h4 lo_ = __builtin_bit_cast(h4, cur[2 * f]), hi_ = __builtin_bit_cast(h4, cur[2 * f + 1]);      \
      const h8 vf = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);                              \
      o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[f >> 1][f & 1], o_acc[db], 0, 0, 0);        \
    }                                                                                                      \
  }

  f16x s_cur[2];
  const bool wave_idle = q0 >= p.Lq;                  // wave uniform

  // one tile: K(kt) and V(kt) sit in ring slot kt % NS; NS - 1 tiles are in flight, so this wave's pieces of tile kt are
  // the OLDEST of its outstanding LDS-DMA operations: the counted wait leaves the younger tiles' pieces in flight
  // (fewer of them near the end of the key range: the wait immediate is chosen among NS - 1 constants)
  constexpr int PIECES = 2 * PPW;                     // LDS-DMA instructions per wave and tile (K + V)
  // (the slot is a compile-time constant: the compiler then PROVES that the K-fragment reads of slot s cannot alias the
  // LDS-DMA writes in flight into the other slots; with a run-time slot it guards every read with s_waitcnt vmcnt(0),
  // which drains the ring)
  auto tile = [&](int kt, auto SLOTC) __attribute__((always_inline)) {
    constexpr int slot = decltype(SLOTC)::value;
    constexpr unsigned sb = (unsigned)slot * ATT_STAGE_BYTES;
    constexpr int VB = 0;
    unsigned vcur[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) vcur[db] = vaddr[db] + sb + ATT_TILE_BYTES;
    {
      const int ahead = nkt - 1 - kt < NS - 2 ? nkt - 1 - kt : NS - 2;   // younger tiles already requested (wave uniform)
      if (NS == 2 || ahead <= 0) __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));                       // vmcnt(0)
      else if (ahead == 1) __builtin_amdgcn_s_waitcnt((PIECES & 15) | (7 << 4) | (15 << 8) | ((PIECES >> 4) << 14));
      else __builtin_amdgcn_s_waitcnt(((2 * PIECES) & 15) | (7 << 4) | (15 << 8) | (((2 * PIECES) >> 4) << 14));
    }
    asm volatile("" ::: "memory");
    ATT_TS(0)
    __builtin_amdgcn_s_barrier();                                   // ... everybody's; slot (kt - 1) % NS is free
    asm volatile("" ::: "memory");
    ATT_TS(1)
    // The next tile's LDS-DMA pieces: 32 KB per workgroup and tile through the CU's 64 B / clk vector-memory path = 512 clocks
    // during which a wave that issues all its pieces in one burst sits at issue (tools/attn_timing.sh: 9 % of the kernel).
    // ATT_DMA = 1: K pieces here, V pieces behind the S^T MFMAs (they then drain beside the matrix pipe); 0 (default): one burst.
    // Measured (tools/attn_ab.sh, profiles/round5_attn_timing.txt): 1.497 -> 1.516 ms per step of self-attention, i.e. no gain — an
    // in-order wave that stalls on its V pieces behind the MFMAs starts its softmax that much later
    // (256-query workgroups only: the 128-query form runs at its register limit, and holding the V offsets across the S^T MFMAs spills)
    constexpr int ns_next = slot == 0 ? NS - 1 : slot - 1;          // (kt + NS - 1) % NS
    constexpr bool dma_split = ATT_DMA != 0 && NW == 8;
    const bool more = !(ATT_ABL & 8) && kt + NS - 1 < nkt;
    if (more) {
      stage_k(ns_next, kt + NS - 1);
      if (!dma_split || wave_idle) stage_v(ns_next, kt + NS - 1);
    }
    ATT_TS(2)
    // a wave whose 32 queries all lie beyond Lq (the last query tile of a short sequence: decoder L = 167 -> waves 2, 3 of
    // the second 128-query tile; SenseVoice T = 171 -> waves 6, 7 of a 256-query tile) only stages and keeps the barriers:
    // its matrix / VALU slots go to the other waves of its SIMD
    if (wave_idle) return;
    if (!(ATT_ABL & 4)) qk(sb, s_cur);
    if (dma_split && more) {
      __builtin_amdgcn_sched_barrier(0);
      stage_v(ns_next, kt + NS - 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    ATT_TS(3)

    // first V^T fragments (d block 0)
    fp4 va[8], vb2[8];
#if !(ATT_ABL & 2)
    ATT_ISSUE(va, 0, VB)
#endif

    // ---- mask the tile tail (keys >= Lk)
    if ((kt + 1) * ATT_BK > p.Lk) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kt * ATT_BK + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (key >= p.Lk) s_cur[kb][e] = -INFINITY;
        }
    }
    // ---- online softmax (lane = one query; its 32 keys here, the other 32 in lane^32).  P is taken
    // against the running max m_run, which is the same in both half-waves of a query and is only
    // raised (slow path, wave-uniform) when some lane's tile max exceeds it by more than 2^8; the
    // two half-wave partial sums are combined once, after the last tile.
    // four independent max chains (one dependent chain of 32 costs ~8 cycles of latency per link)
    float mx[4] = {s_cur[0][0], s_cur[0][1], s_cur[1][0], s_cur[1][1]};
#pragma unroll
    for (int e = 2; e < 16; e += 2) {
      mx[0] = fmaxf(mx[0], s_cur[0][e]); mx[1] = fmaxf(mx[1], s_cur[0][e + 1]);
      mx[2] = fmaxf(mx[2], s_cur[1][e]); mx[3] = fmaxf(mx[3], s_cur[1][e + 1]);
    }
    float mloc = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3])) * LOG2E;
    if (!(ATT_ABL & 16) && __any(mloc > m_run + 8.0f)) {
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float m_new = fmaxf(m_run, mloc);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);     // exp2(-inf) = 0 on the first tile
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o_acc[d][e] *= alpha;
    }
    f2 psum4[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};       // independent partial sums
    const f2 l2 = {LOG2E, LOG2E}, nm2 = {-m_run, -m_run};
    h8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        f2& psum2 = psum4[(kb * 8 + e / 2) & 3];
        f2 t = {s_cur[kb][e], s_cur[kb][e + 1]};
        t = t * l2 + nm2;
#if ATT_ABL & 1
        f2 pv = t;
#else
        f2 pv = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
#endif
        psum2 += pv;
        pf[kb][e >> 3][e & 7] = (half_t)pv.x;
        pf[kb][e >> 3][(e & 7) + 1] = (half_t)pv.y;
      }
    const f2 ps = (psum4[0] + psum4[1]) + (psum4[2] + psum4[3]);
    l_run += ps.x + ps.y;
#ifdef ATT_TIMING
    asm volatile("" :: "v"(pf[0][0]), "v"(pf[1][1]), "v"(l_run));    // the softmax results exist before the stamp
#endif
    ATT_TS(4)

    // ---- O^T += V^T P^T : fragments of d block db+1 are requested before the MFMAs of block db
#if !(ATT_ABL & 2)
    ATT_ISSUE(vb2, 1, VB)
    ATT_WAIT(va, 8)
    ATT_PV(va, 0)
    ATT_ISSUE(va, 2, VB)
    ATT_WAIT(vb2, 8)
    ATT_PV(vb2, 1)
    ATT_ISSUE(vb2, 3, VB)
    ATT_WAIT(va, 8)
    ATT_PV(va, 2)
    ATT_WAIT(vb2, 0)
    ATT_PV(vb2, 3)
#else
    o_acc[0][0] += (float)pf[0][0][0] + (float)pf[1][1][7] + (float)pf[0][1][3] + (float)pf[1][0][5];
#endif
    ATT_TS(5)
  };

#pragma unroll
  for (int st = 0; st < NS - 1; ++st)
    if (st < nkt) { stage_k(st, st); stage_v(st, st); }
  ATT_TS(6)

  for (int kt = 0; kt < nkt; kt += NS) {
    tile(kt, att_ic<0>{});
    if (kt + 1 < nkt) tile(kt + 1, att_ic<1>{});
    if constexpr (NS > 2) { if (kt + 2 < nkt) tile(kt + 2, att_ic<2>{}); }
    if constexpr (NS > 3) { if (kt + 3 < nkt) tile(kt + 3, att_ic<3>{}); }
  }
#undef ATT_TR
#undef ATT_PV
#undef ATT_ISSUE
#undef ATT_WAIT

  // ---- normalise and store: lane = query q0+lc, d = db*32 + (e&3) + 8*(e>>2) + 4*lh
  const int qrow = q0 + lc;
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  float r_lo = 0.f, r_hi = 0.f;
  // 16-byte stores (round 6): lanes l / l + 32 hold columns 8 g + 0..3 / 8 g + 4..7 of query row l; after v_permlane32_swap lane l
  // holds the 8 columns of group 2 gp, lane l + 32 those of group 2 gp + 1 (k_ffn.hip's V pass): 8 dwordx4 instead of 16 dwordx2
  // per lane — the store tail of an attention epilogue is issue-bound, not bandwidth-bound (MI355X guide, T21)
  const bool wide = (p.o_rs & 7) == 0 && ((reinterpret_cast<uintptr_t>(ob) & 15) == 0);
  if (wide) {
    const float inv = qrow < p.Lq ? 1.0f / l_tot : 0.f;
    half_t* op = ob + (int64_t)qrow * p.o_rs + 8 * lh;
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        unsigned x[2], y[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          f2v xa = {o_acc[db][8 * gp + 2 * e + 0], o_acc[db][8 * gp + 2 * e + 1]};
          f2v ya = {o_acc[db][8 * gp + 4 + 2 * e + 0], o_acc[db][8 * gp + 4 + 2 * e + 1]};
          xa *= inv; ya *= inv;
          const h2v xh = __builtin_convertvector(xa, h2v), yh = __builtin_convertvector(ya, h2v);
          if (p.range && qrow < p.Lq) {                 // of the values as stored (f16)
            const float a0 = (float)xh[0], a1 = (float)xh[1], a2 = (float)yh[0], a3 = (float)yh[1];
            r_lo = fminf(fminf(r_lo, a0), fminf(fminf(a1, a2), a3));
            r_hi = fmaxf(fmaxf(r_hi, a0), fmaxf(fmaxf(a1, a2), a3));
          }
          x[e] = __builtin_bit_cast(unsigned, xh);
          y[e] = __builtin_bit_cast(unsigned, yh);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[e]), "+v"(y[e]));
        const h4 lo_ = __builtin_bit_cast(h4, (unsigned long long)x[0] | ((unsigned long long)x[1] << 32));
        const h4 hi_ = __builtin_bit_cast(h4, (unsigned long long)y[0] | ((unsigned long long)y[1] << 32));
        typedef _Float16 h8o __attribute__((ext_vector_type(8)));
        const h8o hv = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);
        if (qrow < p.Lq) *reinterpret_cast<h8o*>(op + db * 32 + 16 * gp) = hv;
      }
  } else if (qrow < p.Lq) {
    const float inv = 1.0f / l_tot;
    half_t* op = ob + (int64_t)qrow * p.o_rs + 4 * lh;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h4 hv = {(half_t)(o_acc[db][4 * g + 0] * inv), (half_t)(o_acc[db][4 * g + 1] * inv),
                 (half_t)(o_acc[db][4 * g + 2] * inv), (half_t)(o_acc[db][4 * g + 3] * inv)};
        *reinterpret_cast<h4*>(op + db * 32 + 8 * g) = hv;
        if (p.range) {                                  // of the values as stored (f16)
          const float a0 = (float)hv[0], a1 = (float)hv[1], a2 = (float)hv[2], a3 = (float)hv[3];
          r_lo = fminf(fminf(r_lo, a0), fminf(fminf(a1, a2), a3));
          r_hi = fmaxf(fmaxf(r_hi, a0), fmaxf(fmaxf(a1, a2), a3));
        }
      }
  }
  if (p.range) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      r_lo = fminf(r_lo, __shfl_xor(r_lo, o, 64));
      r_hi = fmaxf(r_hi, __shfl_xor(r_hi, o, 64));
    }
    __syncthreads();                                    // every wave is done with the K / V buffers
    float* red = reinterpret_cast<float*>(smem);
    if (lane == 0) { red[2 * wave] = r_lo; red[2 * wave + 1] = r_hi; }
    __syncthreads();
    const int wg = blockIdx.x + gridDim.x * blockIdx.y;
    if (tid == 0) {
      float l = 0.f, h = 0.f;
      for (int w = 0; w < NW; ++w) { l = fminf(l, red[2 * w]); h = fmaxf(h, red[2 * w + 1]); }
      p.range[2 * wg] = l;
      p.range[2 * wg + 1] = h;
    }
    if (wg == 0)                                        // the consumer folds 256 pairs whatever this grid was
      for (int i = total + tid; i < 256; i += 64 * NW) { p.range[2 * i] = 0.f; p.range[2 * i + 1] = 0.f; }
  }
#ifdef ATT_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  tm_acc[7] = (unsigned)(__builtin_readcyclecounter() - t_begin);
  if (lane == 0 && lid < 512)                          // 256-query workgroups in the first half, 128-query ones (cross-attention) in the second
    for (int i = 0; i < 8; ++i) att_tm[(((NW == 8 ? 0 : 512) + lid) * 8 + (wave & 7)) * 8 + i] = tm_acc[i];
#endif
}

#ifdef ATT_TIMING
}  // namespace pf
extern "C" int pf_debug_attn_timing(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pf::att_tm), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}
namespace pf {
#endif

// ---------------------------------------------------------------------------------------------------------------
// Ping-pong form for 256-query workgroups (round 4).  Counters on attn_kernel<8> (profiles/round4_feed_gap.md): matrix
// pipe busy 32 %, issue stalls 45 % — the two waves of a SIMD run the SAME phase at the same time (both want the matrix
// pipe for S^T = K Q^T, then both want the VALU for the soft-max, then both want the pipe again for O^T += V^T P^T), so
// each unit idles while the other is oversubscribed.  Here waves 0-3 (group A) and 4-7 (group B; wave w + 4 shares
// wave w's SIMD) run the same per-wave sequence ONE PHASE apart, and the sequence is re-cut so that a phase is all-matrix
// or all-VALU:
//     M(k) = [ O^T += V(k)^T P(k)^T ;  S(k+1) = K(k+1) Q^T ]      32 MFMAs
//     V(k) = soft-max arithmetic of S(k) -> P(k)                   ~130 VALU
//   phase      -1     0      1      2      3      4    ...
//   group A   S(0)  V(0)   M(0)   V(1)   M(1)   V(2)
//   group B         S(0)   V(0)   M(0)   V(1)   M(1)
// One s_barrier per phase.  Tile k (K and V, one 32 KiB ring stage) is read in phases 2k-1 .. 2k+2, so a ring of NS
// stages leaves 2 NS - 4 phases between the barrier that frees a slot and the phase that needs its next content.  All
// LDS reads are inline asm with counted lgkmcnt waits (a run-time ring slot in compiler-visible reads makes the compiler
// guard each of them with s_waitcnt vmcnt(0), which drains the ring).
template <int NS>
__global__ __launch_bounds__(512, 1) void attn_pp_kernel(AttnDev p) {
  constexpr int NW = 8, PPW = 2, PIECES = 2 * PPW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int lh = lane >> 5, lc = lane & 31;
  const int nqt = gridDim.x, total = gridDim.x * gridDim.y;
  int lid = blockIdx.x + nqt * blockIdx.y;
  {
    const int q8 = total >> 3, r8 = total & 7, xcd = lid & 7, idx = lid >> 3;
    lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  const int bh = lid / nqt, qt = lid - bh * nqt;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = qt * (32 * NW) + wave * 32;
  const int Lk = p.Lk, Lq = p.Lq;
  const int k_rs = p.k_rs, v_rs = p.v_rs;

  const half_t* qb = p.q + b * p.q_bs + h * ATT_DK;
  const half_t* kb_ = p.k + b * p.k_bs + h * ATT_DK;
  const half_t* vb = p.v + b * p.v_bs + h * ATT_DK;
  half_t* ob = p.o + b * p.o_bs + h * ATT_DK;

  h8 qf[8];
  {
    int qr = q0 + lc;
    qr = qr < Lq ? qr : Lq - 1;
    const half_t* qp = qb + (int64_t)qr * p.q_rs + 8 * lh;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) qf[ds] = *reinterpret_cast<const h8*>(qp + 16 * ds);
  }

  const int srow = lane >> 4, schunk = lane & 15;
  unsigned k_src[PPW], v_src[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int row = (wave * PPW + i) * 4 + srow;
    k_src[i] = (unsigned)(row * k_rs + ((schunk ^ (row & 15)) << 3)) * 2u;
    v_src[i] = (unsigned)(row * v_rs + ((schunk ^ ((row & 3) << 2)) << 3)) * 2u;
  }
  // both operands of tile kt into ring slot `slot`; rows beyond Lk re-read row Lk - 1 (see attn_kernel)
  auto stage = [&](int slot, int kt) __attribute__((always_inline)) {
    char* kl = smem + slot * ATT_STAGE_BYTES;
    char* vl = kl + ATT_TILE_BYTES;
    const char* kg = reinterpret_cast<const char*>(kb_ + (int64_t)kt * ATT_BK * k_rs);
    const char* vg = reinterpret_cast<const char*>(vb + (int64_t)kt * ATT_BK * v_rs);
    const bool full = (kt + 1) * ATT_BK <= Lk;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      int over = full ? 0 : kt * ATT_BK + (wave * PPW + i) * 4 + srow - (Lk - 1);
      over = over > 0 ? over : 0;
      att_glds16(kg + (k_src[i] - (unsigned)(over * k_rs * 2)), kl + (wave * PPW + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      int over = full ? 0 : kt * ATT_BK + (wave * PPW + i) * 4 + srow - (Lk - 1);
      over = over > 0 ? over : 0;
      att_glds16(vg + (v_src[i] - (unsigned)(over * v_rs * 2)), vl + (wave * PPW + i) * 1024);
    }
  };

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned k_rd[8];
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) k_rd[ds] = lds0 + lc * 256 + (((2 * ds + lh) ^ (lc & 15)) << 4);
  const int vi = lane & 15, vg_ = (lane >> 4) & 1;
  const int v_row_base = 4 * lh + (vi >> 2);
  const int v_col_base = 16 * vg_ + 4 * (vi & 3);
  const int v_rswz = (v_row_base & 3) << 2;
  unsigned vaddr[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
    vaddr[db] = lds0 + ATT_TILE_BYTES + v_row_base * 256 + ((((v_col_base >> 3) + ((4 * db) ^ v_rswz))) << 4) + ((v_col_base & 7) << 1);

  f16x o_acc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o_acc[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float LOG2E = 1.4426950408889634f;
  const f16x zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int nkt = (Lk + ATT_BK - 1) / ATT_BK;
  f16x s_cur[2];
  h8 pf[2][2] = {};

#define APP_KRD(dst, ds, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(kcur[ds]), "n"(off))
#define APP_KWAIT(reg, n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(reg))
  // S^T(kt) = K(kt) Q^T -> s_cur.  Chain 0 (keys 0-31) then chain 1 (keys 32-63); the fragments of chain 1 are requested
  // one MFMA behind chain 0, so eight reads are in flight throughout chain 0.
  auto qk = [&](int slot) __attribute__((always_inline)) {
    unsigned kcur[8];
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) kcur[ds] = k_rd[ds] + (unsigned)slot * ATT_STAGE_BYTES;
    h8 kf0[8], kf1[8];
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) APP_KRD(kf0[ds], ds, 0);
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) {
      APP_KWAIT(kf0[ds], 7);
      s_cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf0[ds], qf[ds], ds == 0 ? zero16 : s_cur[0], 0, 0, 0);
      APP_KRD(kf1[ds], ds, 32 * 256);
    }
    APP_KWAIT(kf1[0], 7); s_cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1[0], qf[0], zero16, 0, 0, 0);
    APP_KWAIT(kf1[1], 6); s_cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1[1], qf[1], s_cur[1], 0, 0, 0);
    APP_KWAIT(kf1[2], 5); s_cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1[2], qf[2], s_cur[1], 0, 0, 0);
    APP_KWAIT(kf1[3], 4); s_cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1[3], qf[3], s_cur[1], 0, 0, 0);
    APP_KWAIT(kf1[4], 3); s_cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1[4], qf[4], s_cur[1], 0, 0, 0);
    APP_KWAIT(kf1[5], 2); s_cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1[5], qf[5], s_cur[1], 0, 0, 0);
    APP_KWAIT(kf1[6], 1); s_cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1[6], qf[6], s_cur[1], 0, 0, 0);
    APP_KWAIT(kf1[7], 0); s_cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1[7], qf[7], s_cur[1], 0, 0, 0);
  };
#undef APP_KRD
#undef APP_KWAIT

#define ATT_TR(dst, db, off) \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(vcur[db]), "n"(off))
#define ATT_ISSUE(dst, db)                                                                                   \
  ATT_TR(dst[0], db, 0);    ATT_TR(dst[1], db, 2048);  ATT_TR(dst[2], db, 4096);  ATT_TR(dst[3], db, 6144); \
  ATT_TR(dst[4], db, 8192); ATT_TR(dst[5], db, 10240); ATT_TR(dst[6], db, 12288); ATT_TR(dst[7], db, 14336);
#define ATT_WAIT(cur, n)                                                                                   \
  asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                 \
               : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]),        \
                 "+v"(cur[6]), "+v"(cur[7]));
#define ATT_PV(cur, db)                                                                                    \
  {                                                                                                        \
    _Pragma("unroll") for (int f = 0; f < 4; ++f) {                                                         \
      const h4 lo_ = __builtin_bit_cast(h4, cur[2 * f]), hi_ = __builtin_bit_cast(h4, cur[2 * f + 1]);      \
      const h8 vf = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);                              \
      o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[f >> 1][f & 1], o_acc[db], 0, 0, 0);        \
    }                                                                                                      \
  }
  // O^T += V(kt)^T P(kt)^T with pf = P(kt): fragments of d block db + 1 are requested before the MFMAs of block db
  auto pv = [&](int slot) __attribute__((always_inline)) {
    unsigned vcur[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) vcur[db] = vaddr[db] + (unsigned)slot * ATT_STAGE_BYTES;
    fp4 va[8], vb2[8];
    ATT_ISSUE(va, 0)
    ATT_ISSUE(vb2, 1)
    ATT_WAIT(va, 8)
    ATT_PV(va, 0)
    ATT_ISSUE(va, 2)
    ATT_WAIT(vb2, 8)
    ATT_PV(vb2, 1)
    ATT_ISSUE(vb2, 3)
    ATT_WAIT(va, 8)
    ATT_PV(va, 2)
    ATT_WAIT(vb2, 0)
    ATT_PV(vb2, 3)
  };
#undef ATT_TR
#undef ATT_PV
#undef ATT_ISSUE
#undef ATT_WAIT

  // soft-max arithmetic of s_cur = S(kt) -> pf = P(kt) (f16), running max / sum, O^T rescale (see attn_kernel)
  auto softmax = [&](int kt) __attribute__((always_inline)) {
    if ((kt + 1) * ATT_BK > Lk) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kt * ATT_BK + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (key >= Lk) s_cur[kb][e] = -INFINITY;
        }
    }
    float mx[4] = {s_cur[0][0], s_cur[0][1], s_cur[1][0], s_cur[1][1]};
#pragma unroll
    for (int e = 2; e < 16; e += 2) {
      mx[0] = fmaxf(mx[0], s_cur[0][e]); mx[1] = fmaxf(mx[1], s_cur[0][e + 1]);
      mx[2] = fmaxf(mx[2], s_cur[1][e]); mx[3] = fmaxf(mx[3], s_cur[1][e + 1]);
    }
    float mloc = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3])) * LOG2E;
    if (__any(mloc > m_run + 8.0f)) {
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float m_new = fmaxf(m_run, mloc);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o_acc[d][e] *= alpha;
    }
    f2 psum4[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    const f2 l2 = {LOG2E, LOG2E}, nm2 = {-m_run, -m_run};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        f2& psum2 = psum4[(kb * 8 + e / 2) & 3];
        f2 t = {s_cur[kb][e], s_cur[kb][e + 1]};
        t = t * l2 + nm2;
        f2 pv2 = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
        psum2 += pv2;
        pf[kb][e >> 3][e & 7] = (half_t)pv2.x;
        pf[kb][e >> 3][(e & 7) + 1] = (half_t)pv2.y;
      }
    const f2 ps = (psum4[0] + psum4[1]) + (psum4[2] + psum4[3]);
    l_run += ps.x + ps.y;
  };

  // ---- prologue: stages 0 .. NS - 2 in flight, stage 0 landed
#pragma unroll
  for (int st = 0; st < NS - 1; ++st)
    if (st < nkt) stage(st, st);
  {
    const int younger = nkt - 1 < NS - 2 ? nkt - 1 : NS - 2;
    if (younger <= 0) __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
    else if (younger == 1) __builtin_amdgcn_s_waitcnt((PIECES & 15) | (7 << 4) | (15 << 8) | ((PIECES >> 4) << 14));
    else __builtin_amdgcn_s_waitcnt(((2 * PIECES) & 15) | (7 << 4) | (15 << 8) | (((2 * PIECES) >> 4) << 14));
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // The phase sequence is written out per group (no per-phase "is there a tile" conditions around the three blocks: a
  // conditionally executed block makes the compiler keep the old and the new S / P registers alive side by side).
  int q = -1, slot_next = NS - 1;                       // global phase; ring slot of the stage the next odd phase requests
  auto begin = [&]() __attribute__((always_inline)) {
    if ((q & 1) && q >= 1) {                            // slot of tile (q - 3) / 2 was released by the barrier behind us
      const int j = ((q - 3) >> 1) + NS;
      if (j < nkt) stage(slot_next, j);
      slot_next = slot_next + 1 == NS ? 0 : slot_next + 1;
    }
  };
  auto end = [&]() __attribute__((always_inline)) {
    if (!(q & 1)) {                                     // tile j is first read in the next phase: this wave's pieces of it have landed
      const int j = (q + 2) >> 1;
      if (j < nkt) {
        const int younger = nkt - 1 - j < NS - 3 ? nkt - 1 - j : NS - 3;
        if (NS <= 3 || younger <= 0) __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
        else __builtin_amdgcn_s_waitcnt((PIECES & 15) | (7 << 4) | (15 << 8) | ((PIECES >> 4) << 14));
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ++q;
  };
  if (grp) { begin(); end(); }                          // group B starts one phase later
  begin(); qk(0); end();
  begin(); softmax(0); end();
  int sl = 0;                                           // ring slot of tile k - 1
  for (int k = 1; k < nkt; ++k) {
    const int sn = sl + 1 == NS ? 0 : sl + 1;
    begin();
    pv(sl);                                             // O^T += V(k-1)^T P(k-1)^T   (P dies here, before S(k) is born)
    qk(sn);                                             // S(k)
    end();
    begin(); softmax(k); end();
    sl = sn;
  }
  begin(); pv(sl); end();
  if (!grp) { begin(); end(); }                         // group A idles in the last phase

  // ---- normalise and store: lane = query q0+lc, d = db*32 + (e&3) + 8*(e>>2) + 4*lh
  const int qrow = q0 + lc;
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  float r_lo = 0.f, r_hi = 0.f;
  if (qrow < Lq) {
    const float inv = 1.0f / l_tot;
    half_t* op = ob + (int64_t)qrow * p.o_rs + 4 * lh;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h4 hv = {(half_t)(o_acc[db][4 * g + 0] * inv), (half_t)(o_acc[db][4 * g + 1] * inv),
                 (half_t)(o_acc[db][4 * g + 2] * inv), (half_t)(o_acc[db][4 * g + 3] * inv)};
        *reinterpret_cast<h4*>(op + db * 32 + 8 * g) = hv;
        if (p.range) {
          const float a0 = (float)hv[0], a1 = (float)hv[1], a2 = (float)hv[2], a3 = (float)hv[3];
          r_lo = fminf(fminf(r_lo, a0), fminf(fminf(a1, a2), a3));
          r_hi = fmaxf(fmaxf(r_hi, a0), fmaxf(fmaxf(a1, a2), a3));
        }
      }
  }
  if (p.range) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      r_lo = fminf(r_lo, __shfl_xor(r_lo, o, 64));
      r_hi = fmaxf(r_hi, __shfl_xor(r_hi, o, 64));
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    if (lane == 0) { red[2 * wave] = r_lo; red[2 * wave + 1] = r_hi; }
    __syncthreads();
    const int wg = blockIdx.x + gridDim.x * blockIdx.y;
    if (tid == 0) {
      float l = 0.f, hh = 0.f;
      for (int w = 0; w < NW; ++w) { l = fminf(l, red[2 * w]); hh = fmaxf(hh, red[2 * w + 1]); }
      p.range[2 * wg] = l;
      p.range[2 * wg + 1] = hh;
    }
    if (wg == 0)
      for (int i = total + tid; i < 256; i += 64 * NW) { p.range[2 * i] = 0.f; p.range[2 * i + 1] = 0.f; }
  }
}

// grid of the launch below: 256-query workgroups when they cover the chip, else 128-query ones
static int attention_grid(const AttnArgs& a, int cus, bool& nw8) {
  static const int force_nw = env_int("PF_ATT_NW", 0);
  const int wg8 = ((a.Lq + 255) / 256) * a.B * a.H;
  nw8 = force_nw ? force_nw == 8 : wg8 >= cus;
  return nw8 ? wg8 : ((a.Lq + ATT_BQ - 1) / ATT_BQ) * a.B * a.H;
}
static int attention_cus() {
  static std::mutex mu;
  static int cus[64] = {0};
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  if (!cus[dev & 63]) {
    hipDeviceProp_t prop;
    PF_HIP(hipGetDeviceProperties(&prop, dev));
    cus[dev & 63] = cu_limit(prop.multiProcessorCount);
  }
  return cus[dev & 63];
}
bool attention_reports_range(const AttnArgs& a) {
  if (a.B == 0 || a.Lq == 0 || a.Lk == 0) return false;
  bool nw8;
  return attention_grid(a, attention_cus(), nw8) <= 256;
}

void launch_attention(hipStream_t s, const AttnArgs& a) {
  if (a.B == 0 || a.Lq == 0 || a.Lk == 0) return;
  AttnDev d;
  d.q = a.q; d.k = a.k; d.v = a.v; d.o = a.o;
  d.q_bs = a.q_bstride; d.k_bs = a.k_bstride; d.v_bs = a.v_bstride; d.o_bs = a.o_bstride;
  d.q_rs = a.q_rstride; d.k_rs = a.k_rstride; d.v_rs = a.v_rstride; d.o_rs = a.o_rstride;
  d.H = a.H; d.Lq = a.Lq; d.Lk = a.Lk;
  d.range = a.range;
  d.blk = a.qk_blocked ? 1 : 0; d.blk_groups = a.blk_groups; d.blk_brows = a.blk_brows; d.blk_kgrp = a.blk_kgrp;
  PF_CHECK(!a.qk_blocked || (a.q == a.k && a.blk_groups > 0 && a.blk_brows >= a.Lq && a.blk_brows >= a.Lk), PF_ERR_INVALID_ARG,
           "attention: blocked Q | K needs one matrix and its geometry");
  PF_CHECK(!a.range || attention_reports_range(a), PF_ERR_INVALID_ARG, "attention: a range output needs a grid of at most 256 workgroups");
  PF_CHECK(a.q_rstride % 8 == 0 && a.k_rstride % 8 == 0 && a.v_rstride % 8 == 0 && a.o_rstride % 4 == 0,
           PF_ERR_INVALID_ARG, "attention: row strides must keep 16-byte alignment");
  static std::mutex init_mu;                         // engines on different devices launch from different threads
  static bool attr_set[64] = {false};
  static int cus[64] = {0};
  static int ns8 = 2, ns4 = 2;                       // ring stages (PF_ATT_NS8 / PF_ATT_NS4: 2 | 3 | 4 for A/B runs); round 4 measured
                                                     // 3 and 4 SLOWER than 2 (self-attention 1.55 vs 1.44 ms per step): DMA latency is not what the tiles wait for
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(init_mu);
    if (!attr_set[dev & 63]) {
      if (const char* e = getenv("PF_ATT_NS8")) { const int v = atoi(e); if (v >= 2 && v <= 4) ns8 = v; }
      if (const char* e = getenv("PF_ATT_NS4")) { const int v = atoi(e); if (v >= 2 && v <= 4) ns4 = v; }
#define PF_ATT_ATTR(NW, NS) PF_HIP(hipFuncSetAttribute((const void*)attn_kernel<NW, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, NS * ATT_STAGE_BYTES))
      PF_ATT_ATTR(4, 2); PF_ATT_ATTR(4, 3); PF_ATT_ATTR(4, 4); PF_ATT_ATTR(8, 2); PF_ATT_ATTR(8, 3); PF_ATT_ATTR(8, 4);
#undef PF_ATT_ATTR
      PF_HIP(hipFuncSetAttribute((const void*)attn_pp_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * ATT_STAGE_BYTES));
      PF_HIP(hipFuncSetAttribute((const void*)attn_pp_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * ATT_STAGE_BYTES));
      hipDeviceProp_t prop;
      PF_HIP(hipGetDeviceProperties(&prop, dev));
      cus[dev & 63] = cu_limit(prop.multiProcessorCount);
      attr_set[dev & 63] = true;
    }
  }
  // 256-query workgroups when they still cover the chip (self-attention at T = 500: 2 x 128 workgroups); PF_ATT_NW forces
  bool nw8;
  (void)attention_grid(a, cus[dev & 63], nw8);
  // round-4 measurement (profiles/round4_attn_pingpong.txt): 1.58 ms per step against 1.48 for the lockstep kernel — the
  // phase offset does NOT pay (the MI355X guide's "moving work between the two waves of a SIMD is zero- or negative-sum"
  // holds here too); kept opt-in for experiments
  static const int use_pp = env_int("PF_ATT_PP", 0) == 1 ? 1 : 0;
  static const int pp_ns = env_int("PF_ATT_PP_NS", 3) == 4 ? 4 : 3;
  if (nw8 && use_pp && !a.qk_blocked) {
    dim3 grid((a.Lq + 255) / 256, a.B * a.H);
    if (pp_ns == 4) hipLaunchKernelGGL((attn_pp_kernel<4>), grid, dim3(512), 4 * ATT_STAGE_BYTES, s, d);
    else hipLaunchKernelGGL((attn_pp_kernel<3>), grid, dim3(512), 3 * ATT_STAGE_BYTES, s, d);
  } else if (nw8) {
    dim3 grid((a.Lq + 255) / 256, a.B * a.H);
    if (ns8 == 4) hipLaunchKernelGGL((attn_kernel<8, 4>), grid, dim3(512), 4 * ATT_STAGE_BYTES, s, d);
    else if (ns8 == 3) hipLaunchKernelGGL((attn_kernel<8, 3>), grid, dim3(512), 3 * ATT_STAGE_BYTES, s, d);
    else hipLaunchKernelGGL((attn_kernel<8, 2>), grid, dim3(512), 2 * ATT_STAGE_BYTES, s, d);
  } else {
    dim3 grid((a.Lq + ATT_BQ - 1) / ATT_BQ, a.B * a.H);
    if (ns4 == 4) hipLaunchKernelGGL((attn_kernel<4, 4>), grid, dim3(256), 4 * ATT_STAGE_BYTES, s, d);
    else if (ns4 == 3) hipLaunchKernelGGL((attn_kernel<4, 3>), grid, dim3(256), 3 * ATT_STAGE_BYTES, s, d);
    else hipLaunchKernelGGL((attn_kernel<4, 2>), grid, dim3(256), 2 * ATT_STAGE_BYTES, s, d);
  }
  PF_HIP(hipGetLastError());
}

}  // namespace pf
