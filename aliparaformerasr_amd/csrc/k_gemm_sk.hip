// k_gemm_sk.hip — ROW-COMPLETE GEMM with a split K for the deep N = 512 projection (encoder FFN-down):
//
//   y = A[M,K] * W[512,K]^T + bias           (f16 operands, fp32 accumulate, MFMA 32x32x16)
//   x = resid + y                             (fp32 residual stream)
//   n = LayerNorm(x) * gamma + beta           (-> f16 operand of the next GEMM and / or fp32)
//
// (MatMul + Add + Add + LayerNormalization of the graph InferenceSession.Run executes,
// AliParaformerAsr/OfflineProjOfParaformer.cs:68.)
//
// Why a third form next to gemm_f16_pp3 (256 x 128 tiles + a LayerNorm launch) and gemm_rc_kernel (64 x 512 tiles):
// a LayerNorm epilogue needs complete rows, but M = 16 000 rows are only 62.5 rows per CU — 64-row tiles stage 72 KB
// of operands per 4.2 MFLOP (58 flop/B; the row-complete kernel measured 67.7 us against 57.5 us for the persistent
// kernel + LayerNorm at K = 2048), and 128-row tiles would leave half the chip idle.  Here a PAIR of workgroups
// shares a 128-row block: each walks HALF of K with a 128 x 512 tile (40 KB per 4.2 MFLOP = 102 flop/B; A is read
// exactly once, by one workgroup, straight from HBM; only the 2 MB W panel is re-read through L2), then the two
// exchange the halves of their fp32 partial tiles they do not own — 128 KB each way, 16-byte write-through stores
// in accumulator layout, one flag per wave (the CDNA guide's hand-off recipe R1: every storing wave drains, one lane
// publishes; the consumer polls relaxed and reads with sc1 loads) — and each finishes 64 complete rows: bias +
// residual -> x, two-pass LayerNorm -> f16.  The pair sits on one XCD (block ids b and b ^ 8), dispatch order
// guarantees progress (a workgroup only ever waits for a partner at most 8 ids later), spins are bounded and raise
// an error word.  P0 + P1 is commutative, so the result does not depend on which partner adds.
//
// K loop: the flat pipeline of gemm_bigp_kernel (k_gemm_big.hip) — k-steps of 32, four 40 KB stages (all 160 KB of
// the CU's LDS), three in flight, one s_barrier per k-step placed mid-step, fragment reads and their counted waits
// in inline asm, five LDS-DMA pieces per wave and step slotted between the MFMAs.  8 waves as 2 (m) x 4 (n), wave
// tile 64 x 128 = 2 x 4 MFMA blocks; the workgroup with k-half h maps its row block i to (i ^ h), so that acc[0] is
// always the half it keeps and acc[1] the half it sends.
#include "kernels.h"

#include <mutex>

namespace pf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float4 __attribute__((may_alias)) float4a;

struct SkDev {
  const half_t* A; const half_t* W; const float* bias;
  const float* resid; float* out_x;
  const float* ln_g; const float* ln_b; half_t* out_n16; float* out_n32;
  float* slab; unsigned* flags; unsigned* err;
  int lda, ldw, ldr, ldx, ldn16, ldn32;
  int M, K, a_blocked, n_rb;
  float eps;
};

constexpr int SK_BM = 128, SK_BN = 512, SK_BK = 32, SK_ROWB = SK_BK * 2, SK_S = 4;
constexpr int SK_A_BYTES = SK_BM * SK_ROWB;                 // 8 KiB
constexpr int SK_STAGE = (SK_BM + SK_BN) * SK_ROWB;         // 40 KiB
constexpr int SK_LDS = SK_S * SK_STAGE;                     // 160 KiB: the whole LDS of a CU
constexpr int SK_XROW = SK_BN * 4 + 16;                     // epilogue tile row: 16-byte skew (conflict-free dump)
constexpr int SK_SLAB = 8 * 16 * 64 * 16;                   // bytes a workgroup hands to its partner (64 rows x 512 fp32)
static_assert(64 * SK_XROW <= SK_LDS, "epilogue tile must fit the ring");

__device__ __forceinline__ void sk_glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void sk_wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ void sk_store16_sc1(void* p, f4v v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ float sk_wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));  // row_mirror
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (a + b) + (c + d);
}

// SPLIT: the pair form described above.  !SPLIT: one workgroup per 128-row block walks ALL of K and finishes its 128 rows in two
// 64-row passes — half as many workgroups as 64-row tiles, each staging 40 KB per 4.2 MFLOP instead of 72 KB: no faster
// alone (125 workgroups on 256 CUs), but it costs ~45 % less CU time, which is what counts when a second engine's
// kernels run on the CUs it leaves free (bench.py --in-flight 2).
template <bool SPLIT>
__global__ __launch_bounds__(512, 1) void gemm_sk_kernel(SkDev p) {
  constexpr int NJ = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, lh = lane >> 5;
  auto swz = [](int row) __attribute__((always_inline)) -> int { return (row >> 2) & 3; };

  // ---- pair schedule: blocks b and b ^ 8 (same XCD, adjacent in its dispatch order) share row block rb
  const int bid = blockIdx.x;
  const int u = bid >> 3, h = SPLIT ? (u & 1) : 0;
  const int rb = SPLIT ? (u >> 1) * 8 + (bid & 7) : bid;
  if (rb >= p.n_rb) return;                                  // both partners of a pair leave together
  const int m0 = rb * SK_BM;
  const int Kh = SPLIT ? (p.K >> 1) : p.K;                   // this workgroup's share of K: [h * Kh, (h + 1) * Kh)
  const int T = Kh / SK_BK;

  // ---- DMA cursor (uniform) and per-lane source offsets
  const int srow = lane >> 2, schunk = lane & 3;
  unsigned a_vo, w_vo[4];
  {
    const int row = wave * 16 + srow;
    a_vo = p.a_blocked ? (unsigned)((wave >> 1) * (p.K >> 3) * 512 + (wave & 1) * 1024 + lane * 16)
                       : (unsigned)(row * p.lda + ((schunk ^ swz(row)) << 3)) * 2u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int wr = (wave + 8 * q) * 16 + srow;
      w_vo[q] = (unsigned)(wr * p.ldw + ((schunk ^ swz(wr)) << 3)) * 2u;
    }
  }
  const char* is_a = p.a_blocked
                         ? reinterpret_cast<const char*>(p.A) + ((size_t)(m0 >> 5) * (size_t)(p.K >> 3) + (size_t)h * (Kh >> 3)) * 512
                         : reinterpret_cast<const char*>(p.A + (size_t)m0 * p.lda + (size_t)h * Kh);
  const char* is_w = reinterpret_cast<const char*>(p.W + (size_t)h * Kh);
  const int a_step = p.a_blocked ? (SK_BK / 8) * 512 : SK_BK * 2;
  int is_t = 0, is_slot = 0;
  auto issue_piece = [&](int q) __attribute__((always_inline)) {
    char* st = smem + (is_slot & (SK_S - 1)) * SK_STAGE + wave * 1024;
    if (q == 0) sk_glds16(is_a + a_vo, st);
    else sk_glds16(is_w + w_vo[q - 1], st + SK_A_BYTES + (q - 1) * 8192);
  };
  // past the last stage the cursor stops advancing (the last stage is re-loaded into a slot nobody reads again), so the
  // loop body is branch-free and every vmcnt immediate a compile-time constant (k_gemm_big.hip explains why that matters)
  auto issue_advance = [&]() __attribute__((always_inline)) {
    if (is_t + 1 < T) { ++is_t; is_a += a_step; is_w += SK_BK * 2; }
    ++is_slot;
  };

  // ---- fragment read offsets inside a stage (bytes); row block i of this wave is tile row block wm*2 + (i ^ h)
  unsigned fa[2][2], fb[2][NJ];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rbl = wm * 2 + (i ^ h);
      const int ra = rbl * 32 + (lane & 31);
      fa[s][i] = p.a_blocked ? (unsigned)(((rbl * 4 + 2 * s + lh) * 32 + (lane & 31)) * 16)
                             : (unsigned)(ra * SK_ROWB + (((2 * s + lh) ^ swz(ra)) << 4));
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int rw = wn * (32 * NJ) + j * 32 + (lane & 31);
      fb[s][j] = (unsigned)(SK_A_BYTES + rw * SK_ROWB + (((2 * s + lh) ^ swz(rw)) << 4));
    }
  }
  f16x acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  h8 a0[2] = {}, b0[NJ] = {}, a1[2] = {}, b1[NJ] = {};
  auto load = [&](unsigned rd, int s, h8 (&af)[2], h8 (&bf)[NJ]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(bf[j]) : "v"(rd + fb[s][j]) : "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(af[i]) : "v"(rd + fa[s][i]) : "memory");
  };
  auto frag_wait6 = [&](h8 (&af)[2], h8 (&bf)[NJ]) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(af[0]), "+v"(af[1]), "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]));
  };
  auto frag_wait0 = [&](h8 (&af)[2], h8 (&bf)[NJ]) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]));
  };
  auto mma = [&](h8 (&af)[2], h8 (&bf)[NJ], bool dma) __attribute__((always_inline)) {
    int q = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        if (q < 5) {
          __builtin_amdgcn_sched_barrier(0);
          if (dma) issue_piece(q);
          ++q;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
  };

  // ---- prologue: three stages in flight
#pragma unroll
  for (int st = 0; st < SK_S - 1; ++st) {
#pragma unroll
    for (int q = 0; q < 5; ++q) issue_piece(q);
    issue_advance();
  }
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
  sk_wait_vmcnt<10>();                                       // stage 0 of this wave has landed
  __builtin_amdgcn_s_barrier();
  load(lds0, 0, a0, b0);

  for (int t = 0; t < T; ++t) {
    const unsigned rd = lds0 + (t & (SK_S - 1)) * SK_STAGE;
    __builtin_amdgcn_sched_barrier(0);
    load(rd, 1, a1, b1);                                     // second half of stage t, under the first half's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    frag_wait6(a0, b0);
    __builtin_amdgcn_s_setprio(1);
    mma(a0, b0, true);                                       // + the five pieces of stage t + 3 (slot freed by the barrier of step t - 1)
    __builtin_amdgcn_s_setprio(0);
    issue_advance();
    __builtin_amdgcn_sched_barrier(0);
    sk_wait_vmcnt<10>();                                     // stage t + 1 of this wave has landed
    frag_wait0(a1, b1);                                      // every read of stage t has retired
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    load(lds0 + ((t + 1) & (SK_S - 1)) * SK_STAGE, 0, a0, b0);   // (after the last step: a slot nobody uses)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma(a1, b1, false);
    __builtin_amdgcn_s_setprio(0);
  }
  frag_wait0(a0, b0);                                        // the dead read behind the last barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the clamped DMA of the last steps has landed ...
  __builtin_amdgcn_s_barrier();                              // ... for every wave: the ring is free for the epilogue tile

  // ---- epilogue of 64 complete rows held in `a` (+ the partner's partials `pv` in the pair form): residual rows are
  // requested first (their round trip runs under the hand-off / the dump), the tile goes row-major into LDS, then wave w
  // finishes rows 8w .. 8w + 7 completely: lane = columns 4*lane and 256 + 4*lane.  ph: which 32-row half of each 64-row
  // wave block these rows are.
  auto finish = [&](int ph, f16x (&a)[NJ], bool with_partner) __attribute__((always_inline)) {
    const int mb = m0 + (wave >> 2) * 64 + ph * 32 + (wave & 3) * 8;
    float4 xv[2][8];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        xv[hh][r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.resid && mb + r < p.M)
          xv[hh][r] = *reinterpret_cast<const float4*>(p.resid + (size_t)(mb + r) * p.ldr + hh * 256 + 4 * lane);
      }
    f4v pv[16];
    if (with_partner) {
      const unsigned* pf = p.flags + (bid ^ 8) * 8 + wave;
      unsigned spins = 0;
      while (__hip_atomic_load(pf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) {
          if (lane == 0) __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      const char* in = reinterpret_cast<const char*>(p.slab) + (size_t)(bid ^ 8) * SK_SLAB + (size_t)(wave * 16) * 1024 + lane * 16;
#pragma unroll
      for (int e = 0; e < 16; ++e)
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(pv[e]) : "v"(in + e * 1024) : "memory");
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]), "+v"(pv[7]),
                     "+v"(pv[8]), "+v"(pv[9]), "+v"(pv[10]), "+v"(pv[11]), "+v"(pv[12]), "+v"(pv[13]), "+v"(pv[14]), "+v"(pv[15])
                   :
                   : "memory");
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) pv[e] = f4v{0.f, 0.f, 0.f, 0.f};
    }
    // D^T fragment: lane = row, 4 consecutive columns per register quad
    char* rowp = smem + (size_t)(wm * 32 + (lane & 31)) * SK_XROW + (wn * (32 * NJ) + 4 * lh) * 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f4v q = pv[j * 4 + g];
        *reinterpret_cast<float4a*>(rowp + (j * 32 + 8 * g) * 4) =
            make_float4(a[j][4 * g + 0] + q[0], a[j][4 * g + 1] + q[1], a[j][4 * g + 2] + q[2], a[j][4 * g + 3] + q[3]);
      }
    __syncthreads();
    const int r0 = wave * 8;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int col = hh * 256 + 4 * lane;
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) b4 = *reinterpret_cast<const float4*>(p.bias + col);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float4 v = *reinterpret_cast<const float4a*>(smem + (size_t)(r0 + r) * SK_XROW + col * 4);
        xv[hh][r].x += v.x + b4.x; xv[hh][r].y += v.y + b4.y; xv[hh][r].z += v.z + b4.z; xv[hh][r].w += v.w + b4.w;
      }
      if (p.out_x) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
          if (mb + r < p.M) *reinterpret_cast<float4*>(p.out_x + (size_t)(mb + r) * p.ldx + col) = xv[hh][r];
      }
    }
    if (!p.ln_g) return;
    // LayerNorm of the complete rows (two-pass statistics; one wave = one row, reductions on the VALU via DPP)
    float4 g4[2], be4[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      g4[hh] = *reinterpret_cast<const float4*>(p.ln_g + hh * 256 + 4 * lane);
      be4[hh] = *reinterpret_cast<const float4*>(p.ln_b + hh * 256 + 4 * lane);
    }
    float mean[8], rstd[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float s = ((xv[0][r].x + xv[0][r].y) + (xv[0][r].z + xv[0][r].w)) + ((xv[1][r].x + xv[1][r].y) + (xv[1][r].z + xv[1][r].w));
      mean[r] = sk_wave_sum(s) * (1.0f / SK_BN);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float m = mean[r];
      xv[0][r].x -= m; xv[0][r].y -= m; xv[0][r].z -= m; xv[0][r].w -= m;
      xv[1][r].x -= m; xv[1][r].y -= m; xv[1][r].z -= m; xv[1][r].w -= m;
      const float q = ((xv[0][r].x * xv[0][r].x + xv[0][r].y * xv[0][r].y) + (xv[0][r].z * xv[0][r].z + xv[0][r].w * xv[0][r].w)) +
                      ((xv[1][r].x * xv[1][r].x + xv[1][r].y * xv[1][r].y) + (xv[1][r].z * xv[1][r].z + xv[1][r].w * xv[1][r].w));
      rstd[r] = 1.0f / sqrtf(sk_wave_sum(q) * (1.0f / SK_BN) + p.eps);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int m = mb + r;
      if (m < p.M) {
        const float k = rstd[r];
        const float4 d0 = xv[0][r], d1 = xv[1][r];
        const float4 y0 = make_float4(d0.x * k * g4[0].x + be4[0].x, d0.y * k * g4[0].y + be4[0].y,
                                      d0.z * k * g4[0].z + be4[0].z, d0.w * k * g4[0].w + be4[0].w);
        const float4 y1 = make_float4(d1.x * k * g4[1].x + be4[1].x, d1.y * k * g4[1].y + be4[1].y,
                                      d1.z * k * g4[1].z + be4[1].z, d1.w * k * g4[1].w + be4[1].w);
        if (p.out_n16) {
          half_t* o = p.out_n16 + (size_t)m * p.ldn16 + 4 * lane;
          *reinterpret_cast<h4*>(o) = h4{(half_t)y0.x, (half_t)y0.y, (half_t)y0.z, (half_t)y0.w};
          *reinterpret_cast<h4*>(o + 256) = h4{(half_t)y1.x, (half_t)y1.y, (half_t)y1.z, (half_t)y1.w};
        }
        if (p.out_n32) {
          float* o = p.out_n32 + (size_t)m * p.ldn32 + 4 * lane;
          *reinterpret_cast<float4*>(o) = y0;
          *reinterpret_cast<float4*>(o + 256) = y1;
        }
      }
    }
  };

  if (SPLIT) {
    // ---- exchange: acc[1] (the rows the partner owns) goes out in accumulator layout, 1 KiB per store instruction
    char* out = reinterpret_cast<char*>(p.slab) + (size_t)bid * SK_SLAB + (size_t)(wave * 16) * 1024 + lane * 16;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f4v v = {acc[1][j][4 * g + 0], acc[1][j][4 * g + 1], acc[1][j][4 * g + 2], acc[1][j][4 * g + 3]};
        sk_store16_sc1(out + (j * 4 + g) * 1024, v);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // R1: EVERY storing wave drains, then one lane publishes its wave's flag
    if (lane == 0) __hip_atomic_store(p.flags + bid * 8 + wave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    finish(h, acc[0], true);
  } else {
    finish(0, acc[0], false);
    __syncthreads();                                         // every wave has read its rows of the first half
    finish(1, acc[1], false);
  }
}

static int sk_grid(int M) { return 16 * cdiv(cdiv(M, SK_BM), 8); }   // pair form (the scratch is sized for it)
size_t gemm_sk_slab_bytes(int M) { return (size_t)sk_grid(M) * SK_SLAB; }
size_t gemm_sk_flag_bytes(int M) { return (size_t)sk_grid(M) * 8 * 4; }

bool gemm_sk_applicable(const GemmRcArgs& a) {
  if (a.fsmn_v) return false;                                // the FSMN memory is the 64-row kernel's epilogue term
  if (a.M <= 0 || a.K < 192 || a.K % 64 != 0 || a.lda % 8 != 0 || a.ldw % 8 != 0) return false;   // (K >= 96 would do without the split)
  if ((a.resid && a.ldr % 4 != 0) || (a.out_x && a.ldx % 4 != 0) || (a.out_n16 && a.ldn16 % 4 != 0) || (a.out_n32 && a.ldn32 % 4 != 0))
    return false;
  if (!a.ln_g != !a.ln_b || (!a.ln_g && (a.out_n16 || a.out_n32))) return false;
  return a.out_x || a.out_n16 || a.out_n32;
}

void launch_gemm_sk(hipStream_t s, const GemmRcArgs& a, float* slab, unsigned* flags, unsigned* err, bool split) {
  PF_CHECK(gemm_sk_applicable(a), PF_ERR_INVALID_ARG, "gemm_sk: shape / epilogue not covered by the 128-row row-complete kernel");
  PF_CHECK(!split || (slab && flags && err), PF_ERR_INVALID_ARG, "gemm_sk: exchange scratch missing");
  SkDev d;
  d.A = a.A; d.W = a.W; d.bias = a.bias; d.resid = a.resid; d.out_x = a.out_x;
  d.ln_g = a.ln_g; d.ln_b = a.ln_b; d.out_n16 = a.out_n16; d.out_n32 = a.out_n32;
  d.slab = slab; d.flags = flags; d.err = err;
  d.lda = a.lda; d.ldw = a.ldw; d.ldr = a.ldr; d.ldx = a.ldx; d.ldn16 = a.ldn16; d.ldn32 = a.ldn32;
  d.M = a.M; d.K = a.K; d.a_blocked = a.a_blocked; d.n_rb = cdiv(a.M, SK_BM);
  d.eps = a.eps;
  static std::mutex init_mu;
  static bool attr_set[64] = {false};
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(init_mu);
    if (!attr_set[dev & 63]) {
      PF_HIP(hipFuncSetAttribute((const void*)gemm_sk_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS));
      PF_HIP(hipFuncSetAttribute((const void*)gemm_sk_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS));
      attr_set[dev & 63] = true;
    }
  }
  note_gemm_kernel(split ? "gemm_sk_kernel<true>" : "gemm_sk_kernel<false>");
  if (split) hipLaunchKernelGGL(gemm_sk_kernel<true>, dim3((unsigned)sk_grid(a.M)), dim3(512), SK_LDS, s, d);
  else hipLaunchKernelGGL(gemm_sk_kernel<false>, dim3((unsigned)d.n_rb), dim3(512), SK_LDS, s, d);
  PF_HIP(hipGetLastError());
}

}  // namespace pf
