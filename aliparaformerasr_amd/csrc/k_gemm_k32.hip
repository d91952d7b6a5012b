// k_gemm_k32.hip — persistent 256 x 128 tile GEMM with fp32 results for deep-K projections (FFN down-projection):
//
//   C[M,N] = A[M,K] * W[N,K]^T + bias [* scale] [+ resid] [ReLU]      -> fp32 (and / or f16, row-major)
//
// (MatMul + Add + Add nodes of the graph InferenceSession.Run executes, AliParaformerAsr/OfflineProjOfParaformer.cs:68.)
//
// Why a third GEMM kernel (round 4).  rocprofv3 counters on the round-3 headline (profiles/round4_pmc_*.json): in
// gemm_f16_pp3<2,2> the matrix pipe is busy 37 % of the time, the waves are parked on waits 40 % of it and the vector L1
// reports "data pending from L2" on 56 % of the cycles — the K loop waits for its operands.  The same tile bytes move
// twice as fast through gemm_bigp_kernel's K loop (64 KB per 0.71 us there, 48 KB per 1.0 us here): what differs is the
// pipeline, not the memory system.  gemm_f16_pp3 keeps two 48 KB stages in flight behind two barriers per k-step; with a
// landing latency of ~1.7 us under load that is 96 KB / 1.7 us = 56 GB/s per CU whatever else is tuned (Little's law —
// the figure tools/ubench/kstep.hip measured for that skeleton with no MFMA at all).  This kernel is the 256 x 256
// kernel's pipeline at half the tile width: k-steps of 32, a SIX-stage ring of 24 KB stages with FIVE in flight
// (120 KB), one barrier per k-step in mid-step, fragment reads and their counted waits in inline asm, branch-free steps
// with constant wait immediates.  8 waves as 4 x 2, wave tile 64 x 64 (2 x 2 MFMA 32x32x16 blocks), so the fp32
// row-segment epilogue of gemm_f16_pp3 (residual loads and result stores as whole 256-byte row pieces through a
// 2 KiB per-wave LDS scratch) applies unchanged.  The A operand may be row-major or in the blocked activation layout
// (kernels.h) the FFN-up kernel writes.
#include "kernels.h"

#include <algorithm>
#include <cstdlib>
#include <mutex>

namespace pf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef float4 __attribute__((may_alias)) float4a;

struct K32Dev {
  const half_t* A; const half_t* W; const float* bias;
  float* out_f32; half_t* out_f16; const float* resid; const float* add2;
  int lda, ldw, ldc32, ldc16, ldr, ld2;
  int M, N, K, tiles_m, tiles_n;
  int relu, scale_cols, a_blocked;
  float scale;
};

constexpr int K3_BM = 256, K3_BN = 128, K3_BK = 32, K3_ROWB = K3_BK * 2, K3_S = 6;
constexpr int K3_A_BYTES = K3_BM * K3_ROWB, K3_W_BYTES = K3_BN * K3_ROWB, K3_STAGE = K3_A_BYTES + K3_W_BYTES;   // 16 + 8 KiB
constexpr int K3_RING = K3_S * K3_STAGE;                       // 144 KiB
constexpr int K3_LDS = K3_RING + 8 * 2048;                     // + 2 KiB epilogue scratch per wave = 160 KiB
constexpr int K3_PIECES = 3;                                   // LDS-DMA instructions per wave and stage (2 of A, 1 of W)

__device__ __forceinline__ void k3_glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void k3_wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

__global__ __launch_bounds__(512, 1) void gemm_k32_kernel(K32Dev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, lh = lane >> 5;
  auto swz = [](int row) __attribute__((always_inline)) -> int { return (row >> 2) & 3; };

  // ---- tile schedule: persistent block b (on XCD b % 8) takes tiles slot, slot + G, ...; XCDs get contiguous runs, n fastest
  const int G = gridDim.x, bid = blockIdx.x;
  const int total_tiles = p.tiles_m * p.tiles_n;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int slot = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int n_my = slot < total_tiles ? (total_tiles - slot + G - 1) / G : 0;
  const int nk = p.K / K3_BK;
  const int T = n_my * nk;
  if (T == 0) return;

  // ---- DMA cursor (uniform): stage is_t = k-step is_k of tile is_tile, going into ring slot is_slot
  const int srow = lane >> 2, schunk = lane & 3;
  unsigned a_vo[2], w_vo;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pi = wave + 8 * i;                             // 1 KiB piece of the A stage
    const int row = pi * 16 + srow;
    a_vo[i] = p.a_blocked ? (unsigned)((pi >> 1) * (p.K >> 3) * 512 + (pi & 1) * 1024 + lane * 16)
                          : (unsigned)(row * p.lda + ((schunk ^ swz(row)) << 3)) * 2u;
  }
  {
    const int row = wave * 16 + srow;
    w_vo = (unsigned)(row * p.ldw + ((schunk ^ swz(row)) << 3)) * 2u;
  }
  const int a_step = p.a_blocked ? (K3_BK / 8) * 512 : K3_BK * 2;
  int is_tile = slot, is_k = 0, is_t = 0, is_slot = 0;
  const char* is_a;
  const char* is_w;
  auto set_issue_tile = [&]() __attribute__((always_inline)) {
    const int tm = is_tile / p.tiles_n, tn = is_tile - tm * p.tiles_n;
    is_a = p.a_blocked ? reinterpret_cast<const char*>(p.A) + (size_t)tm * (K3_BM / 32) * (size_t)(p.K >> 3) * 512
                       : reinterpret_cast<const char*>(p.A + (size_t)tm * K3_BM * p.lda);
    is_w = reinterpret_cast<const char*>(p.W + (size_t)tn * K3_BN * p.ldw);
  };
  set_issue_tile();
  auto issue_piece = [&](int q) __attribute__((always_inline)) {
    char* st = smem + is_slot * K3_STAGE + wave * 1024;
    if (q < 2) k3_glds16(is_a + a_vo[q & 1], st + (q & 1) * 8192);
    else k3_glds16(is_w + w_vo, st + K3_A_BYTES);
  };
  // past the end of this workgroup's tile list the cursor stops advancing (the last stage is re-loaded into a slot nobody
  // reads again): every step issues exactly K3_PIECES operations per wave, every wait immediate is a constant
  auto issue_advance = [&]() __attribute__((always_inline)) {
    if (is_t + 1 < T) {
      ++is_t;
      is_a += a_step; is_w += K3_BK * 2;
      if (++is_k == nk) { is_k = 0; is_tile += G; set_issue_tile(); }
    }
    is_slot = is_slot + 1 == K3_S ? 0 : is_slot + 1;
  };

  // ---- fragment read offsets inside a stage (bytes)
  unsigned fa[2][2], fb[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra = wm * 64 + i * 32 + (lane & 31);
      const int rb = wn * 64 + i * 32 + (lane & 31);
      fa[s][i] = p.a_blocked ? (unsigned)((wm * 2 + i) * 2048 + (2 * s + lh) * 512 + (lane & 31) * 16)
                             : (unsigned)(ra * K3_ROWB + (((2 * s + lh) ^ swz(ra)) << 4));
      fb[s][i] = (unsigned)(K3_A_BYTES + rb * K3_ROWB + (((2 * s + lh) ^ swz(rb)) << 4));
    }
  f16x acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  h8 a0[2] = {}, b0[2] = {}, a1[2] = {}, b1[2] = {};
  // fragment reads and their waits are inline asm (k_gemm_big.hip explains why: the compiler's own wait insertion puts a
  // full lgkmcnt(0) in front of the first MFMA and answers LDS reads behind in-flight LDS-DMA with vmcnt(0))
  auto load = [&](unsigned rd, int s, h8 (&af)[2], h8 (&bf)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(bf[j]) : "v"(rd + fb[s][j]) : "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(af[i]) : "v"(rd + fa[s][i]) : "memory");
  };
  auto frag_wait4 = [&](h8 (&af)[2], h8 (&bf)[2]) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[0]), "+v"(af[1]), "+v"(bf[0]), "+v"(bf[1]));
  };
  auto frag_wait0 = [&](h8 (&af)[2], h8 (&bf)[2]) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(bf[0]), "+v"(bf[1]));
  };
  // four MFMAs of one half-step; with `dma` the three pieces of the cursor's stage go out between them
  auto mma = [&](h8 (&af)[2], h8 (&bf)[2], bool dma) __attribute__((always_inline)) {
    int q = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        if (q < K3_PIECES) {
          __builtin_amdgcn_sched_barrier(0);
          if (dma) issue_piece(q);
          ++q;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
  };

  // ---- tile end: fp32 epilogue through the wave's LDS scratch.  The D^T fragment gives a lane one output ROW; 4 rows at
  // a time are written to the scratch by their owner lanes and read back row-contiguous (16 lanes x 16 B = one 256-byte
  // row segment), so residual / add2 loads and the result stores are whole lines.
  // The scratch accesses are inline asm as well: a compiler-visible LDS read behind in-flight LDS-DMA is answered with
  // s_waitcnt vmcnt(0), which here would serialise the sixteen residual loads of a wave (one HBM round trip per chunk).
  typedef float f4v __attribute__((ext_vector_type(4)));
  const float lo = p.relu ? 0.f : -INFINITY;
  auto epilogue = [&](int tile, unsigned scr) __attribute__((always_inline)) {       // scr: LDS byte address of the wave's 2 KiB
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * K3_BM + wm * 64, n0 = tn * K3_BN + wn * 64;
    const int lc = lane & 31;
    const int rr = lane >> 4, cc = lane & 15;                // reader: row in chunk, 4-column group
    const int n = n0 + cc * 4;
    const unsigned wq = scr + (lc & 3) * 272 + lh * 16;
    const unsigned rq = scr + rr * 272 + cc * 16;
    f4v b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) b4 = *reinterpret_cast<const f4v*>(p.bias + n);
    const float sc = (n < p.scale_cols) ? p.scale : 1.f;
    // all sixteen residual row pieces of this lane are requested up front (64 registers): the asm blocks below are
    // compiler barriers for memory operations, so a load placed inside the chunk loop would wait for its own round trip
    // before the next one is even issued
    f4v r4[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int m = m0 + i * 32 + c * 4 + rr;
        const int ml = m < p.M ? m : p.M - 1;                // rows of the last tile beyond M: clamped loads, no stores
        r4[i][c] = f4v{0.f, 0.f, 0.f, 0.f};
        if (p.resid) r4[i][c] = *reinterpret_cast<const f4v*>(p.resid + (size_t)ml * p.ldr + n);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int m = m0 + i * 32 + c * 4 + rr;
        const bool live = m < p.M;
        if ((lc >> 2) == c) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f4v q4 = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
              asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(wq), "v"(q4), "n"((j * 32 + 8 * g) * 4) : "memory");
            }
        }
        f4v v;
        // (the LDS executes one wave's operations in issue order: the read sees the rows written just above)
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(rq) : "memory");
        v = (v + b4) * sc;
        v += r4[i][c];
        v = f4v{fmaxf(v.x, lo), fmaxf(v.y, lo), fmaxf(v.z, lo), fmaxf(v.w, lo)};
        if (live) {
          if (p.out_f32) *reinterpret_cast<f4v*>(p.out_f32 + (size_t)m * p.ldc32 + n) = v;
          if (p.out_f16)
            *reinterpret_cast<h4*>(p.out_f16 + (size_t)m * p.ldc16 + n) = h4{(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  };

  // ---- prologue: five stages in flight
#pragma unroll
  for (int st = 0; st < K3_S - 1; ++st) {
#pragma unroll
    for (int q = 0; q < K3_PIECES; ++q) issue_piece(q);
    issue_advance();
  }
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;   // LDS byte address of the ring
  k3_wait_vmcnt<(K3_S - 2) * K3_PIECES>();                     // stage 0 of this wave has landed
  __builtin_amdgcn_s_barrier();
  load(lds0, 0, a0, b0);

  int tile = slot, k = 0;
  unsigned rd = lds0;                                          // stage t
  for (int t = 0; t < T; ++t) {
    const unsigned rd_next = rd + K3_STAGE == lds0 + K3_RING ? lds0 : rd + K3_STAGE;
    __builtin_amdgcn_sched_barrier(0);
    load(rd, 1, a1, b1);                                       // second half of stage t, under the first half's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    frag_wait4(a0, b0);                                        // first half (requested behind the last barrier) is in
    __builtin_amdgcn_s_setprio(1);
    mma(a0, b0, true);                                         // + the pieces of stage t + 5 (slot freed by the barrier of step t - 1)
    __builtin_amdgcn_s_setprio(0);
    issue_advance();
    __builtin_amdgcn_sched_barrier(0);
    // stage t + 1 of this wave has landed: the 4 x 3 operations of stages t + 2 .. t + 5 may still be in flight (after a
    // tile end the epilogue's stores are younger than some of them: the constant then over-waits, never under-waits —
    // vmcnt retires in order)
    k3_wait_vmcnt<(K3_S - 2) * K3_PIECES>();
    frag_wait0(a1, b1);                                        // every read of stage t has retired
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    load(rd_next, 0, a0, b0);                                  // (after the last step: a slot nobody uses)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma(a1, b1, false);
    __builtin_amdgcn_s_setprio(0);
    rd = rd_next;
    if (++k == nk) {
      epilogue(tile, lds0 + K3_RING + wave * 2048);
      k = 0; tile += G;
    }
  }
  k3_wait_vmcnt<0>();                                          // the clamped DMA of the last steps must not outlive the workgroup's LDS
}

bool gemm_k32_applicable(const GemmArgs& a) {
  if (!a.out_f32 && !a.out_f16) return false;
  if (a.out_blocked || a.add2) return false;               // (a second fp32 addend would put loads — and their waits — into the store loop)
  if (a.K % 64 != 0 || a.K < 256 || a.N % K3_BN != 0 || a.lda % 8 != 0 || a.ldw % 8 != 0) return false;
  if ((a.out_f32 && a.ldc32 % 4 != 0) || (a.out_f16 && a.ldc16 % 4 != 0) || (a.resid && a.ldr % 4 != 0) || (a.add2 && a.ld2 % 4 != 0)) return false;
  if (a.scale_cols % 64 != 0) return false;
  return true;
}

void launch_gemm_k32(hipStream_t s, const GemmArgs& a, int cus) {
  K32Dev d{};
  d.A = a.A; d.W = a.W; d.bias = a.bias;
  d.out_f32 = a.out_f32; d.out_f16 = a.out_f16; d.resid = a.resid; d.add2 = a.add2;
  d.lda = a.lda; d.ldw = a.ldw; d.ldc32 = a.ldc32; d.ldc16 = a.ldc16; d.ldr = a.ldr; d.ld2 = a.ld2;
  d.M = a.M; d.N = a.N; d.K = a.K;
  d.tiles_m = cdiv(a.M, K3_BM); d.tiles_n = a.N / K3_BN;
  d.relu = a.relu; d.scale_cols = a.scale_cols; d.scale = a.scale_cols > 0 ? a.scale : 1.f;
  d.a_blocked = a.a_blocked;
  static std::mutex init_mu;
  static bool attr_set[64] = {false};
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(init_mu);
    if (!attr_set[dev & 63]) {
      PF_HIP(hipFuncSetAttribute((const void*)gemm_k32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, K3_LDS));
      attr_set[dev & 63] = true;
    }
  }
  const int total = d.tiles_m * d.tiles_n;
  if (total == 0) return;
  note_gemm_kernel("gemm_k32_kernel");
  hipLaunchKernelGGL(gemm_k32_kernel, dim3((unsigned)std::min(total, cus)), dim3(512), K3_LDS, s, d);
  PF_HIP(hipGetLastError());
}

}  // namespace pf
