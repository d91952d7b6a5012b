// k_gemm.hip — f16 x f16 -> f32 GEMM on the gfx950 matrix cores.
//
// Replaces the MatMul/Gemm nodes ONNX Runtime executes inside InferenceSession.Run
// (reference call site AliParaformerAsr/OfflineProjOfParaformer.cs:68): QKV, attention
// output, FFN, CIF conv (as im2col GEMM), decoder projections and the vocabulary layer.
//
//   C[M,N] = A[M,K] * W[N,K]^T   (both operands K-contiguous, "B^T input")
//
// Persistent kernel: one (or two) workgroups per CU walk a static list of output tiles; the
// (tile, k-step) sequence is ONE flat software pipeline, so the HBM->LDS prefetch of the next
// tile's first k-slices runs underneath the current tile's last MFMAs and its epilogue (the
// model's GEMMs have K = 512: only 8 k-steps per tile, a per-tile prologue would dominate).
//   * operand tiles go HBM -> LDS with global_load_lds_dwordx4 into an S-deep ring; a stage is
//     consumed after a COUNTED s_waitcnt vmcnt(N) + raw s_barrier, so S-1 stages stay in flight
//     across barriers (cdna guide T3+T4);
//   * the LDS image of global_load_lds is lane-linear: the bank-conflict XOR swizzle is applied
//     to the SOURCE address and to the ds_read_b128 address (rule 21);
//   * each wavefront owns a 64x64 block as 2x2 v_mfma_f32_32x32x16_f16 accumulators, issued as
//     D^T = W_tile * X_tile^T so that a lane holds 4 consecutive output columns: the epilogue
//     (bias, q-scale, FSMN add, residual, ReLU, fp32 / f16 stores) is 16/8-byte vector I/O;
//   * tiles are dealt so that the 8 XCDs each work on a contiguous run of tiles (n fastest) and
//     share A panels / W tiles in their private L2.
#include "kernels.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace pf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));
// LDS scratch is written with one vector width and read back with another: these accesses must
// not be reordered by type-based alias analysis
typedef h8 __attribute__((may_alias)) h8a;
typedef h4 __attribute__((may_alias)) h4a;
typedef float4 __attribute__((may_alias)) float4a;
typedef float __attribute__((may_alias)) floata;

struct GemmDev {
  const half_t* A; const half_t* W; const float* bias;
  float* out_f32; half_t* out_f16; const float* resid; const float* add2;
  int lda, ldw, ldc32, ldc16, ldr, ld2;
  int M, N, K, tiles_m, tiles_n;
  int relu, scale_cols; float scale;
  int out_padded;  // f16 output buffer has >= round_up(M,256) rows (lets the epilogue store whole tiles)
  long long* ts;   // optional s_memtime trace (PF_GEMM_TS=1): [2 waves][64 steps][8]
  int dbg;   // diagnostic ablation bits (PF_GEMM_DBG): 1 no MFMA, 2 no DMA after prologue, 4 no epilogue, 8 no ds_read
};

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  // gfx9 s_waitcnt simm16: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

// WM x WN wavefronts, each 64x64; block tile (64*WM) x (64*WN); BK in {32, 64}; S-stage ring.
template <int WM, int WN, int BK, int S, bool FRAGS_FIRST, int ABLATE = 0>
__global__ __launch_bounds__(WM * WN * 64, 1) void gemm_f16_persistent(GemmDev p) {
  constexpr int NW = WM * WN;
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int ROWB = BK * 2;                     // bytes per tile row in LDS
  constexpr int CPR = ROWB / 16;                   // 16-byte chunks per row (4 or 8)
  constexpr int RPI = 64 / CPR;                    // rows covered by one wave-instruction (1 KiB)
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = A_BYTES / 1024, B_INSTR = B_BYTES / 1024;
  constexpr int A_PW = A_INSTR / NW, B_PW = B_INSTR / NW;     // wave-instructions per wave
  constexpr int LPS = A_PW + B_PW;                 // glds per wave per step
  constexpr int KSUB = BK / 16;
  static_assert(A_INSTR % NW == 0 && B_INSTR % NW == 0, "tile does not split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lh = lane >> 5;

  // ---- tile schedule: persistent block b (on XCD b % 8) takes tiles r*G + perm(b)
  const int G = gridDim.x, bid = blockIdx.x;
  const int total_tiles = p.tiles_m * p.tiles_n;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int slot = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int n_my = slot < total_tiles ? (total_tiles - slot + G - 1) / G : 0;
  const int nk = p.K / BK;
  const int total_steps = n_my * nk;
  if (total_steps == 0) return;

  // swizzle: BK=64: chunk ^ ((row>>1)&7) ; BK=32: chunk ^ ((row>>2)&3)
  auto swz = [](int row) -> int { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

  // ---- staging: this wave's wave-instruction i covers tile rows (wave + NW*i)*RPI .. +RPI-1
  const int srow = lane / CPR, schunk = lane % CPR;
  auto stage = [&](int step) {
    const int tseq = step / nk, kt = step - tseq * nk;
    const int tile = slot + tseq * G;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    char* ab = smem + (step % S) * STAGE;
    char* wb = ab + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
      const int q = wave + NW * i;
      const int row = q * RPI + srow;
      const int c = schunk ^ swz(row);
      glds16(p.A + (size_t)(tm * BM + row) * p.lda + kt * BK + c * 8, ab + q * 1024);
    }
#pragma unroll
    for (int i = 0; i < B_PW; ++i) {
      const int q = wave + NW * i;
      const int row = q * RPI + srow;
      const int c = schunk ^ swz(row);
      glds16(p.W + (size_t)(tn * BN + row) * p.ldw + kt * BK + c * 8, wb + q * 1024);
    }
  };

  // ---- fragment read offsets
  int a_off[2], b_off[2], a_swz[2], b_swz[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wm * 64 + i * 32 + (lane & 31);
    const int rb = wn * 64 + i * 32 + (lane & 31);
    a_off[i] = ra * ROWB; a_swz[i] = swz(ra);
    b_off[i] = rb * ROWB; b_swz[i] = swz(rb);
  }

  f16x acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- prologue: S-1 stages in flight
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < total_steps) stage(s);

  int kt = 0, tseq = 0;
  for (int step = 0; step < total_steps; ++step) {
    // stage `step` must have landed: at most the S-2 younger stages may still be in flight
    if (step + S - 2 < total_steps) wait_vmcnt<LPS * (S - 2)>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (ABLATE != 2 && step + S - 1 < total_steps) stage(step + S - 1);     // refills the buffer consumed at step-1
    const char* ab = smem + (step % S) * STAGE;
    const char* wb = ab + A_BYTES;
    // all fragment reads of the k-step first, then the MFMA burst: the two wavefronts that share
    // a SIMD alternate between an LDS phase and a matrix phase instead of both stalling on
    // ds_read latency between groups of 4 MFMAs
    h8 af[KSUB][2], bf[KSUB][2];
#pragma unroll
    for (int s = 0; s < KSUB; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[s][i] = *(const h8*)(ab + a_off[i] + (((2 * s + lh) ^ a_swz[i]) << 4));
        bf[s][i] = *(const h8*)(wb + b_off[i] + (((2 * s + lh) ^ b_swz[i]) << 4));
      }
    if (FRAGS_FIRST) __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if (ABLATE == 1) {
      // ablation: keep the operands live, skip the matrix work
#pragma unroll
      for (int s = 0; s < KSUB; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) { asm volatile("" :: "v"(af[s][i]), "v"(bf[s][i])); }
    } else {
#pragma unroll
    for (int s = 0; s < KSUB; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[s][j], af[s][i], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (++kt < nk) continue;
    kt = 0;

    // ---- epilogue of this tile (lane: one row m, 4 consecutive columns per register quad)
    const int tile = slot + tseq * G;
    ++tseq;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + wm * 64 + i * 32 + (lane & 31);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn * 64 + j * 32 + 8 * g + 4 * lh;
          float v[4] = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          acc[i][j][4 * g + 0] = 0.f; acc[i][j][4 * g + 1] = 0.f; acc[i][j][4 * g + 2] = 0.f; acc[i][j][4 * g + 3] = 0.f;
          if (ABLATE == 3) { asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); continue; }
          if (m >= p.M || n >= p.N) continue;
          if (n + 3 < p.N) {
            if (p.bias) {
              const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
              v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
            }
            if (n < p.scale_cols) { v[0] *= p.scale; v[1] *= p.scale; v[2] *= p.scale; v[3] *= p.scale; }
            if (p.add2) {
              const float4 a4 = *reinterpret_cast<const float4*>(p.add2 + (size_t)m * p.ld2 + n);
              v[0] += a4.x; v[1] += a4.y; v[2] += a4.z; v[3] += a4.w;
            }
            if (p.resid) {
              const float4 r4 = *reinterpret_cast<const float4*>(p.resid + (size_t)m * p.ldr + n);
              v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            }
            if (p.relu) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            if (p.out_f32)
              *reinterpret_cast<float4*>(p.out_f32 + (size_t)m * p.ldc32 + n) = make_float4(v[0], v[1], v[2], v[3]);
            if (p.out_f16) {
              h4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
              *reinterpret_cast<h4*>(p.out_f16 + (size_t)m * p.ldc16 + n) = hv;
            }
          } else {
            for (int e = 0; e < 4 && n + e < p.N; ++e) {
              float x = v[e] + (p.bias ? p.bias[n + e] : 0.f);
              if (n + e < p.scale_cols) x *= p.scale;
              if (p.add2) x += p.add2[(size_t)m * p.ld2 + n + e];
              if (p.resid) x += p.resid[(size_t)m * p.ldr + n + e];
              if (p.relu) x = x > 0.f ? x : 0.f;
              if (p.out_f32) p.out_f32[(size_t)m * p.ldc32 + n + e] = x;
              if (p.out_f16) p.out_f16[(size_t)m * p.ldc16 + n + e] = (half_t)x;
            }
          }
        }
      }
    }
    // the epilogue's loads/stores share the vmcnt queue with the in-flight prefetches: drain so
    // that the counted waits above start from a known state
    if (ABLATE != 4) wait_vmcnt<0>();
  }
}


// ------------------------------------------------------------------------------------------
// Ping-pong variant: 256x128 tile, 8 wavefronts = two groups of 4 that share the 4 SIMDs
// pairwise (wave w and w+4 sit on the same SIMD).  While group A issues its 16-MFMA burst for
// k-step k, group B reads its fragments for the same k-step from LDS (and vice versa one phase
// later), so the matrix pipe of every SIMD always has a burst queued and LDS reads never sit
// in front of MFMAs.  One s_barrier per phase; operand stages arrive by LDS-DMA with counted
// vmcnt, S-1 stages in flight.
//   phase 2k   : A = LOAD(k)    (+ issues DMA for step k+S-1)   B = COMPUTE(k-1) (+ DMA k+S-1)
//   phase 2k+1 : A = COMPUTE(k)                                 B = LOAD(k)
// Stage k is read in phases 2k (A) and 2k+1 (B); its buffer is re-filled from phase 2k+2 on.
template <int BK, int S, int ROT>
__global__ __launch_bounds__(512, 1) void gemm_f16_pingpong(GemmDev p) {
  constexpr int WN = 2, NW = 8;
  constexpr int BM = 256, BN = 128;
  constexpr int ROWB = BK * 2, CPR = ROWB / 16, RPI = 64 / CPR;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PW = A_BYTES / 1024 / NW, B_PW = B_BYTES / 1024 / NW;
  constexpr int LPS = A_PW + B_PW;
  constexpr int KSUB = BK / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                         // 0 = A (rows 0..127), 1 = B (rows 128..255)
  const int wm = wave / WN, wn = wave % WN;
  const int lh = lane >> 5;

  const int G = gridDim.x, bid = blockIdx.x;
  const int total_tiles = p.tiles_m * p.tiles_n;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int slot = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int n_my = slot < total_tiles ? (total_tiles - slot + G - 1) / G : 0;
  const int nk = p.K / BK;
  const int T = n_my * nk;
  if (T == 0) return;

  auto swz = [](int row) -> int { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

  // ---- LDS-DMA addressing, hoisted: per-lane constant byte offsets (32 bit) + a uniform base
  // (SGPR pair) that advances by BK*2 bytes per k-step; nothing but an s_add per step in the loop.
  const int srow = lane / CPR, schunk = lane % CPR;
  unsigned a_vo[A_PW], w_vo[B_PW];
#pragma unroll
  for (int i = 0; i < A_PW; ++i) {
    const int row = (wave + NW * i) * RPI + srow;
    a_vo[i] = (unsigned)(row * p.lda + ((schunk ^ swz(row)) << 3)) * 2u;
  }
#pragma unroll
  for (int i = 0; i < B_PW; ++i) {
    const int row = (wave + NW * i) * RPI + srow;
    w_vo[i] = (unsigned)(row * p.ldw + ((schunk ^ swz(row)) << 3)) * 2u;
  }
  const int rot = ROT == 0 ? 0 : (ROT == 1 ? slot % nk : (slot / 8) % nk);
  int is_tseq = 0, is_kt = 0, is_buf = 0;            // cursor of the next step to issue (uniform)
  const char* is_a = nullptr;
  const char* is_w = nullptr;
  auto set_issue_tile = [&]() {
    const int tile = slot + is_tseq * G;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    is_a = reinterpret_cast<const char*>(p.A + (size_t)tm * BM * p.lda);
    is_w = reinterpret_cast<const char*>(p.W + (size_t)tn * BN * p.ldw);
  };
  set_issue_tile();
  auto issue_next = [&]() {
    char* ab = smem + is_buf * STAGE + wave * 1024;
    // K-order rotation: concurrent blocks start at different k-slices so that the CUs of an XCD
    // do not all request the same lines of a shared A panel / W tile in the same cycle window
    int kk = is_kt + rot;
    kk = kk >= nk ? kk - nk : kk;
    const char* ga = is_a + kk * (BK * 2);
    const char* gw = is_w + kk * (BK * 2);
#pragma unroll
    for (int i = 0; i < A_PW; ++i) glds16(ga + a_vo[i], ab + NW * i * 1024);
#pragma unroll
    for (int i = 0; i < B_PW; ++i) glds16(gw + w_vo[i], ab + A_BYTES + NW * i * 1024);
    is_buf = (is_buf + 1 == S) ? 0 : is_buf + 1;
    if (++is_kt == nk) {
      is_kt = 0;
      ++is_tseq;
      if (is_tseq < n_my) set_issue_tile();
    }
  };

  // ---- fragment read offsets inside a stage (bytes), hoisted
  unsigned fa[KSUB][2], fb[KSUB][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wm * 64 + i * 32 + (lane & 31);
    const int rb = wn * 64 + i * 32 + (lane & 31);
#pragma unroll
    for (int s = 0; s < KSUB; ++s) {
      fa[s][i] = (unsigned)(ra * ROWB + (((2 * s + lh) ^ swz(ra)) << 4));
      fb[s][i] = (unsigned)(A_BYTES + rb * ROWB + (((2 * s + lh) ^ swz(rb)) << 4));
    }
  }

  f16x acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nout = (p.out_f32 ? 1 : 0) + (p.out_f16 ? 1 : 0);
  int credit_waits = 0, credit_ns = 0;      // epilogue stores still younger than the awaited DMA

  // wait until this wave's DMA of step `j` has landed; `issued_to` = last step this wave issued
  auto wait_step = [&](int j, int issued_to) {
    const int younger = issued_to - j;               // DMA sets issued after step j
    if (younger < S - 2) { wait_vmcnt<0>(); return; }
    if (credit_waits > 0) {
      --credit_waits;
      if (credit_ns == 16) { wait_vmcnt<LPS * (S - 2) + 16>(); return; }
      if (credit_ns == 32) { wait_vmcnt<LPS * (S - 2) + 32>(); return; }
    }
    wait_vmcnt<LPS * (S - 2)>();
  };

  // ---- prologue: group A has steps 0..S-2 in flight, group B 0..S-1
  int issued = -1;
  for (int s = 0; s < S - 1 + grp; ++s)
    if (s < T) { issue_next(); issued = s; }
  wait_step(0, issued);
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();        // B runs one phase behind A

  h8 af[KSUB][2], bf[KSUB][2];
  int kt = 0, tseq = 0, rd_buf = 0;
  for (int k = 0; k < T; ++k) {
    // ================= LOAD(k) =================
    if (grp == 0 && k + S - 1 < T && !(p.dbg & 2)) { issue_next(); issued = k + S - 1; }
    if (!(p.dbg & 8)) {
      const char* st = smem + rd_buf * STAGE;
      rd_buf = (rd_buf + 1 == S) ? 0 : rd_buf + 1;
#pragma unroll
      for (int s = 0; s < KSUB; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          af[s][i] = *(const h8*)(st + fa[s][i]);
          bf[s][i] = *(const h8*)(st + fb[s][i]);
        }
    }
    if (grp == 1 && k + 1 < T) wait_step(k + 1, issued);     // B: before the barrier ending phase 2k+1
    __builtin_amdgcn_s_barrier();
    // ================= COMPUTE(k) =================
    if (grp == 1 && k + S < T && !(p.dbg & 2)) { issue_next(); issued = k + S; }
    __builtin_amdgcn_s_setprio(1);
    if (!(p.dbg & 1)) {
#pragma unroll
    for (int s = 0; s < KSUB; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[s][j], af[s][i], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (grp == 0 && k + 1 < T) wait_step(k + 1, issued);     // A: before the barrier ending phase 2k+1
    if (!(grp == 1 && k == T - 1)) __builtin_amdgcn_s_barrier();
    if (++kt < nk) continue;
    kt = 0;

    // ================= epilogue (runs at the head of this wave's next LOAD phase) ==========
    const int tile = slot + tseq * G;
    ++tseq;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * BM + wm * 64, n0 = tn * BN + wn * 64;
    const bool interior = (m0 + 64 <= p.M) && (n0 + 64 <= p.N);
    const int nb = n0 + 4 * lh;                       // this lane's first column
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + i * 32 + (lane & 31);
      const float* r_ptr = p.resid ? p.resid + (size_t)m * p.ldr + nb : nullptr;
      const float* a_ptr = p.add2 ? p.add2 + (size_t)m * p.ld2 + nb : nullptr;
      float* o32 = p.out_f32 ? p.out_f32 + (size_t)m * p.ldc32 + nb : nullptr;
      half_t* o16 = p.out_f16 ? p.out_f16 + (size_t)m * p.ldc16 + nb : nullptr;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dn = j * 32 + 8 * g;             // compile-time column offset
          const int n = nb + dn;
          float v[4] = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          acc[i][j][4 * g + 0] = 0.f; acc[i][j][4 * g + 1] = 0.f; acc[i][j][4 * g + 2] = 0.f; acc[i][j][4 * g + 3] = 0.f;
          if (p.dbg & 4) { asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); continue; }
          if (!interior && (m >= p.M || n >= p.N)) continue;
          if (interior || n + 3 < p.N) {
            if (p.bias) {
              const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
              v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
            }
            if (n < p.scale_cols) { v[0] *= p.scale; v[1] *= p.scale; v[2] *= p.scale; v[3] *= p.scale; }
            if (a_ptr) {
              const float4 a4 = *reinterpret_cast<const float4*>(a_ptr + dn);
              v[0] += a4.x; v[1] += a4.y; v[2] += a4.z; v[3] += a4.w;
            }
            if (r_ptr) {
              const float4 r4 = *reinterpret_cast<const float4*>(r_ptr + dn);
              v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            }
            if (p.relu) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            if (o32) *reinterpret_cast<float4*>(o32 + dn) = make_float4(v[0], v[1], v[2], v[3]);
            if (o16) {
              h4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
              *reinterpret_cast<h4*>(o16 + dn) = hv;
            }
          } else {
            for (int e = 0; e < 4 && n + e < p.N; ++e) {
              float x = v[e] + (p.bias ? p.bias[n + e] : 0.f);
              if (n + e < p.scale_cols) x *= p.scale;
              if (a_ptr) x += a_ptr[dn + e];
              if (r_ptr) x += r_ptr[dn + e];
              if (p.relu) x = x > 0.f ? x : 0.f;
              if (o32) o32[dn + e] = x;
              if (o16) o16[dn + e] = (half_t)x;
            }
          }
        }
      }
    }
    // the 16*nout stores just issued are YOUNGER than the DMA sets already in flight: the next
    // waits may leave them outstanding (vmcnt is in-order on gfx9).  Only for interior tiles,
    // where the store count is exact; otherwise stay conservative.
    if (interior && !(p.dbg & 4)) { credit_ns = 16 * nout; credit_waits = S - 2 + grp; }
    else { credit_ns = 0; credit_waits = 0; }
  }
}


// ------------------------------------------------------------------------------------------
// Ping-pong v2: same schedule as above with a branch-free steady state.  The two wave groups
// run their own straight-line loops; every step issues exactly LPS LDS-DMA loads (past the
// end they are clamped to the last step and land in a ring slot nobody reads again), so the
// counted s_waitcnt vmcnt(LPS*(S-2)) is a compile-time constant everywhere.
template <int BK, int S, bool INTERLEAVE, bool DEFER>
__global__ __launch_bounds__(512, 1) void gemm_f16_pp2(GemmDev p) {
  constexpr int WN = 2, NW = 8;
  constexpr int BM = 256, BN = 128;
  constexpr int ROWB = BK * 2, CPR = ROWB / 16, RPI = 64 / CPR;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PW = A_BYTES / 1024 / NW, B_PW = B_BYTES / 1024 / NW;
  constexpr int LPS = A_PW + B_PW;
  constexpr int KSUB = BK / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave / WN, wn = wave % WN;
  const int lh = lane >> 5;

  const int G = gridDim.x, bid = blockIdx.x;
  const int total_tiles = p.tiles_m * p.tiles_n;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int slot = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int n_my = slot < total_tiles ? (total_tiles - slot + G - 1) / G : 0;
  const int nk = p.K / BK;
  const int T = n_my * nk;
  if (T == 0) return;

  auto swz = [](int row) -> int { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
  const int srow = lane / CPR, schunk = lane % CPR;
  unsigned a_vo[A_PW], w_vo[B_PW];
#pragma unroll
  for (int i = 0; i < A_PW; ++i) {
    const int row = (wave + NW * i) * RPI + srow;
    a_vo[i] = (unsigned)(row * p.lda + ((schunk ^ swz(row)) << 3)) * 2u;
  }
#pragma unroll
  for (int i = 0; i < B_PW; ++i) {
    const int row = (wave + NW * i) * RPI + srow;
    w_vo[i] = (unsigned)(row * p.ldw + ((schunk ^ swz(row)) << 3)) * 2u;
  }
  // DMA cursor (uniform): next step to issue = (is_tile, is_kt); clamps at the last step
  int is_tile = slot, is_kt = 0, is_left = T;        // is_left: real steps not yet issued
  char* is_lds = smem + wave * 1024;                 // ring slot write pointer for this wave
  const char* is_a;
  const char* is_w;
  auto set_issue_tile = [&]() {
    const int tm = is_tile / p.tiles_n, tn = is_tile - tm * p.tiles_n;
    is_a = reinterpret_cast<const char*>(p.A + (size_t)tm * BM * p.lda);
    is_w = reinterpret_cast<const char*>(p.W + (size_t)tn * BN * p.ldw);
  };
  set_issue_tile();
  auto issue_piece = [&](int i) {                    // i in [0, LPS): one 1-KiB LDS-DMA
    if (i < A_PW) glds16(is_a + a_vo[i < A_PW ? i : 0], is_lds + NW * i * 1024);
    else glds16(is_w + w_vo[i >= A_PW ? i - A_PW : 0], is_lds + A_BYTES + NW * (i - A_PW) * 1024);
  };
  auto issue_advance = [&]() {
    is_lds = (is_lds + STAGE == smem + wave * 1024 + S * STAGE) ? smem + wave * 1024 : is_lds + STAGE;
    if (--is_left > 0) {
      is_a += BK * 2; is_w += BK * 2;
      if (++is_kt == nk) { is_kt = 0; is_tile += G; set_issue_tile(); }
    }
  };
  auto issue_next = [&]() {
#pragma unroll
    for (int i = 0; i < A_PW; ++i) glds16(is_a + a_vo[i], is_lds + NW * i * 1024);
#pragma unroll
    for (int i = 0; i < B_PW; ++i) glds16(is_w + w_vo[i], is_lds + A_BYTES + NW * i * 1024);
    is_lds = (is_lds + STAGE == smem + wave * 1024 + S * STAGE) ? smem + wave * 1024 : is_lds + STAGE;
    if (--is_left > 0) {                              // past the end: keep re-issuing the last step
      is_a += BK * 2; is_w += BK * 2;
      if (++is_kt == nk) { is_kt = 0; is_tile += G; set_issue_tile(); }
    }
  };

  unsigned fa[KSUB][2], fb[KSUB][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wm * 64 + i * 32 + (lane & 31);
    const int rb = wn * 64 + i * 32 + (lane & 31);
#pragma unroll
    for (int s = 0; s < KSUB; ++s) {
      fa[s][i] = (unsigned)(ra * ROWB + (((2 * s + lh) ^ swz(ra)) << 4));
      fb[s][i] = (unsigned)(A_BYTES + rb * ROWB + (((2 * s + lh) ^ swz(rb)) << 4));
    }
  }

  f16x acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  h8 af[KSUB][2], bf[KSUB][2];
  const char* rd = smem;                             // ring slot read pointer
  auto load_frags = [&]() {
#pragma unroll
    for (int s = 0; s < KSUB; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[s][i] = *(const h8*)(rd + fa[s][i]);
        bf[s][i] = *(const h8*)(rd + fb[s][i]);
      }
    rd = (rd + STAGE == smem + S * STAGE) ? smem : rd + STAGE;
  };
  // 16 MFMAs with the step's LPS LDS-DMA pieces slotted between them (one piece per MFMA pair):
  // the address unit then sees an even stream instead of 8 waves bursting 6 loads each at a
  // phase boundary, and the issuing wave hides the DMA issue latency behind its own MFMAs.
  // `mid` runs after 12 of the 16 MFMAs (all DMA pieces already issued): the counted wait and the
  // phase barrier sit INSIDE the burst, so when the partner group is released there are still 4
  // MFMAs queued on this SIMD and the matrix pipe does not drain at the phase boundary.  Legal
  // because everything the barrier guards (this wave's LDS reads of the stage, its DMA wait) is
  // settled before the burst's tail.
  auto mfma_burst_dma = [&](auto&& mid) {
    __builtin_amdgcn_s_setprio(1);
    int piece = 0;
#pragma unroll
    for (int s = 0; s < KSUB; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (s * 2 + i == (KSUB * 2) - 2) {
          issue_advance();
          mid();
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[s][j], af[s][i], acc[i][j], 0, 0, 0);
        if (piece < LPS) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(piece++);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    __builtin_amdgcn_s_setprio(0);
  };
  auto mfma_burst = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < KSUB; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[s][j], af[s][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  int ep_tile = slot;
  char* const scr = smem + S * STAGE + wave * 2048;      // 2 KiB transpose scratch per wave
  // deferred-epilogue state
  const bool fast_epi = p.out_f16 && !p.out_f32 && !p.resid && !p.add2 && p.out_padded;
  h4 hq[2][2][4];
  half_t* op = nullptr;
  int pend = 0;                                          // passes of the previous tile still to emit
  char* const wp = scr + (lane & 7) * 144 + lh * 8;
  const char* const rp = scr + (lane >> 3) * 144 + (lane & 7) * 16;
  h8 rowv;                                               // pass row in flight between LOAD and COMPUTE
  half_t* rowp = nullptr;
  auto pass_body = [&](auto I, auto Q) {
    constexpr int i = decltype(I)::value, q = decltype(Q)::value;
    if (((lane & 31) >> 3) == q) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<h4a*>(wp + (j * 32 + 8 * g) * 2) = hq[i][j][g];
    }
    asm volatile("" ::: "memory");
    rowv = *reinterpret_cast<const h8a*>(rp);            // consumed by pass_store() one phase later
    asm volatile("" ::: "memory");
    rowp = op + (size_t)((i * 4 + q) * 8) * p.ldc16;
  };
  auto pass_store = [&]() { *reinterpret_cast<h8*>(rowp) = rowv; };   // rows are padded: always issued
  auto do_pass = [&](int idx) {
    using std::integral_constant;
    switch (idx) {
      case 0: pass_body(integral_constant<int, 0>{}, integral_constant<int, 0>{}); break;
      case 1: pass_body(integral_constant<int, 0>{}, integral_constant<int, 1>{}); break;
      case 2: pass_body(integral_constant<int, 0>{}, integral_constant<int, 2>{}); break;
      case 3: pass_body(integral_constant<int, 0>{}, integral_constant<int, 3>{}); break;
      case 4: pass_body(integral_constant<int, 1>{}, integral_constant<int, 0>{}); break;
      case 5: pass_body(integral_constant<int, 1>{}, integral_constant<int, 1>{}); break;
      case 6: pass_body(integral_constant<int, 1>{}, integral_constant<int, 2>{}); break;
      default: pass_body(integral_constant<int, 1>{}, integral_constant<int, 3>{}); break;
    }
  };
  auto flush_passes = [&]() {
    while (pend > 0) { do_pass(8 - pend); --pend; pass_store(); }
  };
  // bias: one float per lane (this wave's 64 columns), fetched a whole tile ahead (vmcnt is
  // in-order: a load issued at tile end would wait for every DMA in front of it) and parked in
  // a 256-byte LDS line per wave; the conversion reads it back as float4 (1 VGPR instead of 32).
  float* const bias_lds = reinterpret_cast<float*>(scr + 1152);      // bytes 1152..1407 of the scratch
  float bias_next = 0.f;
  auto fetch_bias = [&](int tile) {
    const int tn = tile - (tile / p.tiles_n) * p.tiles_n;
    const int n = tn * BN + wn * 64 + lane;
    bias_next = (p.bias && n < p.N) ? p.bias[n] : 0.f;
  };
  auto park_bias = [&]() { reinterpret_cast<floata*>(bias_lds)[lane] = bias_next; asm volatile("" ::: "memory"); };
  if (fast_epi) { fetch_bias(slot); park_bias(); if (slot + G < total_tiles) fetch_bias(slot + G); }
  auto epilogue = [&]() {
    const int tm = ep_tile / p.tiles_n, tn = ep_tile - tm * p.tiles_n;
    ep_tile += G;
    const int m0 = tm * BM + wm * 64, n0 = tn * BN + wn * 64;
    const bool interior = (m0 + 64 <= p.M) && (n0 + 64 <= p.N);
    if (fast_epi && n0 + 64 <= p.N) {
      // ---- f16 result, DEFERRED: convert the tile to packed f16 now (bias, q-scale, ReLU folded
      // in, branch-free) and free the accumulators; the 8 LDS-transpose passes that store it as
      // whole 128-byte lines are emitted one per following LOAD phase (do_pass), i.e. underneath
      // the partner group's MFMA burst.
      flush_passes();
      const float sc = (n0 < p.scale_cols) ? p.scale : 1.f;     // scale_cols is a multiple of 64
      const float lo = p.relu ? 0.f : -INFINITY;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4a*>(bias_lds + 4 * lh + j * 32 + 8 * g);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const float v0 = fmaxf((acc[i][j][4 * g + 0] + b4.x) * sc, lo), v1 = fmaxf((acc[i][j][4 * g + 1] + b4.y) * sc, lo);
            const float v2 = fmaxf((acc[i][j][4 * g + 2] + b4.z) * sc, lo), v3 = fmaxf((acc[i][j][4 * g + 3] + b4.w) * sc, lo);
            hq[i][j][g] = h4{(half_t)v0, (half_t)v1, (half_t)v2, (half_t)v3};
            acc[i][j][4 * g + 0] = 0.f; acc[i][j][4 * g + 1] = 0.f; acc[i][j][4 * g + 2] = 0.f; acc[i][j][4 * g + 3] = 0.f;
          }
        }
      op = p.out_f16 + (size_t)(m0 + (lane >> 3)) * p.ldc16 + n0 + (lane & 7) * 8;
      pend = 8;
      if (!DEFER) flush_passes();
      park_bias();                                         // bias of the next tile (fetched a tile ago)
      if (ep_tile + G < total_tiles) fetch_bias(ep_tile + G);
      return;
    }
    flush_passes();
    const int nb = n0 + 4 * lh;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + i * 32 + (lane & 31);
      const float* r_ptr = p.resid ? p.resid + (size_t)m * p.ldr + nb : nullptr;
      const float* a_ptr = p.add2 ? p.add2 + (size_t)m * p.ld2 + nb : nullptr;
      float* o32 = p.out_f32 ? p.out_f32 + (size_t)m * p.ldc32 + nb : nullptr;
      half_t* o16 = p.out_f16 ? p.out_f16 + (size_t)m * p.ldc16 + nb : nullptr;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dn = j * 32 + 8 * g;
          const int n = nb + dn;
          float v[4] = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          acc[i][j][4 * g + 0] = 0.f; acc[i][j][4 * g + 1] = 0.f; acc[i][j][4 * g + 2] = 0.f; acc[i][j][4 * g + 3] = 0.f;
          if (!interior && (m >= p.M || n >= p.N)) continue;
          if (interior || n + 3 < p.N) {
            if (p.bias) {
              const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
              v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
            }
            if (n < p.scale_cols) { v[0] *= p.scale; v[1] *= p.scale; v[2] *= p.scale; v[3] *= p.scale; }
            if (a_ptr) {
              const float4 a4 = *reinterpret_cast<const float4*>(a_ptr + dn);
              v[0] += a4.x; v[1] += a4.y; v[2] += a4.z; v[3] += a4.w;
            }
            if (r_ptr) {
              const float4 r4 = *reinterpret_cast<const float4*>(r_ptr + dn);
              v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            }
            if (p.relu) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            if (o32) *reinterpret_cast<float4*>(o32 + dn) = make_float4(v[0], v[1], v[2], v[3]);
            if (o16) {
              h4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
              *reinterpret_cast<h4*>(o16 + dn) = hv;
            }
          } else {
            for (int e = 0; e < 4 && n + e < p.N; ++e) {
              float x = v[e] + (p.bias ? p.bias[n + e] : 0.f);
              if (n + e < p.scale_cols) x *= p.scale;
              if (a_ptr) x += a_ptr[dn + e];
              if (r_ptr) x += r_ptr[dn + e];
              if (p.relu) x = x > 0.f ? x : 0.f;
              if (o32) o32[dn + e] = x;
              if (o16) o16[dn + e] = (half_t)x;
            }
          }
        }
      }
    }
  };

  // timeline trace: block 0, waves 0 (group A) and 4 (group B), lane 0
  const bool tr = p.ts != nullptr && bid == 0 && (wave & 3) == 0 && lane == 0;
  long long* trp = p.ts + (size_t)grp * 64 * 8;
  auto stamp = [&](int k, int slot_) {
    if (tr && k < 64) trp[k * 8 + slot_] = (long long)__builtin_readcyclecounter();
  };
  int kt = nk;
  if (grp == 0) {
    // ---------------- group A: LOAD in even phases, COMPUTE in odd phases
#pragma unroll
    for (int s = 0; s < S - 1; ++s) issue_next();
    wait_vmcnt<LPS * (S - 2)>();
    __builtin_amdgcn_s_barrier();
    int sA_prev = 0;
    for (int k = 0; k < T; ++k) {
      stamp(k, 0);
      if (!INTERLEAVE) issue_next();                  // step k+S-1 (clamped)
      const int sA = pend > 0 ? 1 : 0;                // one deferred pass this step
      if (sA) { do_pass(8 - pend); --pend; }           // LDS writes + row read issued; stored after the MFMAs
      load_frags();
      stamp(k, 1);
      __builtin_amdgcn_s_barrier();
      stamp(k, 2);
      auto midA = [&]() {
        if (sA) pass_store();
        // younger than DMA(k+1) [issued in COMPUTE(k-1)]: store(k-1), DMA(k+2), store(k)
        const int ns = sA + sA_prev;
        if (ns == 2) wait_vmcnt<LPS * (S - 2) + 2>();
        else if (ns == 1) wait_vmcnt<LPS * (S - 2) + 1>();
        else wait_vmcnt<LPS * (S - 2)>();
        sA_prev = sA;
        __builtin_amdgcn_s_barrier();
      };
      if (INTERLEAVE) mfma_burst_dma(midA); else { mfma_burst(); midA(); }
      stamp(k, 5);
      if (--kt == 0) { kt = nk; epilogue(); }
      stamp(k, 6);
    }
  } else {
    // ---------------- group B: one phase behind A
#pragma unroll
    for (int s = 0; s < S; ++s) issue_next();
    wait_vmcnt<LPS * (S - 1)>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    int sB_prev = 0, sB_prev2 = 0;
    for (int k = 0; k < T; ++k) {
      stamp(k, 0);
      const int sB = pend > 0 ? 1 : 0;
      if (sB) { do_pass(8 - pend); --pend; }
      load_frags();
      stamp(k, 1);
      // younger than DMA(k+1) [issued in COMPUTE(k-2)]: store(k-2), DMA(k+2), store(k-1)
      {
        const int ns = sB_prev + sB_prev2;
        if (ns == 2) wait_vmcnt<LPS * (S - 2) + 2>();
        else if (ns == 1) wait_vmcnt<LPS * (S - 2) + 1>();
        else wait_vmcnt<LPS * (S - 2)>();
        sB_prev2 = sB_prev; sB_prev = sB;
      }
      stamp(k, 2);
      __builtin_amdgcn_s_barrier();
      stamp(k, 3);
      auto midB = [&]() {
        if (sB) pass_store();
        if (k + 1 < T) __builtin_amdgcn_s_barrier();
      };
      if (INTERLEAVE) mfma_burst_dma(midB); else { issue_next(); mfma_burst(); midB(); }   // DMA of step k+S (clamped)
      stamp(k, 5);
      if (--kt == 0) { kt = nk; epilogue(); }
      stamp(k, 6);
    }
  }
  flush_passes();
  wait_vmcnt<0>();                                    // clamped tail DMA must land before LDS is released
}

template <int BK, int S, bool INTERLEAVE = false, bool DEFER = false>
static void launch_pp2(hipStream_t s, GemmDev d) {
  constexpr int LDS = S * (256 + 128) * BK * 2 + 8 * 2048;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  d.tiles_m = cdiv(d.M, 256);
  d.tiles_n = cdiv(d.N, 128);
  const int total = d.tiles_m * d.tiles_n;
  if (total == 0) return;
  static int cus[64] = {0};
  static bool attr[64] = {false};
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  dev &= 63;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    PF_HIP(hipGetDeviceProperties(&prop, dev));
    cus[dev] = prop.multiProcessorCount;
  }
  auto kfn = gemm_f16_pp2<BK, S, INTERLEAVE, DEFER>;
  if (!attr[dev]) {
    PF_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr[dev] = true;
  }
  int grid = cus[dev];
  if (grid > total) grid = total;
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), LDS, s, d);
  PF_HIP(hipGetLastError());
}

template <int BK, int S, int ROT = 0>
static void launch_pingpong(hipStream_t s, GemmDev d) {
  constexpr int LDS = S * (256 + 128) * BK * 2;
  d.tiles_m = cdiv(d.M, 256);
  d.tiles_n = cdiv(d.N, 128);
  const int total = d.tiles_m * d.tiles_n;
  if (total == 0) return;
  static int cus[64] = {0};
  static bool attr[64] = {false};
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  dev &= 63;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    PF_HIP(hipGetDeviceProperties(&prop, dev));
    cus[dev] = prop.multiProcessorCount;
  }
  auto kfn = gemm_f16_pingpong<BK, S, ROT>;
  if (!attr[dev]) {
    PF_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr[dev] = true;
  }
  int grid = cus[dev];
  if (grid > total) grid = total;
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), LDS, s, d);
  PF_HIP(hipGetLastError());
}

template <int WM, int WN, int BK, int S, bool FRAGS_FIRST = true, int ABLATE = 0>
static void launch_variant(hipStream_t s, GemmDev d, int blocks_per_cu) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int LDS = S * (BM + BN) * BK * 2;
  d.tiles_m = cdiv(d.M, BM);
  d.tiles_n = cdiv(d.N, BN);
  const int total = d.tiles_m * d.tiles_n;
  if (total == 0) return;
  static int cus[64] = {0};
  static bool attr[64] = {false};
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  dev &= 63;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    PF_HIP(hipGetDeviceProperties(&prop, dev));
    cus[dev] = prop.multiProcessorCount;
  }
  auto kfn = gemm_f16_persistent<WM, WN, BK, S, FRAGS_FIRST, ABLATE>;
  if (!attr[dev]) {
    PF_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr[dev] = true;
  }
  int grid = cus[dev] * blocks_per_cu;
  if (grid > total) grid = total;
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(WM * WN * 64), LDS, s, d);
  PF_HIP(hipGetLastError());
}

void launch_gemm(hipStream_t s, const GemmArgs& a) {
  PF_CHECK(a.K % 64 == 0 && a.K > 0, PF_ERR_INVALID_ARG, "gemm: K must be a multiple of 64");
  PF_CHECK(a.lda % 8 == 0 && a.ldw % 8 == 0, PF_ERR_INVALID_ARG, "gemm: lda/ldw must be multiples of 8");
  PF_CHECK((!a.out_f32 || a.ldc32 % 4 == 0) && (!a.out_f16 || a.ldc16 % 4 == 0) && (!a.resid || a.ldr % 4 == 0) &&
               (!a.add2 || a.ld2 % 4 == 0) && a.scale_cols % 64 == 0,
           PF_ERR_INVALID_ARG, "gemm: output leading dimensions must be multiples of 4");
  GemmDev d;
  d.A = a.A; d.W = a.W; d.bias = a.bias;
  d.out_f32 = a.out_f32; d.out_f16 = a.out_f16; d.resid = a.resid; d.add2 = a.add2;
  d.lda = a.lda; d.ldw = a.ldw; d.ldc32 = a.ldc32; d.ldc16 = a.ldc16; d.ldr = a.ldr; d.ld2 = a.ld2;
  d.M = a.M; d.N = a.N; d.K = a.K;
  d.relu = a.relu; d.scale_cols = a.scale_cols; d.scale = a.scale_cols > 0 ? a.scale : 1.f;
  d.out_padded = a.out_padded;
  d.tiles_m = d.tiles_n = 0;
  static int dbg = -1;
  if (dbg < 0) { const char* e = std::getenv("PF_GEMM_DBG"); dbg = e ? std::atoi(e) : 0; }
  d.dbg = dbg;
  static long long* ts_dev = nullptr;
  static int ts_on = -1;
  if (ts_on < 0) { const char* e = std::getenv("PF_GEMM_TS"); ts_on = e ? std::atoi(e) : 0; }
  if (ts_on && !ts_dev) PF_HIP(hipMalloc(&ts_dev, 2 * 64 * 8 * 8));
  d.ts = ts_on ? ts_dev : nullptr;
  if (ts_on) PF_HIP(hipMemsetAsync(ts_dev, 0, 2 * 64 * 8 * 8, s));
  static int variant = -1;
  if (variant < 0) {
    const char* e = std::getenv("PF_GEMM_VARIANT");
    variant = e ? std::atoi(e) : 15;
  }
  switch (variant) {
    case 0: launch_variant<2, 2, 64, 2>(s, d, 2); break;   // 128x128, BK64, 2-stage, 2 blocks/CU
    case 1: launch_variant<2, 2, 32, 4>(s, d, 2); break;   // 128x128, BK32, 4-stage, 2 blocks/CU
    case 2: launch_variant<4, 2, 32, 4>(s, d, 1); break;   // 256x128, BK32, 4-stage (96 KB)
    case 3: launch_variant<4, 2, 32, 6>(s, d, 1); break;   // 256x128, BK32, 6-stage (144 KB)
    case 4: launch_variant<4, 2, 64, 3>(s, d, 1); break;   // 256x128, BK64, 3-stage (144 KB)
    case 5: launch_variant<4, 2, 64, 3, false>(s, d, 1); break;   // as 4, compiler-scheduled frag reads
    case 6: launch_variant<2, 2, 64, 2, false>(s, d, 2); break;   // as 0, compiler-scheduled frag reads
    case 11: launch_pingpong<64, 3>(s, d); break;                   // ping-pong 256x128 BK64 S3
    case 12: launch_pingpong<64, 3, 1>(s, d); break;                // + K rotation by slot
    case 13: launch_pingpong<64, 3, 2>(s, d); break;                // + K rotation by slot/8
    case 14: launch_pp2<64, 3>(s, d); break;                        // ping-pong v2 (branch-free steady state)
    case 15: launch_pp2<64, 3, true, true>(s, d); break;            // + DMA interleaved, deferred epilogue passes
    case 16: launch_pp2<64, 3, true, false>(s, d); break;           // + DMA interleaved, epilogue at tile end
    case 7: launch_variant<4, 2, 64, 3, true, 1>(s, d, 1); break;   // ablation: no MFMA
    case 8: launch_variant<4, 2, 64, 3, true, 2>(s, d, 1); break;   // ablation: no loads after the prologue
    case 9: launch_variant<4, 2, 64, 3, true, 3>(s, d, 1); break;   // ablation: no epilogue
    case 10: launch_variant<4, 2, 64, 3, true, 4>(s, d, 1); break;  // ablation: no drain after the epilogue (unsafe)
    default: throw Error(PF_ERR_INVALID_ARG, "unknown PF_GEMM_VARIANT");
  }
  if (ts_on) {
    std::vector<long long> h(2 * 64 * 8);
    PF_HIP(hipStreamSynchronize(s));
    PF_HIP(hipMemcpy(h.data(), ts_dev, h.size() * 8, hipMemcpyDeviceToHost));
    FILE* f = std::fopen("gpurun_out/gemm_ts.txt", "w");
    if (f) {
      for (int g = 0; g < 2; ++g)
        for (int k = 0; k < 64; ++k) {
          std::fprintf(f, "g%d k%02d", g, k);
          for (int j = 0; j < 7; ++j) std::fprintf(f, " %lld", h[(g * 64 + k) * 8 + j] ? h[(g * 64 + k) * 8 + j] - h[0] : -1);
          std::fprintf(f, "\n");
        }
      std::fclose(f);
    }
  }
}

}  // namespace pf
