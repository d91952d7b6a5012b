// k_gemm_qkv.hip — persistent 256 x 192 tile GEMM for the encoder's fused Q | K | V projection:
//
//   [Q | K | V][M,1536] = A[M,K] * Wqkv[1536,K]^T + bias ;  Q *= 1/sqrt(d_head)
//
// (linear_q_k_v of the SAN-M layer: one MatMul + Add of the graph InferenceSession.Run executes,
// AliParaformerAsr/OfflineProjOfParaformer.cs:68; the q scaling is the Mul in front of the attention MatMul.)
//
// Why not gemm_f16_pp3 (256 x 128 tiles, row-major result): its K loop stages 48 KB per 4.2 MFLOP; the 256 x 256
// pipeline of k_gemm_big.hip stages 32 KB per 4.2 MFLOP and hands its D^T fragments to memory without any LDS
// exchange — but only into the BLOCKED activation layout (kernels.h), and 1536 columns are 6 tiles of 256: 378 tiles
// on 256 CUs = two rounds for 1.48 rounds of work.  Here: 192-column tiles (8 per row block: 504 tiles = 1.97 rounds,
// 28 KB per 3.1 MFLOP), and every tile carries 128 columns of Q | K and 64 columns of V, so that
//   * Q and K leave in the blocked layout ([M, 1024] as 32-row x 8-column blocks of 512 contiguous bytes): whole-line
//     stores straight from the accumulators; the attention kernel reads its Q fragments as whole blocks and stages K
//     tiles as 1 KiB contiguous LDS-DMA pieces whose LDS image is conflict-free without a swizzle (k_attn.hip);
//   * V leaves row-major ([M, 512]): the row-complete out-projection reads its 11-row FSMN window row-wise
//     (k_gemm_rc.hip) and the attention kernel transposes V tiles with ds_read_b64_tr_b16 as before.  A wave's V
//     block is 32 rows x 32 columns; lanes l and l + 32 exchange register pairs (v_permlane32_swap) so that each
//     stores 16 contiguous bytes — 32 rows x 32 B per instruction, plain (write-back) stores: L2 merges the lines.
// The weight rows are permuted once at load (qkv_tile_row) so that a tile's 192 W rows are contiguous:
//   tile t, wave column wn, block j, c:  j < 2 -> Q|K column 128 t + 64 wn + 32 j + c ;  j = 2 -> V column 64 t + 32 wn + c.
// Pipeline: gemm_bigp_kernel's (k-steps of 32, four 28 KB stages, three in flight, one mid-step barrier per k-step,
// branch-free body with compile-time vmcnt immediates, fragment reads and waits in inline asm, accumulators start as
// the tile's bias).  Same accumulation order as gemm_f16_pp3 (k ascending in 16-wide MFMA steps from the bias), so
// the results are bit-identical to the row-major kernel's.
#include "kernels.h"

#include <algorithm>
#include <mutex>

namespace pf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));

struct QkvDev {
  const half_t* A; const half_t* W; const float* bias;       // W, bias: tile-permuted (qkv_tile_row)
  half_t* out_qk; half_t* out_v;                              // blocked [Mpad, 1024] ; row-major [Mpad, ldv]
  int lda, ldw, ldv;
  int M, K, tiles_m;
  float scale;                                                // applied to the Q columns (< 512)
};

constexpr int QV_BM = 256, QV_BK = 32, QV_ROWB = QV_BK * 2, QV_S = 4, QV_NJ = 3, QV_BN = 64 * QV_NJ, QV_TN = 8;
constexpr int QV_A_BYTES = QV_BM * QV_ROWB;                   // 16 KiB
constexpr int QV_STAGE = (QV_BM + QV_BN) * QV_ROWB;           // 28 KiB
constexpr int QV_RING = QV_S * QV_STAGE;                      // 112 KiB
constexpr int QV_LDS = QV_RING + 2 * QV_BN * 4;               // + two bias lines
constexpr int QV_STORES = 2 * 2 * 4 + 2 * 2;                  // result stores per wave and tile end: 16 blocked (8 B) + 4 V (16 B)

__host__ __device__ inline int qkv_tile_row(int r) {          // permuted row r (0..1535) -> row of the [Q | K | V] weight
  const int t = r / QV_BN, q = r - t * QV_BN, wn = q / 96, qq = q - wn * 96, j = qq >> 5, c = qq & 31;
  return j < 2 ? 128 * t + 64 * wn + 32 * j + c : 1024 + 64 * t + 32 * wn + c;
}

__global__ void qkv_permute_kernel(const half_t* __restrict__ w, int ldw, const float* __restrict__ b, half_t* __restrict__ wp,
                                   float* __restrict__ bp, int cols) {
  const int r = blockIdx.x, s = qkv_tile_row(r);
  for (int c = threadIdx.x; c < cols; c += blockDim.x) wp[(size_t)r * ldw + c] = w[(size_t)s * ldw + c];
  if (threadIdx.x == 0) bp[r] = b ? b[s] : 0.f;
}

void launch_qkv_permute(hipStream_t s, const half_t* w, int ldw, const float* bias, half_t* wp, float* bp) {
  hipLaunchKernelGGL(qkv_permute_kernel, dim3(1536), dim3(128), 0, s, w, ldw, bias, wp, bp, ldw);
  PF_HIP(hipGetLastError());
}

__device__ __forceinline__ void qv_glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ void qv_glds4(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}
template <int N>
__device__ __forceinline__ void qv_wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
// Q | K blocks: write-through and dropped from the XCD's L2 (they are read by the attention kernel, after this
// launch; the L2 is needed for the A panels and W tiles — k_gemm_big.hip measured the same)
__device__ __forceinline__ void qv_store8_sc1(void* p, h4 v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// V pieces: 16 bytes of a 64-byte row segment; write-back, so that the L2 merges the four pieces of a line
__device__ __forceinline__ void qv_store16(void* p, h8 v) {
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

__global__ __launch_bounds__(512, 1) void gemm_qkvp_kernel(QkvDev p) {
  constexpr int NJ = QV_NJ, BN = QV_BN, A_BYTES = QV_A_BYTES, STAGE = QV_STAGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, lh = lane >> 5;
  auto swz = [](int row) __attribute__((always_inline)) -> int { return (row >> 2) & 3; };

  // ---- tile schedule: persistent block b (on XCD b % 8) takes tiles slot, slot + G, ...; XCDs get contiguous runs, n fastest
  const int G = gridDim.x, bid = blockIdx.x;
  const int total_tiles = p.tiles_m * QV_TN;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int slot = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int n_my = slot < total_tiles ? (total_tiles - slot + G - 1) / G : 0;
  const int nk = p.K / QV_BK;
  const int T = n_my * nk;
  if (T == 0) return;
  float* const bias_line = reinterpret_cast<float*>(smem + QV_RING);

  // ---- DMA cursor (uniform): stage is_t = k-step is_k of tile is_tile.  Per wave and step: one bias-line piece, two A
  // pieces, two W pieces (the W tile is 12 pieces: waves 4-7 repeat their first one — every wave issues the same count,
  // so every vmcnt immediate is a compile-time constant)
  const int srow = lane >> 2, schunk = lane & 3;
  unsigned a_vo[2], w_vo[2];
  int w_dst[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave + 8 * i) * 16 + srow;
    a_vo[i] = (unsigned)(row * p.lda + ((schunk ^ swz(row)) << 3)) * 2u;
    const int piece = (i == 1 && wave >= 4) ? wave : wave + 8 * i;
    const int wrow = piece * 16 + srow;
    w_vo[i] = (unsigned)(wrow * p.ldw + ((schunk ^ swz(wrow)) << 3)) * 2u;
    w_dst[i] = A_BYTES + piece * 1024;
  }
  int is_tile = slot, is_k = 0, is_t = 0, is_slot = 0, is_round = 0;
  const char* is_a;
  const char* is_w;
  int is_n0 = 0;
  auto set_issue_tile = [&]() __attribute__((always_inline)) {
    const int tm = is_tile / QV_TN, tn = is_tile - tm * QV_TN;
    is_a = reinterpret_cast<const char*>(p.A + (size_t)tm * QV_BM * p.lda);
    is_w = reinterpret_cast<const char*>(p.W + (size_t)tn * BN * p.ldw);
    is_n0 = tn * BN;
  };
  set_issue_tile();
  auto issue_piece = [&](int q) __attribute__((always_inline)) {
    char* st = smem + (is_slot & (QV_S - 1)) * STAGE;
    if (q < 2) qv_glds16(is_a + a_vo[q & 1], st + (wave + 8 * (q & 1)) * 1024);
    else qv_glds16(is_w + w_vo[q & 1], st + w_dst[q & 1]);
  };
  const int bw = wave % 3;
  auto issue_bias = [&]() __attribute__((always_inline)) {
    qv_glds4(p.bias + is_n0 + bw * 64 + lane, bias_line + (is_round & 1) * BN + bw * 64);
  };
  auto issue_advance = [&]() __attribute__((always_inline)) {
    if (is_t + 1 < T) {
      ++is_t;
      is_a += QV_BK * 2; is_w += QV_BK * 2;
      if (++is_k == nk) { is_k = 0; is_tile += G; ++is_round; set_issue_tile(); }
    }
    ++is_slot;
  };
  auto wait_landed = [&](bool burst) __attribute__((always_inline)) {
    if (burst) qv_wait_vmcnt<10 + QV_STORES>(); else qv_wait_vmcnt<10>();
  };

  unsigned fa[2][2], fb[2][NJ];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra = wm * 64 + i * 32 + (lane & 31);
      fa[s][i] = (unsigned)(ra * QV_ROWB + (((2 * s + lh) ^ swz(ra)) << 4));
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int rb = wn * (32 * NJ) + j * 32 + (lane & 31);
      fb[s][j] = (unsigned)(A_BYTES + rb * QV_ROWB + (((2 * s + lh) ^ swz(rb)) << 4));
    }
  }
  f16x acc[2][NJ];
  h8 a0[2] = {}, b0[NJ] = {}, a1[2] = {}, b1[NJ] = {};
  auto load = [&](unsigned rd, int s, h8 (&af)[2], h8 (&bf)[NJ]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(bf[j]) : "v"(rd + fb[s][j]) : "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(af[i]) : "v"(rd + fa[s][i]) : "memory");
  };
  // five reads per half-step: the second half's may stay outstanding while the first half is consumed
  auto frag_wait5 = [&](h8 (&af)[2], h8 (&bf)[NJ]) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(af[0]), "+v"(af[1]), "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]));
  };
  auto frag_wait0 = [&](h8 (&af)[2], h8 (&bf)[NJ]) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]));
  };
  auto mma = [&](h8 (&af)[2], h8 (&bf)[NJ], bool dma) __attribute__((always_inline)) {
    int q = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        if (q < 4) {
          __builtin_amdgcn_sched_barrier(0);
          if (dma) issue_piece(q);
          ++q;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
  };

  // ---- prologue: three stages in flight
#pragma unroll
  for (int st = 0; st < QV_S - 1; ++st) {
    issue_bias();
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_piece(q);
    issue_advance();
  }
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
  wait_landed(false);
  __builtin_amdgcn_s_barrier();
  load(lds0, 0, a0, b0);

  int tile = slot, k = 0, round = 0, since_burst = 3;
  typedef float f4v __attribute__((ext_vector_type(4)));
  auto acc_init = [&](int rnd) __attribute__((always_inline)) {
    const unsigned bl = lds0 + QV_RING + ((rnd & 1) * BN + wn * (32 * NJ) + 4 * lh) * 4;
    f4v b4[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1" : "=v"(b4[j][g]) : "v"(bl + (j * 32 + 8 * g) * 4) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(b4[0][0]), "+v"(b4[0][1]), "+v"(b4[0][2]), "+v"(b4[0][3]), "+v"(b4[1][0]), "+v"(b4[1][1]), "+v"(b4[1][2]), "+v"(b4[1][3]),
                   "+v"(b4[2][0]), "+v"(b4[2][1]), "+v"(b4[2][2]), "+v"(b4[2][3]));
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          acc[i][j][4 * g + 0] = b4[j][g][0]; acc[i][j][4 * g + 1] = b4[j][g][1]; acc[i][j][4 * g + 2] = b4[j][g][2]; acc[i][j][4 * g + 3] = b4[j][g][3];
        }
  };
  typedef float f2v __attribute__((ext_vector_type(2)));
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  auto tile_end = [&]() __attribute__((always_inline)) {
    const int tm = tile / QV_TN, tn = tile - tm * QV_TN;
    const int m0 = tm * QV_BM;
    // ---- Q | K: blocks of the blocked [M, 1024] matrix, 8-byte stores straight from the accumulators
    const int qk0 = 128 * tn + 64 * wn;                        // first Q|K column of this wave (multiple of 64: Q or K, never both)
    const bool scaled = qk0 < 512;
    char* ob = reinterpret_cast<char*>(p.out_qk) + ((size_t)((m0 >> 5) + wm * 2) * 128 + (size_t)(qk0 >> 3)) * 512 + (lane & 31) * 16 + lh * 8;
    constexpr size_t rb_stride = (size_t)128 * 512;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f2v lo2 = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1]}, hi2 = {acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          if (scaled) { lo2 *= p.scale; hi2 *= p.scale; }
          const h2v l = __builtin_convertvector(lo2, h2v), h = __builtin_convertvector(hi2, h2v);
          const h4 hv = {l[0], l[1], h[0], h[1]};
          qv_store8_sc1(ob + i * rb_stride + (size_t)(j * 4 + g) * 512, hv);
        }
    // ---- V: row-major [M, ldv]; lanes l / l + 32 hold columns 8g + 0..3 / 8g + 4..7 of row l: after the swap lane l
    // holds the 8 columns of group g, lane l + 32 those of group g + 1
    {
      const int vc0 = 64 * tn + 32 * wn;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        half_t* vrow = p.out_v + (size_t)(m0 + wm * 64 + i * 32 + (lane & 31)) * p.ldv + vc0;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          unsigned x[2], y[2];                                 // x: group 2gp (this lane's 4 columns), y: group 2gp + 1
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const f2v xa = {acc[i][2][8 * gp + 2 * e + 0], acc[i][2][8 * gp + 2 * e + 1]};
            const f2v ya = {acc[i][2][8 * gp + 4 + 2 * e + 0], acc[i][2][8 * gp + 4 + 2 * e + 1]};
            x[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(xa, h2v));
            y[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(ya, h2v));
          }
          // v_permlane32_swap a, b: a[32..63] <-> b[0..31].  With a = x (group 2gp), b = y (group 2gp + 1):
          //   x = {low lanes: cols +0..3 of 2gp, high lanes: cols +0..3 of 2gp+1}, y = {low: cols +4..7 of 2gp, high: cols +4..7 of 2gp+1}
#pragma unroll
          for (int e = 0; e < 2; ++e) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[e]), "+v"(y[e]));
          const h4 lo_ = __builtin_bit_cast(h4, (unsigned long long)x[0] | ((unsigned long long)x[1] << 32));
          const h4 hi_ = __builtin_bit_cast(h4, (unsigned long long)y[0] | ((unsigned long long)y[1] << 32));
          const h8 hv = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);
          qv_store16(vrow + 16 * gp + 8 * lh, hv);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    k = 0; tile += G; ++round; since_burst = 0;
    acc_init(round);
  };

  acc_init(0);
  for (int t = 0; t < T; ++t) {
    const unsigned rd = lds0 + (t & (QV_S - 1)) * STAGE;
    __builtin_amdgcn_sched_barrier(0);
    load(rd, 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    issue_bias();
    frag_wait5(a0, b0);
    __builtin_amdgcn_s_setprio(1);
    mma(a0, b0, true);
    __builtin_amdgcn_s_setprio(0);
    issue_advance();
    __builtin_amdgcn_sched_barrier(0);
    wait_landed(since_burst < 2);
    frag_wait0(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    load(lds0 + ((t + 1) & (QV_S - 1)) * STAGE, 0, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma(a1, b1, false);
    __builtin_amdgcn_s_setprio(0);
    ++since_burst;
    if (++k == nk) tile_end();
  }
  qv_wait_vmcnt<0>();
}

bool gemm_qkvp_applicable(int M, int K, int lda, int ldw, int ldv) {
  return M > 0 && K >= 128 && K % 64 == 0 && lda % 8 == 0 && ldw % 8 == 0 && ldv % 8 == 0 && ldv >= 512;
}

void launch_gemm_qkvp(hipStream_t s, const half_t* A, int lda, const half_t* Wp, int ldw, const float* bias_p, int M, int K,
                      float qscale, half_t* out_qk, half_t* out_v, int ldv) {
  PF_CHECK(gemm_qkvp_applicable(M, K, lda, ldw, ldv), PF_ERR_INVALID_ARG, "gemm_qkvp: shape not covered");
  QkvDev d{};
  d.A = A; d.W = Wp; d.bias = bias_p; d.out_qk = out_qk; d.out_v = out_v;
  d.lda = lda; d.ldw = ldw; d.ldv = ldv; d.M = M; d.K = K; d.tiles_m = cdiv(M, QV_BM); d.scale = qscale;
  static std::mutex init_mu;
  static bool attr_set[64] = {false};
  static int cus[64] = {0};
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(init_mu);
    if (!attr_set[dev & 63]) {
      PF_HIP(hipFuncSetAttribute((const void*)gemm_qkvp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, QV_LDS));
      hipDeviceProp_t prop;
      PF_HIP(hipGetDeviceProperties(&prop, dev));
      cus[dev & 63] = cu_limit(prop.multiProcessorCount);
      attr_set[dev & 63] = true;
    }
  }
  const int total = d.tiles_m * QV_TN;
  note_gemm_kernel("gemm_qkvp_kernel");
  hipLaunchKernelGGL(gemm_qkvp_kernel, dim3((unsigned)std::min(total, cus[dev & 63])), dim3(512), QV_LDS, s, d);
  PF_HIP(hipGetLastError());
}

}  // namespace pf
