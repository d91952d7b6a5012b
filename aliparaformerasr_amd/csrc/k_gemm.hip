// k_gemm.hip — f16 x f16 -> f32 GEMM on the gfx950 matrix cores.
//
// Replaces the MatMul/Gemm nodes ONNX Runtime executes inside InferenceSession.Run
// (reference call site AliParaformerAsr/OfflineProjOfParaformer.cs:68): QKV, attention
// output, FFN, CIF conv (as im2col GEMM), decoder projections and the vocabulary layer.
//
//   C[M,N] = A[M,K] * W[N,K]^T   (both operands K-contiguous, "B^T input")
//
// Tile 128x128x64, 256 threads = 4 wavefronts (2x2), each wavefront owns a 64x64 block as
// 2x2 v_mfma_f32_32x32x16_f16 accumulators.  Operand tiles are brought HBM -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip), double buffered.  The LDS image of a
// global_load_lds is lane-linear, so the bank-conflict swizzle (16-byte chunk index XOR
// ((row>>1)&7) inside a 128-byte tile row) is applied to the SOURCE address and to the
// ds_read_b128 address (cdna guide §5.4 rule 21).  Workgroup ids are remapped so that the
// 8 XCDs each walk a contiguous range of tiles (n fastest) and share A panels in their L2.
#include "kernels.h"

namespace pf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16x __attribute__((ext_vector_type(16)));

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 64
#define GEMM_STAGE_BYTES (2 * 128 * 64 * 2)   // A tile + W tile = 32 KiB

struct GemmDev {
  const half_t* A; const half_t* W; const float* bias;
  float* out_f32; half_t* out_f16; const float* resid; const float* add2;
  int lda, ldw, ldc32, ldc16, ldr, ld2;
  int M, N, K, tiles_n;
  int relu, scale_cols; float scale;
};

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__global__ __launch_bounds__(256, 2) void gemm_f16_kernel(GemmDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware bijective remap of the workgroup id (block b runs on XCD b % 8).
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
  const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

  // ---- staging addresses: wave-instruction i of this wave covers tile rows (wave*4+i)*8 .. +7
  const int srow = lane >> 3;                 // row inside the 8-row slab
  const int schunk = lane & 7;                // 16-byte chunk inside the 128-byte LDS row
  const half_t* ga[4]; const half_t* gw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + srow;
    const int c = schunk ^ ((row >> 1) & 7);  // source chunk that must land in LDS chunk `schunk`
    ga[i] = p.A + (size_t)(m0 + row) * p.lda + c * 8;
    gw[i] = p.W + (size_t)(n0 + row) * p.ldw + c * 8;
  }

  // ---- fragment read offsets (bytes inside a tile): row*128 + ((2s + lane>>5) ^ swz)*16
  int a_off[2], b_off[2], a_swz[2], b_swz[2];
  const int lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wm * 64 + i * 32 + (lane & 31);
    const int rb = wn * 64 + i * 32 + (lane & 31);
    a_off[i] = ra * 128; a_swz[i] = (ra >> 1) & 7;
    b_off[i] = rb * 128; b_swz[i] = (rb >> 1) & 7;
  }

  f16x acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = p.K / GEMM_BK;

  auto stage = [&](int buf, int kt) {
    char* ab = smem + buf * GEMM_STAGE_BYTES;
    char* wb = ab + 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(ga[i] + kt * GEMM_BK, ab + (wave * 4 + i) * 1024);
      glds16(gw[i] + kt * GEMM_BK, wb + (wave * 4 + i) * 1024);
    }
  };

  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                           // drains vmcnt (LDS-DMA) and orders the buffers
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* ab = smem + (kt & 1) * GEMM_STAGE_BYTES;
    const char* wb = ab + 16384;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      h8 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *(const h8*)(ab + a_off[i] + (((2 * s + lh) ^ a_swz[i]) << 4));
        bf[i] = *(const h8*)(wb + b_off[i] + (((2 * s + lh) ^ b_swz[i]) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue.  The MFMA is issued as D^T = W_tile * X_tile^T, so the 32x32 C/D layout gives
  // every lane ONE output row m (col = lane&31) and, per register quad g, FOUR consecutive
  // output columns n = 8g + 4*(lane>>5) + 0..3: bias / residual / FSMN reads and the result
  // stores are 16-byte (fp32) or 8-byte (f16) vectors straight from registers.
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + (lane & 31);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + j * 32 + 8 * g + 4 * lh;
        if (n >= p.N) continue;
        float v[4] = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
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
          if (p.out_f32) *reinterpret_cast<float4*>(p.out_f32 + (size_t)m * p.ldc32 + n) = make_float4(v[0], v[1], v[2], v[3]);
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
}

void launch_gemm(hipStream_t s, const GemmArgs& a) {
  PF_CHECK(a.K % GEMM_BK == 0 && a.K > 0, PF_ERR_INVALID_ARG, "gemm: K must be a multiple of 64");
  PF_CHECK(a.lda % 8 == 0 && a.ldw % 8 == 0, PF_ERR_INVALID_ARG, "gemm: lda/ldw must be multiples of 8");
  PF_CHECK((!a.out_f32 || a.ldc32 % 4 == 0) && (!a.out_f16 || a.ldc16 % 4 == 0) && (!a.resid || a.ldr % 4 == 0) &&
               (!a.add2 || a.ld2 % 4 == 0) && a.scale_cols % 4 == 0,
           PF_ERR_INVALID_ARG, "gemm: output leading dimensions must be multiples of 4");
  GemmDev d;
  d.A = a.A; d.W = a.W; d.bias = a.bias;
  d.out_f32 = a.out_f32; d.out_f16 = a.out_f16; d.resid = a.resid; d.add2 = a.add2;
  d.lda = a.lda; d.ldw = a.ldw; d.ldc32 = a.ldc32; d.ldc16 = a.ldc16; d.ldr = a.ldr; d.ld2 = a.ld2;
  d.M = a.M; d.N = a.N; d.K = a.K;
  d.relu = a.relu; d.scale_cols = a.scale_cols; d.scale = a.scale_cols > 0 ? a.scale : 1.f;
  const int tiles_m = cdiv(a.M, GEMM_BM);
  d.tiles_n = cdiv(a.N, GEMM_BN);
  if (tiles_m == 0 || d.tiles_n == 0) return;
  static bool attr_set[64] = {false};
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  if (!attr_set[dev & 63]) {
    PF_HIP(hipFuncSetAttribute((const void*)gemm_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               2 * GEMM_STAGE_BYTES));
    attr_set[dev & 63] = true;
  }
  hipLaunchKernelGGL(gemm_f16_kernel, dim3(tiles_m * d.tiles_n), dim3(256), 2 * GEMM_STAGE_BYTES, s, d);
  PF_HIP(hipGetLastError());
}

}  // namespace pf
