// k_decmid.hip — the middle of a SAN-M decoder layer in ONE launch (round 6; VERDICT r5 #6: 5 -> 3 launches per decoder layer).
//
//   t   = LN_F(relu(w_1 norm1(x))) w_2^T          finishing pass of the split FFN form (k_ffn.hip): the S shares'
//                                                  partial rows summed, the LayerNorm over the 2048 hidden columns applied from the
//                                                  row statistics:  t = rstd (sum_s part_s - mean c) + d
//   tn  = norm2(t)
//   x  += FSMN(tn)                                 11- / 21-tap depthwise memory + identity, inputs and outputs masked by token_num
//   xn  = norm3(x)
//   q   = (xn Wq^T + bq) / sqrt(128)               -> f16 [B L, 512], the cross-attention's query
//
// i.e. the Add / LayerNormalization / Conv / Mul / MatMul nodes between a decoder layer's feed-forward block and its
// cross-attention in the graph executed behind AliParaformerAsr/OfflineProjOfParaformer.cs:68.  Until round 5 these were three
// launches per layer (ffn_dec_finish_kernel 9.5 us, fsmn_dec_ln_kernel 12.4 us, the q-projection GEMM 11 us at B L = 5344):
// latency chains of a third of the chip each, with tn (fp32) and xn (f16) written to HBM and read back in between.
//
// Geometry: one 512-thread workgroup per (utterance, 32 consecutive positions): B x ceil(L / 32) workgroups (192 at the benchmark).
//   A  wave w finishes rows w, w + 8, .. of the 32 + K - 1 rows the block's FSMN window needs (one wave per row, a lane owns
//      columns 4 lane .. + 3 and 256 + 4 lane .. + 3, exactly ffn_dec_finish_kernel's arithmetic) -> norm2 -> fp32 rows in LDS;
//      rows outside [0, token_num) are zeros (the graph's masks);
//   B  wave w owns output rows 4 w .. 4 w + 3: FSMN out of the LDS rows (register taps), + x (loaded at kernel start), x stored,
//      norm3 (fsmn_dec_ln_kernel's shifted two-pass form) -> f16 -> the swizzled operand tile of phase C in LDS;
//   C  wave w computes Q^T[64 columns x 32 rows] = Wq[w] xn^T over K = 512: 32 k-steps of 2 MFMA 32x32x16, the Wq fragments
//      straight from a fragment-ordered image (launch_ffn_retile_out, the encoder tail's), 8 in flight; bias in the accumulators
//      from the start, scale, f16, 16-byte row-major stores after a v_permlane32_swap (k_ffn.hip's V pass).
#include "kernels.h"
#include "exact.h"

#include <mutex>

namespace pf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef float4 __attribute__((may_alias)) float4a;

constexpr int DM_R = 32, DM_D = 512, DM_F = 2048;
constexpr int DM_ROWB = DM_D * 4;                         // an fp32 row in LDS
constexpr int DM_TILE = DM_R * DM_D * 2;                  // 32 KiB: 8 k-blocks of [32 rows][128 B]

#ifdef DM_TIMING
__device__ unsigned dm_tm[256 * 8 * 8];
#define DM_TS(i) { tm_[i] = (unsigned)(__builtin_readcyclecounter() - t_begin_); }
#else
#define DM_TS(i)
#endif

struct DecMidDev {
  const float* part; const float* stats; const float* cd; int S, Mp;
  float eps_f;
  const float* n2_g; const float* n2_b; float eps2;
  const float* wT; const int32_t* token_num; int B, L;
  float* x;
  const float* n3_g; const float* n3_b;
  const half_t* Wqt; const float* bq; float qscale; half_t* q16; int ldq;
};

__device__ __forceinline__ float dm_wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (a + b) + (c + d);
}

template <int K>
__global__ __launch_bounds__(512, 1) void dec_mid_kernel(DecMidDev p) {
  constexpr int left = (K - 1) / 2, NROWS = DM_R + K - 1, PF = 8;     // (16 fragments in flight: no difference, profiles/round6_small_ab.txt)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* zrows = smem;                                     // [NROWS][512] fp32: norm2(t) of the window rows
  char* tile = smem + NROWS * DM_ROWB;                    // f16 operand tile of phase C
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5, l31 = lane & 31;
#ifdef DM_TIMING
  unsigned tm_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_begin_ = __builtin_readcyclecounter();
#endif
  const int nblk = (p.L + DM_R - 1) / DM_R;
  const int b = (int)blockIdx.x / nblk, l0 = ((int)blockIdx.x - b * nblk) * DM_R;
  const int nvalid = p.token_num[b];
  const int c0 = 4 * lane, c1 = 256 + 4 * lane;
  auto swz = [](int row) __attribute__((always_inline)) -> int { return (row >> 1) & 7; };

  // ---- weight stream of phase C: primed now, it lands under phases A and B
  const half_t* wqu = p.Wqt + (size_t)wave * (64 * 512);
  const unsigned lane16 = (unsigned)lane * 16u;
  auto wqload = [&](int pos) __attribute__((always_inline)) -> h8 {
    return *reinterpret_cast<const h8*>(reinterpret_cast<const char*>(wqu + pos * 512) + lane16);
  };
  h8 ring[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) ring[i] = wqload(i);
  // the residual rows of phase B (wave w: rows 4 w .. 4 w + 3), requested up front
  float4 xr[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int l = l0 + 4 * wave + q;
    xr[q][0] = xr[q][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (l < p.L) {
      const float* r = p.x + ((size_t)b * p.L + l) * DM_D;
      xr[q][0] = *reinterpret_cast<const float4*>(r + c0);
      xr[q][1] = *reinterpret_cast<const float4*>(r + c1);
    }
  }

  // ---- A: finishing pass + norm2 of the window rows -> LDS
  {
    const float4 cc0 = *reinterpret_cast<const float4*>(p.cd + c0), cc1 = *reinterpret_cast<const float4*>(p.cd + c1);
    const float4 dd0 = *reinterpret_cast<const float4*>(p.cd + 512 + c0), dd1 = *reinterpret_cast<const float4*>(p.cd + 512 + c1);
    const float4 g0 = *reinterpret_cast<const float4*>(p.n2_g + c0), g1 = *reinterpret_cast<const float4*>(p.n2_g + c1);
    const float4 e0 = *reinterpret_cast<const float4*>(p.n2_b + c0), e1 = *reinterpret_cast<const float4*>(p.n2_b + c1);
    // the loads of ALL of this wave's rows go out together (share by share): a row at a time would be RPW dependent round trips
    constexpr int RPW = (NROWS + 7) / 8;
    float4 a0[RPW], a1[RPW];
    float sm[RPW], sq[RPW];
    bool live[RPW];
    size_t mrow[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wave + 8 * i, tt = l0 - left + r;
      live[i] = r < NROWS && tt >= 0 && tt < nvalid && tt < p.L;
      mrow[i] = (size_t)b * p.L + (live[i] ? tt : 0);
      a0[i] = a1[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      sm[i] = sq[i] = 0.f;
    }
    for (int sp = 0; sp < p.S; ++sp) {
      float4 v0[RPW], v1[RPW];
      float2 st[RPW];
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        v0[i] = v1[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        st[i] = make_float2(0.f, 0.f);
        if (live[i]) {
          const float* rp = p.part + ((size_t)sp * p.Mp + mrow[i]) * DM_D;
          v0[i] = *reinterpret_cast<const float4*>(rp + c0);
          v1[i] = *reinterpret_cast<const float4*>(rp + c1);
          st[i] = *reinterpret_cast<const float2*>(p.stats + ((size_t)sp * p.Mp + mrow[i]) * 2);
        }
      }
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        a0[i].x += v0[i].x; a0[i].y += v0[i].y; a0[i].z += v0[i].z; a0[i].w += v0[i].w;
        a1[i].x += v1[i].x; a1[i].y += v1[i].y; a1[i].z += v1[i].z; a1[i].w += v1[i].w;
        sm[i] += st[i].x; sq[i] += st[i].y;
      }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wave + 8 * i;
      if (r >= NROWS) continue;
      float4 z0 = make_float4(0.f, 0.f, 0.f, 0.f), z1 = z0;
      if (live[i]) {
        const double mu_d = (double)sm[i] * (1.0 / DM_F);
        double var = (double)sq[i] * (1.0 / DM_F) - mu_d * mu_d;
        var = var > 0.0 ? var : 0.0;
        const float mu = (float)mu_d, rs = (float)(1.0 / sqrt(var + (double)p.eps_f));
        const float4 x0 = a0[i], x1 = a1[i];
        float4 y0 = make_float4(rs * (x0.x - mu * cc0.x) + dd0.x, rs * (x0.y - mu * cc0.y) + dd0.y, rs * (x0.z - mu * cc0.z) + dd0.z, rs * (x0.w - mu * cc0.w) + dd0.w);
        float4 y1 = make_float4(rs * (x1.x - mu * cc1.x) + dd1.x, rs * (x1.y - mu * cc1.y) + dd1.y, rs * (x1.z - mu * cc1.z) + dd1.z, rs * (x1.w - mu * cc1.w) + dd1.w);
        const float mean = dm_wave_sum(((y0.x + y0.y) + (y0.z + y0.w)) + ((y1.x + y1.y) + (y1.z + y1.w))) * (1.0f / DM_D);
        y0.x -= mean; y0.y -= mean; y0.z -= mean; y0.w -= mean; y1.x -= mean; y1.y -= mean; y1.z -= mean; y1.w -= mean;
        const float k = 1.0f / sqrtf(dm_wave_sum(((y0.x * y0.x + y0.y * y0.y) + (y0.z * y0.z + y0.w * y0.w)) +
                                                 ((y1.x * y1.x + y1.y * y1.y) + (y1.z * y1.z + y1.w * y1.w))) * (1.0f / DM_D) + p.eps2);
        z0 = make_float4(y0.x * k * g0.x + e0.x, y0.y * k * g0.y + e0.y, y0.z * k * g0.z + e0.z, y0.w * k * g0.w + e0.w);
        z1 = make_float4(y1.x * k * g1.x + e1.x, y1.y * k * g1.y + e1.y, y1.z * k * g1.z + e1.z, y1.w * k * g1.w + e1.w);
      }
      *reinterpret_cast<float4a*>(zrows + (size_t)r * DM_ROWB + c0 * 4) = z0;
      *reinterpret_cast<float4a*>(zrows + (size_t)r * DM_ROWB + c1 * 4) = z1;
    }
  }
  DM_TS(0)
  __syncthreads();
  DM_TS(1)

  // ---- B: FSMN + residual + norm3 -> x (HBM) and the f16 operand tile (LDS)
  {
    const int r0 = 4 * wave;                              // output rows r0 .. r0 + 3 of the block; window rows r0 .. r0 + 3 + K - 1
    float4 acc[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q][0] = acc[q][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (l0 + r0 < nvalid) {
      float4 w[K][2];
#pragma unroll
      for (int j = 0; j < K; ++j) {
        w[j][0] = *reinterpret_cast<const float4*>(p.wT + (size_t)j * DM_D + c0);
        w[j][1] = *reinterpret_cast<const float4*>(p.wT + (size_t)j * DM_D + c1);
      }
#pragma unroll
      for (int s = 0; s < 4 + K - 1; ++s) {
        const float4 xa = *reinterpret_cast<const float4a*>(zrows + (size_t)(r0 + s) * DM_ROWB + c0 * 4);
        const float4 xb = *reinterpret_cast<const float4a*>(zrows + (size_t)(r0 + s) * DM_ROWB + c1 * 4);
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const int q = s - j;
          if (q >= 0 && q < 4) {
            acc[q][0].x += w[j][0].x * xa.x; acc[q][0].y += w[j][0].y * xa.y; acc[q][0].z += w[j][0].z * xa.z; acc[q][0].w += w[j][0].w * xa.w;
            acc[q][1].x += w[j][1].x * xb.x; acc[q][1].y += w[j][1].y * xb.y; acc[q][1].z += w[j][1].z * xb.z; acc[q][1].w += w[j][1].w * xb.w;
          }
        }
        const int q0 = s - left;
        if (q0 >= 0 && q0 < 4) {
          acc[q0][0].x += xa.x; acc[q0][0].y += xa.y; acc[q0][0].z += xa.z; acc[q0][0].w += xa.w;
          acc[q0][1].x += xb.x; acc[q0][1].y += xb.y; acc[q0][1].z += xb.z; acc[q0][1].w += xb.w;
        }
      }
    }
    const float4 g0 = *reinterpret_cast<const float4*>(p.n3_g + c0), g1 = *reinterpret_cast<const float4*>(p.n3_g + c1);
    const float4 e0 = *reinterpret_cast<const float4*>(p.n3_b + c0), e1 = *reinterpret_cast<const float4*>(p.n3_b + c1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = r0 + q, l = l0 + row;
      h4 y0 = h4{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f}, y1 = y0;
      if (l < p.L) {
        float4 a = xr[q][0], c = xr[q][1];
        if (l < nvalid) {
          a = make_float4(a.x + acc[q][0].x, a.y + acc[q][0].y, a.z + acc[q][0].z, a.w + acc[q][0].w);
          c = make_float4(c.x + acc[q][1].x, c.y + acc[q][1].y, c.z + acc[q][1].z, c.w + acc[q][1].w);
          xr[q][0] = a; xr[q][1] = c;                     // stored behind phase C: stores retire through vmcnt in order with the
        }                                                 // loads and would sit in front of every counted wait of the Wq stream
        const float x0 = __shfl(a.x, 0, 64);
        a.x = sub_rn(a.x, x0); a.y = sub_rn(a.y, x0); a.z = sub_rn(a.z, x0); a.w = sub_rn(a.w, x0);
        c.x = sub_rn(c.x, x0); c.y = sub_rn(c.y, x0); c.z = sub_rn(c.z, x0); c.w = sub_rn(c.w, x0);
        const float mean = dm_wave_sum(((a.x + a.y) + (a.z + a.w)) + ((c.x + c.y) + (c.z + c.w))) / (float)DM_D;
        a.x -= mean; a.y -= mean; a.z -= mean; a.w -= mean; c.x -= mean; c.y -= mean; c.z -= mean; c.w -= mean;
        const float var = dm_wave_sum(((a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w)) + ((c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w))) / (float)DM_D;
        const float rstd = 1.0f / sqrtf(var + 1e-12f);
        y0 = h4{(half_t)(a.x * rstd * g0.x + e0.x), (half_t)(a.y * rstd * g0.y + e0.y), (half_t)(a.z * rstd * g0.z + e0.z), (half_t)(a.w * rstd * g0.w + e0.w)};
        y1 = h4{(half_t)(c.x * rstd * g1.x + e1.x), (half_t)(c.y * rstd * g1.y + e1.y), (half_t)(c.z * rstd * g1.z + e1.z), (half_t)(c.w * rstd * g1.w + e1.w)};
      }
      // column c -> k-block c >> 6, 16-byte chunk (c & 63) >> 3 (XOR-swizzled by row), 8 bytes inside it
      *reinterpret_cast<h4*>(tile + (c0 >> 6) * (DM_R * 128) + row * 128 + ((((c0 & 63) >> 3) ^ swz(row)) << 4) + (c0 & 7) * 2) = y0;
      *reinterpret_cast<h4*>(tile + (c1 >> 6) * (DM_R * 128) + row * 128 + ((((c1 & 63) >> 3) ^ swz(row)) << 4) + (c1 & 7) * 2) = y1;
    }
  }
  DM_TS(2)
  __syncthreads();
  DM_TS(3)

  // ---- C: Q^T[64 x 32] of this wave; accumulators start as the bias of its 64 columns
  f16x yacc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b4 = *reinterpret_cast<const float4*>(p.bq + wave * 64 + j * 32 + 8 * g + 4 * lh);
      yacc[j][4 * g + 0] = b4.x; yacc[j][4 * g + 1] = b4.y; yacc[j][4 * g + 2] = b4.z; yacc[j][4 * g + 3] = b4.w;
    }
  unsigned xo[4];
#pragma unroll
  for (int ss = 0; ss < 4; ++ss) xo[ss] = (unsigned)(l31 * 128 + (((2 * ss + lh) ^ swz(l31)) << 4));
  {
    constexpr int XD = 2, XR = XD + 1;
    h8 xf[XR];
#pragma unroll
    for (int s0 = 0; s0 < XD; ++s0) xf[s0] = *reinterpret_cast<const h8*>(tile + xo[s0 & 3] + (s0 >> 2) * (DM_R * 128));
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      if (s + XD < 32) xf[(s + XD) % XR] = *reinterpret_cast<const h8*>(tile + xo[(s + XD) & 3] + ((s + XD) >> 2) * (DM_R * 128));
      h8 wq[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        wq[j] = ring[(2 * s + j) % PF];
        if (2 * s + j + PF < 64) ring[(2 * s + j) % PF] = wqload(2 * s + j + PF);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) yacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[j], xf[s % XR], yacc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int l = l0 + 4 * wave + q;
    if (l < nvalid && l < p.L) {
      float* r = p.x + ((size_t)b * p.L + l) * DM_D;
      *reinterpret_cast<float4*>(r + c0) = xr[q][0];
      *reinterpret_cast<float4*>(r + c1) = xr[q][1];
    }
  }
  DM_TS(4)
  // rows l0 + l31: lanes l / l + 32 hold columns 8 g + 0..3 / 8 g + 4..7; after v_permlane32_swap lane l holds the 8 columns of
  // group 2 gp, lane l + 32 those of group 2 gp + 1: 16-byte row-major stores
  typedef float f2v __attribute__((ext_vector_type(2)));
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  const int lrow = l0 + l31;
  half_t* qrow = p.q16 + ((size_t)b * p.L + lrow) * p.ldq + wave * 64;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      unsigned x[2], y[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        f2v xa = {yacc[j][8 * gp + 2 * e + 0], yacc[j][8 * gp + 2 * e + 1]};
        f2v ya = {yacc[j][8 * gp + 4 + 2 * e + 0], yacc[j][8 * gp + 4 + 2 * e + 1]};
        xa *= p.qscale; ya *= p.qscale;
        x[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(xa, h2v));
        y[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(ya, h2v));
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[e]), "+v"(y[e]));
      const h4 lo_ = __builtin_bit_cast(h4, (unsigned long long)x[0] | ((unsigned long long)x[1] << 32));
      const h4 hi_ = __builtin_bit_cast(h4, (unsigned long long)y[0] | ((unsigned long long)y[1] << 32));
      const h8 hv = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);
      if (lrow < p.L) *reinterpret_cast<h8*>(qrow + j * 32 + 16 * gp + 8 * lh) = hv;
    }
#ifdef DM_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DM_TS(5)
  if (lane == 0 && blockIdx.x < 256)
    for (int i = 0; i < 8; ++i) dm_tm[(blockIdx.x * 8 + wave) * 8 + i] = tm_[i];
#endif
}

#ifdef DM_TIMING
}  // namespace pf
extern "C" int pf_debug_decmid_timing(unsigned* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pf::dm_tm), (size_t)n * 4, 0, hipMemcpyDeviceToHost);
}
namespace pf {
#endif

size_t dec_mid_lds_bytes(int k) { return (size_t)(DM_R + k - 1) * DM_ROWB + DM_TILE; }

bool launch_dec_mid(hipStream_t s, const DecMidArgs& a) {
  if (a.k != 11 || !a.Wqt || !a.bq || a.B <= 0 || a.L <= 0) return false;    // (21 taps — the SeACo bias decoder's — would spill: its path keeps the three launches)
  PF_CHECK(a.ws && a.img && a.x && a.q16 && a.token_num && a.n2_g && a.n3_g && a.fsmn_wT && a.ldq % 8 == 0, PF_ERR_INVALID_ARG, "dec_mid: missing operand");
  const int M = a.B * a.L, S = a.splits > 0 ? a.splits : ffn_dec_splits(M), Mp = (int)round_up(M, 64);
  DecMidDev d{};
  d.part = reinterpret_cast<const float*>(a.ws); d.stats = d.part + (size_t)S * Mp * DM_D; d.S = S; d.Mp = Mp;
  d.cd = ffn_dec_image_cd(a.img); d.eps_f = a.eps_hidden;
  d.n2_g = a.n2_g; d.n2_b = a.n2_b; d.eps2 = a.eps2;
  d.wT = a.fsmn_wT; d.token_num = a.token_num; d.B = a.B; d.L = a.L; d.x = a.x;
  d.n3_g = a.n3_g; d.n3_b = a.n3_b; d.Wqt = a.Wqt; d.bq = a.bq; d.qscale = a.qscale; d.q16 = a.q16; d.ldq = a.ldq;
  static std::mutex init_mu;
  static bool attr_set[64] = {false};
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(init_mu);
    if (!attr_set[dev & 63]) {
      PF_HIP(hipFuncSetAttribute((const void*)dec_mid_kernel<11>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dec_mid_lds_bytes(11)));
      attr_set[dev & 63] = true;
    }
  }
  const dim3 grid((unsigned)(a.B * cdiv(a.L, DM_R)));
  hipLaunchKernelGGL(dec_mid_kernel<11>, grid, dim3(512), dec_mid_lds_bytes(11), s, d);
  PF_HIP(hipGetLastError());
  return true;
}

}  // namespace pf
