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

namespace pf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f16x __attribute__((ext_vector_type(16)));

#define ATT_DK 128
#define ATT_BQ 128
#define ATT_BK 64
#define ATT_TILE_BYTES (ATT_BK * ATT_DK * 2)      // 16 KiB
#define ATT_STAGE_BYTES (2 * ATT_TILE_BYTES)      // K + V

struct AttnDev {
  const half_t* q; const half_t* k; const half_t* v; half_t* o;
  int64_t q_bs, k_bs, v_bs, o_bs;
  int q_rs, k_rs, v_rs, o_rs;
  int H, Lq, Lk;
};

__device__ __forceinline__ void att_glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__global__ __launch_bounds__(256, 2) void attn_kernel(AttnDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5, lc = lane & 31;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int q0 = blockIdx.x * ATT_BQ + wave * 32;

  const half_t* qb = p.q + b * p.q_bs + h * ATT_DK;
  const half_t* kb_ = p.k + b * p.k_bs + h * ATT_DK;
  const half_t* vb = p.v + b * p.v_bs + h * ATT_DK;
  half_t* ob = p.o + b * p.o_bs + h * ATT_DK;

  // ---- Q fragments (B operand): lane supplies Q[q0 + lc][16*ds + 8*lh + 0..7]
  h8 qf[8];
  {
    int qr = q0 + lc;
    qr = qr < p.Lq ? qr : p.Lq - 1;
    const half_t* qp = qb + (int64_t)qr * p.q_rs + 8 * lh;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) qf[ds] = *reinterpret_cast<const h8*>(qp + 16 * ds);
  }

  // ---- staging: wave-instruction i of this wave covers tile rows (wave*4+i)*4 .. +3
  const int srow = lane >> 4, schunk = lane & 15;
  int64_t k_src[4], v_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 4 + srow;
    k_src[i] = (int64_t)row * p.k_rs + ((schunk ^ (row & 15)) << 3);
    v_src[i] = (int64_t)row * p.v_rs + ((schunk ^ ((row & 3) << 2)) << 3);
  }
  auto stage = [&](int buf, int kt) {
    char* kl = smem + buf * ATT_STAGE_BYTES;
    char* vl = kl + ATT_TILE_BYTES;
    const half_t* kg = kb_ + (int64_t)kt * ATT_BK * p.k_rs;
    const half_t* vg = vb + (int64_t)kt * ATT_BK * p.v_rs;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      att_glds16(kg + k_src[i], kl + (wave * 4 + i) * 1024);
      att_glds16(vg + v_src[i], vl + (wave * 4 + i) * 1024);
    }
  };

  // ---- per-lane read offsets
  // K (A operand of S^T): key row kb*32 + lc, chunk (2*ds + lh) ^ (row & 15)
  int k_off[2], k_swz[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int row = kb * 32 + lc;
    k_off[kb] = row * 256; k_swz[kb] = row & 15;
  }
  // V tr-read: key row = kb*32 + 16*s2 + 8*jj + 4*lh + (i>>2), element col = db*32 + 16*g + 4*(i&3)
  const int vi = lane & 15, vg_ = (lane >> 4) & 1;
  const int v_row_base = 4 * lh + (vi >> 2);            // + kb*32 + 16*s2 + 8*jj
  const int v_col_base = 16 * vg_ + 4 * (vi & 3);       // + db*32   (halfs)
  const int v_rswz = (v_row_base & 3) << 2;             // (row & 3) << 2: row offsets added are multiples of 4

  f16x o_acc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) o_acc[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float LOG2E = 1.4426950408889634f;

  const int nkt = (p.Lk + ATT_BK - 1) / ATT_BK;
  stage(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
    const char* kl = smem + (kt & 1) * ATT_STAGE_BYTES;
    const char* vl = kl + ATT_TILE_BYTES;

    // ---- S^T = K Q^T
    f16x s_acc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s_acc[kb][e] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 8; ++ds) {
        const h8 kf = *reinterpret_cast<const h8*>(kl + k_off[kb] + (((2 * ds + lh) ^ k_swz[kb]) << 4));
        s_acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ds], s_acc[kb], 0, 0, 0);
      }
    }
    // ---- mask the tile tail (keys >= Lk)
    if ((kt + 1) * ATT_BK > p.Lk) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kt * ATT_BK + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (key >= p.Lk) s_acc[kb][e] = -INFINITY;
        }
    }
    // ---- online softmax (lane = one query; its 32 keys here, the other 32 in lane^32)
    float mloc = s_acc[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) mloc = fmaxf(mloc, s_acc[kb][e]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = exp2f((m_run - m_new) * LOG2E);
    const float mb = m_new * LOG2E;
    float psum = 0.f;
    h8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = exp2f(s_acc[kb][e] * LOG2E - mb);
        psum += pv;
        pf[kb][e >> 3][e & 7] = (half_t)pv;
      }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) o_acc[d][e] *= alpha;

    // ---- O^T += V^T P^T
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          h8 vf;
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int row = v_row_base + kb * 32 + 16 * s2 + 8 * jj;
            const int colh = v_col_base + db * 32;                 // halfs
            const int chunk = (colh >> 3) ^ v_rswz;
            const int addr = row * 256 + (chunk << 4) + ((colh & 7) << 1);
            const fp4 t = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (__attribute__((address_space(3))) fp4*)(vl + addr));
            vf[4 * jj + 0] = (half_t)t[0]; vf[4 * jj + 1] = (half_t)t[1];
            vf[4 * jj + 2] = (half_t)t[2]; vf[4 * jj + 3] = (half_t)t[3];
          }
          o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][s2], o_acc[db], 0, 0, 0);
        }
      }
    }
  }

  // ---- normalise and store: lane = query q0+lc, d = db*32 + (e&3) + 8*(e>>2) + 4*lh
  const int qrow = q0 + lc;
  if (qrow < p.Lq) {
    const float inv = 1.0f / l_run;
    half_t* op = ob + (int64_t)qrow * p.o_rs + 4 * lh;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h4 hv = {(half_t)(o_acc[db][4 * g + 0] * inv), (half_t)(o_acc[db][4 * g + 1] * inv),
                 (half_t)(o_acc[db][4 * g + 2] * inv), (half_t)(o_acc[db][4 * g + 3] * inv)};
        *reinterpret_cast<h4*>(op + db * 32 + 8 * g) = hv;
      }
  }
}

void launch_attention(hipStream_t s, const AttnArgs& a) {
  if (a.B == 0 || a.Lq == 0 || a.Lk == 0) return;
  AttnDev d;
  d.q = a.q; d.k = a.k; d.v = a.v; d.o = a.o;
  d.q_bs = a.q_bstride; d.k_bs = a.k_bstride; d.v_bs = a.v_bstride; d.o_bs = a.o_bstride;
  d.q_rs = a.q_rstride; d.k_rs = a.k_rstride; d.v_rs = a.v_rstride; d.o_rs = a.o_rstride;
  d.H = a.H; d.Lq = a.Lq; d.Lk = a.Lk;
  PF_CHECK(a.q_rstride % 8 == 0 && a.k_rstride % 8 == 0 && a.v_rstride % 8 == 0 && a.o_rstride % 4 == 0,
           PF_ERR_INVALID_ARG, "attention: row strides must keep 16-byte alignment");
  static bool attr_set[64] = {false};
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  if (!attr_set[dev & 63]) {
    PF_HIP(hipFuncSetAttribute((const void*)attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               2 * ATT_STAGE_BYTES));
    attr_set[dev & 63] = true;
  }
  dim3 grid((a.Lq + ATT_BQ - 1) / ATT_BQ, a.B * a.H);
  hipLaunchKernelGGL(attn_kernel, grid, dim3(256), 2 * ATT_STAGE_BYTES, s, d);
  PF_HIP(hipGetLastError());
}

}  // namespace pf
