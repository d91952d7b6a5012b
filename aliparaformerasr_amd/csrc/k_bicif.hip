// k_bicif.hip — BiCIF timestamp head (CifPredictorV3.get_upsample_timestmap of the FunASR export;
// its output `us_cif_peak` is graph output [3] consumed by AliParaformerAsr/OfflineRecognizer.cs:172-183
// and turned into timestamps by time_stamp_lfr6_onnx, OfflineRecognizer.cs:200-302).
//
//   up   = ConvTranspose1d(D, D, k=3, stride=3)(H)           -> GEMM  [M,512] x [512,1536]   (k_gemm.hip)
//   xg   = up W_ih^T + b_ih + b_hh  (both directions)         -> GEMM  [3M,512] x [512,4096]  (k_gemm.hip)
//   h_t  = LSTM cell, forward and reverse direction            -> lstm_step_kernel, one launch per time step
//   a2   = relu(sigmoid(h W_o^T + b_o) * smooth2 - noise2)     -> us_alpha_kernel
//   a2  *= token_num / sum(a2);  peak = running integrate       -> us_peak_kernel
//
// The recurrence is a chain of 3T dependent steps; round 1 runs it as one small launch per step
// (latency bound, ~5 us/step).  Each launch covers both directions; a block owns 8 hidden units
// (x 4 gates = one 32-row MFMA tile of W_hh) for a tile of 32 utterances, its 4 waves split K = 512.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "exact.h"
#include "kernels.h"

namespace pf {

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// Row rho of the MFMA A tile is W_hh row (gate = rho & 3, unit = u0 + (rho >> 2)); the 32x32 result
// then leaves every lane with the 4 gates of 4 units for one utterance (D rows 8q + 4*(lane>>5) + r).
__global__ __launch_bounds__(256) void lstm_step_kernel(LstmArgs a) {
  const int D = a.D;
  const int ub = blockIdx.x, dir = blockIdx.y, bt = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = dir == 0 ? a.step : a.T3 - 1 - a.step;
  const int pp = a.step & 1;
  const half_t* hprev = a.hstate + (size_t)(dir * 2 + pp) * a.B * D;
  half_t* hnext = a.hstate + (size_t)(dir * 2 + (pp ^ 1)) * a.B * D;
  const int r = lane & 31, kg = lane >> 5;
  const half_t* wrow = a.whh + ((size_t)dir * 4 * D + (size_t)(r & 3) * D + ub * 8 + (r >> 2)) * D;
  const int bb = min(bt * 32 + r, a.B - 1);
  const half_t* hrow = hprev + (size_t)bb * D;

  // the cell's own inputs (this thread's unit / utterance) are fetched first: their HBM latency then
  // overlaps the W_hh / h loads and the MFMAs instead of following the LDS reduction
  const int bq = min(bt * 32 + r, a.B - 1);
  const int uq = ub * 8 + 2 * wave + kg;
  const float* xgp = a.xg + ((size_t)bq * a.T3 + t) * (size_t)(a.ndir * 4 * D) + (size_t)dir * 4 * D + uq;
  float* cp = a.cstate + ((size_t)dir * a.B + bq) * D + uq;
  const float xi = xgp[0], xf = xgp[D], xc = xgp[2 * D], xo = xgp[3 * D], cprev = *cp;

  const int kspan = D / 4;               // K slice of this wave
  f16v acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = wave * kspan; k0 < (wave + 1) * kspan; k0 += 128) {
    h8v av[8], bv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      av[s] = *reinterpret_cast<const h8v*>(wrow + k0 + s * 16 + kg * 8);
      bv[s] = *reinterpret_cast<const h8v*>(hrow + k0 + s * 16 + kg * 8);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bv[s], acc, 0, 0, 0);
  }

  __shared__ float red[4][16][64];
#pragma unroll
  for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
  __syncthreads();
  float g[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    g[q] = (red[0][4 * wave + q][lane] + red[1][4 * wave + q][lane]) + (red[2][4 * wave + q][lane] + red[3][4 * wave + q][lane]);

  const int b = bt * 32 + r;
  if (b >= a.B) return;
  const int u = uq;
  const float gi = g[0] + xi, gf = g[1] + xf, gg = g[2] + xc, go = g[3] + xo;
  const float c = sigmoidf_(gf) * cprev + sigmoidf_(gi) * tanhf(gg);
  const float h = sigmoidf_(go) * tanhf(c);
  *cp = c;
  hnext[(size_t)b * D + u] = (half_t)h;
  a.hout[((size_t)b * a.T3 + t) * (size_t)(a.ndir * D) + (size_t)dir * D + u] = h;
}

// ------------------------------------------------------------------------------------------------------------
// PERSISTENT recurrence: ONE launch runs all T3 steps.  Same work split as lstm_step_kernel (a workgroup = 8 hidden
// units x 4 gates x one tile of 32 utterances, its 4 waves split K), but
//   * the workgroup's W_hh slice (32 rows x 512 x f16 = 32 VGPRs per lane) is loaded ONCE and stays in registers,
//     the cell state lives in a register;
//   * h is exchanged between the D/8 workgroups of one (direction, utterance tile) through the global ping-pong
//     buffer, hand-off per the CDNA guide's recipe R1: 16-byte write-through (sc1) stores of the new h slice, every
//     storing wave drains vmcnt, ONE lane adds to a monotonic arrival counter; consumers poll that word (relaxed,
//     bounded spin with s_sleep) and then read h with sc1 loads (L2-served, never this CU's stale L1);
//   * step s may overwrite the buffer step s-1 read: a workgroup arrives at the step-(s-1) counter only AFTER its
//     reads of that step, and nobody starts step s before all arrivals — no further ordering is needed.
// All (D/8) * ndir * tiles workgroups must be resident at once (launcher checks against the CU count); every spin
// is bounded: on time-out the kernel raises *err and every workgroup leaves.
// eight 16-byte sc1 loads (offsets 0, 32, .., 224 bytes) issued back to back, ONE wait: the loads and their wait live
// in one asm statement, so the compiler never sees a destination before the data has landed (CDNA guide §5.7 item 1)
__device__ __forceinline__ void ld8x16_sc1(const half_t* p, h8v (&v)[8]) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\t"
      "global_load_dwordx4 %1, %8, off offset:32 sc1\n\t"
      "global_load_dwordx4 %2, %8, off offset:64 sc1\n\t"
      "global_load_dwordx4 %3, %8, off offset:96 sc1\n\t"
      "global_load_dwordx4 %4, %8, off offset:128 sc1\n\t"
      "global_load_dwordx4 %5, %8, off offset:160 sc1\n\t"
      "global_load_dwordx4 %6, %8, off offset:192 sc1\n\t"
      "global_load_dwordx4 %7, %8, off offset:224 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(p)
      : "memory");
}
__device__ __forceinline__ void st16_sc1(half_t* p, h8v v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

__global__ __launch_bounds__(256) void lstm_persistent_kernel(LstmArgs a, unsigned* __restrict__ cnt, unsigned* __restrict__ err) {
  const int D = a.D;
  const int ub = blockIdx.x, dir = blockIdx.y, bt = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, kg = lane >> 5;
  const int nub = gridDim.x;                                // producers per (direction, tile)
  unsigned* my_cnt = cnt + (dir * gridDim.z + bt);
  __shared__ float red[4][16][64];
  __shared__ _Float16 hx[32][8];                            // new h slice: [utterance][unit] -> 16-byte rows
  __shared__ int s_abort;

  // W_hh rows of this lane, K slice of this wave: resident for the whole launch
  const half_t* wrow = a.whh + ((size_t)dir * 4 * D + (size_t)(r & 3) * D + ub * 8 + (r >> 2)) * D;
  const int kspan = D / 4;                                  // 128 (D = 512)
  h8v av[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) av[s] = *reinterpret_cast<const h8v*>(wrow + wave * kspan + s * 16 + kg * 8);

  const int bb = min(bt * 32 + r, a.B - 1);                 // utterance whose h row this lane feeds to the MFMA
  const int b = bt * 32 + r;                                // utterance of this lane's cell
  const int uq = ub * 8 + 2 * wave + kg;                    // hidden unit of this lane's cell
  float c = 0.f;
  if (threadIdx.x == 0) s_abort = 0;
  __syncthreads();

  for (int step = 0; step < a.T3; ++step) {
    const int t = dir == 0 ? step : a.T3 - 1 - step;
    const int pp = step & 1;
    const half_t* hprev = a.hstate + (size_t)(dir * 2 + pp) * a.B * D;
    half_t* hnext = a.hstate + (size_t)(dir * 2 + (pp ^ 1)) * a.B * D;
    // the cell's own inputs do not depend on h: requested before the wait
    const float* xgp = a.xg + ((size_t)bb * a.T3 + t) * (size_t)(a.ndir * 4 * D) + (size_t)dir * 4 * D + uq;
    const float xi = xgp[0], xf = xgp[D], xc = xgp[2 * D], xo = xgp[3 * D];
    if (step > 0) {
      if (threadIdx.x == 0) {
        const unsigned want = (unsigned)step * (unsigned)nub;
        unsigned spins = 0;
        while (__hip_atomic_load(my_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 22) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_abort = 1;
            break;
          }
        }
      }
      __syncthreads();
      if (s_abort) return;
    }
    const half_t* hrow = hprev + (size_t)bb * D + wave * kspan + kg * 8;
    f16v acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    h8v bv[8];
    // one wait per load measured FASTER than eight loads behind one wait (6.5-6.8 vs 7.8 ms for the 1500 steps, same
    // session, tools/lstm_ab.sh): 512 waves issuing 8 write-through-coherent loads at once queue behind each other and
    // behind the pollers.  a.step is free in the persistent form: PF_LSTM_VAR=1 selects the batched form for experiments
    if (a.step == 1) ld8x16_sc1(hrow, bv);
    else {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        h8v v;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(hrow + s * 16) : "memory");
        bv[s] = v;
      }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bv[s], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    float g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      g[q] = (red[0][4 * wave + q][lane] + red[1][4 * wave + q][lane]) + (red[2][4 * wave + q][lane] + red[3][4 * wave + q][lane]);
    const float gi = g[0] + xi, gf = g[1] + xf, gg = g[2] + xc, go = g[3] + xo;
    c = sigmoidf_(gf) * c + sigmoidf_(gi) * tanhf(gg);
    const float h = sigmoidf_(go) * tanhf(c);
    hx[r][2 * wave + kg] = (_Float16)h;
    if (b < a.B) a.hout[((size_t)b * a.T3 + t) * (size_t)(a.ndir * D) + (size_t)dir * D + uq] = h;
    __syncthreads();
    if (wave == 0 && lane < 32 && bt * 32 + lane < a.B)       // 32 x 16 bytes, write-through
      st16_sc1(hnext + (size_t)(bt * 32 + lane) * D + ub * 8, *reinterpret_cast<const h8v*>(&hx[lane][0]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains (R1)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(my_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- ring form of the persistent recurrence (default): the h exchange carries its own arrival information.
// h lives in FOUR slots per direction.  Step s reads slot s % 4 (h_{s-1}), writes h_s into slot (s + 1) % 4 and — once the
// workgroup holds all of h_{s-1}, i.e. every workgroup of its group has finished reading slot (s + 3) % 4 (the input of
// step s - 1) — overwrites its own 16-byte granules of slot (s + 3) % 4 with a POISON pattern (f16 NaNs; a hidden state
// is sigmoid * tanh, never NaN).  That slot receives data two steps later (h_{s+2}), so the poison store has a whole
// step to complete and nothing waits for it (with three slots the data store of the NEXT step had to wait for it).  A consumer simply loads the granules it needs (write-through-coherent loads) and
// repeats while any of them is still poison: no arrival counter, no separate poll, no serialised loads — the chain of a
// step is producer store -> consumer load instead of store -> drain -> counter add -> counter poll -> eight dependent
// loads.  Why a consumer polling for h_{s+2} can only see poison or h_{s+2} in a producer's granule, never the stale
// h_{s-2}: it received that producer's h_{s+1}, which wave 0 of the producer stored in step s + 1 after that step's
// polls, and every poll ends in s_waitcnt vmcnt(0) — which also retired wave 0's poison store of step s (memory
// operations of a wave retire in order).  A granule is written by one 16-byte store of one lane (observed untorn on
// gfx950; the poison test reads its first word).  Bounded spins + the error word as in the counter form.
// hstate: [ndir][4][B][D], slot 0 zero, slots 1-3 poison.
// X3 (math_mode 3, the exact mode's timestamp head): W_hh and h travel as (hi, lo') f16 pairs — 22 mantissa bits — and a step's
// product is hi_W lo'_h + lo'_W hi_h (16 MFMAs), the accumulators x 2^-11, + hi_W hi_h (8 MFMAs) in one fp32 accumulator, as the
// Linears of that mode.  whh rows are then [hi (D) | lo' (D)] (launch_split_x3), a slot row of hstate
// likewise: a producer publishes TWO 16-byte granules per utterance (hi at column ub * 8, lo' at D + ub * 8), each poisoned and
// polled like the single granule of the f16 form (the argument in the header holds per granule: both poison stores of a step
// precede the counted wait, both value stores follow it).
template <bool X3>
__global__ __launch_bounds__(320) void lstm_ring_kernel(LstmArgs a, unsigned* __restrict__ err) {
  const int D = a.D;
  const int RS = X3 ? 2 * D : D;                            // row stride of whh and of a slot row (f16 elements)
  const int ub = blockIdx.x, dir = blockIdx.y, bt = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, kg = lane >> 5;
  __shared__ float red[4][16][64];
  __shared__ _Float16 hx[32][8];
  __shared__ _Float16 hxl[X3 ? 32 : 1][8];                  // X3: the lo' halves of the new slice
  __shared__ float hf[32][8];                               // the same slice in fp32 for hout
  __shared__ int s_abort;
  // wave 4 is the STORE wave: poison, h and hout stores are its only memory operations, so the compute waves' polls
  // (s_waitcnt vmcnt(0) each) never sit behind the acknowledgement of a write-through store
  const bool storer = wave == 4;
  const int cw = storer ? 0 : wave;                          // K slice index of a compute wave
  const half_t* wrow = a.whh + ((size_t)dir * 4 * D + (size_t)(r & 3) * D + ub * 8 + (r >> 2)) * RS;
  const int kspan = D / 4;
  h8v av[8], avl[X3 ? 8 : 1];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    av[s] = *reinterpret_cast<const h8v*>(wrow + cw * kspan + s * 16 + kg * 8);
    if constexpr (X3) avl[s] = *reinterpret_cast<const h8v*>(wrow + D + cw * kspan + s * 16 + kg * 8);
  }
  const int bb = min(bt * 32 + r, a.B - 1);
  const int uq = ub * 8 + 2 * cw + kg;
  float c = 0.f;
  if (threadIdx.x == 0) s_abort = 0;
  __syncthreads();
  h8v poison;
#pragma unroll
  for (int e = 0; e < 8; ++e) poison[e] = __builtin_bit_cast(_Float16, (unsigned short)0xFFFFu);
  half_t* const slots = a.hstate + (size_t)dir * 4 * a.B * RS;
  for (int step = 0; step < a.T3; ++step) {
    const int t = dir == 0 ? step : a.T3 - 1 - step;
    const int si = step & 3, so = (step + 1) & 3, sp = (step + 3) & 3;
    if (storer) {
      __syncthreads();                                         // (1) the workgroup holds all of h_{step-1}
      if (s_abort) return;
      if (lane < 32 && bt * 32 + lane < a.B) {                 // re-arm this workgroup's granules of the slot read one step ago
        st16_sc1(slots + ((size_t)sp * a.B + bt * 32 + lane) * RS + ub * 8, poison);
        if constexpr (X3) st16_sc1(slots + ((size_t)sp * a.B + bt * 32 + lane) * RS + D + ub * 8, poison);
      }
      __syncthreads();                                         // (2) hx / hf hold the new slice
      // the poison stores of the PREVIOUS step must have completed before this step's h becomes visible (see above): of
      // this wave's stores only {poison(step), hout(step-1), h(step-1)} may still be in flight (X3: two granules each)
      if constexpr (X3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      if (lane < 32 && bt * 32 + lane < a.B) {
        st16_sc1(slots + ((size_t)so * a.B + bt * 32 + lane) * RS + ub * 8, *reinterpret_cast<const h8v*>(&hx[lane][0]));
        if constexpr (X3) st16_sc1(slots + ((size_t)so * a.B + bt * 32 + lane) * RS + D + ub * 8, *reinterpret_cast<const h8v*>(&hxl[lane][0]));
      }
      const int ob = bt * 32 + (lane >> 1);
      if (ob < a.B)
        *reinterpret_cast<float4*>(a.hout + ((size_t)ob * a.T3 + t) * (size_t)(a.ndir * D) + (size_t)dir * D + ub * 8 + (lane & 1) * 4) =
            *reinterpret_cast<const float4*>(&hf[lane >> 1][(lane & 1) * 4]);
      continue;                                                // (the next step's barrier (1) orders these reads of hx / hf before their rewrite)
    }
    const float* xgp = a.xg + ((size_t)bb * a.T3 + t) * (size_t)(a.ndir * 4 * D) + (size_t)dir * 4 * D + uq;
    const float xi = xgp[0], xf = xgp[D], xc = xgp[2 * D], xo = xgp[3 * D];
    const half_t* hrow = slots + ((size_t)si * a.B + bb) * RS + wave * kspan + kg * 8;
    h8v bv[8], bvl[X3 ? 8 : 1];
    unsigned spins = 0;
    // cheap poll first: lane j < 16 watches the first word of producer (wave * 16 + j)'s granule for the tile's first
    // utterance (64 bytes per wave and poll; polling with the eight full loads — 8 KB per wave — kept 4 MB per round in
    // flight on the fabric and made a step 7 us instead of 4.2); the full loads follow and are re-checked
    const unsigned* watch = reinterpret_cast<const unsigned*>(slots + ((size_t)si * a.B + bt * 32) * RS + wave * kspan + (lane & 15) * 8);
    auto load_all = [&]() __attribute__((always_inline)) -> bool {      // true = some granule is still poison
      ld8x16_sc1(hrow, bv);
      bool bad = false;
#pragma unroll
      for (int s = 0; s < 8; ++s) bad |= (__builtin_bit_cast(uint4, bv[s]).x == 0xFFFFFFFFu);
      if constexpr (X3) {
        ld8x16_sc1(hrow + D, bvl);
#pragma unroll
        for (int s = 0; s < 8; ++s) bad |= (__builtin_bit_cast(uint4, bvl[s]).x == 0xFFFFFFFFu);
      }
      return __any(bad);
    };
    // the workgroups run in lockstep, so the others' h_{step-1} becomes visible about one store latency after this
    // workgroup published its own: wait that long (a.step x 64 clocks), then ask for the real thing at once — when it is
    // there the step has ONE load round trip after arrival instead of two (successful cheap poll + the full loads)
    bool got = false;
    if (step > 0 && a.step > 0) {
      for (int z = 0; z < a.step; ++z) __builtin_amdgcn_s_sleep(2);
      got = !load_all();
    }
    while (!got) {
      unsigned w0;
      asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(w0) : "v"(watch) : "memory");
      const bool bad = w0 == 0xFFFFFFFFu;
      if (!__any(bad)) {
        if (!load_all()) break;
      }
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22) || (lane == 0 && (spins & 63) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
        if (lane == 0) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_abort = 1; }
        break;
      }
    }
    f16v acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if constexpr (X3) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bvl[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(avl[s], bv[s], acc, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] *= (1.0f / 2048.0f);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s], bv[s], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();                                           // (1) all four K slices are in
    if (s_abort) return;
    float g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      g[q] = (red[0][4 * wave + q][lane] + red[1][4 * wave + q][lane]) + (red[2][4 * wave + q][lane] + red[3][4 * wave + q][lane]);
    const float gi = g[0] + xi, gf = g[1] + xf, gg = g[2] + xc, go = g[3] + xo;
    c = sigmoidf_(gf) * c + sigmoidf_(gi) * tanhf(gg);
    const float h = sigmoidf_(go) * tanhf(c);
    const _Float16 hh = (_Float16)h;
    hx[r][2 * wave + kg] = hh;
    if constexpr (X3) hxl[r][2 * wave + kg] = (_Float16)((h - (float)hh) * 2048.0f);
    hf[r][2 * wave + kg] = h;
    __syncthreads();                                           // (2)
  }
}

// returns false when the persistent form cannot be used (grid larger than the device, or D != 512): caller falls
// back to one launch per step
bool launch_lstm_persistent(hipStream_t s, const LstmArgs& a, unsigned* sync_words /* >= 64 words, device */) {
  if (a.D != 512 || (a.ndir != 1 && a.ndir != 2)) return false;
  const int tiles = cdiv(a.B, 32);
  const int wgs = (a.D / 8) * a.ndir * tiles;
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  int cus = 0;
  PF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  if (wgs > cus || a.ndir * tiles > 60) return false;       // every workgroup must be resident: one per CU at most
  PF_HIP(hipMemsetAsync(sync_words, 0, 64 * sizeof(unsigned), s));
  LstmArgs b = a;
  static const int var = env_int("PF_LSTM_VAR", 2);          // PF_LSTM_VAR: 2 = ring form (default), 0 / 1 = arrival-counter form
  if (var == 2) {
    static const int delay = env_int("PF_LSTM_DELAY", 13);   // PF_LSTM_DELAY: x 2 x 64 clocks before the first (full) load of a step
    b.step = delay;
    // hstate [ndir][4][B][D]: slot 0 = h_{-1} = 0, slots 1 to 3 poison
    const size_t slot = (size_t)a.B * a.D * 2;
    PF_HIP(hipMemsetAsync(a.hstate, 0xFF, (size_t)a.ndir * 4 * slot, s));
    for (int d = 0; d < a.ndir; ++d) PF_HIP(hipMemsetAsync(reinterpret_cast<char*>(a.hstate) + (size_t)d * 4 * slot, 0, slot, s));
    hipLaunchKernelGGL(lstm_ring_kernel<false>, dim3(a.D / 8, a.ndir, tiles), dim3(320), 0, s, b, sync_words + 63);
  } else {
    PF_HIP(hipMemsetAsync(a.hstate, 0, (size_t)a.ndir * 2 * a.B * a.D * 2, s));
    b.step = var;
    hipLaunchKernelGGL(lstm_persistent_kernel, dim3(a.D / 8, a.ndir, tiles), dim3(256), 0, s, b, sync_words, sync_words + 63);
  }
  PF_HIP(hipGetLastError());
  return true;
}

// The ring form with (hi, lo') pair operands (math_mode 3): a.whh = [ndir][4D][2D] pair rows (launch_split_x3), a.hstate has room for
// [ndir][4][B][2D] f16 and is initialised here (slot 0 zero, slots 1-3 poison).  false = not applicable (as above).
bool launch_lstm_persistent_x3(hipStream_t s, const LstmArgs& a, unsigned* sync_words) {
  if (a.D != 512 || (a.ndir != 1 && a.ndir != 2)) return false;
  const int tiles = cdiv(a.B, 32);
  const int wgs = (a.D / 8) * a.ndir * tiles;
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  int cus = 0;
  PF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  if (wgs > cus || a.ndir * tiles > 60) return false;
  PF_HIP(hipMemsetAsync(sync_words, 0, 64 * sizeof(unsigned), s));
  LstmArgs b = a;
  static const int delay = env_int("PF_LSTM_DELAY", 13);
  b.step = delay;
  const size_t slot = (size_t)a.B * 2 * a.D * 2;
  PF_HIP(hipMemsetAsync(a.hstate, 0xFF, (size_t)a.ndir * 4 * slot, s));
  for (int d = 0; d < a.ndir; ++d) PF_HIP(hipMemsetAsync(reinterpret_cast<char*>(a.hstate) + (size_t)d * 4 * slot, 0, slot, s));
  hipLaunchKernelGGL(lstm_ring_kernel<true>, dim3(a.D / 8, a.ndir, tiles), dim3(320), 0, s, b, sync_words + 63);
  PF_HIP(hipGetLastError());
  return true;
}

void launch_lstm_step(hipStream_t s, const LstmArgs& a) {
  PF_CHECK(a.D % 512 == 0, PF_ERR_UNSUPPORTED, "lstm: hidden size must be a multiple of 512");
  PF_CHECK(a.ndir == 1 || a.ndir == 2, PF_ERR_INVALID_ARG, "lstm: ndir must be 1 or 2");
  hipLaunchKernelGGL(lstm_step_kernel, dim3(a.D / 8, a.ndir, cdiv(a.B, 32)), dim3(256), 0, s, a);
  PF_HIP(hipGetLastError());
}

// a2raw[row] = relu(sigmoid(dot(hout[row, 0:W], w) + b0) * smooth - noise); one wave per row
__global__ __launch_bounds__(256) void us_alpha_kernel(const float* __restrict__ hout, int64_t rows, int W,
                                                       const float* __restrict__ w, const float* __restrict__ b0,
                                                       float smooth, float noise, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* x = hout + row * W;
  float s = 0.f;
  for (int c = lane * 4; c < W; c += 256) {
    const float4 xv = *reinterpret_cast<const float4*>(x + c);
    const float4 wv = *reinterpret_cast<const float4*>(w + c);
    s += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) {
    const float sg = sigmoidf_(s + b0[0]);
    out[row] = fmaxf(sub_rn(mul_rn(sg, smooth), noise), 0.f);
  }
}

// per utterance: alphas *= token_num / sum(alphas) (sum carried in double), then cif_wo_hidden with
// threshold thr: integrate += alpha; peak[t] = integrate; integrate -= thr once it reaches thr.
__global__ __launch_bounds__(64) void us_peak_kernel(float* __restrict__ alphas, const int32_t* __restrict__ token_num,
                                                     int T3, float thr, float* __restrict__ peak) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float* a = alphas + (int64_t)b * T3;
  float* p = peak + (int64_t)b * T3;
  double s = 0.0;
  for (int t = lane; t < T3; t += 64) s += (double)a[t];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float sum = (float)s;
  const float ratio = (float)token_num[b] / sum;
  for (int t = lane; t < T3; t += 64) a[t] = mul_rn(a[t], ratio);
  __threadfence_block();
  __syncthreads();
  if (lane != 0) return;
  float integrate = 0.f;
  for (int t = 0; t < T3; ++t) {
    integrate = add_rn(integrate, a[t]);
    p[t] = integrate;
    if (integrate >= thr) integrate = sub_rn(integrate, thr);
  }
}

void launch_us_alpha(hipStream_t s, const float* hout, int64_t rows, int W, const float* w, const float* b0,
                     float smooth, float noise, float* out) {
  if (rows == 0) return;
  PF_CHECK(W % 4 == 0, PF_ERR_UNSUPPORTED, "us_alpha: width must be a multiple of 4");
  hipLaunchKernelGGL(us_alpha_kernel, dim3((unsigned)cdiv(rows, (int64_t)4)), dim3(256), 0, s, hout, rows, W, w, b0,
                     smooth, noise, out);
  PF_HIP(hipGetLastError());
}

void launch_us_peak(hipStream_t s, float* alphas, const int32_t* token_num, int B, int T3, float thr, float* peak) {
  if (B == 0) return;
  hipLaunchKernelGGL(us_peak_kernel, dim3(B), dim3(64), 0, s, alphas, token_num, T3, thr, peak);
  PF_HIP(hipGetLastError());
}

}  // namespace pf
