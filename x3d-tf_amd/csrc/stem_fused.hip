// X3D stem as ONE kernel each way (reference model.py:202-210: conv_s -> conv_t with NOTHING in between, then BN + ReLU).
// The two-kernel form (stem.hip) writes the conv_s output s_raw to HBM, reads it back for conv_t, reads it again in the
// backward pass, and writes the conv_t input gradient ds only for the conv_s weight gradient to read it: on X3D-M B = 64
// that is 616 MB four times over.  Here neither tensor exists:
//
//   forward   x -> t_raw (+ BatchNorm sums)              reads x, writes t_raw                      0.92 GB instead of 2.16
//   backward  g, t_raw, x -> dW_s, dW_t                  reads g, t_raw, x, writes two small tables  1.54 GB instead of 3.39
//
// A workgroup (eight waves) owns FOUR 64-point segments of output rows and walks T.  Per plane t:
//   * the im2col tile of conv_s ([tap][point], 27 taps padded to 32, bf16) is built in LDS from 16-byte loads of the
//     channels-last input rows, as in stem.hip;
//   * s[t] = W_s x im2col on the matrix cores (v_mfma_f32_32x32x16, two k-steps; wave w = points 32w .. 32w+31), rounded
//     to the storage type exactly as the stored s_raw was, and handed through an LDS slab to the TIME-DOMAIN layout:
//     wave w owns channels w, w+8, w+16 (wave-uniform: every per-channel constant is an SGPR), lane = (segment, 4 points);
//   * forward: conv_t as a SCATTER into a five-slot ring of output accumulators (out[t+2-k] += w[k] s[t]; the products
//     and their order per output are those of the gather form in dwt_fwd_kernel, so t_raw is the same bits), the plane
//     t-2 leaves as 8-byte stores with its BatchNorm sums;
//   * backward: g and t_raw run TWO planes ahead of x, so that with dY[t-2 .. t+2] in registers (fp32, as in
//     dwt_bwd_kernel) both ds[t] = sum_k w[k] dY[t+2-k] and dW_t[k] += dY[t+2-k] . s[t] need nothing older; ds[t],
//     rounded to the storage type as the stored ds was, is the A operand of the conv_s weight-gradient tile
//     dW_s[co][tap] += ds[t] x im2col[t]^T (K = the 256 points, 32 per wave) -- the im2col tile already in LDS.
// The ring / window slots and the register buffers of the prefetched planes (distance 2) are compile-time: six slots,
// the time loop unrolled by six (one instantiation per phase).  One workgroup per CU; both LDS images exist twice (plane
// parity): two barriers per plane in the backward kernel, one in the forward kernel (whose three stages run skewed by a plane).
// LDS images: the im2col tile is read transposed (ds_read_b64_tr_b16, pitch = 64 banks mod 256 B: conflict-free) AND as
// k-contiguous 16-byte rows (dW_s); 16-byte chunks are XOR-swizzled with (tap >> 2) & 3, which is uniform over the lane
// group of a transposed read and distinct over the four row quads of a 16-byte read group.
#include <type_traits>

#include "common.h"

#ifndef SF_EXP
#define SF_EXP 0     // timing experiments (tools/ab_stem_parts.sh): 1 no time-domain math, 2 no commit, 4 no conv_s, 8 no barrier, 16 no MFMA in conv_s, 32 no slab write
#endif

namespace {

constexpr int SF_THREADS = 512;
constexpr int SF_SEGS = 4;          // 64-point segments per workgroup step
constexpr int SF_PTS = SF_SEGS * 64;
constexpr int SF_KT = 5;
constexpr int SF_LP = SF_PTS + 32;  // im2col pitch (elements): 576 B = 64 mod 256
constexpr int SF_AP = SF_PTS + 8;   // ds pitch: 528 B, 16-byte rows conflict-free
constexpr int SF_SP = SF_PTS + 4;   // s slab pitch: 520 B
constexpr int SF_OOB = 0x7fffff00;  // buffer offset past every tensor (host check): loads return 0, stores are dropped

typedef __attribute__((ext_vector_type(2))) unsigned int sf_u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int sf_u32x4;
typedef __attribute__((ext_vector_type(4))) short sf_s16x4;
typedef __attribute__((ext_vector_type(8))) short sf_s16x8;
typedef sf_s16x4 __attribute__((address_space(3))) * sf_lds_s16x4_ptr;

__device__ __forceinline__ int bs_at(int row, int col) { return row * SF_LP + (col ^ (((row >> 2) & 3) << 3)); }

template <typename HT> __device__ __forceinline__ float sf_get(const sf_u32x2& w, int e) {
  const unsigned int h = (w[e >> 1] >> (16 * (e & 1))) & 0xffffu;
  if constexpr (__is_same(HT, bf16)) return __uint_as_float(h << 16);
  else return (float)__builtin_bit_cast(HT, (unsigned short)h);
}
template <typename HT> __device__ __forceinline__ sf_u32x2 sf_pack(const float (&v)[4]) {
  typename HV<HT>::x4 h;
#pragma unroll
  for (int e = 0; e < 4; e++) h[e] = (HT)v[e];
  return __builtin_bit_cast(sf_u32x2, h);
}

// x staging, channels-last [N][T][H][W][3]: thread (q = segment, kh, v) owns the 8 input pixels 2 wo0 + 8 v .. of input row
// 2 ho + kh - 1 (48 contiguous bytes) and commits its three channels to the im2col tile: even pixels -> tap kw = 1, odd ->
// kw = 2 and, one point later, kw = 0; the pixel left of the segment for point 0 of kw = 0 (stem.hip, StemNhwcStage).
template <typename HT>
struct SfXBuf {                 // one plane of a thread's input vector: 8 pixels x 3 channels, and the 8 bytes in front of it
  sf_u32x4 r[3];                // (their last 6 are the pixel left of the segment)
  sf_u32x2 l;
};
// Every load is an UNCONDITIONAL bounds-checked buffer instruction (no vector / plane past T / pixel left of the row: offset
// SF_OOB, zeros, no traffic): vmcnt retires in order and conditional memory operations cannot be counted, so a single
// `if (ok) load` would put an s_waitcnt vmcnt(0) -- the newest stores and the prefetched planes included -- in front of
// every commit (stem.hip, dwt_bwd_kernel: same rule).
template <typename HT>
struct SfStage {
  typedef typename HV<HT>::x8 hx8; typedef typename HV<HT>::x4 hx4;
  int q, kh, v;
  bool on;
  int boff, loff;       // byte offsets of the thread's vector / of the 8 bytes in front of it at t = 0 (SF_OOB: none)
  __device__ __forceinline__ void roles(int tid) {
    q = tid / 48;
    const int rem = tid - q * 48;
    kh = rem >> 4; v = rem & 15;
    on = tid < 192;
  }
  __device__ __forceinline__ void group(int seg0, int seg_end, int nws, int Ho, int Tn, int H, int W) {
    boff = SF_OOB; loff = SF_OOB;
    const int seg = seg0 + q;
    if (!on || seg >= seg_end) return;
    const int ws = seg % nws;
    const int tmp = seg / nws;
    const int ho = tmp % Ho, n = tmp / Ho;
    const int wo0 = ws * 64;
    const int hi = 2 * ho + kh - 1;
    if (hi >= 0 && hi < H && 2 * wo0 + 8 * v < W) {
      boff = ((((n * Tn) * H + hi) * W + 2 * wo0 + 8 * v) * 3) * 2;
      if (v == 0 && wo0 > 0) loff = boff - 8;
    }
  }
  // (the plane offset -- or SF_OOB for a plane past T -- is wave-uniform: the scalar offset operand.  SF_OOB + SF_OOB < 2^32: a
  // lane without a vector stays out of range whatever the scalar part is)
  __device__ __forceinline__ void issue(SfXBuf<HT>& b, __amdgpu_buffer_rsrc_t rx, int t, int Tn, int plane3_bytes) const {
    const int so = t < Tn ? t * plane3_bytes : SF_OOB;
    b.r[0] = __builtin_amdgcn_raw_buffer_load_b128(rx, boff, so, 0);
    b.r[1] = __builtin_amdgcn_raw_buffer_load_b128(rx, boff + 16, so, 0);
    b.r[2] = __builtin_amdgcn_raw_buffer_load_b128(rx, boff + 32, so, 0);
    b.l = __builtin_amdgcn_raw_buffer_load_b64(rx, loff, so, 0);
  }
  static __device__ __forceinline__ HT at(const SfXBuf<HT>& b, int i) {
    const unsigned int wd = b.r[i >> 3][(i & 7) >> 1];
    return __builtin_bit_cast(HT, (unsigned short)(wd >> (16 * (i & 1))));
  }
  __device__ __forceinline__ void commit(const SfXBuf<HT>& b, HT* Bs) const {
    if (!on) return;
    const int col = q * 64 + 4 * v;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int tap1 = c * 9 + kh * 3 + 1;
      hx4 ev, od;
#pragma unroll
      for (int e = 0; e < 4; e++) { ev[e] = at(b, 3 * (2 * e) + c); od[e] = at(b, 3 * (2 * e + 1) + c); }
      *(hx4*)&Bs[bs_at(tap1, col)] = ev;          // kw = 1: wi = 2wo
      *(hx4*)&Bs[bs_at(tap1 + 1, col)] = od;      // kw = 2: wi = 2wo + 1
#pragma unroll
      for (int e = 0; e < 4; e++)                 // kw = 0: wi = 2wo - 1  (one point later)
        if (4 * v + 1 + e < 64) Bs[bs_at(tap1 - 1, col + 1 + e)] = od[e];
      if (v == 0) {                               // the pixel left of the segment: elements 1 + c of the four in front
        const unsigned int wd = b.l[(1 + c) >> 1];
        Bs[bs_at(tap1 - 1, col)] = __builtin_bit_cast(HT, (unsigned short)(wd >> (16 * ((1 + c) & 1))));
      }
    }
  }
};

// s[t] for the 32 points of wave w: D[point][co] = im2col^T x W_s^T (operands swapped against stem_s_fwd_bf16_kernel, same
// products and the same two k-steps per element), so a lane holds channel r = lane & 31 and FOUR CONSECUTIVE points per
// accumulator quad: rounded and written to the slab [co][point] as 8-byte pieces.
template <typename HT>
__device__ __forceinline__ void sf_conv_s(const HT* Bs, HT* Ss, const typename HV<HT>::x8 (&wfrag)[2], int w, int lane) {
  typedef typename HV<HT>::x8 hx8; typedef typename HV<HT>::x4 hx4;
  const int r = lane & 31, half = lane >> 5;
  const int g16 = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
  const int tr_row = 8 * (g16 >> 1) + q4;
  const int tr_col = 32 * w + 16 * (g16 & 1) + 4 * pp;
  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 16; j++) acc[j] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    const sf_s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sf_lds_s16x4_ptr)(&Bs[bs_at(ks * 16 + tr_row, tr_col)]));
    const sf_s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sf_lds_s16x4_ptr)(&Bs[bs_at(ks * 16 + tr_row + 4, tr_col)]));
    const sf_s16x8 bs = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#if !(SF_EXP & 16)
    acc = mfma16<HT>(__builtin_bit_cast(hx8, bs), wfrag[ks], acc);
#else
    for (int e = 0; e < 8; e++) acc[e + 8 * ks] += (float)__builtin_bit_cast(hx8, bs)[e] * (float)wfrag[ks][e];   // (timing experiment: no MFMA)
#endif
  }
#if (SF_EXP & 32)
  if (acc[0] != 12345.678f) return;      // (timing experiment: the slab is not written)
#endif
#pragma unroll
  for (int g = 0; g < 4; g++) {
    hx4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] = (HT)acc[4 * g + e];
    *(hx4*)&Ss[r * SF_SP + 32 * w + 8 * g + 4 * half] = o;
  }
}

// conv_s weights as the B operand of sf_conv_s: column co = lane & 31, taps 16 ks + 8 half .. + 7 (zero beyond Cout / 27)
template <typename HT>
__device__ __forceinline__ void sf_wfrag(const float* __restrict__ ws, int Cout, int lane, typename HV<HT>::x8 (&wfrag)[2]) {
  const int r = lane & 31, half = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 2; ks++)
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int tap = ks * 16 + 8 * half + e;
      wfrag[ks][e] = (HT)((r < Cout && tap < 27) ? ws[r * 27 + tap] : 0.f);
    }
}

// time-domain roles of a lane for one group of four segments: byte offset of its vector in channel 0 at t = 0 (SF_OOB: no such
// vector -- segment past the end, columns past Wo); channel `row` is row * Tn * plane_bytes further (wave-uniform)
__device__ __forceinline__ int sf_lane_off(int seg0, int seg_end, int lane, int nws, int Ho, int Wo, int Cout, int Tn) {
  const int q = lane >> 4, v4 = lane & 15;
  const int seg = seg0 + q;
  if (seg >= seg_end) return SF_OOB;
  const int ws = seg % nws;
  const int tmp = seg / nws;
  const int ho = tmp % Ho, n = tmp / Ho;
  const int wo = ws * 64 + 4 * v4;
  if (wo >= Wo) return SF_OOB;
  return (((n * Cout * Tn) * Ho + ho) * Wo + wo) * 2;
}

template <int PH> using Phase = std::integral_constant<int, PH>;
constexpr int SF_RING = 6;          // ring / window slots: five live planes + one, so that the slot and the prefetch buffer of a plane are
                                    // both compile-time inside a time loop unrolled by six (prefetch distance 2 divides 6, 5 would need 10)

// ------------------------------------------------------------------------------------------------
// forward.  ONE barrier per plane: tick k commits the im2col tile of plane k, runs conv_s of plane k - 1 (tile committed one
// tick earlier) and the time-domain step of plane k - 2 (slab written one tick earlier) -- three independent instruction
// streams per wave between two barriers; both LDS images exist twice (plane parity).
// INFER: y = out_act(os * conv + ot), the stem's BatchNorm from the moving statistics + ReLU (no statistics).
// ------------------------------------------------------------------------------------------------
template <typename HT, int NR, bool INFER>
__global__ __launch_bounds__(SF_THREADS) void stem_fwd_fused_kernel(const HT* __restrict__ x, const float* __restrict__ ws,
                                                                    const float* __restrict__ wt, HT* __restrict__ y,
                                                                    double* stats, const float* __restrict__ oss, int oact,
                                                                    int Cout, int Tn, int H, int W, int Ho, int Wo, int nws,
                                                                    int total_segs, int ngroups, int groups_per_block) {
  typedef typename HV<HT>::x8 hx8; typedef typename HV<HT>::x4 hx4;
  __shared__ __attribute__((aligned(16))) HT Bs2[2][32 * SF_LP];
  __shared__ __attribute__((aligned(16))) HT Ss2[2][32 * SF_SP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * 32 * SF_LP / 8; i += SF_THREADS) {
    hx8 z;
#pragma unroll
    for (int e = 0; e < 8; e++) z[e] = (HT)0.f;
    ((hx8*)&Bs2[0][0])[i] = z;
  }
  hx8 wfrag[2];
  sf_wfrag<HT>(ws, Cout, lane, wfrag);
  float wk[NR][SF_KT], os[NR], ot[NR];
#pragma unroll
  for (int i = 0; i < NR; i++) {
    const int row = i * 8 + w;
    const bool ok = row < Cout;
#pragma unroll
    for (int k = 0; k < SF_KT; k++) wk[i][k] = ok ? wt[row * SF_KT + k] : 0.f;
    os[i] = (INFER && ok) ? oss[row * 2] : 1.f;
    ot[i] = (INFER && ok) ? oss[row * 2 + 1] : 0.f;
  }
  const float olo = (INFER && oact == X3D_ACT_RELU) ? 0.f : -__builtin_inff();
  const int plane3 = H * W * 3 * 2;           // bytes of one input plane
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((HT*)x, 0, (int)((long long)(total_segs / (Ho * nws)) * Tn * plane3), 0x00020000);
  const int plane_bytes = Ho * Wo * 2;
  const long long ybytes = (long long)(total_segs / (Ho * nws)) * Cout * Tn * plane_bytes;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)ybytes, 0x00020000);
  SfStage<HT> st;
  st.roles(tid);
  __syncthreads();
  const int g_begin = blockIdx.x * groups_per_block, g_end = min(g_begin + groups_per_block, ngroups);
  for (int gi = g_begin; gi < g_end; gi++) {
    const int seg0 = gi * SF_SEGS, seg_end = min(seg0 + SF_SEGS, total_segs);
    st.group(seg0, seg_end, nws, Ho, Tn, H, W);
    const int off0 = sf_lane_off(seg0, seg_end, lane, nws, Ho, Wo, Cout, Tn);
    float ring[SF_RING][NR][4], red[NR][2];
#pragma unroll
    for (int s = 0; s < SF_RING; s++)
#pragma unroll
      for (int i = 0; i < NR; i++)
#pragma unroll
        for (int e = 0; e < 4; e++) ring[s][i][e] = 0.f;
#pragma unroll
    for (int i = 0; i < NR; i++) red[i][0] = red[i][1] = 0.f;
    SfXBuf<HT> xb[2];
    st.issue(xb[0], rx, 0, Tn, plane3);
    st.issue(xb[1], rx, 1, Tn, plane3);

    // tick k (PH = k % 6).  Planes past T are zeros (bounds-checked loads): their terms vanish, their stores are dropped.
    auto tick = [&](auto ph, int k, auto do_conv, auto do_time) {
      constexpr int PH = decltype(ph)::value;
#if !(SF_EXP & 2)
      st.commit(xb[PH & 1], Bs2[PH & 1]);           // plane k  (last readers of this image: conv_s of plane k - 2, last tick)
#else
      if (xb[PH & 1].r[0][0] == 0x12345678u) Bs2[0][tid] = (HT)1.f;      // (timing experiment: the loads stay, the commit goes)
#endif
      st.issue(xb[PH & 1], rx, k + 2, Tn, plane3);
#if !(SF_EXP & 4)
      if constexpr (decltype(do_conv)::value)       // plane k - 1
        sf_conv_s<HT>(Bs2[(PH + 1) & 1], Ss2[(PH + 1) & 1], wfrag, w, lane);
#endif
      if constexpr (decltype(do_time)::value) {     // plane j = k - 2: s[j] into the ring, plane j - 2 out
        constexpr int PJ = (PH + 4) % SF_RING;
        const HT* Ss = Ss2[PH & 1];
        const int tau = k - 4;
        const bool t_ok = tau >= 0 && tau < Tn;
#pragma unroll
        for (int i = 0; i < NR; i++) {
          const hx4 sv = *(const hx4*)&Ss[(i * 8 + w) * SF_SP + 4 * lane];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float s = (float)sv[e];
            // out[j + 2 - kk] += w[kk] s[j]; kk = 0 opens the slot of plane j + 2  (products and order of dwt_fwd_kernel)
#if !(SF_EXP & 1)
            ring[(PJ + 2) % SF_RING][i][e] = wk[i][0] * s;
#pragma unroll
            for (int kk = 1; kk < SF_KT; kk++)
              ring[(PJ + 2 - kk + SF_RING) % SF_RING][i][e] = __builtin_fmaf(wk[i][kk], s, ring[(PJ + 2 - kk + SF_RING) % SF_RING][i][e]);
#else
            ring[(PJ + SF_RING - 2) % SF_RING][i][e] = s;
#endif
          }
          // plane j - 2 is complete: its last term was s[j]
          // (live: a column past Wo holds the row's last pixel in its kw = 0 taps -- stored nowhere, not a sum term)
          const bool live = off0 != SF_OOB && i * 8 + w < Cout && t_ok;
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            o[e] = ring[(PJ + SF_RING - 2) % SF_RING][i][e];
            // (the fp32 sum as dwt_fwd_kernel rounds it: without this the last fma and the fp16 conversion fuse into one
            // v_fma_mixlo_f16 -- a single rounding, 1 ulp away from the two-kernel path on ~1e-4 of the elements)
            asm volatile("" : "+v"(o[e]));
            if constexpr (INFER) o[e] = fmaxf(os[i] * o[e] + ot[i], olo);
          }
          hx4 h;
#pragma unroll
          for (int e = 0; e < 4; e++) h[e] = (HT)o[e];
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(sf_u32x2, h), ry, off0,
                                                (i * 8 + w < Cout && t_ok) ? ((i * 8 + w) * Tn + tau) * plane_bytes : SF_OOB, 0);
          if constexpr (!INFER) {   // BatchNorm sums of the tensor as stored
            const float v0 = (float)h[0], v1 = (float)h[1], v2 = (float)h[2], v3 = (float)h[3];
            const float s1 = (v0 + v1) + (v2 + v3);
            const float s2 = __builtin_fmaf(v3, v3, __builtin_fmaf(v2, v2, __builtin_fmaf(v1, v1, v0 * v0)));
            red[i][0] += live ? s1 : 0.f;
            red[i][1] += live ? s2 : 0.f;
          }
        }
      }
#if !(SF_EXP & 8)
      __syncthreads();
#endif
    };
    constexpr std::true_type yes{};
    constexpr std::false_type no{};
    tick(Phase<0>{}, 0, no, no);
    tick(Phase<1>{}, 1, yes, no);
    // the last plane leaves at tick Tn + 3: Tn + 2 more ticks, rounded up to whole turns of the ring -- NO guard between the
    // phases (a guarded chain compiles to a dispatch block every phase returns to, where the wait-count pass merges all their
    // states and drains vmcnt in front of every commit)
    const int turns = (Tn + 2 + SF_RING - 1) / SF_RING;
    for (int r = 0, k0 = 2; r < turns; r++, k0 += SF_RING) {
      tick(Phase<2>{}, k0, yes, yes);
      tick(Phase<3>{}, k0 + 1, yes, yes);
      tick(Phase<4>{}, k0 + 2, yes, yes);
      tick(Phase<5>{}, k0 + 3, yes, yes);
      tick(Phase<0>{}, k0 + 4, yes, yes);
      tick(Phase<1>{}, k0 + 5, yes, yes);
    }
    if constexpr (!INFER) {
      if (stats) {
        double* sp = stats_replica(stats, Cout, (unsigned)gi);
#pragma unroll
        for (int i = 0; i < NR; i++) {
          const float s0 = wave_sum(red[i][0]), s1 = wave_sum(red[i][1]);
          const int row = i * 8 + w;
          if (lane == 0 && row < Cout) {
            atomic_add_d(&sp[row * 2], (double)s0);
            atomic_add_d(&sp[row * 2 + 1], (double)s1);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward: dY = A*[ms*yraw + mt > 0]*g + B*yraw + C ; dW_t[c][k] += dY[t] . s[t + k - 2] ; ds[t] = sum_k w_t[k] dY[t + 2 - k] ;
// dW_s[co][tap] += ds[t] . im2col(x)[t]
// ------------------------------------------------------------------------------------------------
template <typename HT, int NR>
__global__ __launch_bounds__(SF_THREADS) void stem_bwd_fused_kernel(const HT* __restrict__ g, const HT* __restrict__ yraw,
                                                                    const float* __restrict__ rss, const float* __restrict__ coef,
                                                                    const HT* __restrict__ x, const float* __restrict__ ws,
                                                                    const float* __restrict__ wt, float* dws, float* dwt,
                                                                    int Cout, int Tn, int H, int W, int Ho, int Wo, int nws,
                                                                    int total_segs, int ngroups, int groups_per_block) {
  typedef typename HV<HT>::x8 hx8; typedef typename HV<HT>::x4 hx4;
  // per plane parity: im2col tile | ds tile | s slab.  The first two of parity 0 end as the 8 x 32 x 32 fp32 reduction buffer of dW_s
  constexpr int BS_BYTES = 32 * SF_LP * 2, AS_BYTES = 32 * SF_AP * 2, SS_BYTES = 32 * SF_SP * 2;
  constexpr int IMG_BYTES = BS_BYTES + AS_BYTES + SS_BYTES;
  static_assert(BS_BYTES + AS_BYTES >= 8 * 32 * 32 * 4, "dW_s reduction buffer");
  static_assert((BS_BYTES % 16) == 0 && (AS_BYTES % 16) == 0 && (SS_BYTES % 16) == 0, "tile alignment");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * IMG_BYTES + 2 * 64 * 16];
  hx8* Wf = (hx8*)(smem + 2 * IMG_BYTES);          // conv_s weight fragments [ks][lane]: read per plane, not held
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, half = lane >> 5;
  for (int i = tid; i < 2 * IMG_BYTES / 16; i += SF_THREADS) {
    hx8 z;
#pragma unroll
    for (int e = 0; e < 8; e++) z[e] = (HT)0.f;
    ((hx8*)smem)[i] = z;
  }
  if (w == 0) {
    hx8 wfrag[2];
    sf_wfrag<HT>(ws, Cout, lane, wfrag);
    Wf[lane] = wfrag[0]; Wf[64 + lane] = wfrag[1];
  }
  float wk[NR][SF_KT], cA[NR], cB[NR], cC[NR], ms[NR], mt[NR];
#pragma unroll
  for (int i = 0; i < NR; i++) {
    const int row = i * 8 + w;
    const bool ok = row < Cout;
#pragma unroll
    for (int k = 0; k < SF_KT; k++) wk[i][k] = ok ? wt[row * SF_KT + k] : 0.f;
    cA[i] = ok ? coef[row * 4] : 0.f; cB[i] = ok ? coef[row * 4 + 1] : 0.f; cC[i] = ok ? coef[row * 4 + 2] : 0.f;
    // rss: g is the unmasked gradient, the ReLU mask [ms*yraw + mt > 0] is applied here (ms = 0, mt = 1: always on)
    ms[i] = (rss && ok) ? rss[row * 2] : 0.f;
    mt[i] = (rss && ok) ? rss[row * 2 + 1] : 1.f;
  }
  const int plane3 = H * W * 3 * 2;           // bytes of one input plane
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((HT*)x, 0, (int)((long long)(total_segs / (Ho * nws)) * Tn * plane3), 0x00020000);
  const int plane_bytes = Ho * Wo * 2;
  const long long ybytes = (long long)(total_segs / (Ho * nws)) * Cout * Tn * plane_bytes;
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((HT*)g, 0, (int)ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((HT*)yraw, 0, (int)ybytes, 0x00020000);
  SfStage<HT> st;
  st.roles(tid);
  f32x16 acc2;                       // dW_s[co][tap], this wave's 32 points of every step
#pragma unroll
  for (int j = 0; j < 16; j++) acc2[j] = 0.f;
  float dwk[NR][SF_KT];
#pragma unroll
  for (int i = 0; i < NR; i++)
#pragma unroll
    for (int k = 0; k < SF_KT; k++) dwk[i][k] = 0.f;
  __syncthreads();

  const int g_begin = blockIdx.x * groups_per_block, g_end = min(g_begin + groups_per_block, ngroups);
  for (int gi = g_begin; gi < g_end; gi++) {
    const int seg0 = gi * SF_SEGS, seg_end = min(seg0 + SF_SEGS, total_segs);
    st.group(seg0, seg_end, nws, Ho, Tn, H, W);
    const int off0 = sf_lane_off(seg0, seg_end, lane, nws, Ho, Wo, Cout, Tn);
    float dwin[SF_RING][NR][4];        // dwin[t % 6] = dY[t]
#pragma unroll
    for (int s = 0; s < SF_RING; s++)
#pragma unroll
      for (int i = 0; i < NR; i++)
#pragma unroll
        for (int e = 0; e < 4; e++) dwin[s][i][e] = 0.f;
    sf_u32x2 bg[2][NR], by[2][NR];
    // dY of plane t from the loaded g / yraw vectors of row i
    auto dy_of = [&](const sf_u32x2& gq, const sf_u32x2& yq, int i, int t, float (&d)[4]) {
      const bool in = off0 != SF_OOB && i * 8 + w < Cout && t < Tn;   // past T the loads returned zeros: dY must be 0 there, not C
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float yv = sf_get<HT>(yq, e);
        const float gv = (ms[i] * yv + mt[i] > 0.f) ? sf_get<HT>(gq, e) : 0.f;
        d[e] = in ? cA[i] * gv + cB[i] * yv + cC[i] : 0.f;
      }
    };
    auto load_plane = [&](int b, int i, int t) {
      const int so = (i * 8 + w < Cout && t < Tn) ? ((i * 8 + w) * Tn + t) * plane_bytes : SF_OOB;   // wave-uniform: scalar operand
      bg[b][i] = __builtin_amdgcn_raw_buffer_load_b64(rg, off0, so, 0);
      by[b][i] = __builtin_amdgcn_raw_buffer_load_b64(ry, off0, so, 0);
    };
    // g and yraw run two planes ahead of x: dY[0], dY[1] enter the window here, planes 2 and 3 are in flight when the loop starts
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int i = 0; i < NR; i++) load_plane(b, i, b);
    SfXBuf<HT> xb;                // (one register set, the next plane issued right behind the commit: the register budget)
    st.issue(xb, rx, 0, Tn, plane3);
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int i = 0; i < NR; i++) {
        dy_of(bg[b][i], by[b][i], i, b, dwin[b][i]);
        load_plane(b, i, b + 2);
      }

    // step u: dY[u + 2] enters the window, then the x plane u
    auto iter = [&](auto ph, int u) {
      constexpr int PH = decltype(ph)::value;      // u % 6
      constexpr int SV = (PH + 2) % SF_RING;       // slot of dY[u + 2]
#pragma unroll
      for (int i = 0; i < NR; i++) {
        dy_of(bg[PH & 1][i], by[PH & 1][i], i, u + 2, dwin[SV][i]);
        load_plane(PH & 1, i, u + 4);
      }
      unsigned char* img = smem + (PH & 1) * IMG_BYTES;
      HT* Bs = (HT*)img;
      HT* As = (HT*)(img + BS_BYTES);
      HT* Ss = (HT*)(img + BS_BYTES + AS_BYTES);
      // ds[u] = sum_k w[k] dY[u + 2 - k] = sum_k w[k] dwin[(SV - k) mod 6], in the order of dwt_bwd_kernel
#pragma unroll
      for (int i = 0; i < NR; i++) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          float acc = 0.f;
#pragma unroll
          for (int k = 0; k < SF_KT; k++) acc += wk[i][k] * dwin[(SV - k + SF_RING) % SF_RING][i][e];
          o[e] = acc;
        }
        *(sf_u32x2*)&As[(i * 8 + w) * SF_AP + 4 * lane] = sf_pack<HT>(o);
      }
      st.commit(xb, Bs);
      st.issue(xb, rx, u + 1, Tn, plane3);
      __syncthreads();
      {
        const hx8 wfrag[2] = {Wf[lane], Wf[64 + lane]};
        sf_conv_s<HT>(Bs, Ss, wfrag, w, lane);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        const int col = 32 * w + 16 * ks + 8 * half;
        const hx8 af = *(const hx8*)&As[r * SF_AP + col];
        const hx8 bf = *(const hx8*)&Bs[bs_at(r, col)];
        acc2 = mfma16<HT>(af, bf, acc2);
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NR; i++) {
        const hx4 sv = *(const hx4*)&Ss[(i * 8 + w) * SF_SP + 4 * lane];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float s = (float)sv[e];
          // dW_t[k] += dY[t] s[t + k - 2] with t + k - 2 = u: t = u + 2 - k
#pragma unroll
          for (int k = 0; k < SF_KT; k++) dwk[i][k] = __builtin_fmaf(dwin[(SV - k + SF_RING) % SF_RING][i][e], s, dwk[i][k]);
        }
      }
    };
    // Tn steps rounded up to whole turns of the window (planes past T are zeros: no traffic, no contribution), no guard between
    // the phases (see the forward kernel)
    const int steps = (Tn + SF_RING - 1) / SF_RING * SF_RING;
    for (int u0 = 0; u0 < steps; u0 += SF_RING) {
      iter(Phase<0>{}, u0);
      iter(Phase<1>{}, u0 + 1);
      iter(Phase<2>{}, u0 + 2);
      iter(Phase<3>{}, u0 + 3);
      iter(Phase<4>{}, u0 + 4);
      iter(Phase<5>{}, u0 + 5);
    }
    __syncthreads();     // the next group's first planes reuse the images of this group's last two
  }
  // dW_t: every lane of the wave shares its channels -- wave sums, lane k adds tap k
#pragma unroll
  for (int i = 0; i < NR; i++) {
    float mine = 0.f;
#pragma unroll
    for (int k = 0; k < SF_KT; k++) {
      const float s = wave_sum(dwk[i][k]);
      mine = lane == k ? s : mine;
    }
    const int row = i * 8 + w;
    if (lane < SF_KT && row < Cout) atomicAdd(&dwt[row * SF_KT + lane], mine);
  }
  // dW_s: the eight waves' partial tiles meet in LDS, one atomic per (co, tap) and workgroup
  float* Ds = (float*)smem;     // [8][32][32]
#pragma unroll
  for (int j = 0; j < 16; j++) Ds[(w * 32 + (j & 3) + 8 * (j >> 2) + 4 * half) * 32 + r] = acc2[j];
  __syncthreads();
  for (int i = tid; i < 32 * 32; i += SF_THREADS) {
    const int co = i >> 5, tap = i & 31;
    if (co < Cout && tap < 27) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 8; k++) s += Ds[k * 1024 + i];
      atomicAdd(&dws[co * 27 + tap], s);
    }
  }
}

int sf_slots() {
  return x3d_device_cus();      // one workgroup of eight waves per CU (register count): a persistent grid of that many
}

bool sf_shape_ok(int Cin, int Cout, int KT, int N, int T, int H, int W, int dtype, int x_layout) {
  if (!x3d_is_half(dtype) || x_layout != X3D_LAYOUT_NTHWC || Cin != 3 || KT != SF_KT || Cout < 1 || Cout > 32) return false;
  if (N < 1 || T < 1 || H < 1 || W < 8 || (W % 8) != 0) return false;
  const long long Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  if ((long long)N * Cout * T * Ho * Wo * 2 >= (long long)SF_OOB) return false;     // 32-bit buffer offsets
  if ((long long)N * T * H * W * 3 * 2 >= (long long)SF_OOB) return false;
  if ((long long)N * Ho * ((Wo + 63) / 64) + SF_SEGS >= (1ll << 31)) return false;
  return true;
}

}  // namespace

// bit 0: x3d_stem_fwd takes the shape; bit 1: x3d_stem_bwd takes it AND is the faster path.  With more than 24 channels (X3D-XL:
// 32, four rows per wave) the backward kernel's window no longer fits 256 registers and spills inside the time loop: 715 us
// against 493 us for x3d_dwt_bwd + x3d_stem_s_wgrad at 16 x 16 x 312^2 (profiles/r06_stem_bench.txt), so a training plan keeps
// the two-kernel path there; the forward kernel (254 us against 367 us) serves inference.
extern "C" int x3d_stem_fused_supported(int Cin, int Cout, int KT, int N, int T, int H, int W, int dtype, int x_layout) {
  if (!sf_shape_ok(Cin, Cout, KT, N, T, H, W, dtype, x_layout)) return 0;
  return Cout <= 24 ? 3 : 1;
}

extern "C" int x3d_stem_fwd(const void* x, const float* w_s, const float* w_t, void* y, double* stats,
                            const float* out_scale_shift, int out_act, int N, int Cin, int T, int H, int W, int Cout, int KT,
                            int dtype, int x_layout, void* stream) {
  X3D_REQUIRE(x && w_s && w_t && y, "stem_fwd: bad args");
  X3D_REQUIRE(sf_shape_ok(Cin, Cout, KT, N, T, H, W, dtype, x_layout),
              "stem_fwd: needs 16-bit storage, a channels-last batch with Cin = 3 and W %% 8 == 0, Cout <= 32, KT = 5 and an output "
              "under 2 GB (x3d_stem_fused_supported); run x3d_stem_s_fwd + x3d_dwt_fwd otherwise");
  X3D_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 8) == 0, "stem_fwd: x must be 16-byte and y 8-byte aligned");
  X3D_REQUIRE(!(out_scale_shift && stats), "stem_fwd: the inference epilogue (out_scale_shift) takes no statistics");
  X3D_REQUIRE(out_act == X3D_ACT_NONE || out_act == X3D_ACT_RELU, "stem_fwd: out_act must be none or ReLU");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, nws = ceil_div(Wo, 64);
  const long long total_segs = (long long)N * Ho * nws;
  const int ngroups = (int)ceil_div_ll(total_segs, SF_SEGS);
  const int gpb = (int)ceil_div_ll(ngroups, sf_slots());
  const int grid = ceil_div(ngroups, gpb);
  hipStream_t st = (hipStream_t)stream;
#define SF_FWD(TT, NR_, INF_) hipLaunchKernelGGL((stem_fwd_fused_kernel<TT, NR_, INF_>), dim3(grid), dim3(SF_THREADS), 0, st, (const TT*)x, w_s, \
                                                 w_t, (TT*)y, stats, out_scale_shift, out_act, Cout, T, H, W, Ho, Wo, nws,              \
                                                 (int)total_segs, ngroups, gpb)
#define SF_FWD_NR(TT, INF_) do { if (Cout <= 24) SF_FWD(TT, 3, INF_); else SF_FWD(TT, 4, INF_); } while (0)
  if (dtype == X3D_F16) { if (out_scale_shift) SF_FWD_NR(f16, true); else SF_FWD_NR(f16, false); }
  else { if (out_scale_shift) SF_FWD_NR(bf16, true); else SF_FWD_NR(bf16, false); }
#undef SF_FWD_NR
#undef SF_FWD
  X3D_LAUNCH_CHECK("stem_fwd");
  return X3D_OK;
}

extern "C" int x3d_stem_bwd(const void* g, const void* yraw, const float* relu_scale_shift, const float* coef, const void* x,
                            const float* w_s, const float* w_t, float* dw_s, float* dw_t, int N, int Cin, int T, int H, int W,
                            int Cout, int KT, int dtype, int x_layout, void* stream) {
  X3D_REQUIRE(g && yraw && coef && x && w_s && w_t && dw_s && dw_t, "stem_bwd: bad args");
  X3D_REQUIRE(sf_shape_ok(Cin, Cout, KT, N, T, H, W, dtype, x_layout),
              "stem_bwd: needs 16-bit storage, a channels-last batch with Cin = 3 and W %% 8 == 0, Cout <= 32, KT = 5 and an output "
              "under 2 GB (x3d_stem_fused_supported); run x3d_dwt_bwd + x3d_stem_s_wgrad otherwise");
  X3D_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)g % 8) == 0 && ((uintptr_t)yraw % 8) == 0,
              "stem_bwd: x must be 16-byte, g and yraw 8-byte aligned");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, nws = ceil_div(Wo, 64);
  const long long total_segs = (long long)N * Ho * nws;
  const int ngroups = (int)ceil_div_ll(total_segs, SF_SEGS);
  const int gpb = (int)ceil_div_ll(ngroups, sf_slots());
  const int grid = ceil_div(ngroups, gpb);
  hipStream_t st = (hipStream_t)stream;
#define SF_BWD(TT, NR_) hipLaunchKernelGGL((stem_bwd_fused_kernel<TT, NR_>), dim3(grid), dim3(SF_THREADS), 0, st, (const TT*)g, (const TT*)yraw, \
                                           relu_scale_shift, coef, (const TT*)x, w_s, w_t, dw_s, dw_t, Cout, T, H, W, Ho, Wo, nws,        \
                                           (int)total_segs, ngroups, gpb)
  if (dtype == X3D_F16) { if (Cout <= 24) SF_BWD(f16, 3); else SF_BWD(f16, 4); }
  else { if (Cout <= 24) SF_BWD(bf16, 3); else SF_BWD(bf16, 4); }
#undef SF_BWD
  X3D_LAUNCH_CHECK("stem_bwd");
  return X3D_OK;
}
