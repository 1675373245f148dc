// x3d_pw_pack_weights: fp32 master weights -> 16-bit (bf16 / f16) LDS-image panels for the 16-bit pointwise GEMMs
// (include/x3d_hip.h).  One launch for the whole model: blockIdx.x = item, blockIdx.y strides the panel.
#include "common.h"

static inline int panel_pitch(int cols) { return ((cols + 15) & ~15) + 8; }
static inline int panel_rows(int rows) { return (rows + 31) & ~31; }

extern "C" long long x3d_pw_panel_elems(int rows, int cols) {
  if (rows <= 0 || cols <= 0) return 0;
  // row-major LDS image + the MFMA-operand-tiled image the weights-streamed kernel reads (pw_gemm_ws.h)
  return (long long)panel_rows(rows) * panel_pitch(cols) + (long long)panel_rows(rows) * ((cols + 15) & ~15);
}

// tiled image: [row block of 32][k-step of 16][lane 0..63][8] -- lane (r, half) holds W[32*mi + r][16*ks + 8*half ..+7],
// i.e. exactly the A operand of one 32x32x16 MFMA as ONE contiguous 1 KB wave load (the row-major image makes that
// load touch 32 cache lines and use a quarter of each)
template <typename H>
__device__ inline void pack_tiled(H* dst, const float* w, int rows, int Kp, int R, int C, long long sr, long long sc,
                                  int start, int step) {
  const int ksteps = Kp >> 4;
  for (int i = start; i < rows * Kp; i += step) {
    const int e = i & 7, lane = (i >> 3) & 63, blk = i >> 9;
    const int mi = blk / ksteps, ks = blk - mi * ksteps;
    const int r = mi * 32 + (lane & 31), c = ks * 16 + 8 * (lane >> 5) + e;
    dst[i] = (H)((r < R && c < C) ? w[r * sr + c * sc] : 0.f);
  }
}

template <typename H>
__global__ __launch_bounds__(256) void pw_pack_kernel(const x3d_pw_pack_item* __restrict__ items) {
  const x3d_pw_pack_item it = items[blockIdx.x];
  const int Cout = it.Cout, Cin = it.Cin;
  {
    const int pitch = ((Cin + 15) & ~15) + 8, rows = (Cout + 31) & ~31;
    H* dst = (H*)it.fwd_panel;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < rows * pitch; i += gridDim.y * 256) {
      const int r = i / pitch, c = i - r * pitch;
      dst[i] = (H)((r < Cout && c < Cin) ? it.w[(long long)r * Cin + c] : 0.f);
    }
    pack_tiled(dst + (long long)rows * pitch, it.w, rows, pitch - 8, Cout, Cin, Cin, 1, blockIdx.y * 256 + threadIdx.x,
               gridDim.y * 256);
  }
  if (it.dgrad_panel) {
    const int pitch = ((Cout + 15) & ~15) + 8, rows = (Cin + 31) & ~31;
    H* dst = (H*)it.dgrad_panel;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < rows * pitch; i += gridDim.y * 256) {
      const int r = i / pitch, c = i - r * pitch;   // r = ci, c = co
      dst[i] = (H)((r < Cin && c < Cout) ? it.w[(long long)c * Cin + r] : 0.f);
    }
    pack_tiled(dst + (long long)rows * pitch, it.w, rows, pitch - 8, Cin, Cout, 1, Cin, blockIdx.y * 256 + threadIdx.x,
               gridDim.y * 256);
  }
}

extern "C" int x3d_pw_pack_weights(const x3d_pw_pack_item* items, int n_items, int dtype, void* stream) {
  X3D_REQUIRE(items && n_items > 0, "pw_pack_weights: no items");
  X3D_REQUIRE(x3d_is_half(dtype), "pw_pack_weights: panels exist for the 16-bit storage types only");
  if (dtype == X3D_F16) hipLaunchKernelGGL(pw_pack_kernel<f16>, dim3(n_items, 16), dim3(256), 0, (hipStream_t)stream, items);
  else hipLaunchKernelGGL(pw_pack_kernel<bf16>, dim3(n_items, 16), dim3(256), 0, (hipStream_t)stream, items);
  X3D_LAUNCH_CHECK("pw_pack_weights");
  return X3D_OK;
}
