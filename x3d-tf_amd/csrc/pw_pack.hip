// x3d_pw_pack_weights: fp32 master weights -> bf16 LDS-image panels for the bf16 pointwise GEMMs
// (include/x3d_hip.h).  One launch for the whole model: blockIdx.x = item, blockIdx.y strides the panel.
#include "common.h"

static inline int panel_pitch(int cols) { return ((cols + 15) & ~15) + 8; }
static inline int panel_rows(int rows) { return (rows + 31) & ~31; }

extern "C" long long x3d_pw_panel_elems(int rows, int cols) {
  if (rows <= 0 || cols <= 0) return 0;
  return (long long)panel_rows(rows) * panel_pitch(cols);
}

__global__ __launch_bounds__(256) void pw_pack_kernel(const x3d_pw_pack_item* __restrict__ items) {
  const x3d_pw_pack_item it = items[blockIdx.x];
  const int Cout = it.Cout, Cin = it.Cin;
  {
    const int pitch = ((Cin + 15) & ~15) + 8, rows = (Cout + 31) & ~31;
    bf16* dst = (bf16*)it.fwd_panel;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < rows * pitch; i += gridDim.y * 256) {
      const int r = i / pitch, c = i - r * pitch;
      dst[i] = (bf16)((r < Cout && c < Cin) ? it.w[(long long)r * Cin + c] : 0.f);
    }
  }
  if (it.dgrad_panel) {
    const int pitch = ((Cout + 15) & ~15) + 8, rows = (Cin + 31) & ~31;
    bf16* dst = (bf16*)it.dgrad_panel;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < rows * pitch; i += gridDim.y * 256) {
      const int r = i / pitch, c = i - r * pitch;   // r = ci, c = co
      dst[i] = (bf16)((r < Cin && c < Cout) ? it.w[(long long)c * Cin + r] : 0.f);
    }
  }
}

extern "C" int x3d_pw_pack_weights(const x3d_pw_pack_item* items, int n_items, void* stream) {
  X3D_REQUIRE(items && n_items > 0, "pw_pack_weights: no items");
  hipLaunchKernelGGL(pw_pack_kernel, dim3(n_items, 16), dim3(256), 0, (hipStream_t)stream, items);
  X3D_LAUNCH_CHECK("pw_pack_weights");
  return X3D_OK;
}
