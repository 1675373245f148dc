"""Shapes of the `-m gpu` oracle-parity cases, as data.

tests/test_kernels_gpu.py and tests/test_model_gpu.py parametrise over these lists; tests/test_dispatch_coverage.py
(CPU) asks the library's dry-run dispatch which kernel instantiation every case runs and asserts that every
instantiation the BASELINE configurations launch at full size appears among them.  A case belongs here only if its GPU
test compares the kernel with the CPU oracle / an fp64 restatement (not with another device kernel).
"""
import torch

F32, BF16, F16 = torch.float32, torch.bfloat16, torch.float16
DTYPES = [F32, BF16, F16]
HALF_DTYPES = [BF16, F16]

# ---- x3d_pw_fwd: N, Cin, Cout, T, H, W, stride, prologue ------------------------------------------------------------
PW_FWD = [
    (2, 48, 108, 2, 5, 13, 1, None), (1, 108, 48, 1, 3, 43, 1, "swish"),   # P = 130 / 129: a row's partial 16-byte vector is vector 0 of a 128-point tile (fp32 pipelined kernel: early start in front of the tile)
    (2, 24, 54, 4, 12, 12, 1, None),       # bottleneck a (stage 2 widths)
    (2, 54, 24, 3, 10, 10, 1, "swish"),    # bottleneck c with BN_b + SE gate + swish folded
    (2, 24, 48, 4, 12, 12, 2, None),       # strided shortcut (valid, samples pixels 0,2,..)
    (1, 48, 108, 13, 5, 5, 1, "relu"),     # P = 325 (odd): scalar path
    (2, 96, 216, 2, 7, 7, 1, None),        # Cout > 128: two row blocks
    (1, 216, 96, 2, 7, 7, 1, "swish"),     # K chunking (does not fit LDS resident)
    (1, 24, 24, 2, 9, 11, 2, None),        # odd extents with stride 2
    (1, 200, 40, 1, 4, 8, 1, "swish"),     # widths off the 32-grid
    (1, 24, 48, 2, 56, 56, 2, None), (2, 48, 96, 2, 28, 28, 2, None), (2, 96, 192, 4, 14, 14, 2, None),  # gather groups 4 / 2 / 1
    (1, 24, 24, 2, 32, 32, 2, None),       # Wo % 8 == 0
    (2, 432, 192, 2, 8, 8, 1, "swish"), (2, 192, 432, 2, 8, 8, 1, None), (1, 440, 200, 1, 8, 5, 1, "relu"),  # weights-streamed / -stationary paths
    (2, 216, 96, 2, 14, 14, 1, "swish"), (2, 96, 216, 2, 14, 14, 1, None), (2, 200, 90, 1, 8, 8, 1, "relu"),  # stage-4 weights-stationary shapes (two workgroups per CU)
    (2, 96, 432, 2, 14, 14, 1, None),      # stage-5 block 0 `a` conv (7 waves x 2 row blocks x 6 k-steps)
    (1, 24, 24, 1, 156, 156, 2, None), (1, 24, 48, 1, 78, 78, 2, None), (1, 48, 96, 2, 39, 39, 2, None),   # X3D-L / XL shortcuts
    (1, 96, 192, 2, 20, 20, 2, None),
    (1, 96, 192, 8, 14, 14, 2, None), (1, 24, 48, 8, 78, 78, 2, None),   # odd Wo with P % 8 == 0: gather group 1 on the vector path
    (1, 24, 24, 8, 156, 156, 2, None),                                   # Wo = 78: gather group 2
    (1, 24, 54, 1, 32, 32, 1, None), (1, 108, 48, 1, 16, 16, 1, "swish"), (1, 48, 108, 1, 16, 16, 1, None),  # aligned stage-2/3 layers
    (1, 24, 108, 1, 16, 16, 1, None), (1, 48, 216, 1, 16, 16, 1, None),
    # X3D-S stages 4 / 5 (13 frames: rows of 1300 / 325 points, not multiples of 8): 16-bit storage takes the vector kernels
    # with ragged row ends (unaligned row starts, a row's last vector element by element)
    (2, 216, 96, 13, 10, 10, 1, "swish"), (2, 96, 216, 13, 10, 10, 1, None), (1, 432, 192, 13, 5, 5, 1, "swish"),
    (1, 192, 432, 13, 5, 5, 1, None), (1, 24, 54, 1, 3, 4, 1, None),   # (... and a row shorter than two vectors: P = 12)
    # strided shortcut with an ODD input width on the vector gather (groups of 4 / 2 / 1, the row's last group loaded early)
    (1, 24, 48, 2, 11, 23, 2, None), (1, 32, 32, 4, 9, 27, 2, None), (1, 24, 48, 8, 13, 13, 2, None),
    # ... whose rows are ONE gather group wide (7 -> 4, 3 -> 2: stage 5 of 112-pixel crops): the early load of the last
    # group would start before the row (before the tensor for its first row) -- these take the next smaller group
    (1, 96, 192, 8, 7, 7, 2, None), (2, 96, 192, 8, 3, 3, 2, None),
    # round 6: the shortcut convs as DENSE launches on the even-pixel copy of the block input (x3d_subsample2): the compact planes
    # of X3D-XS .. M (40 / 56, 28, 14, 7 wide) and X3D-L / XL (78, 39, 20, 10 wide; 32 -> 32)
    (1, 24, 24, 2, 40, 40, 1, None), (2, 24, 24, 1, 56, 56, 1, None), (1, 24, 48, 2, 28, 28, 1, None), (2, 48, 96, 2, 14, 14, 1, None),
    (2, 96, 192, 4, 7, 7, 1, None), (1, 24, 24, 2, 78, 78, 1, None), (1, 32, 32, 1, 78, 78, 1, None), (1, 24, 48, 2, 39, 39, 1, None),
    (1, 48, 96, 2, 20, 20, 1, None), (1, 96, 192, 2, 10, 10, 1, None),
]
# ... with the residual tail of the block below folded into the prologue (16-bit storage; prologue "tail": identity shortcut,
# "tail_conv": shortcut conv with its own BN): x = raw c output, in_add = shortcut, in_store = the block output y
PW_FWD_TAIL = [
    (2, 24, 54, 4, 12, 12, 1, "tail"), (1, 48, 108, 2, 16, 16, 1, "tail_conv"), (1, 24, 108, 1, 16, 16, 1, "tail"),   # stages 2 / 3
    (1, 48, 216, 1, 16, 16, 1, "tail"), (1, 32, 72, 1, 16, 16, 1, "tail"), (1, 72, 162, 1, 8, 8, 1, "tail_conv"),     # stage-4 block 0; X3D-XL
    (1, 48, 108, 13, 5, 5, 1, "tail"), (2, 96, 216, 13, 10, 10, 1, "tail_conv"), (1, 24, 54, 1, 3, 4, 1, "tail"),     # odd / ragged point counts
    (3, 40, 72, 2, 7, 8, 1, "tail_conv"),
    # stages 4 / 5: the weights-stationary kernel carries the fold (shapes 4, 5, 2 of pw_gemm_wst.h; with a weight panel), whole and
    # ragged rows (13 frames of 10 x 10 / 5 x 5), two samples, a partial last tile
    (2, 96, 216, 4, 14, 14, 1, "tail"), (1, 96, 432, 2, 14, 14, 1, "tail_conv"), (2, 192, 432, 4, 7, 7, 1, "tail"), (1, 192, 432, 8, 7, 7, 1, "tail_conv"),
    (1, 96, 216, 13, 10, 10, 1, "tail"), (2, 192, 432, 13, 5, 5, 1, "tail_conv"),
    # "tail1": no Add -- the stem's BatchNorm + ReLU folded into the first block's `a` conv (x = raw conv_t output, in_store = y0)
    (2, 24, 54, 2, 16, 16, 1, "tail1"), (1, 32, 72, 1, 16, 16, 1, "tail1"), (1, 24, 54, 1, 3, 4, 1, "tail1"), (2, 24, 54, 13, 5, 5, 1, "tail1"),
]
# X3D-XL widths (configs/kinetics/X3D_XL.yaml: width factor 2.9, bottleneck 2.25): 32/72, 72/162, 136/306, 280/630, conv5 630
PW_FWD_XL = [
    (1, 32, 72, 1, 16, 16, 1, None), (1, 72, 32, 1, 16, 16, 1, "swish"), (1, 32, 32, 1, 16, 16, 2, None),
    (1, 32, 162, 1, 16, 16, 1, None), (1, 162, 72, 1, 8, 8, 1, "swish"), (1, 72, 162, 1, 8, 8, 1, None), (1, 32, 72, 1, 16, 16, 2, None),
    (1, 72, 306, 1, 8, 8, 1, None), (1, 306, 136, 1, 8, 8, 1, "swish"), (1, 136, 306, 1, 8, 8, 1, None), (1, 72, 136, 1, 16, 16, 2, None),
    (1, 136, 630, 1, 8, 8, 1, None), (1, 630, 280, 1, 8, 8, 1, "swish"), (1, 280, 630, 1, 8, 8, 1, None), (1, 136, 280, 1, 16, 16, 2, None),
    (1, 32, 72, 8, 78, 78, 2, None), (1, 32, 32, 8, 156, 156, 2, None),
]

# ---- x3d_pw_fwd with the INFERENCE epilogue (folded BN + residual Add + ReLU on the accumulators): the stride-1 cases above
# that carry the `c` conv's prologue, with the shortcut alternating between the block input ("identity") and a raw
# shortcut-conv output with its own BN ("conv"); plus plain BN (+ ReLU) epilogues without a residual.
#   N, Cin, Cout, T, H, W, prologue, residual, out_act
PW_FWD_INFER = [(n, ci, co, t, h, w, pro, ("identity", "conv")[i % 2], "relu")
                for i, (n, ci, co, t, h, w, st, pro) in enumerate(PW_FWD + PW_FWD_XL) if st == 1 and pro == "swish"] + [
    (2, 24, 54, 4, 12, 12, None, None, "relu"), (1, 48, 108, 13, 5, 5, "relu", None, None), (2, 192, 432, 2, 8, 8, None, "identity", None),
    (2, 96, 216, 2, 14, 14, None, None, "relu"), (1, 280, 630, 1, 8, 8, None, "conv", "relu"),
    (1, 54, 24, 2, 20, 20, "swish", "identity", "relu"), (1, 108, 48, 2, 10, 10, "swish", "conv", "relu"),   # fp32 panels of 1 / 2 row tiles
    (1, 216, 96, 2, 10, 10, "swish", "identity", "relu"), (1, 432, 192, 4, 5, 5, "swish", "identity", "relu"),   # ... 3 / 4 (X3D-XS, config 1)
]

# ---- x3d_pw_dgrad: N, Cin, Cout, T, H, W  x  epilogue -----------------------------------------------------------------
PW_DGRAD = [
    (2, 48, 108, 2, 5, 13), (1, 108, 48, 1, 3, 43),   # P = 130 / 129 (see PW_FWD)
    (2, 24, 54, 4, 12, 12), (1, 54, 24, 3, 10, 10), (1, 48, 108, 13, 5, 5), (1, 96, 216, 2, 7, 7),
    (1, 216, 96, 2, 7, 7), (1, 200, 40, 1, 4, 8),
    (1, 54, 24, 4, 14, 14), (1, 48, 108, 2, 28, 28),   # strided add with rows of 2k / 4k points (pair / quad groups)
    (1, 192, 432, 2, 8, 8), (1, 432, 192, 2, 8, 8), (1, 96, 192, 2, 8, 8), (1, 96, 432, 2, 14, 14),   # stage-5 weights-stationary dgrads
    (1, 24, 48, 1, 16, 16), (1, 48, 96, 1, 16, 16), (1, 48, 216, 1, 16, 16),
    (2, 96, 216, 13, 10, 10), (1, 192, 432, 13, 5, 5), (1, 96, 432, 13, 10, 10), (1, 432, 192, 13, 5, 5),   # X3D-S stages 4 / 5: ragged rows
]
PW_DGRAD_EPI = ["store", "add", "add_strided", "swish_bwd"]

# ---- x3d_pw_wgrad: N, Cin, Cout, T, H, W, stride, prologue --------------------------------------------------------------
PW_WGRAD = [
    (1, 32, 32, 1, 2, 17, 1, None), (2, 48, 108, 1, 33, 2, 1, "swish"),   # P = 34 / 66: the partial vector is vector 0 of the last 32-point step
    (2, 24, 54, 4, 12, 12, 1, None), (2, 54, 24, 3, 10, 10, 1, "swish"), (2, 24, 48, 4, 12, 12, 2, None),
    (1, 48, 108, 13, 5, 5, 1, None), (1, 96, 216, 2, 7, 7, 1, None), (1, 216, 96, 2, 7, 7, 1, "swish"),
    (1, 192, 432, 1, 7, 7, 1, None), (1, 432, 192, 1, 7, 7, 1, "swish"), (1, 24, 24, 2, 9, 11, 2, None),
    (1, 24, 48, 2, 56, 56, 2, None), (2, 48, 96, 2, 28, 28, 2, None), (2, 96, 192, 4, 14, 14, 2, None),  # gather groups 4 / 2 / 1
    (2, 96, 216, 2, 8, 8, 1, None), (2, 192, 432, 3, 8, 8, 1, None), (2, 432, 192, 2, 8, 8, 1, "swish"),  # wide layers: 12-tile groups (4x3 / 3x4)
    (2, 192, 432, 8, 7, 7, 1, None), (3, 432, 192, 8, 7, 7, 1, "swish"), (2, 96, 216, 2, 14, 14, 1, None),  # ragged last 64-point step
    (1, 24, 24, 2, 32, 32, 2, None), (1, 48, 216, 1, 16, 16, 1, None), (1, 24, 24, 1, 156, 156, 2, None),
    (1, 24, 48, 1, 78, 78, 2, None), (1, 48, 96, 2, 39, 39, 2, None), (1, 96, 192, 2, 20, 20, 2, None),
    (1, 96, 192, 8, 14, 14, 2, None), (1, 24, 48, 8, 78, 78, 2, None), (1, 24, 24, 8, 156, 156, 2, None),
    (1, 48, 108, 1, 16, 16, 1, None), (1, 108, 48, 1, 16, 16, 1, "swish"),
    (2, 24, 108, 3, 12, 12, 1, None), (1, 24, 216, 2, 10, 10, 1, None), (2, 54, 24, 13, 10, 10, 1, "swish"),   # fp32 tile groups 4x1 / 8x1 / 1x2 with several point chunks
    (2, 96, 216, 13, 10, 10, 1, None), (2, 216, 96, 13, 10, 10, 1, "swish"), (1, 192, 432, 13, 5, 5, 1, None),   # X3D-S stages 4 / 5: ragged rows
    (1, 432, 192, 13, 5, 5, 1, "swish"),
    (1, 24, 48, 2, 11, 23, 2, None), (1, 32, 32, 4, 9, 27, 2, None), (1, 24, 48, 8, 13, 13, 2, None),   # strided, odd input width: vector gather
    (1, 96, 192, 8, 7, 7, 2, None), (2, 96, 192, 8, 3, 3, 2, None),   # rows one gather group wide (7 -> 4, 3 -> 2)
    # round 6: the shortcut convs' weight gradient as a dense launch on the even-pixel copy (x3d_subsample2)
    (1, 24, 24, 2, 40, 40, 1, None), (2, 24, 24, 1, 56, 56, 1, None), (1, 24, 48, 2, 28, 28, 1, None), (2, 48, 96, 2, 14, 14, 1, None),
    (2, 96, 192, 4, 7, 7, 1, None), (1, 24, 48, 2, 39, 39, 1, None), (1, 48, 96, 2, 20, 20, 1, None), (1, 96, 192, 2, 10, 10, 1, None),
    (2, 48, 96, 13, 10, 10, 1, None), (2, 24, 24, 13, 40, 40, 1, None),
]

# ---- x3d_pw_bwd (fused dgrad + wgrad): N, Cin, Cout, T, H, W, epilogue ---------------------------------------------------
PW_BWD = [
    (2, 24, 54, 4, 16, 16, "add"), (2, 48, 108, 2, 28, 28, "add"), (1, 24, 108, 3, 16, 16, "add_strided"),
    (2, 24, 54, 2, 28, 28, "add_strided"),     # rows of 28 points: element loads of the shortcut gradient in the epilogue
    (1, 24, 54, 2, 16, 16, "add_strided"),     # rows of 8k points: 8-byte loads (a separate instantiation)
    (2, 54, 24, 4, 16, 16, "swish_bwd"), (2, 108, 48, 3, 12, 12, "swish_bwd"), (3, 40, 20, 1, 7, 8, "swish_bwd"),
    (1, 96, 32, 2, 10, 12, "swish_bwd"),
    (2, 216, 96, 2, 14, 14, "swish_bwd"),      # stage-4 `c` conv: the weights-stationary fused kernel (pw_bwd_wst.hip)
    (24, 216, 96, 8, 14, 14, "swish_bwd"),     # ... several tiles per persistent workgroup, across sample boundaries
    (3, 200, 90, 1, 10, 12, "swish_bwd"),      # ... widths off the grid, a partial last tile per sample (P = 120)
    (1, 200, 40, 2, 8, 8, "add"), (1, 136, 72, 1, 8, 16, "add_strided"),   # sliced `a`-type layers, widths off the grid
    (2, 96, 216, 2, 14, 14, "add"), (24, 96, 216, 8, 14, 14, "add"), (3, 90, 210, 1, 10, 12, "add"),   # stage-4 `a` conv: pw_bwd_wsta.hip
    (1, 32, 72, 1, 16, 16, "add"), (1, 32, 72, 1, 16, 16, "add_strided"), (1, 72, 32, 1, 16, 16, "swish_bwd"),   # X3D-XL stage 2
    # stage-5 `c` conv (432 <-> 192 on 7 x 7 planes): pw_bwd_wst.hip with two slices of seven row blocks over blockIdx.y
    (2, 432, 192, 8, 7, 7, "swish_bwd"), (10, 432, 192, 16, 7, 7, "swish_bwd"), (3, 420, 180, 1, 10, 12, "swish_bwd"),
    (2, 300, 192, 2, 8, 8, "swish_bwd"),       # ... a short second slice (ten row blocks: 7 + 3)
]

# ... without the conv's raw output (pw_bwd_rc.hip: y = W x folded into the BatchNorm backward): N, Cin, Cout, T, H, W, epilogue, tail
PW_BWD_RC = [
    (2, 24, 54, 4, 16, 16, "add", 0), (2, 24, 54, 4, 16, 16, "add", 1), (2, 24, 54, 4, 16, 16, "add", 2),          # stage 2 (X3D-S / M / L)
    (1, 24, 54, 2, 16, 16, "add_strided", 0), (1, 24, 54, 2, 16, 16, "add_strided", 1),                            # block 0 of stage 2 (stem fold = tail 1)
    (2, 24, 54, 2, 28, 28, "add_strided", 0), (2, 24, 54, 2, 28, 28, "add_strided", 2),                            # rows of 28 points: element form
    (1, 24, 108, 3, 16, 16, "add_strided", 0), (1, 24, 108, 3, 16, 16, "add_strided", 1), (1, 24, 108, 3, 16, 16, "add_strided", 2),   # stage 3 block 0
    (1, 24, 108, 2, 12, 12, "add_strided", 2), (1, 24, 108, 2, 12, 12, "add_strided", 1), (2, 24, 54, 2, 28, 28, "add_strided", 1),   # ... X3D-L / XL rows (element form)
    (1, 32, 72, 1, 16, 16, "add", 0), (1, 32, 72, 1, 16, 16, "add", 1), (1, 32, 72, 1, 16, 16, "add_strided", 2),  # X3D-XL stage 2 (Cin = 32: no spare row)
    (3, 20, 40, 1, 7, 8, "add", 1), (1, 24, 20, 2, 10, 12, "add", 2), (2, 8, 31, 1, 9, 8, "add", 0),               # ragged tiles, widths off the grid, Cout + 1 = 32
    (1, 16, 95, 1, 12, 12, "add", 0), (1, 24, 127, 1, 8, 8, "add", 2),                                             # three / four row tiles of g
    (5, 24, 54, 8, 28, 28, "add", 1),                                                                              # several tiles per workgroup, across samples
    (2, 48, 108, 2, 28, 28, "add", 0), (24, 48, 108, 8, 28, 28, "add", 0),                                         # stage 3: two row tiles of x (48 rows in the image)
    (1, 40, 100, 1, 12, 12, "add", 0), (1, 48, 90, 2, 8, 16, "add_strided", 0), (3, 48, 108, 4, 20, 20, "add", 0), # ... widths off the grid, three g tiles, X3D-L planes
    # first `a` conv of stage 4 (48 -> 216: seven row tiles of g, one workgroup per CU): strided add (the model's case, vector and
    # element forms), plain add, several tiles per workgroup across samples, Cout off the grid
    (2, 48, 216, 2, 16, 16, "add_strided", 0), (1, 48, 216, 2, 28, 28, "add_strided", 0), (1, 48, 216, 1, 16, 16, "add", 0),
    (9, 48, 216, 4, 28, 28, "add_strided", 0), (1, 40, 200, 1, 12, 12, "add", 0),
    (1, 48, 216, 8, 13, 13, "add_strided", 0),      # X3D-L / XL: rows of odd length (39 x 39) take the element form of the strided add
    # ... and rows of 4 k + 2 points (78 x 78): element form too (rows of 4 k take the group form: 12, 28, 156)
    (1, 24, 108, 2, 10, 10, "add_strided", 1), (1, 24, 108, 2, 10, 10, "add_strided", 0), (1, 24, 108, 2, 10, 10, "add_strided", 2),
    (1, 24, 54, 2, 6, 14, "add_strided", 1), (2, 24, 54, 2, 9, 8, "add_strided", 0),
]

# ... of the strided shortcut conv (x_stride = 2, epilogue STORE): N, Cin, Cout, T, xH, xW (input extents)
PW_BWD_RC_STRIDED = [
    (2, 24, 24, 4, 32, 32), (2, 24, 24, 2, 112, 112),          # stage 2 (gather groups of 4)
    (1, 24, 48, 4, 28, 28), (1, 24, 48, 2, 56, 56),            # stage 3: rows of 14 outputs (groups of 2) / 28
    (1, 48, 96, 8, 14, 14), (2, 48, 96, 4, 28, 28),            # stage 4: two row tiles of x; rows of 7 outputs (groups of 1)
    (1, 24, 48, 2, 39, 39), (1, 24, 24, 8, 78, 78), (1, 48, 96, 2, 39, 39),
    (1, 24, 24, 2, 12, 156), (1, 24, 48, 8, 6, 78),            # X3D-L stage 2 / 3 rows: 156 -> 78 (groups of 2), 78 -> 39 (groups of 1)            # X3D-L: odd input rows (39 -> 20: the row's last group loaded early)
    (1, 32, 32, 2, 16, 24), (1, 20, 40, 1, 12, 16),            # X3D-XL stage 2 (Cin = Cout = 32), widths off the grid
]

# ... with the residual-tail backward of the block below folded into the epilogue (the `a` convs: ADD epilogues, panels of
# one or two row tiles): N, Cin, Cout, T, H, W, epilogue, tail (1 = identity shortcut below, 2 = shortcut conv below)
PW_BWD_TAIL = [
    (2, 24, 54, 4, 16, 16, "add", 1), (2, 24, 54, 2, 28, 28, "add_strided", 1), (2, 24, 54, 4, 16, 16, "add", 2),   # stage 2 (X3D-S / M / L)
    (1, 24, 54, 2, 16, 16, "add_strided", 1), (1, 24, 54, 2, 16, 16, "add_strided", 2),
    (1, 24, 108, 3, 16, 16, "add_strided", 1), (1, 24, 108, 3, 16, 16, "add_strided", 2),                          # stage 3 block 0
    (1, 24, 108, 2, 12, 12, "add_strided", 1), (1, 24, 108, 2, 12, 12, "add_strided", 2),                          # ... X3D-L / XL rows (78: element form)
    (2, 48, 54, 2, 28, 28, "add", 1), (2, 48, 54, 2, 28, 28, "add", 2),                                            # two row tiles x two dY tiles
    (1, 32, 72, 1, 16, 16, "add", 1), (1, 32, 72, 1, 16, 16, "add", 2), (1, 32, 72, 1, 16, 16, "add_strided", 1),  # X3D-XL stage 2
    (3, 20, 40, 1, 7, 8, "add", 1), (1, 24, 20, 2, 10, 12, "add", 2),                                              # ragged tiles, widths off the grid
    (2, 96, 216, 2, 14, 14, "add", 1), (24, 96, 216, 8, 14, 14, "add", 1), (3, 90, 210, 1, 10, 12, "add", 1),      # stage-4 `a` conv (pw_bwd_wsta.hip)
]

# ---- x3d_dw3d_fwd / x3d_dw3d_bwd: N, C, T, H, W, stride --------------------------------------------------------------------
DW = [
    (2, 5, 4, 16, 16, 1), (2, 5, 4, 16, 16, 2),       # even, SW=2
    (1, 3, 3, 7, 7, 1), (1, 3, 3, 14, 14, 2),         # 7x7 planes, SW=1
    (1, 2, 5, 39, 39, 2), (1, 2, 2, 20, 20, 1),       # X3D-L odd case 39 -> 20 (pads 1/1)
    (2, 3, 5, 39, 39, 1), (1, 2, 4, 78, 78, 2), (1, 3, 3, 25, 37, 1), (1, 2, 2, 41, 43, 1),   # odd row length of 16-bit outputs, strips of four: every second row starts 2 mod 4, several samples / channels (the plane is an odd number of elements: the parity alternates), last strip of 3 / 1 / 3 outputs
    (1, 2, 3, 56, 56, 1), (1, 2, 3, 112, 112, 2),     # X3D-M stage-2 planes, SW=4, H-tiled
    (1, 2, 1, 9, 23, 2), (1, 1, 2, 10, 13, 1),        # T=1 / non-square / odd widths
    (1, 2, 16, 28, 28, 1),                            # vec 4 path for bf16
    (2, 3, 16, 14, 14, 1), (1, 2, 6, 7, 7, 1),        # deep-prefetch variants (dw_pd.hip): T = 4k, T % 4 != 0,
    (1, 2, 1, 7, 7, 1), (1, 2, 7, 12, 12, 2),         #   T < depth, stride 2
    (1, 2, 4, 28, 28, 2),                             # X3D-M stage-4 first block (216 ch 28 -> 14): <2, 2, 2, 4> in bf16
    (1, 2, 3, 56, 56, 2),                             # X3D-M stage-3 first block (56 -> 28)
    # dw3d_bwd_s2_kernel (dw_s2.hip; 16-bit storage, rows of whole 16-byte vectors): T loop unrolled by 6 (T = 6 / 7 / 13 / 16 / 2 / 1),
    # several samples and channels, a partial last H-tile, non-square planes, odd H (one pad row on top)
    (2, 3, 7, 112, 112, 2), (1, 2, 13, 56, 56, 2), (1, 2, 16, 48, 48, 2), (2, 2, 6, 64, 40, 2), (1, 3, 2, 96, 112, 2), (1, 2, 1, 56, 56, 2),
    (1, 2, 8, 45, 48, 2),
    (1, 2, 7, 120, 120, 2),   # planes too large for the ring kernel's constant LDS strides: dw3d_bwd_s2_kernel (two barriers per plane)
    # dw3d_bwd_s1r_kernel (dw_s1.hip; stride 1, strips of four, 16-bit storage): every T mod 6 of the unrolled loop and its drain,
    # several samples / channels, a partial last H-tile, non-square planes
    (2, 3, 7, 56, 56, 1), (1, 2, 13, 40, 40, 1), (1, 2, 16, 48, 48, 1), (1, 2, 6, 80, 80, 1), (1, 2, 1, 56, 56, 1), (1, 2, 2, 56, 56, 1),
    (1, 3, 8, 64, 40, 1), (1, 2, 4, 56, 56, 1), (1, 2, 5, 56, 56, 1), (1, 2, 9, 32, 32, 1),
    (1, 2, 3, 156, 156, 2), (1, 2, 3, 78, 78, 1), (1, 2, 3, 78, 78, 2), (1, 2, 3, 39, 39, 1),   # X3D-L / XL planes (16 x 312 x 312 clips)
    (1, 2, 3, 20, 20, 2), (1, 2, 3, 10, 10, 1), (1, 2, 3, 80, 80, 2), (1, 2, 3, 40, 40, 1), (1, 2, 3, 40, 40, 2),  # + X3D-S planes
    (1, 2, 3, 10, 10, 2), (1, 2, 3, 5, 5, 1),
    (6, 3, 5, 14, 14, 1), (11, 2, 4, 10, 10, 1), (3, 2, 7, 12, 12, 1),   # packed backward (dw_pk.hip): one plane per wave, strips of 4 / 2
    (9, 3, 5, 7, 7, 1),                                                   #   7x7: four planes per wave (4 + 4 + 1), strips 4 + 3
    (3, 4, 8, 14, 14, 1), (2, 3, 1, 14, 14, 1), (2, 2, 2, 14, 14, 1),     # matrix-core kernels (dw_mx.hip, 16-bit storage): T % 4 == 0 runs the
    (2, 3, 13, 14, 14, 1),                                                #   exit-free loop (ring of 4), else ring of 3; T below the prefetch depth
    (5, 3, 4, 7, 7, 1), (2, 2, 8, 7, 7, 1), (4, 2, 16, 7, 7, 1),          #   7x7 planes four to a tile (partial last group), T % 4 == 0
    (2, 2, 4, 12, 14, 1), (1, 2, 5, 14, 12, 1),                           #   12 / 14 rows and columns
    (2, 3, 5, 28, 28, 1), (1, 2, 4, 26, 30, 1), (2, 2, 8, 28, 26, 1),     #   H-tiled backward for rows of 26 .. 30 elements (dw3d_bwd_mxw_kernel)
    # H- and W-tiled matrix-core backward for ragged rows (dw3d_bwd_mxg_kernel<bf16, NT, W-tiled, odd, ...>): one window of 3 column
    # tiles (39, 45, 44), two / three W-tiles (78 = 40 + 38, 91 = 32 + 32 + 27 odd), windows of 2 column tiles (30, 29, 54 = 28 + 26,
    # 55 = 28 + 27), both T loops, a partial last H-tile
    (2, 3, 4, 39, 39, 1), (1, 2, 8, 78, 78, 1), (1, 2, 5, 39, 39, 1), (1, 2, 4, 36, 45, 1), (1, 2, 4, 36, 44, 1), (1, 2, 4, 28, 91, 1),
    (1, 2, 4, 22, 30, 1), (1, 2, 4, 28, 29, 1), (2, 2, 4, 28, 54, 1), (1, 2, 5, 28, 55, 1), (1, 2, 7, 22, 78, 1),
    # ragged rows (flat staging, CV < 0): X3D-S 182-pixel test crops (91 / 46 / 23), vectors that cross rows and H-tiles,
    # rows shorter than a 16-byte vector (13 -> 7: 8-byte vectors), short planes
    (1, 2, 3, 91, 91, 1), (1, 2, 3, 91, 91, 2), (1, 2, 4, 46, 46, 1), (1, 2, 3, 23, 23, 1), (1, 2, 3, 23, 23, 2),
    (2, 2, 3, 13, 13, 2), (1, 2, 3, 6, 21, 1), (1, 1, 3, 30, 11, 1), (1, 2, 2, 3, 37, 1),
]

# ---- whole-model cases (tests/test_model_gpu.py): variant, N, T, S --------------------------------------------------------
MODEL_TRAIN_FP32 = [
    ("XS", 4, 4, 64), ("S", 2, 13, 64), ("M", 2, 4, 64), ("S", 3, 5, 96),
    ("XS", 2, 4, 78),     # odd extents end to end: 78 -> 39 -> 20 -> 10 -> 5 -> 3 (X3D-L's 39 -> 20 TF-SAME pads, odd stride-2 planes)
    ("M", 2, 16, 112),    # T = 16 and 56 / 28 / 14 / 7 planes: the deep-prefetch depthwise variants (dw_pd.hip) inside the model
    ("S", 1, 2, 160),     # BASELINE config 2's real planes (80 / 40 / 20 / 10 / 5)
    ("S", 8, 13, 160),    # BASELINE config 2 at its real clip size (13 x 160^2), a quarter of its batch: the CPU oracle takes ~10 s
]
MODEL_TRAIN_HALF = [     # teacher-forced block by block, bf16 and fp16 storage
    ("S", 3, 5, 96),      # odd point counts: scalar / generic kernel paths
    ("M", 2, 4, 128),     # every P a multiple of 8, 16-byte aligned rows: the fast paths the benchmark runs
    ("XL", 2, 4, 64),     # XL widths (72/162/306/630...: off the 32-grid, K > 432), 55 blocks, SE parity across stages
    ("L", 1, 2, 312),     # BASELINE config 4's real planes: 156 / 78 / 39 / 20 / 10 (odd 39 -> 20)
    ("S", 1, 13, 91),     # 13 frames on half a 182-pixel test crop: ragged planes (46 / 23 / 12 / 6 / 3) and rows of P % 8 != 0 points at every stage
    ("M", 1, 4, 224),     # BASELINE config 3's own planes (112 / 56 / 28 / 14 / 7) with T % 4 == 0: the exact dw_mx variants, the
                          #   stage-2/3 forward tail fold on real rows, the stem fold at 112^2 -- what bench.py times
]
MODEL_INFER = [           # variant, views, crops, T, S, dtype
    ("XS", 10, 1, 4, 160, F32), ("S", 2, 1, 13, 96, F32),
    ("XL", 10, 3, 2, 96, F32), ("XL", 10, 3, 2, 96, F16), ("XL", 10, 3, 2, 96, BF16),   # BASELINE config 5: 30 views per video
    ("XL", 2, 1, 8, 96, F16),     # stage 5 with P = 8 x 3 x 3 = 72 points (P % 8 == 0): the sliced 630 -> 280 conv on whole vectors
]


# ----------------------------------------------------------------------------------------------------------------------
# argument structs of the kernel-level cases over address-only operands (the dry-run dispatch only looks at alignment)
# ----------------------------------------------------------------------------------------------------------------------
class _Addr:
    nxt = 0x5000_0000_0000

    @classmethod
    def new(cls):
        cls.nxt += 1 << 32
        return cls.nxt


def _code(dtype):
    from x3d_tf_amd import hip
    return hip.dtype_code(dtype)


def pw_fwd_struct(shape, dtype, panel):
    from x3d_tf_amd import hip
    n, cin, cout, t, h, w, stride, pro = shape
    A = _Addr.new
    if pro in ("tail", "tail_conv", "tail1"):
        return hip.PwFwdArgs(A(), A(), A(), A(), A(), None, 1, n, cin, cout, t, h, w, stride, _code(dtype), A() if panel else None,
                             in_add=None if pro == "tail1" else A(), in_add_scale_shift=A() if pro == "tail_conv" else None, in_store=A())
    return hip.PwFwdArgs(A(), A(), A(), A(), A() if pro else None, A() if pro == "swish" else None,
                         {None: 0, "relu": 1, "swish": 2}[pro], n, cin, cout, t, h, w, stride, _code(dtype),
                         A() if panel else None)


def pw_fwd_infer_struct(shape, dtype, panel):
    from x3d_tf_amd import hip
    n, cin, cout, t, h, w, pro, res, oact = shape
    A = _Addr.new
    return hip.PwFwdArgs(A(), A(), A(), None, A() if pro else None, A() if pro == "swish" else None,
                         {None: 0, "relu": 1, "swish": 2}[pro], n, cin, cout, t, h, w, 1, _code(dtype),
                         A() if panel else None, out_scale_shift=A(), out_add=A() if res else None,
                         out_add_scale_shift=A() if res == "conv" else None, out_act=1 if oact == "relu" else 0)


def pw_dgrad_struct(shape, epi, dtype, panel):
    from x3d_tf_amd import hip
    n, cin, cout, t, h, w = shape
    A = _Addr.new
    e = PW_DGRAD_EPI.index(epi)
    sw = epi == "swish_bwd"
    return hip.PwDgradArgs(A(), A(), A(), A(), A(), e, A() if epi in ("add", "add_strided") else None,
                           A() if sw else None, A() if sw else None, A() if sw else None, A() if sw else None,
                           n, cin, cout, t, h, w, _code(dtype), A() if panel else None)


def pw_wgrad_struct(shape, dtype):
    from x3d_tf_amd import hip
    n, cin, cout, t, h, w, stride, pro = shape
    A = _Addr.new
    return hip.PwWgradArgs(A(), A(), A(), A(), A() if pro else None, A() if pro else None, 2 if pro else 0, A(),
                           n, cin, cout, t, h, w, stride, _code(dtype))


def pw_bwd_struct(shape, dtype):
    from x3d_tf_amd import hip
    n, cin, cout, t, h, w, epi = shape[:7]
    tail = shape[7] if len(shape) > 7 else 0
    A = _Addr.new
    e = PW_DGRAD_EPI.index(epi)
    sw = epi == "swish_bwd"
    return hip.PwBwdArgs(A(), A(), A(), A(), A(), e, None if sw else A(), A() if sw else None, A() if sw else None,
                         A() if sw else None, A() if sw else None, None if sw else A(), A(), n, cin, cout, t, h, w,
                         _code(dtype), A() if tail else None, A() if tail == 2 else None, A() if tail else None,
                         A() if tail == 2 else None)


def pw_bwd_rc_struct(shape, dtype):
    from x3d_tf_amd import hip
    n, cin, cout, t, h, w, epi, tail = shape
    A = _Addr.new
    return hip.PwBwdArgs(A(), None, None, None, A(), PW_DGRAD_EPI.index(epi), None if epi == "store" else A(), None, None, None, None, A(), None, n, cin,
                         cout, t, h, w, _code(dtype), A() if tail else None, A() if tail == 2 else None, A() if tail else None,
                         A() if tail == 2 else None, A(), A(), A())


def pw_bwd_rc_strided_struct(shape, dtype):
    from x3d_tf_amd import hip
    n, cin, cout, t, xh, xw = shape
    A = _Addr.new
    return hip.PwBwdArgs(A(), None, None, None, A(), 0, None, None, None, None, None, A(), None, n, cin, cout, t, (xh + 1) // 2,
                         (xw + 1) // 2, _code(dtype), None, None, None, None, A(), A(), A(), 2, xh, xw)


def dw_fwd_struct(shape, dtype):
    from x3d_tf_amd import hip
    n, c, t, h, w, stride = shape
    A = _Addr.new
    return hip.Dw3dFwdArgs(A(), A(), A(), A(), 1, A(), A(), n, c, t, h, w, stride, _code(dtype))


def dw_bwd_struct(shape, dtype):
    from x3d_tf_amd import hip
    n, c, t, h, w, stride = shape
    A = _Addr.new
    return hip.Dw3dBwdArgs(A(), A(), A(), A(), A(), A(), A(), A(), A(), n, c, t, h, w, stride, _code(dtype))
