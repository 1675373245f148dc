# ---- the `a` conv without its output tensor (ab_fused.hip): N, Cin, C, T, H, W, stride --------------------------------------
AB = [
    (2, 24, 54, 3, 16, 16, 2), (2, 24, 54, 3, 16, 16, 1),       # strips of 2, one H-tile, four channel groups (the last: 6 of 16)
    (1, 24, 54, 4, 112, 112, 2),                                # X3D-M block 0: 28 H-tiles of 2 output rows, strips of 4
    (2, 24, 54, 2, 56, 56, 1), (1, 24, 108, 3, 56, 56, 2),      # stage 2 stride 1 (left pad 1: unaligned plane writes), stage 3 block 0
    (1, 48, 108, 2, 32, 32, 1), (1, 48, 216, 2, 24, 24, 2),     # Cin = 48: two k blocks per tile
    (1, 32, 72, 2, 24, 24, 2), (1, 32, 72, 1, 40, 40, 1),       # X3D-XL widths (Cin = 32: every k block real)
    (1, 24, 20, 3, 40, 40, 1), (1, 8, 9, 2, 44, 48, 2),         # partial channel groups, a short last H-tile, Cin = 8
    (3, 24, 54, 1, 24, 32, 2), (1, 24, 54, 5, 8, 8, 1),         # T = 1, non-square, tiny planes
]

