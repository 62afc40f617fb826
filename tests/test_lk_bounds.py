"""The tracker's staging reads rows below a window that touches a level's bottom border: masked taps of the lane-items, the J region's
rows rounded up to a whole staging round, the template's image rows.  Slots of a frame batch are contiguous, so such a read lands in
the next slot's plane -- except for the LAST slot, where it must stay inside the slack rows dr3hip_batch_create allocates behind the
planes (3dr_amd/csrc/dr3hip_api.cpp: pad_y = win_h + 6 rows of border above and below, + 17 rows of slack).  This test restates the row
counts of lk_trackg_kernel's shapes (3dr_amd/csrc/lk_kernels.hip: lk_group_shape, RPR, IROWS, the J rounds) and checks them against that
allocation for every window the ABI accepts."""
import itertools

RM, SLACK_ROWS = 4, 17


def group_shape(w, h):
    n8 = w * ((h + 7) // 8)
    if w <= 16 and h <= 16:
        return 8, 2
    if (w, h) == (21, 21):
        return 4, 4
    if n8 <= 32:
        return 2, 1
    if n8 <= 64:
        return 2, 2
    return 1, 2


def rows_read_below_origin(w, h):
    """(J rows from the region's first row, template image rows and derivative rows from the window's first row)"""
    ppw, ni = group_shape(w, h)
    fixed = w == h and w in (15, 21, 30)
    colf = ppw == 8 or (ppw == 4 and ni == 1)                                      # whole-column lane-items
    lanes = 64 // ppw
    cols = 8 if colf else (((w + 2 * RM + 4 + 3) // 4 + 1) & ~1) if (fixed and ppw == 4) else 16
    rpr = lanes // (cols // 2)
    nblk = 1 if colf else (h + 7) // 8
    rh = h + 9
    j_rows = -(-rh // rpr) * rpr + (1 if rpr * (cols // 2) != lanes else 0) if fixed else rh   # whole rounds (+ the spare lanes' row: loaded, not stored)
    need = (h + 1 if fixed else 17) if colf else 8 * nblk + 1                      # rows the lane-items read (masked taps included)
    i_rows = -(-need // rpr) * rpr + (1 if fixed and rpr * (cols // 2) != lanes else 0) if fixed else need   # run-time windows stop at `need`
    return j_rows, i_rows, need


def test_staging_rows_stay_inside_the_slack_behind_the_last_slot():
    for w, h in itertools.product(range(3, 33), repeat=2):
        pad_y = h + 6
        j_rows, i_rows, d_rows = rows_read_below_origin(w, h)
        for lh in (4, h + 1, 47, 376):                                             # level heights: the smallest the ABI accepts (an image may be smaller than the window), a coarse level, level 0
            rows_available = lh + 2 * pad_y + SLACK_ROWS
            # the J region starts at most RM rows above the last image row (the staging centre is clamped to the level)
            last_j = (lh - 1 - RM) + pad_y + j_rows - 1
            # a window's first row is at most the last image row (the bounds test lets windows start anywhere in [-win, size))
            last_i = (lh - 1) + pad_y + i_rows - 1
            last_d = (lh - 1) + pad_y + d_rows - 1
            assert max(last_j, last_i, last_d) < rows_available, (w, h, lh, last_j, last_i, last_d, rows_available)


def test_compile_time_shapes_match_the_launcher():
    assert group_shape(15, 15) == (8, 2) and group_shape(21, 21) == (4, 4) and group_shape(30, 30) == (1, 2)
    assert rows_read_below_origin(15, 15) == (24, 16, 16)
    assert rows_read_below_origin(21, 21) == (31, 28, 25)        # J: 10 rounds of 3 rows + the spare lane's row; I: 9 rounds + 1
    assert rows_read_below_origin(30, 30) == (40, 40, 33)
