"""CPU check (kernel emulator) of the experimental option "s_mask": any mix of quarter-tile and full-tile kernels over the stages of a tree level gives the same bits."""
import pytest

import parity_cases as pc
from test_emu_parity import env  # noqa: F401  (fixture)


@pytest.mark.parametrize("mask", [31, 5, 10, 17])
def test_conv_then_pack_with_quarter_tile_stages_on_big_levels(env, mask):
    ctx, O = env
    ctx.set_option("small_levels", 0)
    ctx.set_option("s_mask", mask)
    try:
        pc.case_conv(ctx, O, 16)
    finally:
        ctx.set_option("s_mask", 0)
        ctx.set_option("small_levels", 16)
