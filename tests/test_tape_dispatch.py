"""CPU: the launch plan of the training tape's dense products (itr_amd/autograd.py:_peel_plan) -- pure host arithmetic.  The tile
kernel keeps 2 x 256 workgroups resident, so a product of 257-1023 output tiles is planned as a main launch of exactly one or two
workgroups per CU plus a K-sliced tail, when the model says that pays."""
from itr_amd.autograd import _peel_plan


def test_peel_plan_on_the_shapes_it_was_built_for():
    assert _peel_plan(4608, 2048, 2048) == ('rows', 4096)          # 576 tiles = 512 + 64 (VSRN: 36 products per step)
    assert _peel_plan(4608, 1024, 2048) == ('rows', 4096)          # 288 = 256 + 32
    assert _peel_plan(4096, 3072, 768) == ('cols', 2048)           # 768 = 512 + 256 column-wise (BERT's FFN at 4 096 rows; no row cut gives whole rounds)
    kind, main = _peel_plan(2048, 2304, 768)                       # 288 = 252 + 36 (rows) or 256 + 32 (columns)
    assert (kind, main) in (('rows', 1792), ('cols', 2048))


def test_peel_plan_leaves_the_rest_alone():
    assert _peel_plan(2048, 3072, 768) is None                     # 384 tiles: a 128-tile tail is a round of its own
    assert _peel_plan(7808, 2048, 2048) is None                    # 976 tiles: 3.8 rounds of 256, nothing to win
    assert _peel_plan(4608, 2048, 256) is None                     # short K: the extra launch is not paid back
    assert _peel_plan(300, 300, 300) is None
    assert _peel_plan(128 * 64, 128 * 16, 2048) is None            # 1 024 tiles: the streaming kernel's
    for M, N, K in [(4608, 2048, 2048), (4700, 2000, 1024), (2560, 1664, 512)]:
        plan = _peel_plan(M, N, K)
        assert plan is not None and plan[1] % 128 == 0 and 0 < plan[1] < (M if plan[0] == 'rows' else N)
