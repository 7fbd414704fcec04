"""Explicit switches of the Python layer.

The product path reads NO environment variable (tests/test_abi.py greps itr_amd/ for it; train.py / test.py read only the launcher's
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* like the reference reads CUDA_VISIBLE_DEVICES, itr/config.py:412): a stray variable cannot
change the arithmetic or the order of work of an evaluation.  Every switch is a plain attribute with its production default; a test,
a study tool or bench.py sets it in its own process (`from itr_amd.settings import SETTINGS; SETTINGS.x = ...`), where a reader of that
program sees it."""


class _Settings(object):
    __slots__ = ("sgr_group_rows", "sgraf_image_block", "agsa_fused", "vsrn_residual_in_epilogue", "force_collectives", "virtual_split",
                 "exchange")

    def __init__(self):
        self.reset()

    def reset(self):
        self.sgr_group_rows = 64                # SGR node groups of <= 64 rows; 32 = the two-class plan kept as a bit-identical cross-check
        self.sgraf_image_block = None           # None = memory-aware choice (ops._sgraf_workspace); an int pins the pair stage's image block
        self.agsa_fused = True                  # CAMERA's gate as one kernel (csrc/agsa_gate.hip); False = the three-GEMM composition
        self.vsrn_residual_in_epilogue = True   # Rs_GCN's `W(y) + v` read by the GEMM epilogue; False = copy v, accumulate onto the copy
        self.force_collectives = False          # run the N > 1 path's collectives in a 1-rank process group (1-GPU smoke of RCCL)
        self.virtual_split = None               # "k[:v]": treat the caption axis as owned by k ranks of which this process is owner v
        self.exchange = "all_gather"            # "p2p": the exchange as batched point-to-point sends / receives (unmeasured on xGMI)


SETTINGS = _Settings()
