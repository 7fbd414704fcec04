"""EncoderSimilarity state_dict with the reference's layout (Fusionmodule.py:373-404), Xavier-uniform linears and non-trivial
BatchNorm running statistics -- shared by the SGRAF parity tests."""
import numpy as np
import torch


def make(D, S, sgr_step=3, seed=5):
    g = torch.Generator().manual_seed(seed)
    w = {}

    def lin(name, o, i):
        r = float(np.sqrt(6.0 / (i + o)))
        w[name + ".weight"] = (torch.rand(o, i, generator=g) * 2 - 1) * r
        w[name + ".bias"] = torch.randn(o, generator=g) * 0.02

    def bn(name, n):
        w[name + ".weight"] = torch.rand(n, generator=g) * 0.4 + 0.8
        w[name + ".bias"] = torch.randn(n, generator=g) * 0.05
        w[name + ".running_mean"] = torch.randn(n, generator=g) * 0.1
        w[name + ".running_var"] = torch.rand(n, generator=g) + 0.5

    lin("v_global_w.embedding_local.0", D, D); bn("v_global_w.embedding_local.1", 36)
    lin("v_global_w.embedding_global.0", D, D); bn("v_global_w.embedding_global.1", D)
    lin("v_global_w.embedding_common.0", 1, D)
    lin("t_global_w.embedding_local.0", D, D); lin("t_global_w.embedding_global.0", D, D); lin("t_global_w.embedding_common.0", 1, D)
    lin("sim_tranloc_w", S, D); lin("sim_tranglo_w", S, D); lin("sim_eval_w", 1, S)
    lin("SAF_module.attn_sim_w", 1, S); bn("SAF_module.bn", 1)
    for k in range(sgr_step):
        for nm in ("graph_query_w", "graph_key_w", "sim_graph_w"):
            lin("SGR_module.sgr%d.%s" % (k, nm), S, S)
    return w
