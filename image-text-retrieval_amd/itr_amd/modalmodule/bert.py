"""BERT with the reference's module / parameter names (itr/modalmodule/bert.py: 2018 pytorch-pretrained-BERT
layout: LayerNorm.gamma / .beta, attention.self.query ...), so `pytorch_model.bin` checkpoints load unchanged.
The modules are parameter containers; `forward` runs on the HIP path: fp32 MFMA GEMMs with fused bias / erf-GELU /
tanh, fused embedding-sum + LayerNorm, residual + TF-style LayerNorm, and a small-sequence attention kernel."""
import copy
import json
import math

import torch
from torch import nn

from .. import ops


class BertConfig(object):
    """Same fields and JSON loader as bert.py:37-110."""

    def __init__(self, vocab_size, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
                 attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=16,
                 initializer_range=0.02):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.hidden_act = hidden_act
        self.intermediate_size = intermediate_size
        self.hidden_dropout_prob = hidden_dropout_prob
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        self.max_position_embeddings = max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.initializer_range = initializer_range

    @classmethod
    def from_dict(cls, json_object):
        config = BertConfig(vocab_size=None)
        for (key, value) in json_object.items():
            config.__dict__[key] = value
        return config

    @classmethod
    def from_json_file(cls, json_file):
        with open(json_file, "r") as reader:
            return cls.from_dict(json.loads(reader.read()))

    def to_dict(self):
        return copy.deepcopy(self.__dict__)


class BERTLayerNorm(nn.Module):
    def __init__(self, config, variance_epsilon=1e-12):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(config.hidden_size))
        self.beta = nn.Parameter(torch.zeros(config.hidden_size))
        self.variance_epsilon = variance_epsilon

    def forward(self, x, residual=None):
        return ops.add_layernorm(x, residual, self.gamma.detach(), self.beta.detach(), self.variance_epsilon)


class BERTEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = BERTLayerNorm(config)

    def forward(self, input_ids, token_type_ids=None):
        return ops.bert_embed_ln(input_ids, token_type_ids, self.word_embeddings.weight.detach(),
                                 self.position_embeddings.weight.detach(), self.token_type_embeddings.weight.detach(),
                                 self.LayerNorm.gamma.detach(), self.LayerNorm.beta.detach(),
                                 self.LayerNorm.variance_epsilon)


class BERTSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self._fused = None

    def _qkv(self):
        # one GEMM for Q, K, V: concatenated [3H, H] weight (parameter prep, cached; frozen tower)
        if self._fused is None or self._fused[0].device != self.query.weight.device:
            w = torch.cat([self.query.weight, self.key.weight, self.value.weight], 0).detach().contiguous()
            b = torch.cat([self.query.bias, self.key.bias, self.value.bias], 0).detach().contiguous()
            self._fused = (w, b)
        return self._fused

    def forward(self, hidden_states, mask01):
        B, L, H = hidden_states.shape
        w, b = self._qkv()
        qkv = ops.linear(hidden_states.reshape(B * L, H), w, b)          # (B*L, 3H)
        A = self.all_head_size
        ctx = ops.mha_small(qkv[:, :A], qkv[:, A:2 * A], qkv[:, 2 * A:], mask01, B, L, self.num_attention_heads,
                            self.attention_head_size, 1.0 / math.sqrt(self.attention_head_size))
        return ctx.view(B, L, A)


class BERTSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BERTLayerNorm(config)

    def forward(self, hidden_states, input_tensor):
        h = ops.linear(hidden_states, self.dense.weight.detach(), self.dense.bias.detach())
        return self.LayerNorm(h, input_tensor)


class BERTAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BERTSelfAttention(config)
        self.output = BERTSelfOutput(config)

    def forward(self, input_tensor, mask01):
        return self.output(self.self(input_tensor, mask01), input_tensor)


class BERTIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)

    def forward(self, hidden_states):
        return ops.linear(hidden_states, self.dense.weight.detach(), self.dense.bias.detach(), act='gelu')


class BERTOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = BERTLayerNorm(config)

    def forward(self, hidden_states, input_tensor):
        h = ops.linear(hidden_states, self.dense.weight.detach(), self.dense.bias.detach())
        return self.LayerNorm(h, input_tensor)


class BERTLayer(nn.Module):
    """bert.py:262-273.  `attention_mask` is accepted either as the reference's extended additive mask
    (B, 1, 1, L) of {0, -10000} or as a (B, L) 0/1 mask."""

    def __init__(self, config):
        super().__init__()
        self.attention = BERTAttention(config)
        self.intermediate = BERTIntermediate(config)
        self.output = BERTOutput(config)
        # nn.Dropout sites of the reference (bert.py:174, :221, :255); used by forward_train only
        self.p_attn, self.p_hidden = float(config.attention_probs_dropout_prob), float(config.hidden_dropout_prob)

    def forward_train(self, hidden_states, mask01, seeds, training=True):
        """The layer on the autograd tape (train_emb): hidden_states (B, L, H) -> (B, L, H).  Same arithmetic as forward plus
        the reference's three dropout sites when `training`; under torch.no_grad() this is the frozen-tower forward in
        training mode (the reference keeps BERT's dropout active although its weights are frozen, TextEncoder.py:83 + model.train())."""
        from .. import autograd as ag
        B, L, H = hidden_states.shape
        at, so, io, oo = self.attention.self, self.attention.output, self.intermediate, self.output
        x2 = hidden_states.reshape(B * L, H)
        w = torch.cat([at.query.weight, at.key.weight, at.value.weight], 0)
        b = torch.cat([at.query.bias, at.key.bias, at.value.bias], 0)
        qkv = ag.linear(x2, w, b)
        ctx = ag.mha(qkv, mask01, B, L, at.num_attention_heads, self.p_attn if training else 0.0, seeds.next())
        h = ag.dropout(ag.linear(ctx, so.dense.weight, so.dense.bias), self.p_hidden, seeds, training)
        att = ag.add_layernorm(h, x2, so.LayerNorm.gamma, so.LayerNorm.beta, so.LayerNorm.variance_epsilon)
        inter = ag.gelu(ag.linear(att, io.dense.weight, io.dense.bias))
        h2 = ag.dropout(ag.linear(inter, oo.dense.weight, oo.dense.bias), self.p_hidden, seeds, training)
        return ag.add_layernorm(h2, att, oo.LayerNorm.gamma, oo.LayerNorm.beta, oo.LayerNorm.variance_epsilon).view(B, L, H)

    def forward(self, hidden_states, attention_mask):
        mask01 = attention_mask
        if mask01 is not None and mask01.dim() == 4:
            mask01 = (mask01.reshape(mask01.shape[0], -1) > -5000.0).to(torch.float32)
        attention_output = self.attention(hidden_states, mask01)
        intermediate_output = self.intermediate(attention_output)
        return self.output(intermediate_output, attention_output)


class BERTEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layer = nn.ModuleList([BERTLayer(config) for _ in range(config.num_hidden_layers)])

    def forward(self, hidden_states, mask01):
        all_encoder_layers = []
        for layer_module in self.layer:
            hidden_states = layer_module(hidden_states, mask01)
            all_encoder_layers.append(hidden_states)
        return all_encoder_layers


class BERTPooler(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)

    def forward(self, hidden_states):
        return ops.linear(hidden_states[:, 0].contiguous(), self.dense.weight.detach(), self.dense.bias.detach(), act='tanh')


class BertModel(nn.Module):
    """bert.py:305-358: returns (all_encoder_layers, pooled_output)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = BERTEmbeddings(config)
        self.encoder = BERTEncoder(config)
        self.pooler = BERTPooler(config)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None):
        input_ids = input_ids.cuda() if not input_ids.is_cuda else input_ids
        mask01 = None
        if attention_mask is not None:
            mask01 = attention_mask.to(input_ids.device).to(torch.float32)
        if token_type_ids is not None:
            token_type_ids = token_type_ids.to(input_ids.device)
        embedding_output = self.embeddings(input_ids, token_type_ids)
        all_encoder_layers = self.encoder(embedding_output, mask01)
        pooled_output = self.pooler(all_encoder_layers[-1])
        return all_encoder_layers, pooled_output

    def forward_frozen_train(self, input_ids, token_type_ids, attention_mask, seeds):
        """Last encoder layer of the FROZEN tower in training mode (SAEM / CAMERA text towers under model.train(): the
        weights receive no gradient but the dropout sites are live -- embeddings :151-158, every layer :174/:221/:255)."""
        from .. import autograd as ag
        input_ids = input_ids.cuda() if not input_ids.is_cuda else input_ids
        mask01 = attention_mask.to(input_ids.device).to(torch.float32) if attention_mask is not None else None
        if token_type_ids is not None:
            token_type_ids = token_type_ids.to(input_ids.device)
        with torch.no_grad():
            h = self.embeddings(input_ids, token_type_ids)
            h = ag.dropout(h, float(self.config.hidden_dropout_prob), seeds)
            for layer in self.encoder.layer:
                h = layer.forward_train(h, mask01, seeds, training=True)
        return h
