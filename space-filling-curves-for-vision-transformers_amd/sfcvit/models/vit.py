"""VisionTransformer / VisionTransformer1D on HIP kernels, drop-in for
src/models/vit.py:177-458 of the reference: same constructor signatures, module
tree and state_dict keys, including the unused token-mix
parameters and the tokenizer registered a second time under `encoder`.

Parameters live in stock torch containers (nn.Linear, nn.LayerNorm,
nn.TransformerEncoder as a *parameter holder*: identical keys and identical
initialisation for a given seed); the arithmetic never goes through them -- every
forward here calls sfcvit.functional, i.e. libsfcvit_hip.so.
"""
import math

import torch
import torch.nn as nn

from .. import functional as F
from ..tokenizers.base_patch_embedding import BasePatchEmbedding


class TransformerSeqEncoder(nn.Module):
    """vit.py:177-242: `depth` post-norm nn.TransformerEncoderLayer (relu, eps 1e-5)."""

    def __init__(self, input_dim, max_len, n_head, hidden_dim, method, dropout_p=0.1, n_layers=1):
        super().__init__()
        self.max_len = max_len
        self.grid_size = int(math.sqrt(max_len))
        self.n_head = n_head
        self.dropout_p = dropout_p
        encoder_layer = nn.TransformerEncoderLayer(d_model=input_dim, nhead=n_head,
                                                   dim_feedforward=hidden_dim, dropout=dropout_p,
                                                   batch_first=True)
        self.transformer = nn.TransformerEncoder(encoder_layer, num_layers=n_layers,
                                                 enable_nested_tensor=False)
        self.to_patch_embedding = method

    def forward(self, x):
        p = self.dropout_p if self.training else 0.0
        for layer in self.transformer.layers:
            a = layer.self_attn
            x = F.encoder_layer(x, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                                layer.norm1.weight, layer.norm1.bias, layer.linear1.weight, layer.linear1.bias,
                                layer.linear2.weight, layer.linear2.bias, layer.norm2.weight, layer.norm2.bias,
                                self.n_head, layer.norm1.eps, dropout_p=p)
        return x


class MixerBlock(nn.Module):
    """vit.py:250-273: only the channel-mix branch is live; the token-mix parameters
    exist (state_dict compatibility) and never receive a gradient."""

    def __init__(self, seq_len, embed_dim, hidden_dim, out_dim):
        super().__init__()
        self.token_mix_ln = nn.LayerNorm(embed_dim)
        self.channel_mix_ln = nn.LayerNorm(embed_dim)
        self.token_mix = nn.Sequential(nn.Linear(seq_len, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, seq_len))
        self.channel_mix = nn.Sequential(nn.Linear(embed_dim, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, out_dim))

    def forward(self, x):
        ln, fc1, fc2 = self.channel_mix_ln, self.channel_mix[0], self.channel_mix[2]
        return F.mixer_block(x, ln.weight, ln.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias, ln.eps)


class FactorisedLinear(nn.Module):
    """vit.py:276-292: y[b, o] = sum_{n, r} (x[b, n, :] . W_emb[r, :]) * W_seq[o, n, r]."""

    def __init__(self, seq_len, embed_dim, rank, out_dim):
        super().__init__()
        self.W_emb = nn.Parameter(torch.empty(rank, embed_dim))
        self.W_seq = nn.Parameter(torch.empty(out_dim, seq_len, rank))
        nn.init.xavier_normal_(self.W_emb)
        nn.init.xavier_normal_(self.W_seq)

    def forward(self, x):
        b, n, _ = x.shape
        h = F.linear(x, self.W_emb)
        return F.linear(h.reshape(b, n * self.W_emb.shape[0]), self.W_seq.reshape(self.W_seq.shape[0], -1))


class MultiLayerPredictor(nn.Sequential):
    """vit.py:295-319.  The n_layers = 2 form the models use runs as one fused function."""

    def __init__(self, embed_dim, seq_len, n_layers=2, rank=64, dropout_p=0.5, num_classes=10, mix=False):
        super().__init__()
        if mix:
            raise TypeError("MultiLayerPredictor(mix=True) cannot be constructed in the reference either "
                            "(vit.py:301 passes 3 of MixerBlock's 4 arguments)")
        self.append(nn.LayerNorm(embed_dim))
        fact_out = embed_dim * 2
        self.append(FactorisedLinear(seq_len, embed_dim, rank, fact_out))
        self.append(nn.GELU())
        self.append(nn.Dropout(dropout_p))
        prev_dim = fact_out
        for _ in range(n_layers - 2):
            next_dim = prev_dim // 2
            self.append(nn.Linear(prev_dim, next_dim))
            self.append(nn.GELU())
            self.append(nn.Dropout(dropout_p))
            prev_dim = next_dim
        self.append(nn.Linear(prev_dim, num_classes))
        self._n_layers = n_layers
        self._dropout_p = dropout_p

    def forward(self, x):
        p = self._dropout_p if self.training else 0.0
        if self._n_layers == 2:
            ln, fact, fc = self[0], self[1], self[4]
            return F.predictor_head(x, ln.weight, ln.bias, fact.W_emb, fact.W_seq, fc.weight, fc.bias, ln.eps,
                                    dropout_p=p)
        # any depth (vit.py:310-318): every nn.GELU here is followed by its nn.Dropout -- one fused pass per pair
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.LayerNorm):
                x = F.layer_norm(x, m.weight, m.bias, m.eps)
            elif isinstance(m, nn.Linear):
                x = F.linear(x, m.weight, m.bias)
            elif isinstance(m, nn.GELU) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.Dropout):
                x = F.gelu_dropout(x, p)
                i += 1
            elif isinstance(m, nn.GELU):
                x = F.gelu(x)
            elif isinstance(m, nn.Dropout):
                if p > 0:
                    raise NotImplementedError("a Dropout that does not follow a GELU")
            else:
                x = m(x)
            i += 1
        return x


class VisionTransformer(nn.Module):
    """vit.py:325-385 (`embed_dim` is ignored there too: taken from the tokenizer, :351)."""

    def __init__(self, patch_embed: BasePatchEmbedding, embed_dim=128, depth=6, n_heads=4, mlp_dim=256,
                 num_classes=10, dropout_p=0.1, head_dropout_p=0.5):
        super().__init__()
        self.patch_embed = patch_embed
        embed_dim = patch_embed.embed_dim
        self.encoder = TransformerSeqEncoder(input_dim=embed_dim, max_len=self.patch_embed.n_patches,
                                             method=self.patch_embed, n_head=n_heads, hidden_dim=mlp_dim,
                                             n_layers=depth, dropout_p=dropout_p)
        self.mlp_head = MultiLayerPredictor(embed_dim, self.patch_embed.n_patches, n_layers=2,
                                            num_classes=num_classes, dropout_p=head_dropout_p)

    def forward(self, x):
        x = self.patch_embed(x)
        x = self.encoder(x)
        return self.mlp_head(x)


class VisionTransformer1D(nn.Module):
    """vit.py:392-458: tokenizer -> channel-mix block -> encoder stack -> factorised head."""

    def __init__(self, patch_embed: BasePatchEmbedding, embed_dim=128, depth=6, n_heads=4, mlp_dim=256,
                 num_classes=10, dropout_p=0.1, head_dropout_p=0.5):
        super().__init__()
        self.patch_embed = patch_embed
        embed_dim = patch_embed.embed_dim
        self.mlp_mixer = MixerBlock(seq_len=self.patch_embed.n_patches, embed_dim=embed_dim,
                                    hidden_dim=embed_dim * 2, out_dim=embed_dim)
        self.encoder = TransformerSeqEncoder(input_dim=embed_dim, max_len=self.patch_embed.n_patches,
                                             n_head=n_heads, hidden_dim=mlp_dim, n_layers=depth,
                                             method=self.patch_embed, dropout_p=dropout_p)
        self.mlp_head = MultiLayerPredictor(embed_dim, self.patch_embed.n_patches, n_layers=2,
                                            dropout_p=head_dropout_p, num_classes=num_classes)

    def forward(self, x):
        x = self.patch_embed(x)
        x = self.mlp_mixer(x)
        x = self.encoder(x)
        return self.mlp_head(x)
