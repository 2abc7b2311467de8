"""cls_only_last_layer (opt-in): token 0 of the final hidden state and EVERY parameter gradient equal the full model's.
Runs on the CPU: the patched forwards are plain torch there (the HIP paths inside them are covered by the GPU test)."""

import copy

import pytest
import torch


def _grads(model):
    return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}


def test_clip_vision_last_layer_for_token_0_only():
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

    from mmlearn_amd import fused

    torch.manual_seed(0)
    cfg = CLIPVisionConfig(patch_size=8, image_size=32, projection_dim=16, hidden_size=64, intermediate_size=128, num_hidden_layers=3,
                           num_attention_heads=2)
    full = CLIPVisionModelWithProjection(cfg).train()
    cls = copy.deepcopy(full)
    assert fused.cls_only_last_layer(cls) == 1 and fused.cls_only_last_layer(cls) == 0
    x = torch.randn(5, 3, 32, 32)
    w = torch.randn(5, 16)
    outs = []
    for m in (full, cls):
        e = m(pixel_values=x).image_embeds
        (e * w).sum().backward()
        outs.append((e.detach(), _grads(m)))
    assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-6)
    assert outs[0][1].keys() == outs[1][1].keys() and len(outs[0][1]) > 40
    for k in outs[0][1]:
        assert torch.allclose(outs[0][1][k], outs[1][1][k], rtol=1e-4, atol=1e-6), k
    # the encoder's last hidden state is now one token long
    assert cls.vision_model(pixel_values=x).last_hidden_state.shape == (5, 1, 64)


def test_bert_last_layer_for_cls_only():
    from transformers import BertConfig, BertModel

    from mmlearn_amd import fused

    torch.manual_seed(1)
    cfg = BertConfig(hidden_size=64, num_hidden_layers=3, num_attention_heads=2, intermediate_size=128, vocab_size=200,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    full = BertModel(cfg, add_pooling_layer=False).train()
    cls = copy.deepcopy(full)
    assert fused.cls_only_last_layer(cls) == 1
    ids = torch.randint(0, 200, (4, 11))
    w = torch.randn(4, 64)
    outs = []
    for m in (full, cls):
        h = m(input_ids=ids).last_hidden_state[:, 0]
        (h * w).sum().backward()
        outs.append((h.detach(), _grads(m)))
    assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-6)
    assert outs[0][1].keys() == outs[1][1].keys()
    for k in outs[0][1]:
        assert torch.allclose(outs[0][1][k], outs[1][1][k], rtol=1e-4, atol=1e-6), k
    assert cls(input_ids=ids).last_hidden_state.shape == (4, 1, 64)
    # a call the CLS-only form cannot serve (an attention mask) runs the full layer
    mask = torch.ones(4, 11, dtype=torch.long)
    mask[:, -3:] = 0
    a = full(input_ids=ids, attention_mask=mask).last_hidden_state
    b = cls(input_ids=ids, attention_mask=mask).last_hidden_state
    assert b.shape == a.shape and torch.allclose(a, b, rtol=1e-5, atol=1e-6)
