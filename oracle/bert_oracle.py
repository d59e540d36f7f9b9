"""TEST INFRASTRUCTURE -- not part of the product path.

Oracle for the encoder site (classifier.py:1271-1275): the installed transformers `BertModel`
(v5.15.0, modeling_bert.py) in fp32 / eval / eager attention on CPU, randomly initialised from a
seed because no pretrained weights are available offline, followed by CLS pooling and
F.normalize exactly as the reference does.
"""
import torch
import torch.nn.functional as F


def make_bert(hidden=768, layers=12, heads=12, intermediate=3072, vocab=30522, max_pos=512, seed=0, qk_scale=1.0,
              ln_outlier=1.0):
    """qk_scale / ln_outlier: see sharpen_attention (1.0 = transformers' own init: std 0.02, near-uniform softmax)."""
    from transformers import BertConfig, BertModel
    cfg = BertConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                     intermediate_size=intermediate, max_position_embeddings=max_pos)
    try:
        cfg._attn_implementation = "eager"
    except Exception:
        pass
    torch.manual_seed(seed)
    model = BertModel(cfg, add_pooling_layer=False).eval()
    if qk_scale != 1.0 or ln_outlier != 1.0:
        sharpen_attention(model, qk_scale, ln_outlier, seed=seed + 17)
    return model


@torch.no_grad()
def sharpen_attention(model, qk_scale=6.0, ln_outlier=1.0, seed=17):
    """Random init at std 0.02 gives attention logits of ~0.3, i.e. a softmax within a few percent of uniform: the online
    max / rescale path of an attention kernel and its key masks are then hardly exercised.  Scale the query and key
    projections (weights and freshly drawn biases) by qk_scale so the distributions are peaked, as in trained
    checkpoints; ln_outlier > 1 additionally gives a few LayerNorm channels a large gain and offset (trained BERTs
    carry such outlier channels).  Works on BertModel and DistilBertModel (module names query/key, q_lin/k_lin)."""
    g = torch.Generator().manual_seed(seed)
    for name, mod in model.named_modules():
        leaf = name.rsplit(".", 1)[-1]
        if leaf in ("query", "key", "q_lin", "k_lin") and isinstance(mod, torch.nn.Linear):
            mod.weight.mul_(qk_scale)
            mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.02 * qk_scale)
        elif isinstance(mod, torch.nn.LayerNorm) and ln_outlier != 1.0:
            idx = torch.randperm(mod.weight.numel(), generator=g)[:4]
            mod.weight[idx] *= ln_outlier
            mod.bias[idx] += 0.5 * ln_outlier * torch.randn(4, generator=g)
    return model


@torch.no_grad()
def attention_peak_stats(model, ids, mask, types=None):
    """(mean, min over layers of the mean) of max_k softmax probability over the valid queries -- how far the attention
    of this model on this batch is from uniform (uniform = 1 / valid keys)."""
    kw = dict(input_ids=ids, attention_mask=mask, output_attentions=True)
    if types is not None:
        kw["token_type_ids"] = types
    att = model(**kw).attentions                      # per layer [b, heads, S, S]
    valid = mask.bool()[:, None, :, None]
    per_layer = []
    for a in att:
        mx = a.max(dim=-1, keepdim=True).values       # [b, h, S, 1]
        per_layer.append(float(mx[valid.expand_as(mx)].mean()))
    return sum(per_layer) / len(per_layer), min(per_layer)


def synthetic_batch(b, S, vocab=30522, seed=1234, ragged=True):
    """Token ids ~U[1000, vocab), lengths ~U[S/4, S] (padding id 0, mask 0), CLS-like id 101 first."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1000, vocab, (b, S), generator=g)
    ids[:, 0] = 101
    mask = torch.ones((b, S), dtype=torch.int64)
    if ragged:
        lens = torch.randint(max(1, S // 4), S + 1, (b,), generator=g)
        lens[0] = S
        for i in range(b):
            mask[i, lens[i]:] = 0
            ids[i, lens[i]:] = 0
    types = torch.zeros((b, S), dtype=torch.int64)
    return ids, types, mask


@torch.no_grad()
def encode_cls(model, ids, types, mask):
    out = model(input_ids=ids, token_type_ids=types, attention_mask=mask)
    emb = out.last_hidden_state[:, 0, :]            # classifier.py:1272
    return F.normalize(emb, p=2, dim=1)             # classifier.py:1275


def make_roberta(hidden=128, layers=2, heads=2, intermediate=512, vocab=2000, max_pos=66, seed=0, model_type="roberta"):
    """transformers RobertaModel / XLMRobertaModel (modeling_roberta.py: the BERT block; positions count from
    padding_idx + 1 over the non-pad tokens), fp32 / eval / eager attention, random init."""
    if model_type == "xlm-roberta":
        from transformers import XLMRobertaConfig as Cfg, XLMRobertaModel as Mdl
    else:
        from transformers import RobertaConfig as Cfg, RobertaModel as Mdl
    cfg = Cfg(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
              intermediate_size=intermediate, max_position_embeddings=max_pos, pad_token_id=1, bos_token_id=0, eos_token_id=2,
              hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layer_norm_eps=1e-5)
    try:
        cfg._attn_implementation = "eager"
    except Exception:
        pass
    torch.manual_seed(seed)
    return Mdl(cfg, add_pooling_layer=False).eval()


def roberta_batch(b, S, vocab=2000, seed=1234, ragged=True, pad_id=1):
    """Right-padded RoBERTa inputs: <s> = 0 first, ids ~U[3, vocab), </s> = 2 last, pad id in the padding (the reference
    model derives the position ids from `input_ids != pad_id`, so the padding must really hold it)."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, vocab, (b, S), generator=g)
    ids[:, 0] = 0
    lens = torch.randint(max(2, S // 4), S + 1, (b,), generator=g) if ragged else torch.full((b,), S)
    lens[0] = S
    mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64)
    for i in range(b):
        ids[i, lens[i] - 1] = 2
        ids[i, lens[i]:] = pad_id
    return ids, mask


@torch.no_grad()
def encode_cls_roberta(model, ids, mask):
    out = model(input_ids=ids, attention_mask=mask)
    return F.normalize(out.last_hidden_state[:, 0, :], p=2, dim=1)


def make_modernbert(hidden=768, layers=22, heads=12, intermediate=1152, vocab=50368, max_pos=8192, local_attention=128,
                    global_every=3, seed=0, init_scale=1.0):
    """transformers ModernBertModel (modeling_modernbert.py), fp32 / eval / eager attention, random init.
    init_scale > 1 widens the Linear weights so the attention logits are not all ~0 (a uniform softmax would
    hide RoPE / window mistakes)."""
    from transformers import ModernBertConfig, ModernBertModel
    cfg = ModernBertConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=intermediate,
                           num_hidden_layers=layers, num_attention_heads=heads, max_position_embeddings=max_pos,
                           local_attention=local_attention, global_attn_every_n_layers=global_every,
                           pad_token_id=0, bos_token_id=1, eos_token_id=2, cls_token_id=1, sep_token_id=2)
    try:
        cfg._attn_implementation = "eager"
    except Exception:
        pass
    torch.manual_seed(seed)
    model = ModernBertModel(cfg).eval()
    if init_scale != 1.0:
        with torch.no_grad():
            for name, p in model.named_parameters():
                if name.endswith("Wqkv.weight") or name.endswith("Wi.weight") or name.endswith("Wo.weight"):
                    p.mul_(init_scale)
    return model


@torch.no_grad()
def encode_cls_modernbert(model, ids, mask):
    out = model(input_ids=ids, attention_mask=mask)
    return F.normalize(out.last_hidden_state[:, 0, :], p=2, dim=1)
