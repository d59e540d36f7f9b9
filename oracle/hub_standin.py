"""Offline stand-in for the Hugging Face Hub (TEST INFRASTRUCTURE -- never imported by the product).

The reference builds its encoder with `AutoModel.from_pretrained(name)` / `AutoTokenizer.from_pretrained(name)`
(/root/reference/src/adaptive_classifier/classifier.py:83,85) and its own tests name Hub checkpoints
(`tests/test_classifier.py:11` "bert-base-uncased", `tests/test_order_independence.py:10` "answerdotai/ModernBERT-base",
`tests/test_single_example_confidence.py:11` "google-bert/bert-large-cased", `tests/test_multilabel.py:35`
"distilbert/distilbert-base-cased", `tests/test_ewc.py:94` "distilbert-base-uncased").  There is no network here or on the GPU
box, so `install()` patches those two class methods to return

  * a **seeded random-init model of the NAMED ARCHITECTURE** (transformers' own config defaults for that checkpoint: layer
    count, widths, heads, vocabulary size) -- built under `torch.random.fork_rng`, so the caller's RNG stream is what it would be
    after a real `from_pretrained` (which draws nothing);
  * a real transformers **`BertTokenizer` (WordPiece) over a synthetic vocabulary** of the checkpoint's size: BERT's special ids
    (0 [PAD], 100 [UNK], 101 [CLS], 102 [SEP], 103 [MASK]), ASCII characters and their `##` continuations, a few hundred
    English words, and `t<id>` filler tokens for every other id -- so ordinary sentences tokenise into words / characters and
    the text `"t2001 t17000"` tokenises to exactly the ids 2001, 17000 (how bench.py hands its synthetic id batches to the
    reference's text API).

The same stand-in serves both sides of every differential: the unmodified reference on the CPU (tests/golden/gen_golden.py,
bench.py's `cpu_baseline`) and the product on the GPU (tests/, which build `AdaptiveClassifier(name)` exactly as the reference's
tests do).  Same name + same torch build -> bit-identical weights on both sides.

As a pytest plugin (`-p oracle.hub_standin` / `-p hub_standin`) it installs itself at configure time; with
`AC_STANDIN_FAISS_SHIM=1` it also installs oracle/faiss_shim.py when real faiss is absent (reference-on-CPU runs).
"""
import os
import sys
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (family, config overrides, lower_case).  Widths are the published architectures' (transformers config defaults
# where the checkpoint uses them); nothing here is a trained weight.
ARCHITECTURES = {
    "bert-base-uncased": ("bert", {}, True),
    "google-bert/bert-base-uncased": ("bert", {}, True),
    "bert-base-cased": ("bert", {"vocab_size": 28996}, False),
    "google-bert/bert-large-cased": ("bert", {"vocab_size": 28996, "hidden_size": 1024, "num_hidden_layers": 24,
                                              "num_attention_heads": 16, "intermediate_size": 4096}, False),
    "bert-large-uncased": ("bert", {"hidden_size": 1024, "num_hidden_layers": 24, "num_attention_heads": 16,
                                    "intermediate_size": 4096}, True),
    "intfloat/e5-large-v2": ("bert", {"hidden_size": 1024, "num_hidden_layers": 24, "num_attention_heads": 16,
                                      "intermediate_size": 4096}, True),
    "distilbert-base-uncased": ("distilbert", {}, True),
    "distilbert/distilbert-base-uncased": ("distilbert", {}, True),
    "distilbert/distilbert-base-cased": ("distilbert", {"vocab_size": 28996}, False),
    "distilbert-base-cased": ("distilbert", {"vocab_size": 28996}, False),
    "answerdotai/ModernBERT-base": ("modernbert", {}, True),
    # small stand-ins of our own for fixtures that must stay small (tests/golden/e2e_*)
    "standin/bert-mini-4l": ("bert", {"vocab_size": 4000, "hidden_size": 128, "num_hidden_layers": 4, "num_attention_heads": 2,
                                      "intermediate_size": 512, "max_position_embeddings": 128}, True),
    "standin/bert-small-2l-d64": ("bert", {"vocab_size": 3000, "hidden_size": 64, "num_hidden_layers": 2,
                                           "num_attention_heads": 1, "intermediate_size": 256,
                                           "max_position_embeddings": 128}, True),
    "standin/distilbert-mini-3l": ("distilbert", {"vocab_size": 4000, "dim": 128, "n_layers": 3, "n_heads": 2, "hidden_dim": 512,
                                                  "max_position_embeddings": 128}, True),
    "standin/modernbert-mini-4l": ("modernbert", {"vocab_size": 4000, "hidden_size": 128, "num_hidden_layers": 4,
                                                  "num_attention_heads": 2, "intermediate_size": 192,
                                                  "max_position_embeddings": 256, "pad_token_id": 0, "bos_token_id": 101,
                                                  "cls_token_id": 101, "eos_token_id": 102, "sep_token_id": 102}, True),
}

WORDS = """the of and to in is that for it as was with be by on not he this are or his from at which but have an had they you
were their one all we can her has there been if more when will would who so no out up into do any your what some my me just
about good bad great terrible amazing okay love hate like best worst product service experience quality price time help need
problem issue error system crash bug code python data model test user account login password reset order refund money payment
bill charge shipping delivery late fast slow broken works working excellent awful fine average nice poor happy sad angry
positive negative neutral technical support billing question answer how why where when high low medium simple complex hard easy
new old first last next other many much very really too also only even still again never always sometimes today now soon
computer software hardware network server database file memory null pointer exception function class method variable loop
science sports politics business technology health food travel music movie book game weather news market stock economy
team player win lose score match season coach election government policy law court company customer sales profit growth
doctor patient medicine hospital disease treatment study research scientists discovered space planet energy climate
river mountain city country world people man woman child family friend school student teacher learn write read speak
buy sell pay cost cheap expensive free offer deal recommend return cancel subscription update install download upload
email message phone call chat website app mobile screen button click page link search find open close start stop run
i am was be being been did does done get got make made take took see saw know knew think thought come came want
this these those here then than them she him its our us over under after before between through during against
amazing fantastic wonderful horrible disappointing satisfied unhappy pleased delighted impressed mediocre decent""".split()


def vocabulary(size, lower_case=True):
    """The synthetic WordPiece vocabulary of `size` entries (deterministic; ids documented in the module docstring)."""
    toks = ["[PAD]"] + [f"[unused{i}]" for i in range(1, 100)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    seen = set(toks)

    def add(t):
        if t not in seen and len(toks) < size:
            seen.add(t)
            toks.append(t)

    chars = [chr(c) for c in range(33, 127)]
    if lower_case:
        chars = [c for c in chars if not ("A" <= c <= "Z")]
    for c in chars:
        add(c)
    for c in chars:
        if c.isalnum():
            add("##" + c)
    for w in WORDS:
        add(w)
        if not lower_case:
            add(w.capitalize())
    i = len(toks)
    while len(toks) < size:              # filler: the token for id i is literally "t<i>"
        t = f"t{i}"
        if t in seen:                    # (cannot happen: fillers are the only t<digits> tokens; kept for safety)
            t = f"t{i}x"
        seen.add(t)
        toks.append(t)
        i += 1
    assert len(toks) == size and toks[101] == "[CLS]" and toks[102] == "[SEP]"
    return toks


def text_for_ids(ids):
    """A text whose tokenisation (without the specials) is exactly `ids` -- every id must be a filler id (>= first_filler_id)."""
    return " ".join(f"t{int(i)}" for i in ids)


def first_filler_id(size, lower_case=True):
    v = vocabulary(size, lower_case)
    for i, t in enumerate(v):
        if t == f"t{i}":
            return i
    return size


def _config(name):
    import transformers
    if name not in ARCHITECTURES:
        raise OSError(f"hub_standin: {name!r} is not one of the architectures this offline stand-in knows "
                      f"({sorted(ARCHITECTURES)}); there is no network to fetch it")
    family, over, lower = ARCHITECTURES[name]
    cls = {"bert": transformers.BertConfig, "distilbert": transformers.DistilBertConfig,
           "modernbert": transformers.ModernBertConfig}[family]
    cfg = cls(**over)
    cfg._name_or_path = name
    return family, cfg, lower


def make_model(name, seed=0):
    """Seeded random-init transformers model of the named architecture (eval mode, fp32, CPU); the caller's torch RNG is untouched."""
    import transformers
    family, cfg, _ = _config(name)
    cls = {"bert": transformers.BertModel, "distilbert": transformers.DistilBertModel,
           "modernbert": transformers.ModernBertModel}[family]
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(1000003 * seed + sum(name.encode()))
        model = cls(cfg) if family != "bert" else cls(cfg, add_pooling_layer=True)
    model.config._name_or_path = name
    return model.eval()


_TOK_DIRS = {}


def make_tokenizer(name):
    """transformers BertTokenizer over the synthetic vocabulary of the named checkpoint's size."""
    import transformers
    _, cfg, lower = _config(name)
    key = (cfg.vocab_size, lower)
    d = _TOK_DIRS.get(key)
    if d is None:
        d = _TOK_DIRS[key] = tempfile.mkdtemp(prefix="hub_standin_vocab_")
        with open(os.path.join(d, "vocab.txt"), "w") as f:
            f.write("\n".join(vocabulary(cfg.vocab_size, lower)) + "\n")
    tok = transformers.BertTokenizer(os.path.join(d, "vocab.txt"), do_lower_case=lower)
    tok.model_max_length = min(512, getattr(cfg, "max_position_embeddings", 512))
    return tok


_installed = {}


def install():
    """Patch AutoModel.from_pretrained / AutoTokenizer.from_pretrained (idempotent).  Returns the module for chaining."""
    import transformers
    if _installed:
        return sys.modules[__name__]
    _installed["model"] = transformers.AutoModel.__dict__.get("from_pretrained")
    _installed["tok"] = transformers.AutoTokenizer.__dict__.get("from_pretrained")

    def model_from_pretrained(cls, name, *args, **kwargs):
        return make_model(str(name))

    def tok_from_pretrained(cls, name, *args, **kwargs):
        return make_tokenizer(str(name))

    transformers.AutoModel.from_pretrained = classmethod(model_from_pretrained)
    transformers.AutoTokenizer.from_pretrained = classmethod(tok_from_pretrained)
    return sys.modules[__name__]


def uninstall():
    import transformers
    if not _installed:
        return
    for cls, key in ((transformers.AutoModel, "model"), (transformers.AutoTokenizer, "tok")):
        orig = _installed[key]
        if orig is None:
            try:
                delattr(cls, "from_pretrained")
            except AttributeError:
                pass
        else:
            setattr(cls, "from_pretrained", orig)
    _installed.clear()


def pytest_configure(config):            # `pytest -p hub_standin`: the child pytest of tests/test_reference_suite_gpu.py
    install()
    if os.environ.get("AC_STANDIN_FAISS_SHIM") == "1":
        try:
            import faiss  # noqa: F401
        except ImportError:
            sys.path.insert(0, os.path.dirname(HERE))
            from oracle import faiss_shim
            faiss_shim.install()
