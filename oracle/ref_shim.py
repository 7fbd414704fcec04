"""TEST INFRASTRUCTURE ONLY -- import shim for the *reference* package.

Makes `/root/reference/itr` importable on a CPU-only host so that
`oracle/make_goldens.py` can (1) generate golden input/output vectors and
(2) cross-check the CPU restatement in `oracle/itr_oracle.py`.

The reference never travels to the GPU box: nothing outside `oracle/make_goldens.py`
and `oracle/check_oracle_vs_reference.py` may import this file, and both only run in
the build container (they exit early when /root/reference is absent).

What is shimmed:
  * stub modules for packages the image lacks: torchvision(.models,.transforms), nltk,
    pycocotools(.coco.COCO), tensorboard_logger, sacred;
  * CUDA presence: torch.cuda.is_available -> True, Tensor.cuda / Module.cuda -> identity,
    torch.cuda.synchronize -> no-op, nn.DataParallel -> identity
    (the reference only defines the loss mask under `if torch.cuda.is_available()`,
    Objectives.py:105-109).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("ITR_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "itr"))


def install():
    """Install the shims and put the reference on sys.path. Idempotent."""
    import torch
    from torch import nn

    if getattr(install, "_done", False):
        return
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)

    def _stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models")
    tv.transforms = _stub("torchvision.transforms")
    # nltk is absent: the reference's `nltk.tokenize.word_tokenize` (data_loader.py:113, vocab.py:87) is replaced by
    # "words | single punctuation marks"; fixtures made through it say so (tests/golden/g14_data_layer.npz).
    import re
    _word_re = re.compile(r"\w+|[^\w\s]", re.UNICODE)
    nl = _stub("nltk")
    nl.tokenize = _stub("nltk.tokenize", word_tokenize=lambda text: _word_re.findall(text))
    pc = _stub("pycocotools")
    pc.coco = _stub("pycocotools.coco", COCO=object)
    _stub("tensorboard_logger")

    torch.cuda.is_available = lambda: True
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.is_current_stream_capturing = lambda: False   # torch.optim's capture health check (is_available is faked)

    class _IdentityDP(nn.Module):
        def __init__(self, module, *a, **k):
            super().__init__()
            self.module = module

        def forward(self, *a, **k):
            return self.module(*a, **k)

    nn.DataParallel = _IdentityDP
    torch.nn.parallel.DataParallel = _IdentityDP

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    install._done = True


def import_reference():
    """Returns (Objectives, ImgEncoder, TextEncoder, Fusionmodule, Models, evaluation, utils)."""
    install()
    from itr.modalmodule import Objectives, ImgEncoder, TextEncoder, Fusionmodule, Models
    from itr.modalmodule import utils as mutils
    from itr.metricmodule import evaluation
    return Objectives, ImgEncoder, TextEncoder, Fusionmodule, Models, evaluation, mutils
