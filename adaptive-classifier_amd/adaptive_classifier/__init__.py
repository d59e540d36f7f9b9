"""MI355X-native hot path of codelion/adaptive-classifier.

Same public names as the reference package (/root/reference/src/adaptive_classifier/__init__.py:1-16)
for the path in scope: predict()/add_examples() = encoder forward -> prototype kNN -> adaptive head
(+ EWC-regularised AdamW training).  Arithmetic runs in libacamd.so (HIP, gfx950); see DESIGN.md.
The multi-label classes are the N3 widening; strategic classes are outside this build.
"""
from .classifier import AdaptiveClassifier
from .ewc import EWC
from .memory import PrototypeMemory
from .models import AdaptiveHead, Example, ModelConfig
from .multilabel import MultiLabelAdaptiveClassifier, MultiLabelAdaptiveHead

__version__ = "0.1.0"

__all__ = ["AdaptiveClassifier", "MultiLabelAdaptiveClassifier", "MultiLabelAdaptiveHead", "Example", "AdaptiveHead",
           "ModelConfig", "PrototypeMemory", "EWC"]
