"""MI355X-native hot path of codelion/adaptive-classifier.

Same public names as the reference package (/root/reference/src/adaptive_classifier/__init__.py:1-16)
for the path in scope: predict()/add_examples() = encoder forward -> prototype kNN -> adaptive head
(+ EWC-regularised AdamW training).  Arithmetic runs in libacamd.so (HIP, gfx950); see DESIGN.md.
MultiLabel* and strategic classes of the reference are outside this build.
"""
from .classifier import AdaptiveClassifier
from .ewc import EWC
from .memory import PrototypeMemory
from .models import AdaptiveHead, Example, ModelConfig

__version__ = "0.1.0"

__all__ = ["AdaptiveClassifier", "Example", "AdaptiveHead", "ModelConfig", "PrototypeMemory", "EWC"]
