"""MI355X-native hot path of codelion/adaptive-classifier."""
__version__ = "0.1.0"
