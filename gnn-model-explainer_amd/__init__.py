"""MI355X-native GNNExplainer mask-optimisation engine (drop-in for the reference's
explainer/explain.py hot path).  See DESIGN.md."""
__version__ = "0.1.0"
