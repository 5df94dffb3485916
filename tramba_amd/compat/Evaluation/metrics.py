"""`import Evaluation.metrics as M` (train.py:13, test_TSOD.py:4) -> the GPU-reduced metric objects."""
from tramba_amd.evaluate import MAE, Emeasure, Fmeasure_and_FNR, Smeasure, WeightedFmeasure  # noqa: F401
