"""Shim for Models/SS2D/csms6s.py: the scan-order plugin classes and the selective-scan wrapper.
Nothing runs at import time (the reference builds every table and calls .cuda() here)."""
from tramba_amd.ops import (CrossMerge, CrossMerge_Dilation, CrossMerge_Line, CrossMerge_Window, CrossScan,  # noqa: F401
                            CrossScan_Dilation, CrossScan_Line, CrossScan_Window, SelectiveScanOflex,
                            flops_selective_scan_fn, selective_scan_cuda_oflex)
