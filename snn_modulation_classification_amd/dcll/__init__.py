"""DCLL layers computed by the HIP extension (see pytorch_libdcll.py)."""
